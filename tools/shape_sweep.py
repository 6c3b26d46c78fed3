#!/usr/bin/env python3
"""Development tool: step() throughput and algorithmic bandwidth over a sweep of board shapes."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv

SHAPES = [(3, 1, 0), (4, 2, 2), (4, 4, 2), (5, 2, 3), (5, 6, 3), (6, 3, 4), (7, 5, 6), (8, 4, 8), (8, 12, 8), (8, 20, 10), (9, 1, 9), (9, 4, 9),
          (10, 5, 10), (11, 6, 8), (12, 8, 16), (13, 3, 10), (14, 20, 20), (15, 32, 24), (16, 4, 24), (16, 16, 24), (20, 1, 1), (20, 10, 40), (24, 30, 60),
          (32, 4, 100), (32, 32, 100)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print(f"# static launch policy, output_memory={os.environ.get('TS_SWEEP_MEM', 'contiguous')}")
print("   S    T    K    boards   out MB    us/step   steps/s    alg GB/s   % of 8 TB/s")
for S, T, K in SHAPES:
    n = max(4096, min(1 << 20, (500_000_000 // (12 * S * S)) // 256 * 256))
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30,
                                   auto_reset=True, placement_trials=int(os.environ.get('TS_SWEEP_TRIALS', '0')), output_memory=os.environ.get('TS_SWEEP_MEM', 'contiguous'))
    env.reset()
    act = torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device)
    def timed():
        ts = []
        for r in range(4):
            for i in range(3 if r else 60):  # the first round also ramps the clocks up after the host-side set-up
                env.step_async(act)
            e0.record()
            for i in range(20):
                env.step_async(act)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        return statistics.median(ts)

    us = timed()
    extra = ""
    for piece in [int(x) for x in os.environ.get("TS_SWEEP_PIECES", "").split(",") if x]:  # the same buffers with another block mapping
        env._dims.xcd_piece = piece
        extra += f"   piece {piece}: {timed():7.1f}"
    env._dims.xcd_piece = 0
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0) * 1  # 16-bit cells above 16x16
    gbs = bps * n / us / 1e3
    print(f"{S:4d} {T:4d} {K:4d} {n:9d} {12 * S * S * n / 1e6:8.0f} {us:10.1f} {n / us * 1e6:10.3e} {gbs:10.0f} {gbs / 80:10.1f}{extra}", flush=True)
    del env
