#!/usr/bin/env python3
"""Round 4: block -> board-range mapping (xcd_piece) by board shape on physically contiguous output buffers, ~500 MB batches
(the shapes of tools/shape_sweep.py).  1 = one contiguous eighth of the batch per XCD; P = pieces of P one-wave blocks."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

PIECES = (0, 1, 8, 16, 32, 64, 128)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def rate(env, act, reps=20, rounds=3):
    ts = []
    for r in range(rounds):
        for i in range(3):
            env.step_async(act[i & 3])
        e0.record()
        for i in range(reps):
            env.step_async(act[i & 3])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)


print(f"   S    T    K    boards   out MB | us per step at xcd_piece {PIECES} (0 = the library's policy) | best frac of 8 TB/s")
SHAPES = [(4, 2, 2, 4194304), (4, 4, 2, 4194304), (5, 2, 3, 1677721), (5, 6, 3, 1677721), (6, 3, 4, 1048576), (7, 5, 6, 850176), (8, 4, 8, 651008),
          (8, 12, 8, 651008), (8, 20, 10, 651008), (9, 1, 9, 514304), (9, 4, 9, 514304), (10, 5, 10, 416512), (11, 6, 8, 344320), (12, 8, 16, 289280),
          (13, 3, 10, 246528), (14, 20, 20, 212480), (15, 32, 24, 185088), (16, 4, 24, 162560), (16, 16, 24, 162560), (20, 1, 1, 103936),
          (20, 10, 40, 103936), (24, 30, 60, 72192), (32, 4, 100, 40448), (32, 32, 100, 40448)]
for S, T, K, n in SHAPES:
    n -= n % 2
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False)
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=bench.LEVEL_SEED, multi_color=True, max_steps=2**30, auto_reset=True)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
    row = f"{S:4d} {T:4d} {K:4d} {n:9d} {12 * S * S * n / 1e6:8.0f} |"
    best = 1e9
    for p in PIECES:
        env._dims.xcd_piece = p
        us = rate(env, act)
        best = min(best, us)
        row += f" {us:6.1f}"
    env._dims.xcd_piece = 0
    print(row + f" | {bps * n / best / 1e3 / 8000:.3f}", flush=True)
    del env, act
    torch.cuda.empty_cache()
