#!/usr/bin/env python3
"""Round 4: do the write-back edge stores (a cached share of ~2 of 11 store instructions per chunk) explain the loss of rate beyond ~1 GB?
emit_edges policy / none / first / last / both at 0.7 - 2.4 GB."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("S T MB | us per step with emit_edges = policy / none / first / last / both | best frac")
for S, T, K in [tuple(int(v) for v in x.split(',')) for x in os.environ.get('SHAPES', '15,32,24;16,16,24;8,20,10;24,30,60;12,8,16;6,3,4;8,4,8').split(';')]:
    for mb in [int(x) for x in os.environ.get('MB', '700,1400,2400').split(',')]:
        n = (mb * 1_000_000 // (12 * S * S)) // 256 * 256
        env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True)
        env.reset()
        act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
        for i in range(40):
            env.step_async(act[i & 3])
        row = []
        for ed in (0, 1, 2, 3, 4):
            env._dims.emit_edges = ed
            ts = []
            for r in range(3):
                for i in range(3):
                    env.step_async(act[i & 3])
                e0.record()
                for i in range(12):
                    env.step_async(act[i & 3])
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 12 * 1e3)
            row.append(statistics.median(ts))
        bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
        print(f"{S:2d} {T:2d} {mb:4d} | " + " ".join(f"{u:7.1f}" for u in row) + f" | {bps * n / min(row) / 8e6:.3f} (policy {bps * n / row[0] / 8e6:.3f})", flush=True)
        del env, act
