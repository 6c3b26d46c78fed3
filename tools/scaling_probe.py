#!/usr/bin/env python3
"""Round 4: fraction of the HBM roofline by batch size (0.7 / 1.4 / 2.1 GB of observation per launch) for every kernel family, static policy."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
MB = [int(x) for x in os.environ.get("MB", "700,1400,2100").split(",")]
print("   S    T | " + " | ".join(f"{mb:5d} MB: us (frac)" for mb in MB))
for S, T, K in ((3, 1, 0), (4, 2, 2), (5, 2, 3), (5, 6, 3), (6, 3, 4), (6, 12, 4), (7, 5, 6), (8, 4, 8), (8, 12, 8), (8, 20, 10), (9, 4, 9), (10, 5, 10), (12, 8, 16), (13, 3, 10), (15, 32, 24), (15, 8, 24),
                (16, 16, 24), (20, 10, 40), (24, 30, 60), (28, 8, 60), (32, 4, 100), (32, 32, 100)):
    row = f"{S:4d} {T:4d} |"
    for mb in MB:
        n = (mb * 1_000_000 // (12 * S * S)) // 256 * 256
        env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True)
        env.reset()
        act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
        for i in range(60):
            env.step_async(act[i & 3])
        ts = []
        for r in range(3):
            for i in range(3):
                env.step_async(act[i & 3])
            e0.record()
            for i in range(15):
                env.step_async(act[i & 3])
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 15 * 1e3)
        us = statistics.median(ts)
        bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
        row += f" {us:8.1f} ({bps * n / us / 8e6:.3f}) |"
        del env, act
    print(row, flush=True)
