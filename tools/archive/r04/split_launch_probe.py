#!/usr/bin/env python3
"""Round 4: is the slowdown of launches beyond ~1 GB a property of the LAUNCH (grid size) or of the footprint?  One environment of N
boards against P environments of N / P boards stepped back to back on the same stream (same total boards, state and output)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("S T total MB | us per step of ALL boards (frac) with the batch in 1 / 2 / 3 / 4 launches")
for S, T, K, mb in ((15, 32, 24, 1400), (15, 32, 24, 2100), (16, 16, 24, 1400), (9, 4, 9, 1400), (24, 30, 60, 1400), (8, 20, 10, 1400), (5, 6, 3, 1400), (13, 3, 10, 2100), (3, 1, 0, 2100), (7, 5, 6, 2100), (4, 2, 2, 2100)):
    n = (mb * 1_000_000 // (12 * S * S)) // 3072 * 3072
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
    row = f"{S:2d} {T:2d} {mb:4d} |"
    for parts in (1, 2, 3, 4):
        envs = [VecTilerSliderEnv.random(n // parts, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True, board_offset=p * (n // parts))
                for p in range(parts)]
        acts = [torch.randint(0, 4, (n // parts,), dtype=torch.uint8, device=envs[0].device) for _ in range(4)]
        for e in envs:
            e.reset()
        for i in range(40):
            for e in envs:
                e.step_async(acts[i & 3])
        ts = []
        for r in range(3):
            e0.record()
            for i in range(12):
                for e in envs:
                    e.step_async(acts[i & 3])
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 12 * 1e3)
        us = statistics.median(ts)
        row += f" {us:7.1f} ({bps * n / us / 8e6:.3f})"
        del envs, acts
    print(row, flush=True)
