#!/usr/bin/env python3
"""Round 4: finer sweep of the XCD piece size (ts_dims.xcd_piece: pieces of P consecutive blocks per XCD; P need not be a
power of two) around the policy's value, on physically contiguous buffers, static policy otherwise."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv

PIECES = [0, 6, 8, 10, 12, 14, 16, 18, 20, 24, 28, 32, 40, 48, 64, 96]
SHAPES = [("cfg2", 5, 2, 3, 1 << 20, True), ("cfg4", 15, 32, 24, 1 << 18, False), ("4x4 4M", 4, 2, 2, 1 << 22, False), ("8x8/20", 8, 20, 10, 650_000, False),
          ("8x8/4", 8, 4, 8, 650_000, False), ("12x12/8", 12, 8, 16, 400_000, False), ("24x24/30", 24, 30, 60, 100_000, False), ("32x32/32", 32, 32, 100, 60_000, False)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("us per step by xcd_piece (0 = the policy): " + " ".join(f"{p:6d}" for p in PIECES))
for name, S, T, K, n, extras in SHAPES:
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True,
                                   with_onehot=extras, with_reward=extras)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
    for i in range(300):
        env.step_async(act[i & 3])
    row = []
    for p in PIECES:
        env._dims.xcd_piece = p
        ts = []
        for r in range(3):
            for i in range(5):
                env.step_async(act[i & 3])
            e0.record()
            for i in range(30):
                env.step_async(act[i & 3])
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 30 * 1e3)
        row.append(statistics.median(ts))
    best = min(range(len(row)), key=row.__getitem__)
    print(f"{name:10s} " + " ".join(f"{u:6.1f}" for u in row) + f"   best {PIECES[best]} ({(row[0] / row[best] - 1) * 100:+.1f} % vs policy)", flush=True)
    del env
