#!/usr/bin/env python3
"""Round 4 experiment: cache-resident launches (cfg1: 222 MB in a 256 MiB Infinity Cache) with every N-th wave streaming its
observation PAST the cache (nontemporal), the mirror image of the cached waves of launches beyond it.  ts_tuning(10, N)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi
L = _cabi.lib()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for S, T, K, n in ((4, 2, 2, 1 << 20), (4, 2, 2, 1_200_000), (5, 2, 3, 800_000), (3, 1, 0, 1 << 21), (4, 2, 2, 1 << 19)):
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(16)]
    for i in range(2000):
        env.step_async(act[i & 15])
    out = []
    for N in (0, 2, 3, 4, 6, 8, 16, 0, 2, 4, 8):
        L.ts_tuning(10, N)
        ts = []
        for r in range(5):
            for i in range(20):
                env.step_async(act[i & 15])
            e0.record()
            for i in range(200):
                env.step_async(act[i & 15])
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 200 * 1e3)
        out.append(f"{N}: {statistics.median(ts):.2f}")
    L.ts_tuning(10, 0)
    print(f"{S}x{S} T={T} {n} boards ({12 * S * S * n / 1e6:.0f} MB): " + " | ".join(out), flush=True)
    del env
