#!/usr/bin/env python3
"""cfg2 (two streams, observation buffer chosen among 16 candidates): never / every 32nd / 16th / 8th wave's observation with the cached stores."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi
L = _cabi.lib()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for S, T, K, n in ((5, 2, 3, 1 << 20), (5, 3, 3, 1 << 20), (4, 2, 2, 1 << 21), (6, 3, 4, 1 << 19)):
    L.ts_tuning(_cabi.TUNE_CACHED_EVERY, 1)
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True, with_reward=True, with_onehot=True)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    for i in range(300):
        env.step_async(act[i & 3])
    out = []
    for setting in (1, 32, 16, 8, 1, 32, 16, 8):
        L.ts_tuning(_cabi.TUNE_CACHED_EVERY, setting)
        ts = []
        for r in range(3):
            for i in range(5):
                env.step_async(act[i & 3])
            e0.record()
            for i in range(40):
                env.step_async(act[i & 3])
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 40 * 1e3)
        out.append(f"{setting}: {statistics.median(ts):.2f}")
    L.ts_tuning(_cabi.TUNE_CACHED_EVERY, 0)
    print((S, T, n), env.observation_placement_report, " | ".join(out), flush=True)
    del env
