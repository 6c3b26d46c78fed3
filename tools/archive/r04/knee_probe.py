#!/usr/bin/env python3
"""Round 4: where does a launch start to lose its rate as the batch grows, and is it the OUTPUT or the STATE footprint that counts?
float32 observations (12 B per cell) against uint8 observations (3 B per cell) of the same boards."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for S, T, K in ((15, 32, 24), (16, 16, 24), (8, 20, 10)):
    print(f"{S}x{S} T={T}: boards | state MB | float32 obs: MB, us, ns per board | uint8 obs: MB, us, ns per board")
    for mb in (500, 700, 850, 1000, 1150, 1300, 1500, 1800, 2400):
        n = (mb * 1_000_000 // (12 * S * S)) // 256 * 256
        row = f"  {n:8d} |"
        for dt in ("float32", "uint8"):
            env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True, obs_dtype=dt)
            env.reset()
            act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
            for i in range(40):
                env.step_async(act[i & 3])
            ts = []
            for r in range(3):
                for i in range(3):
                    env.step_async(act[i & 3])
                e0.record()
                for i in range(12):
                    env.step_async(act[i & 3])
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 12 * 1e3)
            us = statistics.median(ts)
            if dt == "float32":
                lines = getattr(env, "_lines", None)
                st = sum(t.numel() * t.element_size() for t in (env._pos, env._tgt, env._init, env._blk, env._step_count, env._done) if t is not None) + (lines.numel() * 4 if lines is not None else 0)
                row += f" {st / 1e6:5.0f} |"
            row += f" {(12 if dt == 'float32' else 3) * S * S * n / 1e6:6.0f} MB {us:7.1f} us {us / n * 1e3:6.3f} |"
            del env, act
        print(row, flush=True)
