#!/usr/bin/env python3
"""Round 4: boards up to 8x8 with more than 8 tiles - one lane per board (k_small's any-tile-count path, TS_TUNE_DEAL = 0)
against a board's tiles dealt over 4 / 8 lanes (k_deal), out-of-cache batches of ~500 MB and cache-resident ones."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

L = _cabi.lib()
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


print("   S    T    K    boards   out MB |  one lane   |  deal, policy |  deal, 8 lanes |  + hint -2 / +2 / +4 (policy lanes)      (us per step, frac of 8 TB/s)")
for S, T, K, n in ((8, 20, 10, 651008), (8, 12, 8, 651008), (8, 9, 8, 651008), (8, 16, 8, 651008), (8, 32, 8, 651008), (8, 40, 4, 651008),
                   (7, 12, 6, 850176), (6, 12, 4, 1048576), (6, 9, 4, 1048576), (5, 12, 3, 1048576), (5, 9, 3, 1048576), (4, 10, 2, 1048576), (4, 10, 2, 4194304),
                   (8, 20, 10, 262144), (6, 12, 4, 262144)):
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False)
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=bench.LEVEL_SEED, multi_color=True, max_steps=2**30, auto_reset=True)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
    frac = lambda us: bps * n / us / 1e3 / 8000
    row = f"{S:4d} {T:4d} {K:4d} {n:9d} {12 * S * S * n / 1e6:8.0f} |"
    L.ts_tuning(_cabi.TUNE_DEAL, 0)
    us = rate(env, act); row += f" {us:6.1f} {frac(us):.3f} |"
    L.ts_tuning(_cabi.TUNE_DEAL, 1)
    us = rate(env, act); row += f" {us:6.1f} {frac(us):.3f}  |"
    env._dims.lines_lanes = 8
    us = rate(env, act); row += f" {us:6.1f} {frac(us):.3f}   |"
    env._dims.lines_lanes = 0
    for h in (-2, 2, 4):
        env._dims.launch_hint = h
        us = rate(env, act, reps=12, rounds=2); row += f" {us:6.1f}"
    env._dims.launch_hint = 0
    print(row, flush=True)
    del env, act
    torch.cuda.empty_cache()
