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


print("   S    T    K    boards   out MB |  one lane   |  deal, policy | per lanes-per-board: us at launch_hint 0 / +4 / +8 (no effect without a bound), then with the large-board kernel's bound   (us per step, frac of 8 TB/s)")
for S, T, K, n in ((8, 20, 10, 651008), (8, 12, 8, 651008), (8, 9, 8, 651008), (8, 16, 8, 651008), (8, 17, 8, 651008), (8, 24, 8, 651008), (8, 28, 6, 651008),
                   (7, 12, 6, 850176), (7, 20, 4, 850176), (8, 20, 10, 262144), (7, 12, 6, 262144)):
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
    for lanes in (4, 8):
        if lanes == 4 and T > 32:
            row += "      -       |"
            continue
        env._dims.lines_lanes = lanes
        row += f" {lanes} lanes:"
        for h in (0, 4, 8):
            env._dims.launch_hint = h
            us = rate(env, act, reps=12, rounds=2); row += f" {us:6.1f}"
        env._dims.launch_hint = 0
        L.ts_tuning(_cabi.TUNE_DEAL, 3)  # experiment: the large-board kernel's bound on the resident blocks
        us = rate(env, act, reps=12, rounds=2); row += f" bounded {us:6.1f} {frac(us):.3f} |"
        L.ts_tuning(_cabi.TUNE_DEAL, 1)
    env._dims.lines_lanes = 0
    print(row, flush=True)
    del env, act
    torch.cuda.empty_cache()
