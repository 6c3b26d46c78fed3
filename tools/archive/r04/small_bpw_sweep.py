#!/usr/bin/env python3
"""Round 4 experiment: boards per wave (64 / 32 / 16) of k_small's register path beyond the Infinity Cache x launch_hint, on
physically contiguous output buffers, 600 MB batches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

L = _cabi.lib()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def rate(env, act, reps=40, warm=50):
    for i in range(warm):
        env.step_async(act[i & 3])
    e0.record()
    for i in range(reps):
        env.step_async(act[i & 3])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


HINTS = (-4, 0, 4, 8)
print(f"   S    T    K oh   boards | policy | per boards-per-wave 64 / 32 / 16: us at launch_hint {HINTS}")
SHAPES = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(4, 2, 2, 0), (5, 2, 3, 0), (5, 2, 3, 1), (6, 3, 4, 0), (6, 8, 4, 0), (7, 5, 6, 0), (8, 4, 8, 0), (8, 8, 8, 0), (8, 1, 10, 0)]
for S, T, K, oh in SHAPES:
    per = 12 * S * S + (4 * S * S * (1 + 2 * T) if oh else 0)
    n = (600_000_000 // per) // 256 * 256
    bps = bench.algorithmic_bytes_per_board_step(S, T, bool(oh), bool(oh))
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True,
                                   with_onehot=bool(oh), with_reward=bool(oh))
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    rate(env, act)
    base = rate(env, act)
    row = f"{S:4d} {T:4d} {K:4d} {oh:2d} {n:9d} | {base:6.1f} ({bps * n / base / 1e3 / 8000:.3f}) |"
    best = base
    for bpw in (64, 32, 16):
        L.ts_tuning(_cabi.TUNE_SMALL_BPW, bpw)
        row += f" {bpw}:"
        for h in HINTS:
            env._dims.launch_hint = h
            us = rate(env, act)
            best = min(best, us)
            row += f" {us:6.1f}"
        row += " |"
    L.ts_tuning(_cabi.TUNE_SMALL_BPW, 0)
    env._dims.launch_hint = 0
    print(row + f" best {bps * n / best / 1e3 / 8000:.3f}", flush=True)
    del env, act
    torch.cuda.empty_cache()
