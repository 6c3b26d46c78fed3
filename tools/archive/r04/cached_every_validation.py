#!/usr/bin/env python3
"""Round 4: the shipped policy (every 16th wave of a single-stream k_small launch of up to 768 MiB writes with the cached
stores) against all-nontemporal (ts_tuning(TS_TUNE_CACHED_EVERY, 1)), same buffers, per entry point: step, step + reward +
legality, reset, encode; and, for reference, launches the policy leaves alone (one-hot planes: forced 16 against never)."""
import ctypes as C
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

L = _cabi.lib()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
SHAPES = [(5, 2, 3, 1 << 20), (5, 6, 3, 1 << 20), (6, 3, 4, 1 << 20), (6, 12, 4, 1 << 20), (8, 4, 8, 650_000), (8, 12, 8, 650_000), (3, 1, 0, 3_000_000), (4, 2, 2, 2_000_000),
          (4, 5, 2, 2_000_000), (7, 5, 6, 850_000), (7, 12, 6, 850_000), (2, 1, 0, 8_000_000), (6, 5, 4, 700_000), (5, 3, 3, 1_500_000), (8, 8, 8, 500_000), (3, 2, 0, 4_000_000), (6, 1, 6, 1_500_000)]


def rate(fn, reps=30):
    ts = []
    for r in range(3):
        for i in range(5):
            fn(i)
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)


print("   S    T   boards  out MB | us: never -> policy (change) for step | step + reward + legality | reset | encode | step + one-hot planes (never -> forced 16)")
for S, T, K, n in SHAPES:
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    row = f"{S:4d} {T:4d} {n:8d} {12 * S * S * n / 1e6:7.0f} |"
    for kind in ("step", "extras", "reset", "encode", "onehot"):
        env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True,
                                       with_reward=kind in ("extras", "onehot"), with_valid_moves=kind == "extras", with_onehot=kind == "onehot", obs_candidates=0)
        env.reset()
        if kind in ("step", "extras", "onehot"):
            fn = lambda i: env.step_async(act[i & 3])
        elif kind == "reset":
            fn = lambda i: env.reset()
        else:
            fn = lambda i: env.encode()
        for i in range(100):
            fn(i)
        res = {}
        for setting in ((1, 16, 1, 16) if kind == "onehot" else (1, 0, 1, 0)):
            L.ts_tuning(_cabi.TUNE_CACHED_EVERY, setting)
            res.setdefault(setting, []).append(rate(fn))
        L.ts_tuning(_cabi.TUNE_CACHED_EVERY, 0)
        off, on = min(res[1]), min(res[16 if kind == "onehot" else 0])
        row += f" {off:6.1f} -> {on:6.1f} ({(on / off - 1) * 100:+5.1f} %) |"
        del env
    print(row, flush=True)
