#!/usr/bin/env python3
"""Round 4: resident one-wave blocks per CU (ts_dims.launch_hint, relative to the library's policy) by board shape on
physically contiguous output buffers, ~500 MB batches; then the XCD piece at the best hint.  Long warm-up per cell."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

HINTS = tuple(int(x) for x in os.environ.get("TS_SWEEP_HINTS", "-2,0,2,4,6,8").split(","))
MEM = os.environ.get("TS_SWEEP_MEM", "contiguous")
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


SHAPES = [(3, 1, 0, 0), (4, 2, 2, 0), (4, 4, 2, 0), (5, 2, 3, 0), (5, 2, 3, 1), (5, 6, 3, 0), (6, 3, 4, 0), (7, 5, 6, 0), (8, 4, 8, 0), (8, 12, 8, 0), (8, 20, 10, 0), (9, 1, 9, 0), (9, 4, 9, 0),
          (10, 5, 10, 0), (11, 6, 8, 0), (12, 8, 16, 0), (13, 3, 10, 0), (14, 20, 20, 0), (15, 32, 24, 0), (16, 4, 24, 0), (16, 16, 24, 0), (20, 1, 1, 0), (20, 10, 40, 0),
          (24, 30, 60, 0), (32, 4, 100, 0), (32, 32, 100, 0)]
print(f"output_memory={MEM}\n   S    T    K oh    boards   out MB | us per step at launch_hint {HINTS} | at the best hint: xcd_piece 1 / 16 / 32 / 64 | best frac of 8 TB/s")
for S, T, K, oh in SHAPES:
    per = 12 * S * S + (4 * S * S * (1 + 2 * T) if oh else 0)
    n = (int(os.environ.get("TS_SWEEP_BYTES", "600000000")) // per) // 256 * 256
    bps = bench.algorithmic_bytes_per_board_step(S, T, bool(oh), bool(oh)) + (T * 2 if S > 16 else 0)
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True,
                                   with_onehot=bool(oh), with_reward=bool(oh), output_memory=MEM)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    rate(env, act)
    row = f"{S:4d} {T:4d} {K:4d} {oh:2d} {n:9d} {per * n / 1e6:8.0f} |"
    res = {}
    for h in HINTS:
        env._dims.launch_hint = h
        res[h] = rate(env, act)
        row += f" {res[h]:6.1f}"
    best = min(res, key=res.get)
    env._dims.launch_hint = best
    row += f" | hint {best:+d}:"
    allus = [res[best]]
    for p in (1, 16, 32, 64):
        env._dims.xcd_piece = p
        us = rate(env, act)
        allus.append(us)
        row += f" {us:6.1f}"
    print(row + f" | {bps * n / min(allus) / 1e3 / 8000:.3f}", flush=True)
    del env, act
    torch.cuda.empty_cache()
