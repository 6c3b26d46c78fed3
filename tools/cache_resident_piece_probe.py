#!/usr/bin/env python3
"""Round 5 (needs a build that honours ts_dims.xcd_piece for cache-resident launches - measured once, flat, not shipped): block -> XCD mapping of CACHE-RESIDENT launches (one contiguous eighth of the batch per XCD by default): pieces of P blocks
per XCD instead, at cfg1 and a few more cache-resident shapes.     python tools/cache_resident_piece_probe.py      (GPU box)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def rate(env, act, steps=200):
    ts = []
    for r in range(5):
        for i in range(20):
            env.step_async(act[i & 3])
        e0.record()
        for i in range(steps):
            env.step_async(act[i & 3])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e3)
    return statistics.median(ts)


for S, T, K, N in ((4, 2, 2, 1 << 20), (3, 1, 0, 1 << 20), (5, 2, 3, 1 << 19), (8, 4, 8, 1 << 17), (15, 32, 24, 1 << 16), (20, 10, 40, 1 << 15)):
    env = VecTilerSliderEnv.random(N, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True)
    env.reset()
    act = [torch.randint(0, 4, (N,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    for i in range(2000):
        env.step_async(act[i & 3])
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
    d = _cabi.describe_launch(env._dims, _cabi.OP_STEP, _cabi.OUT_OBS | _cabi.OUT_FLAGS)
    out = [f"eighths: {rate(env, act):.2f}"]
    for piece in (2, 4, 8, 16, 32, 64, 128):
        env._dims.xcd_piece = piece
        out.append(f"{piece}: {rate(env, act):.2f}")
    env._dims.xcd_piece = 0
    out.append(f"eighths again: {rate(env, act):.2f}")
    print(f"{S}x{S}, {T} tiles, {N} boards ({bps * N / 1e6:.0f} MB, {d['name']}, {d['blocks']} blocks): " + "   ".join(out), flush=True)
    del env, act
    torch.cuda.empty_cache()
