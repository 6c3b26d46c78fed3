#!/usr/bin/env python3
"""Round 4: boards per wave of the large-board kernel (ts_tuning TS_TUNE_LINES_BPW; the remaining lanes idle, the wave's
contiguous chunk of output shrinks) x launch_hint on physically contiguous output buffers, 600 MB batches.  The store-only
probe (profiles/r04_big_chunk_probe.log) writes 49,152-B private chunks at 5.8 TB/s at best and 12,288-B chunks at 7.5."""
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


SHAPES = [(14, 20, 20), (15, 32, 24), (16, 4, 24), (16, 16, 24), (18, 4, 10), (20, 1, 1), (20, 10, 40), (24, 4, 60), (24, 30, 60), (28, 8, 60), (32, 4, 100), (32, 32, 100), (32, 100, 100)]
HINTS = (-2, 0, 2, 4, 8)
print(f"   S    T    K    boards | policy us (frac) | boards per wave 1: us at launch_hint {HINTS} | 2: ... | 4: ...")
for S, T, K in SHAPES:
    n = (600_000_000 // (12 * S * S)) // 256 * 256
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    rate(env, act)
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
    base = rate(env, act)
    row = f"{S:4d} {T:4d} {K:4d} {n:9d} | {base:6.1f} ({bps * n / base / 1e3 / 8000:.3f}) |"
    best = base
    for bpw in (1, 2, 4):
        L.ts_tuning(_cabi.TUNE_LINES_BPW, bpw)
        row += f" {bpw}:"
        for h in HINTS:
            env._dims.launch_hint = h
            us = rate(env, act)
            best = min(best, us)
            row += f" {us:6.1f}"
        row += " |"
    L.ts_tuning(_cabi.TUNE_LINES_BPW, 0)
    env._dims.launch_hint = 0
    print(row + f" best {bps * n / best / 1e3 / 8000:.3f}", flush=True)
    del env, act
    torch.cuda.empty_cache()
