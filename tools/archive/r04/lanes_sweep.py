#!/usr/bin/env python3
"""Round 4: lanes per board of the large-board kernel (ts_dims.lines_lanes) by shape on physically contiguous output buffers,
600 MB batches, at launch_hint -2 / 0 / +4 each (the residency policy keys on the wave's chunk, which the lanes decide)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv

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


SHAPES = [(9, 1, 9), (9, 4, 9), (10, 5, 10), (11, 6, 8), (12, 8, 16), (12, 3, 16), (13, 3, 10), (14, 20, 20), (14, 6, 20), (15, 32, 24), (15, 8, 24), (16, 4, 24), (16, 16, 24),
          (18, 4, 10), (20, 1, 1), (20, 10, 40), (24, 4, 60), (24, 30, 60), (32, 4, 100), (32, 32, 100)]
print("   S    T    K    boards | us per step: policy | 4 lanes at hint -2 / 0 / +4 | 8 lanes | 16 lanes   (a form that does not exist for the shape repeats the policy's)")
for S, T, K in SHAPES:
    n = (600_000_000 // (12 * S * S)) // 256 * 256
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    rate(env, act)
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
    base = rate(env, act)
    row = f"{S:4d} {T:4d} {K:4d} {n:9d} | {base:6.1f} ({bps * n / base / 1e3 / 8000:.3f}) |"
    for lanes in (4, 8, 16):
        env._dims.lines_lanes = lanes
        for h in (-2, 0, 4):
            env._dims.launch_hint = h
            row += f" {rate(env, act):6.1f}"
        row += " |"
    env._dims.lines_lanes = env._dims.launch_hint = 0
    print(row, flush=True)
    del env, act
    torch.cuda.empty_cache()
