#!/usr/bin/env python3
"""Round 4: boards per wave of the large-board kernel at its policy's lanes per board (ts_tuning TS_TUNE_LINES_BPW) for the
shapes whose chunk falls outside the 9-14 KB window, on contiguous memory, 600 MB batches."""
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


print("   S    T    K    boards | policy us (frac) | per boards-per-wave: us at launch_hint -4 / 0 / +4")
for S, T, K, bpws in ((9, 1, 9, (8, 12, 16)), (9, 4, 9, (8, 12, 16)), (10, 5, 10, (4, 8)), (11, 6, 8, (4, 8)), (12, 8, 16, (4, 6, 8)), (13, 3, 10, (4, 8)), (13, 12, 10, (4, 8)), (14, 6, 20, (2, 4)),
                      (15, 8, 24, (4,)), (17, 4, 10, (2, 4)), (18, 4, 10, (2, 4)), (19, 4, 10, (4,)), (20, 10, 40, (1, 2)), (22, 6, 40, (1, 2)), (26, 6, 60, (1, 2))):
    n = (600_000_000 // (12 * S * S)) // 256 * 256
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    rate(env, act)
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
    base = rate(env, act)
    row = f"{S:4d} {T:4d} {K:4d} {n:9d} | {base:6.1f} ({bps * n / base / 1e3 / 8000:.3f}) |"
    for bpw in bpws:
        L.ts_tuning(_cabi.TUNE_LINES_BPW, bpw)
        row += f" {bpw}:"
        for h in (-4, 0, 4):
            env._dims.launch_hint = h
            row += f" {rate(env, act):6.1f}"
        row += " |"
    L.ts_tuning(_cabi.TUNE_LINES_BPW, 0)
    env._dims.launch_hint = 0
    print(row, flush=True)
    del env, act
    torch.cuda.empty_cache()
