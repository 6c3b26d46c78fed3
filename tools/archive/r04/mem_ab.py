#!/usr/bin/env python3
"""Round 4: the same shape on output buffers from torch's caching allocator and from physically contiguous memory, both
environments alive in one process, rated alternately (three rounds each, long warm-up), ~500 MB batches."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def rate(env, act, reps=40, warm=60):
    for i in range(warm):
        env.step_async(act[i & 3])
    e0.record()
    for i in range(reps):
        env.step_async(act[i & 3])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


SHAPES = [(5, 2, 3), (8, 4, 8), (9, 4, 9), (11, 6, 8), (12, 8, 16), (13, 3, 10), (15, 32, 24), (16, 16, 24), (20, 1, 1), (24, 30, 60), (32, 4, 100)]
print("   S    T    K    boards | torch allocator (three rounds) | contiguous (three rounds) | contiguous with xcd_piece 1 / 16 / 64, launch_hint -4 / +4")
for S, T, K in SHAPES:
    n = (500_000_000 // (12 * S * S)) // 256 * 256
    envs = {}
    for mem in ("torch", "contiguous"):
        envs[mem] = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True,
                                             output_memory=mem)
        envs[mem].reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    res = {"torch": [], "contiguous": []}
    for r in range(3):
        for mem in ("torch", "contiguous"):
            res[mem].append(rate(envs[mem], act))
    row = f"{S:4d} {T:4d} {K:4d} {n:9d} |" + "".join(f" {u:6.1f}" for u in res["torch"]) + "  |" + "".join(f" {u:6.1f}" for u in res["contiguous"]) + "  |"
    c = envs["contiguous"]
    for p in (1, 16, 64):
        c._dims.xcd_piece = p
        row += f" {rate(c, act):6.1f}"
    c._dims.xcd_piece = 0
    for h in (-4, 4):
        c._dims.launch_hint = h
        row += f" {rate(c, act):6.1f}"
    print(row, flush=True)
    del envs, act, c
    torch.cuda.empty_cache()
