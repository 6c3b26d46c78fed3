#!/usr/bin/env python3
"""Round 4: time ts_generate_mt19937 (the reference's own seed -> level map, "scramble") and ts_reset at a million boards."""
import os, sys, statistics, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiler_slider_amd import VecTilerSliderEnv, _cabi

L = _cabi.lib()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for S, T, K, n in ((4, 2, 2, 1 << 20), (5, 2, 3, 1 << 20), (8, 20, 10, 1 << 19), (15, 32, 24, 1 << 18), (15, 32, 24, 1 << 20), (18, 8, 20, 1 << 18), (20, 8, 20, 1 << 17), (24, 8, 20, 1 << 16), (32, 64, 100, 1 << 15), (32, 64, 100, 1 << 18)):
    seeds = torch.arange(n, dtype=torch.int64).to(torch.int32).cuda()
    env = VecTilerSliderEnv.from_seeds(torch.arange(64), size=S, num_tiles=T, num_obstacles=K, multi_color=True)  # warm
    dt = torch.uint8 if S <= 16 else torch.int16
    blk = torch.zeros(((S * S + 31) // 32, n), dtype=torch.int32, device="cuda")
    init = torch.zeros((T, n), dtype=dt, device="cuda")
    tgt = torch.zeros((T, n), dtype=dt, device="cuda")
    dims = _cabi.Dims(n, S, T, T, 1, 100, 0)
    st = _cabi.State(None, init.data_ptr(), tgt.data_ptr(), blk.data_ptr(), None, None, None)
    stream = torch.cuda.current_stream().cuda_stream
    ts = []
    for r in range(4):
        e0.record()
        _cabi.check(L.ts_generate_mt19937(C.byref(dims), C.byref(st), seeds.data_ptr(), K, stream), "gen")
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"ts_generate_mt19937 {S}x{S} T={T} K={K} {n} seeds: {statistics.median(ts[1:]):10.1f} us  ({n / statistics.median(ts[1:]) :.1f} levels/us)", flush=True)
