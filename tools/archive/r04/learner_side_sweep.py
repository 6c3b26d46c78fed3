#!/usr/bin/env python3
"""Round 4: ts_encode over many gathered 4x4 boards (cfg3's learner side) by boards per wave and resident blocks."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiler_slider_amd import VecTilerSliderEnv, _cabi
from tiler_slider_amd.vec_env import _contiguous_zeros

L = _cabi.lib()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
stream = torch.cuda.current_stream().cuda_stream
print("boards | us per ts_encode at (boards per wave: launch_hint -4 / 0 / +4)  policy first")
for n in (1 << 21, 1 << 22, 1 << 23):
    env = VecTilerSliderEnv.random(n, size=4, num_tiles=2, num_obstacles=2, seed=1, multi_color=True, max_steps=2**30, obs_dtype="uint8")
    env.reset()
    out = _contiguous_zeros((n, 4, 4, 3), torch.float32, env.device)
    d, st = env._dims, env._state
    def rate():
        for i in range(40):
            L.ts_encode(C.byref(d), C.byref(st), out.data_ptr(), stream)
        e0.record()
        for i in range(20):
            L.ts_encode(C.byref(d), C.byref(st), out.data_ptr(), stream)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    row = f"{n:8d} | policy {rate():6.1f} |"
    for bpw in (64, 32):
        L.ts_tuning(_cabi.TUNE_SMALL_BPW, bpw)
        row += f" {bpw}:"
        for h in (-4, 0, 4):
            d.launch_hint = h
            row += f" {rate():6.1f}"
        d.launch_hint = 0
    L.ts_tuning(_cabi.TUNE_SMALL_BPW, 0)
    print(row, flush=True)
    del env, out
    torch.cuda.empty_cache()
