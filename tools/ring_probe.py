#!/usr/bin/env python3
"""Round 5, VERDICT weak 2 / next 1c: why is the overlapped hand-off loop slower than the serial one on one GPU?

Times ts_step alone (HIP events, 200 steps, cfg1 = 1,048,576 4x4 boards) while the observation goes to a ring of 1 / 2 / 3
buffers, float32 and uint8, with the library's own classification of the launch (201 MB of output: cache-resident forms,
agent-scope stores) and with the out-of-cache forms forced (ts_tuning(TS_TUNE_NT_THRESHOLD_BYTES, 0)); plus the step that
writes no observation at all (ts_step_out.obs = NULL: what an actor rank of the compact hand-off needs).
argv: [boards]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiler_slider_amd import VecTilerSliderEnv, _cabi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda", 0)
L = _cabi.lib()
stream = torch.cuda.current_stream(dev).cuda_stream
acts = []
for i in range(8):
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    L.ts_fill_actions(n, 0xAC710005, 0, i, a.data_ptr(), stream)
    acts.append(a)


def timed(fn, steps=200, rounds=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(rounds):
        for i in range(20):
            fn(i)
        e0.record()
        for i in range(steps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / steps)
    return sorted(best)[len(best) // 2]


default_nt = L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, -1)
# clocks up
w = VecTilerSliderEnv.random(n, size=4, num_tiles=2, num_obstacles=2, seed=1, multi_color=True, max_steps=2**30, auto_reset=True)
w.reset()
for i in range(3000):
    w.step_async(acts[i & 7])
torch.cuda.synchronize()
del w
for dtype in ("float32", "uint8"):
    for bufs in (1, 2, 3):
        for mem in ("contiguous", "torch"):
            env = VecTilerSliderEnv.random(n, size=4, num_tiles=2, num_obstacles=2, seed=0x715311DE, multi_color=True, max_steps=2**30,
                                           auto_reset=True, obs_dtype=dtype, obs_buffers=bufs, output_memory=mem)
            env.reset()
            row = []
            for nt in (default_nt, 0):
                L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, nt)
                row.append(timed(lambda i: env.step_async(acts[i & 7])))
            L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, default_nt)
            print(f"obs {dtype:8s} ring of {bufs}  memory {mem:10s}  library classification {row[0]:7.2f} us   out-of-cache forms forced {row[1]:7.2f} us", flush=True)
            del env
            torch.cuda.empty_cache()

# no observation at all: obs = NULL in ts_step_out
env = VecTilerSliderEnv.random(n, size=4, num_tiles=2, num_obstacles=2, seed=0x715311DE, multi_color=True, max_steps=2**30, auto_reset=True)
env.reset()
out = _cabi.StepOut(env._flags.data_ptr(), None, None, None, None, None, None)
rw = torch.zeros(n, dtype=torch.int32, device=dev)
out_r = _cabi.StepOut(env._flags.data_ptr(), None, rw.data_ptr(), None, None, None, None)
for name, o in (("flags only", out), ("flags + reward", out_r)):
    us = timed(lambda i: L.ts_step(C.byref(env._dims), C.byref(env._state), acts[i & 7].data_ptr(), _cabi.MODE_AUTORESET, C.byref(o), stream))
    print(f"step without observation ({name}): {us:7.2f} us", flush=True)
