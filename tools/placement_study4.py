#!/usr/bin/env python3
"""Development tool: does the KIND of allocation decide the step time?  The observation buffer comes from
(a) torch's allocator (hipMalloc), (b) hipMalloc directly, (c) hipExtMallocWithFlags(fine-grained),
(d) hipExtMallocWithFlags(uncached), (e) hipMallocManaged — several instances each, timed with the same state.

    python tools/placement_study4.py cfg4 [instances]
"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from tiler_slider_amd import VecTilerSliderEnv, _cabi  # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
inst = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = bench.CONFIGS[cfgname]
n = cfg["boards"]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
L = _cabi.lib()
hip = C.CDLL("libamdhip64.so")
ring = []
for i in range(16):
    t = torch.empty(n, dtype=torch.uint8, device=dev)
    _cabi.check(L.ts_fill_actions(n, bench.ACTION_SEED, 0, i, t.data_ptr(), stream), "fill")
    ring.append(t)
env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"],
                               seed=bench.LEVEL_SEED, multi_color=True, max_steps=2**30, device=dev, auto_reset=True)
env.reset()
nbytes = env._obs.numel() * 4


def timed(obs_ptr, reps=60):
    out = _cabi.StepOut(env._flags.data_ptr(), obs_ptr, None, None, None, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        for i in range(3):
            L.ts_step(C.byref(env._dims), C.byref(env._state), ring[i & 15].data_ptr(), env._mode, C.byref(out), stream)
        e0.record()
        for i in range(reps):
            L.ts_step(C.byref(env._dims), C.byref(env._state), ring[i & 15].data_ptr(), env._mode, C.byref(out), stream)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts)


def hip_alloc(kind):
    p = C.c_void_p()
    if kind == "hipMalloc":
        rc = hip.hipMalloc(C.byref(p), C.c_size_t(nbytes))
    elif kind == "finegrained":
        rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(nbytes), C.c_uint(0x1))   # hipDeviceMallocFinegrained
    elif kind == "uncached":
        rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(nbytes), C.c_uint(0x3))   # hipDeviceMallocUncached
    elif kind == "managed":
        rc = hip.hipMallocManaged(C.byref(p), C.c_size_t(nbytes), C.c_uint(1))
    else:
        raise ValueError(kind)
    return rc, p.value


print(f"{cfgname}: obs {nbytes >> 20} MiB; env's own buffer {timed(env._obs.data_ptr()):.2f} us", flush=True)
keep = []
for k in range(inst):
    t = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    keep.append(t)
    print(f"torch.empty #{k}: {timed(t.data_ptr()):8.2f} us  @{t.data_ptr():#x}", flush=True)
for kind in ("hipMalloc", "finegrained", "uncached", "managed"):
    for k in range(inst):
        rc, p = hip_alloc(kind)
        if rc != 0 or not p:
            print(f"{kind} #{k}: allocation failed rc={rc}", flush=True)
            break
        if kind == "managed":
            hip.hipMemPrefetchAsync(C.c_void_p(p), C.c_size_t(nbytes), C.c_int(0), C.c_void_p(stream))
            torch.cuda.synchronize()
        print(f"{kind} #{k}: {timed(p):8.2f} us  @{p:#x}", flush=True)
