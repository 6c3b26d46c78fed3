#!/usr/bin/env python3
"""Diagnostic: dynamic-LDS sizes a launch on this device honours (see tools/lds_probe.hip).

    python tools/lds_probe.py build      # here (CPU box): hipcc cross-compiles
    python tools/lds_probe.py run        # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "build", "lds_probe.so")

if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO,
                    os.path.join(ROOT, "tools", "lds_probe.hip")], check=True)
    print("built", SO)
    sys.exit(0)

import torch  # noqa: E402

L = C.CDLL(SO)
L.probe_run.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
dev = torch.device("cuda", 0)
counter = torch.zeros(1, dtype=torch.int64, device=dev)
a, b, c = C.c_int(), C.c_int(), C.c_int()
L.probe_device(C.byref(a), C.byref(b), C.byref(c))
print(f"device: maxSharedMemoryPerBlock={a.value} B, maxSharedMemoryPerMultiprocessor={b.value} B, CUs={c.value}", flush=True)


def func_attr():
    m, s, r = C.c_int(), C.c_int(), C.c_int()
    rc = L.probe_func_attr(C.byref(m), C.byref(s), C.byref(r))
    return f"hipFuncGetAttributes rc={rc} maxDynamicSharedSizeBytes={m.value} sharedSizeBytes={s.value} numRegs={r.value}"


print("before any hipFuncSetAttribute:", func_attr(), flush=True)
BLOCKS, ROUNDS, SPIN = 16384, 6, 400
KIB = 1024
sizes = [32 * KIB, 60 * KIB, 64 * KIB - 64, 64 * KIB, 64 * KIB + 64, 80 * KIB, 96 * KIB, 128 * KIB, 160 * KIB, 160 * KIB + 64]
print("--- pass 1: plain launches, attribute untouched", flush=True)
for sz in sizes:
    n = C.c_ulonglong(0)
    rc = L.probe_run(counter.data_ptr(), BLOCKS, sz, ROUNDS, SPIN, C.byref(n))
    print(f"dynamic LDS {sz:7d} B/block: launch rc={rc} mismatched words={n.value if rc == 0 else 'n/a'}", flush=True)
    if rc >= 1000:
        print("device error after launch; stopping", flush=True)
        sys.exit(1)
print("--- pass 2: hipFuncSetAttribute(MaxDynamicSharedMemorySize = request) before each launch", flush=True)
for sz in sizes:
    rs = L.probe_set_max_dynamic(sz)
    n = C.c_ulonglong(0)
    rc = L.probe_run(counter.data_ptr(), BLOCKS, sz, ROUNDS, SPIN, C.byref(n))
    print(f"dynamic LDS {sz:7d} B/block: setAttribute rc={rs}; launch rc={rc} mismatched words={n.value if rc == 0 else 'n/a'}; "
          f"{func_attr()}", flush=True)
    if rc >= 1000:
        print("device error after launch; stopping", flush=True)
        sys.exit(1)
