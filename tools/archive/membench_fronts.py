#!/usr/bin/env python3
"""Calibration: store rate as a function of the NUMBER OF FRONTS per CU (one-wave blocks, resident blocks bounded by
the LDS request) for large private chunks - the pattern a block with dedicated emitter waves would produce.

    python tools/membench.py build ; python tools/membench_fronts.py     (GPU box)
"""
import ctypes as C
import os
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

L = C.CDLL(os.path.join(ROOT, "build", "membench.so"))
L.mb_fill2.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
nbytes = 708 * 1000 * 1000 // 1024 * 1024
buf = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)


def lds_for(bpc):
    return 0 if bpc == 32 else 96 * 1024 if bpc == 1 else (160 * 1024 // (bpc + 1) + 16) & ~15


def rate(policy, chunk, lds, mode, wpb, reps=10):
    ts = []
    for r in range(3):
        for i in range(2):
            L.mb_fill2(buf.data_ptr(), nbytes, 1, policy, chunk, lds, mode, wpb, st)
        e0.record()
        for i in range(reps):
            L.mb_fill2(buf.data_ptr(), nbytes, 1, policy, chunk, lds, mode, wpb, st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return nbytes / statistics.median(ts) / 1e3


print("708 MB, observation-like data, XCD-contiguous one-wave blocks; GB/s by fronts (resident waves) per CU")
for policy in (1, 0):
    for chunk in (11, 22, 44, 88, 176, 352):
        row = [f"{bpc:2d}:{rate(policy, chunk, lds_for(bpc), 1, 1):5.0f}" for bpc in (1, 2, 3, 4, 5, 6, 8, 12)]
        print(f"{'nt   ' if policy else 'plain'} chunk {chunk:3d} KiB/wave  " + "  ".join(row), flush=True)
print("two-wave blocks (both stream private chunks)")
for chunk in (22, 44, 88):
    row = [f"{bpc:2d}:{rate(1, chunk, lds_for(bpc), 1, 2):5.0f}" for bpc in (1, 2, 3, 4)]
    print(f"nt    chunk {chunk:3d} KiB/wave  " + "  ".join(row), flush=True)
