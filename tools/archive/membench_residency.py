#!/usr/bin/env python3
"""Calibration: plain fill (tools/membench.hip) with blocks of ONE wave and the resident blocks per CU bounded by the
dynamic-LDS request, for the chunk sizes the step kernels write per wave.  What store bandwidth does the pattern of the
residency-bounded step kernels admit when nothing but the stores runs?

    python tools/membench.py build ; python tools/membench_residency.py      (GPU box)
"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

L = C.CDLL(os.path.join(ROOT, "build", "membench.so"))
L.mb_fill2.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
nbytes = 708 * 1000 * 1000 // 1024 * 1024
buf = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
print("708 MB, observation-like data, nontemporal stores, XCD-contiguous blocks, ONE wave per block")
for chunk in (3, 6, 11, 12, 24, 48):
    row = []
    for bpc in (2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 32):
        lds = 0 if bpc == 32 else ((160 * 1024 // (bpc + 1) + 16) & ~15)
        ts = []
        for r in range(3):
            for i in range(3):
                L.mb_fill2(buf.data_ptr(), nbytes, 1, 1, chunk, lds, 1, 1, st)
            e0.record()
            for i in range(20):
                L.mb_fill2(buf.data_ptr(), nbytes, 1, 1, chunk, lds, 1, 1, st)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        us = statistics.median(ts)
        row.append(f"{bpc:2d}:{nbytes / us / 1e3:5.0f}")
    print(f"chunk {chunk:2d} KiB/wave  GB/s by waves per CU  " + "  ".join(row), flush=True)
