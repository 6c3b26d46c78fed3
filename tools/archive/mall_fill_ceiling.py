#!/usr/bin/env python3
"""Development tool (round 3): how fast can ANY store kernel fill a cache-resident buffer of cfg1's observation size
(201 MB) - the floor under cfg1's 29.9 us?  tools/membench.hip fills with observation-like data; policy sc1 (what the step
kernels use in cache), plain and nontemporal; chunk per wave, waves per block, block -> address mapping."""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

L = C.CDLL(os.path.join(ROOT, "build", "membench.so"))
L.mb_fill2.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
nbytes = 1048576 * 192
buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
POL = {0: "plain", 1: "nt", 2: "sc1"}
print(f"{nbytes / 1e6:.0f} MB, observation-like payload; us per fill (TB/s)")
for pol in (2, 0, 1):
    for mode in (0, 1, 2):
        for wpb in (1, 4, 16):
            if mode == 2 and wpb == 1:
                continue
            row = f"{POL[pol]:5s} mode {mode} waves/block {wpb:2d}:"
            for chunk in (1, 2, 4, 6, 12, 24):
                ts = []
                for r in range(3):
                    for i in range(3):
                        L.mb_fill2(buf.data_ptr(), nbytes, 1, pol, chunk, 0, mode, wpb, st)
                    e0.record()
                    for i in range(30):
                        L.mb_fill2(buf.data_ptr(), nbytes, 1, pol, chunk, 0, mode, wpb, st)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 30 * 1e3)
                us = statistics.median(ts)
                row += f"  {chunk:2d}K {us:5.1f} ({nbytes / us / 1e6:4.2f})"
            print(row, flush=True)
