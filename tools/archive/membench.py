#!/usr/bin/env python3
"""Calibration (development tool): streaming-store bandwidth of a trivial kernel as a function of
cache policy, bytes per wave, occupancy and block->address mapping.  `build` here (ships with the
snapshot), `run` on the GPU box."""
import ctypes as C
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "build", "membench.so")
POLICY = {0: "plain", 1: "nt", 2: "sc1", 3: "sc0sc1"}

if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO,
                    os.path.join(ROOT, "tools", "membench.hip")], check=True)
    print("built", SO)
else:
    import torch
    L = C.CDLL(SO)
    L.mb_fill.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [708]
    L.mb_fill2.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    for mb in sizes:
        nbytes = mb * 1000 * 1000 // 1024 * 1024
        buf = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        for mode in (0, 1):
            for wpb in (1, 2, 4):
                for chunk in (1, 2, 3, 4, 6, 12):
                    ts = []
                    for r in range(4):
                        for i in range(3):
                            L.mb_fill2(buf.data_ptr(), nbytes, 1, 0, chunk, 0, mode, wpb, st)
                        e0.record()
                        for i in range(20):
                            L.mb_fill2(buf.data_ptr(), nbytes, 1, 0, chunk, 0, mode, wpb, st)
                        e1.record()
                        torch.cuda.synchronize()
                        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
                    us = statistics.median(ts)
                    print(f"{mb:4d} MB plain {'xcd-contig' if mode else 'blockIdx  '} waves/block={wpb} chunk={chunk:2d} KiB/wave "
                          f"(block region {wpb * chunk:2d} KiB): {us:8.2f} us  {nbytes / us / 1e3:8.1f} GB/s", flush=True)
