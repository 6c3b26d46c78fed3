#!/usr/bin/env python3
"""Development tool (round 3): tools/align_probe.hip driver.  `build` here, `run` on the GPU box."""
import ctypes as C
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "build", "align_probe.so")
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO, os.path.join(ROOT, "tools", "align_probe.hip")])
    sys.exit(0)
import torch  # noqa: E402

L = C.CDLL(SO)
L.ap_fill.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
nbytes = 708 << 20
bufs = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def lds_for(blocks_per_cu):
    return ((160 * 1024 // (blocks_per_cu + 1)) + 16) & ~15


MODES = tuple(int(x) for x in os.environ.get("AP_MODES", "0,1,2").split(","))
print(f"chunk B  (%128)  blocks/CU |  TB/s for modes {MODES} (see align_probe.hip)   (two buffers)")
for chunk in tuple(int(x) for x in os.environ.get("AP_CHUNKS", "3840,3888,3968,4096,7776,15552,15616,10752,10800,10880,9408,9472,4800,4864,21600,21632").split(",")):
    for bpc in (7, 10):
        row = f"{chunk:7d} ({chunk % 128:4d}) {bpc:9d}   |"
        for buf in bufs:
            for mode in MODES:
                ts = []
                for r in range(5):
                    for i in range(2):
                        L.ap_fill(buf.data_ptr(), nbytes, chunk, mode, lds_for(bpc), st)
                    e0.record()
                    for i in range(10):
                        L.ap_fill(buf.data_ptr(), nbytes, chunk, mode, lds_for(bpc), st)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 10)
                row += f" {nbytes / statistics.median(ts) / 1e9:6.2f}"
            row += "  |"
        print(row, flush=True)
