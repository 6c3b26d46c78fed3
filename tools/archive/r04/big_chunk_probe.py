#!/usr/bin/env python3
"""Round 4: store-only probe (tools/align_probe.hip) for the chunk sizes of the large-board kernel - how fast can one-wave
blocks write private chunks of 12 .. 49 KB at all, by resident blocks per CU, on physically contiguous memory?
mode 0 = all nontemporal, 9 = whole-line instructions + shared pieces write-back (what the kernels do), 3 = first / last
instruction write-back (edge stores)."""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiler_slider_amd.vec_env import _contiguous_zeros

L = C.CDLL(os.path.join(ROOT, "build", "align_probe.so"))
L.ap_fill2.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
nbytes = 600 << 20
buf = _contiguous_zeros((nbytes,), torch.uint8, torch.device("cuda", 0))
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
lds_for = lambda b: ((160 * 1024 // (b + 1)) + 16) & ~15


def rate(chunk, mode, bpc, piece):
    for i in range(30):
        L.ap_fill2(buf.data_ptr(), nbytes, chunk, mode, lds_for(bpc), piece, st)
    e0.record()
    for i in range(20):
        L.ap_fill2(buf.data_ptr(), nbytes, chunk, mode, lds_for(bpc), piece, st)
    e1.record()
    torch.cuda.synchronize()
    return nbytes / (e0.elapsed_time(e1) / 20) / 1e9


print("chunk B | blocks/CU: TB/s for (mode 0, mode 9, mode 3) with eighths, then mode 3 with pieces of 16")
for chunk in (6144, 9600, 12288, 19200, 27648, 49152, 24576, 98304):
    row = f"{chunk:7d} |"
    for bpc in (4, 6, 8, 12, 18):
        row += f"  {bpc:2d}:" + "".join(f" {rate(chunk, m, bpc, 0):5.2f}" for m in (0, 9, 3)) + f" {rate(chunk, 3, bpc, 16):5.2f}"
    print(row, flush=True)
