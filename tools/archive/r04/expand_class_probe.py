#!/usr/bin/env python3
"""Round 4 experiment: ts_expand_u8 reads one large stream and writes another.  Does its time depend on which physically
contiguous block the output (or the input) lies in?  8,388,608 4x4 boards: 403 MB in, 1.6 GB out."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiler_slider_amd import _cabi
from tiler_slider_amd.vec_env import _contiguous_zeros

L = _cabi.lib()
dev = torch.device("cuda", 0)
n = 1 << 23
count = n * 48
stream = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
srcs = [_contiguous_zeros((count,), torch.uint8, dev) for _ in range(4)]
outs = [_contiguous_zeros((count,), torch.float32, dev) for _ in range(10)]
tsrc = torch.zeros(count, dtype=torch.uint8, device=dev)
tout = torch.empty(count, dtype=torch.float32, device=dev)


def rate(s, o):
    for i in range(12):
        L.ts_expand_u8(s.data_ptr(), o.data_ptr(), count, stream)
    e0.record()
    for i in range(10):
        L.ts_expand_u8(s.data_ptr(), o.data_ptr(), count, stream)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


rate(srcs[0], outs[0])
print("rows: input buffer (4 contiguous, then torch); columns: output buffer (10 contiguous, then torch); us per ts_expand_u8")
for s in srcs + [tsrc]:
    print(" ".join(f"{rate(s, o):6.1f}" for o in outs + [tout]), flush=True)
