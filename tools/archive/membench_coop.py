#!/usr/bin/env python3
"""Calibration: does a block that writes its region COOPERATIVELY (all waves advance through the block's region as one
front, tools/membench.hip mode 2) reach a higher store rate than waves that each stream a private chunk (mode 1)?
The step kernels write 5-12 KiB per wave; a cooperative block would have `waves` times fewer, `waves` times wider fronts.

The store rate of the private-chunk pattern depends on the allocation (profiles/r02_placement_study.log), so the script
first allocates several buffers, rates each with cfg4's pattern (one-wave blocks, 11 KiB per wave, 7 per CU) and then
runs the table on the slowest and on the fastest of them.

    python tools/membench.py build ; python tools/membench_coop.py [buffers]     (GPU box)
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
nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def lds_for(bpc):
    return 0 if bpc == 32 else 96 * 1024 if bpc == 1 else (160 * 1024 // (bpc + 1) + 16) & ~15


def rate(buf, policy, chunk, lds, mode, wpb, reps=20):
    ts = []
    for r in range(3):
        for i in range(3):
            L.mb_fill2(buf.data_ptr(), nbytes, 1, policy, chunk, lds, mode, wpb, st)
        e0.record()
        for i in range(reps):
            L.mb_fill2(buf.data_ptr(), nbytes, 1, policy, chunk, lds, mode, wpb, st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return nbytes / statistics.median(ts) / 1e3


bufs = [torch.empty(nbytes // 4, dtype=torch.float32, device=dev) for _ in range(nbuf)]
print("708 MB, observation-like data, nontemporal stores, XCD-contiguous blocks")
rated = []
for k, b in enumerate(bufs):
    g = rate(b, 1, 11, lds_for(7), 1, 1)
    d = rate(b, 1, 1, 0, 1, 4)
    rated.append((g, k))
    print(f"buffer {k} @{b.data_ptr():#x}: cfg4's pattern (1 wave/block, 11 KiB/wave, 7/CU) {g:5.0f} GB/s   dense (1 KiB/wave) {d:5.0f} GB/s", flush=True)
rated.sort()
for tag, (g, k) in (("SLOWEST", rated[0]), ("FASTEST", rated[-1])):
    buf = bufs[k]
    print(f"--- {tag} buffer {k} ({g:.0f} GB/s); GB/s by resident blocks per CU")
    for wpb in (1, 4, 8, 16):
        for chunk in (3, 6, 11):
            for mode in (1, 2):
                if wpb == 1 and mode == 2:
                    continue
                row = []
                for bpc in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 32):
                    if (bpc * wpb > 32 and bpc != 32) or (wpb == 1 and bpc < 4):
                        continue
                    row.append(f"{bpc:2d}:{rate(buf, 1, chunk, lds_for(bpc), mode, wpb, 10):5.0f}")
                print(f"waves/block {wpb:2d}  chunk {chunk:2d} KiB/wave (block region {wpb * chunk:3d} KiB)  "
                      f"{'cooperative' if mode == 2 else 'private    '}  " + "  ".join(row), flush=True)
