#!/usr/bin/env python3
"""Calibration: PERSISTENT emitter waves (tools/membench.hip k_persist) - a fixed number of store fronts per CU for the
whole launch, units handed out by per-XCD counters - against the dispatcher-ordered one-wave blocks of membench_fronts.py.
This is the store pattern a step kernel with dedicated emitter waves (fed through LDS by compute waves) would produce.

    python tools/membench.py build ; python tools/membench_persist.py     (GPU box)
"""
import ctypes as C
import os
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

L = C.CDLL(os.path.join(ROOT, "build", "membench.so"))
L.mb_fill2.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
L.mb_persist.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
nbytes = 708 * 1000 * 1000 // 1024 * 1024
buf = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
ctr = torch.zeros(128, dtype=torch.int32, device=dev)
ncu = torch.cuda.get_device_properties(0).multi_processor_count


def timed(fn, reps=10):
    ts = []
    for r in range(3):
        for i in range(2):
            fn()
        e0.record()
        for i in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return nbytes / statistics.median(ts) / 1e3


def persist(policy, chunk, emitters, blocks, lds, from_lds):
    def go():
        rc = L.mb_persist(buf.data_ptr(), nbytes, policy, chunk, emitters, blocks, lds, from_lds, ctr.data_ptr(), st)
        assert rc == 0, rc
    return timed(go)


print(f"708 MB, {ncu} CUs; reference: one-wave blocks, 11 KiB/wave, 4 per CU (dispatcher order): "
      f"{timed(lambda: L.mb_fill2(buf.data_ptr(), nbytes, 1, 1, 11, (160 * 1024 // 5 + 16) & ~15, 1, 1, st)):5.0f} GB/s;  "
      f"7 per CU: {timed(lambda: L.mb_fill2(buf.data_ptr(), nbytes, 1, 1, 11, (160 * 1024 // 8 + 16) & ~15, 1, 1, st)):5.0f};  "
      f"dense: {timed(lambda: L.mb_fill2(buf.data_ptr(), nbytes, 1, 1, 1, 0, 1, 4, st)):5.0f}", flush=True)
# check once that a persistent launch writes every byte
buf.fill_(-1.0)
assert L.mb_persist(buf.data_ptr(), nbytes, 1, 11, 4, ncu, 96 * 1024, 0, ctr.data_ptr(), st) == 0
torch.cuda.synchronize()
assert int((buf < 0).sum()) == 0, "persistent launch left bytes unwritten"
print("persistent launch covers the buffer: ok", flush=True)
for from_lds in (0, 1):
    for policy in (1, 0):
        for chunk in (3, 6, 11, 22, 43):
            row = []
            for emitters in (2, 3, 4, 5, 6, 8):
                row.append(f"{emitters}:{persist(policy, chunk, emitters, ncu, 128 * 1024, from_lds):5.0f}")
            print(f"{'lds image' if from_lds else 'registers'} {'nt   ' if policy else 'plain'} unit {chunk:2d} KiB  one block per CU, GB/s by emitter waves  "
                  + "  ".join(row), flush=True)
print("two blocks per CU (LDS 64 KiB each), 2 emitters each")
for chunk in (6, 11, 22):
    print(f"registers nt unit {chunk:2d} KiB: {persist(1, chunk, 2, 2 * ncu, 64 * 1024, 0):5.0f}", flush=True)
