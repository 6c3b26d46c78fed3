#!/usr/bin/env python3
"""Development tool (round 3, time-boxed): what separates a "fast" from a "slow" output buffer?  The store-only probe
(tools/align_probe.hip, one-wave blocks streaming private 10.5 KiB chunks, nontemporal) sees the same two speeds as the
step kernels, so it can rate many allocations cheaply: torch's caching allocator, hipMalloc directly, offsets inside one
large slab, and re-rating after other allocations were freed."""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

L = C.CDLL(os.path.join(ROOT, "build", "align_probe.so"))
L.ap_fill.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemGetAddressRange.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
NB = 708 << 20
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
LDS10 = ((160 * 1024 // 11) + 16) & ~15


def rate(ptr, chunk=10752, mode=0):
    ts = []
    for r in range(3):
        for i in range(2):
            L.ap_fill(ptr, NB, chunk, mode, LDS10, st)
        e0.record()
        for i in range(10):
            L.ap_fill(ptr, NB, chunk, mode, LDS10, st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    return NB / statistics.median(ts) / 1e9


def base_of(ptr):
    b, n = C.c_void_p(), C.c_size_t()
    rc = hip.hipMemGetAddressRange(C.byref(b), C.byref(n), C.c_void_p(ptr))
    return (b.value or 0, n.value) if rc == 0 else (0, 0)


def show(tag, ptr):
    b, n = base_of(ptr)
    print(f"  {tag:28s} ptr {ptr:#016x}  %2MiB {ptr % (2 << 20):#09x}  %1GiB {ptr % (1 << 30):#011x}  range base {b:#016x} size {n / 2**20:9.1f} MiB  "
          f"chunked {rate(ptr):5.2f} TB/s  dense front {rate(ptr, 1024):5.2f} TB/s", flush=True)


hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
if len(sys.argv) > 1 and sys.argv[1] == "pieces":
    # does some block -> address mapping make a physically contiguous buffer (where the mapping's physical offsets are the
    # virtual ones) as fast as a lucky ordinary allocation?  pieces of P chunks per XCD round-robin; 0 = one eighth each
    L.ap_fill2.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]

    def rate2(ptr, piece, chunk=10752):
        ts = []
        for r in range(3):
            for i in range(2):
                L.ap_fill2(ptr, NB, chunk, 0, LDS10, piece, st)
            e0.record()
            for i in range(10):
                L.ap_fill2(ptr, NB, chunk, 0, LDS10, piece, st)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        return NB / statistics.median(ts) / 1e9

    bufs = []
    p = C.c_void_p()
    assert hip.hipExtMallocWithFlags(C.byref(p), NB, 4) == 0
    bufs.append(("contiguous", p.value))
    for i in range(3):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), NB) == 0
        bufs.append((f"hipMalloc #{i}", p.value))
    pieces = (-1, 0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096)
    print("TB/s by XCD piece (chunks of 10.5 KiB; -1 = blockIdx order, 0 = one contiguous eighth per XCD):", pieces)
    for tag, ptr in bufs:
        print(f"  {tag:14s}", " ".join(f"{rate2(ptr, pc):5.2f}" for pc in pieces), flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "contiguous":
    # hipDeviceMallocContiguous (0x4): physically contiguous device memory
    for rnd in range(2):
        print(f"round {rnd}: alternating hipMalloc / hipExtMallocWithFlags(hipDeviceMallocContiguous), 708 MiB each:")
        held = []
        for i in range(6):
            for flag, tag in ((None, "hipMalloc"), (4, "contiguous")):
                p = C.c_void_p()
                rc = hip.hipMalloc(C.byref(p), NB) if flag is None else hip.hipExtMallocWithFlags(C.byref(p), NB, flag)
                if rc != 0:
                    print(f"  {tag} #{i}: error {rc}")
                    continue
                held.append(p.value)
                show(f"{tag} #{i}", p.value)
        for p in held:
            hip.hipFree(C.c_void_p(p))
    sys.exit(0)
print("torch caching allocator, 10 buffers of 708 MiB:")
bufs = [torch.empty(NB, dtype=torch.uint8, device="cuda") for _ in range(10)]
for i, b in enumerate(bufs):
    show(f"torch #{i}", b.data_ptr())
print("the same buffers again (is the speed a stable property of the buffer?):")
for i, b in enumerate(bufs[:4]):
    show(f"torch #{i} again", b.data_ptr())
del bufs
torch.cuda.empty_cache()
print("hipMalloc directly, 8 buffers of 708 MiB:")
raw = []
for i in range(8):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), NB) == 0
    raw.append(p.value)
    show(f"hipMalloc #{i}", p.value)
for p in raw:
    hip.hipFree(C.c_void_p(p))
print("one hipMalloc slab of 8 x 708 MiB, offsets inside it:")
p = C.c_void_p()
assert hip.hipMalloc(C.byref(p), 8 * NB) == 0
for i in range(8):
    show(f"slab + {i} x 708 MiB", p.value + i * NB)
show("slab + 708 MiB + 4 KiB", p.value + NB + 4096)
show("slab + 708 MiB + 64 KiB", p.value + NB + 65536)
show("slab + 708 MiB + 1 MiB", p.value + NB + (1 << 20))
hip.hipFree(p)
print("again 6 torch buffers after everything was freed:")
bufs = [torch.empty(NB, dtype=torch.uint8, device="cuda") for _ in range(6)]
for i, b in enumerate(bufs):
    show(f"torch #{i}", b.data_ptr())
