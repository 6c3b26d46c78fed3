#!/usr/bin/env python3
"""Development tool: is XCD == blockIdx % 8 for the launch shapes the step kernels use?  (tools/xcc_probe.hip)

    hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o build/xcc_probe.so tools/xcc_probe.hip   (here)
    python tools/xcc_probe.py                                                                   (GPU box)
"""
import ctypes as C
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "build", "xcc_probe.so"))
L.xcc_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
# (blocks, threads, dynamic LDS, spin): cfg1 cache-resident (4-wave blocks), cfg4 / cfg2 beyond the cache (one-wave
# blocks, 7 / 4 per CU through the LDS request), a short and a long block life each
for name, blocks, threads, lds in (("cfg1-like, 4-wave blocks", 2048, 256, 24 * 1024), ("cfg4-like, one-wave blocks, 7 per CU", 65536, 64, (160 * 1024 // 8 + 16) & ~15),
                                   ("cfg2-like, one-wave blocks, 4 per CU", 32768, 64, (160 * 1024 // 5 + 16) & ~15), ("8 blocks", 8, 64, 0), ("257 blocks", 257, 64, 0)):
    for spin in (0, 2000, 20000):
        out = torch.full((blocks,), 255, dtype=torch.uint8, device=dev)
        assert L.xcc_probe(out.data_ptr(), blocks, threads, lds, spin, st) == 0
        torch.cuda.synchronize()
        x = out.cpu().numpy().astype(int)
        b = torch.arange(blocks).numpy()
        # the ids may be a permutation of 0..7 relative to blockIdx % 8: what matters is that blocks with equal
        # blockIdx % 8 share an XCD
        import numpy as np
        groups = [np.bincount(x[b % 8 == r], minlength=16) for r in range(8)]
        pure = sum(int(g.max()) for g in groups) / blocks
        mapping = [int(g.argmax()) for g in groups]
        print(f"{name:40s} spin {spin:6d}: {pure * 100:6.2f} % of the blocks run on the XCD their blockIdx % 8 group mostly runs on; group -> XCD {mapping}", flush=True)
