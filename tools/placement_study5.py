#!/usr/bin/env python3
"""Development tool: is the allocation-dependent step time a matter of PHYSICAL CONTIGUITY?  The observation buffer
(and the one-hot buffer, for configs that have one) is assembled from separately created physical chunks through
HIP's virtual-memory API (tools/vmm_alloc.hip), mapped in creation order, reversed or shuffled, for several chunk
sizes, and timed against ordinary torch allocations with the same state.

    hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o build/vmm_alloc.so tools/vmm_alloc.hip      (here)
    python tools/placement_study5.py cfg2 [instances]                                             (GPU box)
"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from tiler_slider_amd import VecTilerSliderEnv, _cabi  # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
inst = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = bench.CONFIGS[cfgname]
n = cfg["boards"]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
L = _cabi.lib()
V = C.CDLL(os.path.join(ROOT, "build", "vmm_alloc.so"))
V.vmm_granularity.restype = C.c_size_t
V.vmm_alloc.restype = C.c_void_p
V.vmm_alloc.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_uint32, C.POINTER(C.c_int)]
V.vmm_free.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
ring = []
for i in range(16):
    t = torch.empty(n, dtype=torch.uint8, device=dev)
    _cabi.check(L.ts_fill_actions(n, bench.ACTION_SEED, 0, i, t.data_ptr(), stream), "fill")
    ring.append(t)
env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"],
                               seed=bench.LEVEL_SEED, multi_color=True, max_steps=2**30, device=dev, auto_reset=True,
                               with_reward=cfg["reward"], with_onehot=cfg["onehot"])
env.reset()
obs_bytes = env._obs.numel() * 4
oh_bytes = env._onehot.numel() * 4 if env._onehot is not None else 0
want_obs = env._obs.clone()


def timed(obs_ptr, oh_ptr, reps=60):
    out = _cabi.StepOut(env._flags.data_ptr(), obs_ptr, env._reward.data_ptr() if env._reward is not None else None,
                        oh_ptr if oh_bytes else None, None, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        for i in range(3):
            L.ts_step(C.byref(env._dims), C.byref(env._state), ring[i & 15].data_ptr(), env._mode, C.byref(out), stream)
        e0.record()
        for i in range(reps):
            L.ts_step(C.byref(env._dims), C.byref(env._state), ring[i & 15].data_ptr(), env._mode, C.byref(out), stream)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts)


gran = V.vmm_granularity(0)
print(f"{cfgname}: obs {obs_bytes >> 20} MiB, one-hot {oh_bytes >> 20} MiB; VMM granularity {gran} B; "
      f"env's own buffers {timed(env._obs.data_ptr(), env._onehot.data_ptr() if oh_bytes else None):.2f} us", flush=True)
keep = []
for k in range(inst):
    o = torch.empty(obs_bytes // 4, dtype=torch.float32, device=dev)
    h = torch.empty(max(oh_bytes // 4, 1), dtype=torch.float32, device=dev)
    keep += [o, h]
    print(f"torch.empty #{k}: {timed(o.data_ptr(), h.data_ptr()):8.2f} us", flush=True)


def vmm(nbytes, chunk, order, seed):
    err = C.c_int(0)
    p = V.vmm_alloc(0, nbytes, chunk, order, seed, C.byref(err))
    if not p:
        raise RuntimeError(f"vmm_alloc failed: step {err.value >> 16} hipError {err.value & 0xffff}")
    return p


ORDER = {0: "in order", 1: "shuffled", 2: "reversed"}
CHUNKS = [int(x) << 10 for x in os.environ.get('VMM_CHUNKS_KIB', '64,2048,32768,262144').split(',')]
for chunk in CHUNKS:
    for order in (0, 1, 2):
        for k in range(inst if order == 1 else 1):
            try:
                po = vmm(obs_bytes, chunk, order, 17 + k)
                ph = vmm(oh_bytes, chunk, order, 117 + k) if oh_bytes else None
            except RuntimeError as e:
                print(f"chunks of {chunk >> 10} KiB {ORDER[order]}: {e}", flush=True)
                break
            us = timed(po, ph)
            print(f"VMM chunks of {chunk >> 10:7d} KiB, {ORDER[order]:9s} #{k}: {us:8.2f} us", flush=True)
            V.vmm_free(po, obs_bytes, chunk)
            if ph:
                V.vmm_free(ph, oh_bytes, chunk)
