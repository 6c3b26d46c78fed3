#!/usr/bin/env python3
"""Development tool: step time vs the position of the observation buffer inside ONE large arena
(is the placement effect periodic in the address, i.e. a function of physical address bits?).

    python tools/placement_study3.py cfg4 [arena_GiB] [stride_MiB]
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

cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
arena_gib = int(sys.argv[2]) if len(sys.argv) > 2 else 12
stride_mib = int(sys.argv[3]) if len(sys.argv) > 3 else 256
cfg = dict(bench.CONFIGS[cfgname])
if len(sys.argv) > 4:
    cfg["boards"] = int(sys.argv[4])
n = cfg["boards"]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
L = _cabi.lib()
ring = []
for i in range(16):
    t = torch.empty(n, dtype=torch.uint8, device=dev)
    _cabi.check(L.ts_fill_actions(n, bench.ACTION_SEED, 0, i, t.data_ptr(), stream), "fill")
    ring.append(t)
env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"],
                               seed=bench.LEVEL_SEED, multi_color=True, max_steps=2**30, device=dev, auto_reset=True)
env.reset()


def timed(obs_ptr, reps=60):
    out = _cabi.StepOut(env._flags.data_ptr(), obs_ptr, None, None, None, None)
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


nbytes = env._obs.numel() * 4
print(f"{cfgname}: {n} boards, obs {nbytes / 2**20:.0f} MiB; own buffer: {timed(env._obs.data_ptr()):.2f} us", flush=True)
arena = torch.empty(arena_gib << 30, dtype=torch.uint8, device=dev)
print(f"arena {arena_gib} GiB @ {arena.data_ptr():#x}", flush=True)
off = 0
res = []
while off + nbytes <= arena.numel():
    us = timed(arena.data_ptr() + off)
    res.append((off, us))
    print(f"  obs at +{off >> 20:6d} MiB: {us:8.2f} us", flush=True)
    off += stride_mib << 20
best = min(res, key=lambda r: r[1])
print(f"best +{best[0] >> 20} MiB {best[1]:.2f} us; worst {max(r[1] for r in res):.2f} us", flush=True)
# finer sweep around the first fast and the first slow offset
fast = [o for o, u in res if u < best[1] * 1.03]
slow = [o for o, u in res if u > best[1] * 1.08]
for label, lst in (("fast", fast), ("slow", slow)):
    if not lst:
        continue
    base = lst[0]
    for d in (2, 4, 8, 16, 32, 64, 128):
        o = base + (d << 20)
        if o + nbytes <= arena.numel():
            print(f"  {label} base +{base >> 20} MiB shifted by {d:4d} MiB: {timed(arena.data_ptr() + o):8.2f} us", flush=True)
