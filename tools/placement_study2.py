#!/usr/bin/env python3
"""Development tool, follow-up of placement_study.py: WHICH buffer's placement decides the step time?

Several environments of one config are built in one process (different addresses each).  Then the
fastest and the slowest are crossed: state of one with the observation buffer of the other, and
the observation buffer is shifted inside a larger allocation by a few offsets.

    python tools/placement_study2.py cfg4 [trials]
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
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = bench.CONFIGS[cfgname]
n = cfg["boards"]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
L = _cabi.lib()
ring = []
for i in range(16):
    t = torch.empty(n, dtype=torch.uint8, device=dev)
    _cabi.check(L.ts_fill_actions(n, bench.ACTION_SEED, 0, i, t.data_ptr(), stream), "fill")
    ring.append(t)


def timed_step(env, out, reps=150):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        for i in range(5):
            L.ts_step(C.byref(env._dims), C.byref(env._state), ring[i & 15].data_ptr(), env._mode, C.byref(out), stream)
        e0.record()
        for i in range(reps):
            L.ts_step(C.byref(env._dims), C.byref(env._state), ring[i & 15].data_ptr(), env._mode, C.byref(out), stream)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts)


def make():
    env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"],
                                   seed=bench.LEVEL_SEED, multi_color=True, max_steps=2**30, device=dev, auto_reset=True)
    env.reset()
    return env


def out_with_obs(env, obs_ptr):
    return _cabi.StepOut(env._flags.data_ptr(), obs_ptr, None, None, None, None)


envs = [make() for _ in range(trials)]
times = []
for k, env in enumerate(envs):
    us = timed_step(env, env._out)
    times.append(us)
    print(f"env {k}: {us:8.2f} us   obs@{env._obs.data_ptr():#x}  pos@{env._pos.data_ptr():#x}  lines@{(env._lines.data_ptr() if env._lines is not None else 0):#x}", flush=True)
f, s = times.index(min(times)), times.index(max(times))
print(f"fastest env {f} ({times[f]:.2f} us), slowest env {s} ({times[s]:.2f} us)", flush=True)
print(f"state of {f} + obs of {s}: {timed_step(envs[f], out_with_obs(envs[f], envs[s]._obs.data_ptr())):8.2f} us", flush=True)
print(f"state of {s} + obs of {f}: {timed_step(envs[s], out_with_obs(envs[s], envs[f]._obs.data_ptr())):8.2f} us", flush=True)
print(f"state of {f} + obs of {f}: {timed_step(envs[f], envs[f]._out):8.2f} us (control)", flush=True)
print(f"state of {s} + obs of {s}: {timed_step(envs[s], envs[s]._out):8.2f} us (control)", flush=True)

# the observation buffer shifted inside one larger allocation
nbytes = envs[0]._obs.numel() * 4
big = torch.empty(nbytes + (64 << 20), dtype=torch.uint8, device=dev)
print(f"one allocation of {big.numel() >> 20} MiB @ {big.data_ptr():#x}; state of env {f}", flush=True)
for off in (0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20, 48 << 20):
    us = timed_step(envs[f], out_with_obs(envs[f], big.data_ptr() + off))
    print(f"  obs at +{off >> 10:6d} KiB: {us:8.2f} us", flush=True)
