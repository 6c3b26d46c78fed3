#!/usr/bin/env python3
"""Development tool: cfg2 streams two big outputs per wave (observation + one-hot).  Which of them - or their
COMBINATION - decides whether the step runs at the fast or the slow speed?  M candidates of each are allocated and
every pair is timed with the same state; also each buffer alone (plain step / stand-alone one-hot encode).

    python tools/placement_study6.py [M]
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

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = bench.CONFIGS["cfg2"]
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
                               seed=bench.LEVEL_SEED, multi_color=True, max_steps=2**30, device=dev, auto_reset=True,
                               with_reward=True, with_onehot=True)
env.reset()


def timed(obs_ptr, oh_ptr, reps=40):
    out = _cabi.StepOut(env._flags.data_ptr(), obs_ptr, env._reward.data_ptr(), oh_ptr, None, None)
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


obs = [env._obs] + [torch.empty_like(env._obs) for _ in range(M - 1)]
oh = [env._onehot] + [torch.empty_like(env._onehot) for _ in range(M - 1)]
# interleave the order of allocation for the second half: obs and one-hot candidates alternate
print("addresses: obs " + " ".join(f"{o.data_ptr():#x}" for o in obs))
print("           oh  " + " ".join(f"{o.data_ptr():#x}" for o in oh))
print("observation alone (step without one-hot):  " + "  ".join(f"{timed(o.data_ptr(), None):7.2f}" for o in obs), flush=True)
print("one-hot alone (step without observation):  " + "  ".join(f"{timed(None, h.data_ptr()):7.2f}" for h in oh), flush=True)
print("pairs, rows = observation buffer, columns = one-hot buffer")
for i, o in enumerate(obs):
    print(f"  obs {i}: " + "  ".join(f"{timed(o.data_ptr(), h.data_ptr()):7.2f}" for h in oh), flush=True)
