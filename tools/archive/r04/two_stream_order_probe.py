#!/usr/bin/env python3
"""Round 4 experiment: cfg2 writes two large streams (observation, one-hot planes).  With the library's store order
(observation first) the OBSERVATION buffer's place decides between two speeds (profiles/r04_cross_probe.log).  Which buffer
decides when a wave writes its planes first (build/variants/pf.so: -DTS_TWO_STREAM=1), and how fast are the classes then?
One process, both libraries on the same buffers: 12 candidate observation buffers x 6 candidate plane buffers."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi
from tiler_slider_amd.vec_env import _contiguous_zeros

VDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "variants")
names = sys.argv[1:] or ["base", "pf"]
cfg = bench.CONFIGS["cfg2"]
n, dev = cfg["boards"], torch.device("cuda", 0)
env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=bench.LEVEL_SEED, multi_color=True,
                               max_steps=2**30, auto_reset=True, with_reward=True, with_onehot=True, obs_candidates=0)
env.reset()
stream = torch.cuda.current_stream(dev).cuda_stream
act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev) for _ in range(4)]
obs_c = [env._obs] + [_contiguous_zeros(tuple(env._obs.shape), torch.float32, dev) for _ in range(11)]
oh_c = [env._onehot] + [_contiguous_zeros(tuple(env._onehot.shape), torch.float32, dev) for _ in range(5)]
libs = {}
for nm in names:
    L = C.CDLL(os.path.join(VDIR, nm + ".so"))
    L.ts_step.argtypes = [C.POINTER(_cabi.Dims), C.POINTER(_cabi.State), C.c_void_p, C.c_uint32, C.POINTER(_cabi.StepOut), C.c_void_p]
    L.ts_step.restype = C.c_int32
    libs[nm] = L
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def rate(L, obs, oh, reps=30, warm=10):
    env._obs_ring, env._onehot = [obs], oh
    env._bind_outputs()
    for i in range(warm + reps):
        if i == warm:
            e0.record()
        assert L.ts_step(C.byref(env._dims), C.byref(env._state), act[i & 3].data_ptr(), env._mode, C.byref(env._out), stream) == 0
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for nm in names:
    rate(libs[nm], obs_c[0], oh_c[0], reps=300)  # clocks
print("rows: observation buffer 0..11; columns: plane buffer 0..5; us per step")
for nm in names:
    print(f"-- {nm}")
    for i, o in enumerate(obs_c):
        print(f"  obs {i:2d} | " + " ".join(f"{rate(libs[nm], o, h):6.1f}" for h in oh_c), flush=True)
