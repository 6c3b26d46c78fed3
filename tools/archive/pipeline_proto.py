#!/usr/bin/env python3
"""Experiment (development tool): would splitting the out-of-cache step into a compute kernel that
writes the compact uint8 observation and a linear u8->f32 expand kernel on a second stream beat
the fused kernel?  cfg4 shape, boards split into CH chunks (one small env each)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]
N = cfg["boards"]
dev = torch.device("cuda", 0)
S = cfg["size"]
fused = VecTilerSliderEnv.random(N, size=S, num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=1,
                                 multi_color=True, max_steps=2**30, auto_reset=True)
fused.reset()
act = torch.zeros(N, dtype=torch.uint8, device=dev)
L = _cabi.lib()
_cabi.check(L.ts_fill_actions(N, 5, 0, 0, act.data_ptr(), torch.cuda.current_stream().cuda_stream), "fill")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def expand(src, dst):
    _cabi.check(L.ts_expand_u8(src.data_ptr(), dst.data_ptr(), src.numel(), torch.cuda.current_stream().cuda_stream), "expand")


def timeit(fn, reps=30):
    ts = []
    for r in range(5):
        for i in range(3):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for i in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)


print(f"fused f32 step: {timeit(lambda: fused.step_async(act)):.1f} us")
for CH in (1, 4, 8, 16, 32):
    n = N // CH
    envs = [VecTilerSliderEnv.random(n, size=S, num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=1,
                                     multi_color=True, max_steps=2**30, auto_reset=True, obs_dtype="uint8",
                                     board_offset=k * n) for k in range(CH)]
    for e in envs:
        e.reset()
    acts = [act[k * n:(k + 1) * n].contiguous() for k in range(CH)]
    big = torch.empty((N, S, S, 3), dtype=torch.float32, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in range(CH)]
    done2 = torch.cuda.Event()
    main = torch.cuda.current_stream()

    def piped():
        s1.wait_stream(main)
        s2.wait_stream(main)
        for k in range(CH):
            with torch.cuda.stream(s1):
                envs[k].step_async(acts[k])
                evs[k].record(s1)
            with torch.cuda.stream(s2):
                s2.wait_event(evs[k])
                expand(envs[k]._obs, big[k * n:(k + 1) * n])
        main.wait_stream(s1)
        main.wait_stream(s2)

    def only_a():
        for k in range(CH):
            envs[k].step_async(acts[k])

    def only_b():
        for k in range(CH):
            expand(envs[k]._obs, big[k * n:(k + 1) * n])

    print(f"chunks={CH:2d}: u8 step alone {timeit(only_a):7.1f} us, expand alone {timeit(only_b):7.1f} us, "
          f"two-stream pipeline {timeit(piped):7.1f} us", flush=True)
    del envs, big
