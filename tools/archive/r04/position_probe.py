#!/usr/bin/env python3
"""Round 4 experiment: does the POSITION of physically contiguous output buffers inside VRAM matter?  Before the environment is
built a contiguous spacer of X MiB is allocated (and kept), which moves where the runtime places the environment's buffers;
the step is rated for a list of X.  usage: position_probe.py cfg2|cfg4 [repeat]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv
from tiler_slider_amd.vec_env import _contiguous_zeros

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
n = cfg["boards"]
dev = torch.device("cuda", 0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
bps = bench.algorithmic_bytes_per_board_step(cfg["size"], cfg["tiles"], cfg["onehot"], cfg["reward"])
act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev) for _ in range(4)]


def rate(env, reps=30, warm=60):
    for i in range(warm):
        env.step_async(act[i & 3])
    e0.record()
    for i in range(reps):
        env.step_async(act[i & 3])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


SPACERS = [0, 64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 65536, 131072]
print("spacer MiB | us per step (frac) | obs ptr, onehot ptr (virtual)")
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
    for x in SPACERS:
        spacer = _contiguous_zeros((x << 20,), torch.uint8, dev) if x else None
        env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=bench.LEVEL_SEED, multi_color=True,
                                       max_steps=2**30, auto_reset=True, with_reward=cfg["reward"], with_onehot=cfg["onehot"])
        env.reset()
        us = rate(env)
        print(f"{x:10d} | {us:6.1f} ({bps * n / us / 1e3 / 8000:.3f}) | {env._obs.data_ptr():#x} {env._onehot.data_ptr() if env._onehot is not None else 0:#x}", flush=True)
        del env, spacer
        torch.cuda.empty_cache()
