#!/usr/bin/env python3
"""Round 4 experiment: K physically contiguous blocks of `MB` MiB allocated back to back (all alive); cfg2's observation
buffer is placed at the start of each in turn and the step rated: do the fast blocks come with a period?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv
from tiler_slider_amd.vec_env import _contiguous_zeros

K = int(sys.argv[1]) if len(sys.argv) > 1 else 48
MB = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = bench.CONFIGS["cfg2"]
n = cfg["boards"]
env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=bench.LEVEL_SEED, multi_color=True,
                               max_steps=2**30, auto_reset=True, with_reward=True, with_onehot=True)
env.reset()
act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def rate(reps=24, warm=30):
    for i in range(warm):
        env.step_async(act[i & 3])
    e0.record()
    for i in range(reps):
        env.step_async(act[i & 3])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


obs_b = env._obs.numel() * 4
shape = env._obs.shape
print(f"own buffer: {rate():.1f}")
blocks = [_contiguous_zeros((MB << 20,), torch.uint8, env.device) for _ in range(K)]
row = ""
for k, b in enumerate(blocks):
    env._obs_ring = [b[:obs_b].view(torch.float32).view(shape)]
    env._bind_outputs()
    us = rate()
    row += f" {'F' if us < 116 else 's'}{us:5.1f}"
    if k % 8 == 7:
        print(f"blocks {k - 7:2d}..{k:2d} (va {blocks[k - 7].data_ptr() >> 20:#x} MiB):" + row, flush=True)
        row = ""
if row:
    print("rest:" + row)
