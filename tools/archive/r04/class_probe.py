#!/usr/bin/env python3
"""Round 4 experiment: which launches see the two classes of physically contiguous observation buffers?  Per shape the
observation buffer is placed on each of K contiguous 1 GiB blocks (all alive) and the step rated."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv
from tiler_slider_amd.vec_env import _contiguous_zeros

K = 20
dev = torch.device("cuda", 0)
blocks = [_contiguous_zeros((1 << 30,), torch.uint8, dev) for _ in range(K)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("shape (S, T, K, boards, onehot) | us per step with the observation on each of %d blocks" % K)
for S, T, Kk, n, oh in ((5, 2, 3, 1 << 20, 1), (5, 2, 3, 1 << 20, 0), (4, 2, 2, 1 << 22, 0), (6, 3, 4, 1 << 20, 0), (8, 4, 8, 650_000, 0), (8, 12, 8, 650_000, 0),
                        (12, 8, 16, 400_000, 0), (15, 32, 24, 1 << 18, 0), (16, 4, 24, 200_000, 0), (24, 30, 60, 100_000, 0), (32, 4, 100, 60_000, 0)):
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=Kk, seed=3, multi_color=True, max_steps=2**30, auto_reset=True,
                                   with_onehot=bool(oh), with_reward=bool(oh))
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev) for _ in range(4)]
    obs_b, shape = env._obs.numel() * 4, env._obs.shape
    row, res = "", []
    for b in blocks:
        env._obs_ring = [b[:obs_b].view(torch.float32).view(shape)]
        env._bind_outputs()
        for i in range(30):
            env.step_async(act[i & 3])
        e0.record()
        for i in range(20):
            env.step_async(act[i & 3])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res.append(us)
        row += f" {us:5.1f}"
    print(f"{(S, T, Kk, n, oh)} |{row} | min {min(res):.1f} max {max(res):.1f} (+{(max(res) / min(res) - 1) * 100:.1f} %)", flush=True)
    del env, act
    torch.cuda.empty_cache()
