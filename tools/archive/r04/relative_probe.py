#!/usr/bin/env python3
"""Round 4 experiment: cfg2 writes two streams (observation, one-hot planes).  Both are placed inside ONE physically contiguous
slab, the observation at offset 0 and the planes at offset X, so that the PHYSICAL distance between the two streams is exactly
X; the step is rated for a list of X (and again with the slab re-allocated)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv
from tiler_slider_amd.vec_env import _contiguous_zeros

cfg = bench.CONFIGS["cfg2"]
n = cfg["boards"]
env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=bench.LEVEL_SEED, multi_color=True,
                               max_steps=2**30, auto_reset=True, with_reward=True, with_onehot=True)
env.reset()
act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def rate(reps=30, warm=50):
    for i in range(warm):
        env.step_async(act[i & 3])
    e0.record()
    for i in range(reps):
        env.step_async(act[i & 3])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"as allocated (two allocations): {rate():.1f} us")
obs_b, oh_b = env._obs.numel() * 4, env._onehot.numel() * 4
MB = 1 << 20
XS = [320, 352, 384, 448, 512, 576, 640, 704, 768, 832, 896, 960, 1024, 1088, 1152, 1280, 1536, 1792, 2048, 2560, 3072, 4096]
for rep in range(3):
    slab = _contiguous_zeros(((XS[-1] + 1) * MB + oh_b,), torch.uint8, env.device)
    env._obs_ring = [slab[:obs_b].view(torch.float32).view(env._obs.shape)]
    row = f"slab {rep} ({slab.data_ptr():#x}): "
    for x in XS:
        env._onehot = slab[x * MB:x * MB + oh_b].view(torch.float32).view(env._onehot.shape)
        env._bind_outputs()
        row += f" {x}:{rate():.1f}"
    print(row, flush=True)
    # the observation moved instead (planes at the slab's end)
    env._onehot = slab[XS[-1] * MB:XS[-1] * MB + oh_b].view(torch.float32).view(env._onehot.shape)
    row = "   observation at offset: "
    for x in (0, 64, 256, 1024, 2048, 3072):
        env._obs_ring = [slab[x * MB:x * MB + obs_b].view(torch.float32).view(env._obs.shape)]
        env._bind_outputs()
        row += f" {x}:{rate():.1f}"
    print(row, flush=True)
    del slab
    env._obs_ring = [torch.zeros(1, device=env.device)]
    torch.cuda.empty_cache()
