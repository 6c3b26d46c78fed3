#!/usr/bin/env python3
"""Round 4 experiment: cfg2 on K physically contiguous observation buffers x K physically contiguous one-hot buffers (all alive,
allocated alternately): is the fast / slow class a property of one buffer, of the other, or of the pair?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv
from tiler_slider_amd.vec_env import _contiguous_zeros

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
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


obs0, oh0 = env._obs, env._onehot
print(f"as constructed: {rate():.1f} us   obs {obs0.data_ptr():#x} onehot {oh0.data_ptr():#x}")
obs_bufs, oh_bufs = [obs0], [oh0]
for k in range(1, K):
    obs_bufs.append(_contiguous_zeros(tuple(obs0.shape), torch.float32, env.device))
    oh_bufs.append(_contiguous_zeros(tuple(oh0.shape), torch.float32, env.device))
print("rows: observation buffer, columns: one-hot buffer (buffer 0 = the environment's own)")
print("            " + "".join(f" {b.data_ptr() >> 20 & 0xfffff:>7x}" for b in oh_bufs))
for i, ob in enumerate(obs_bufs):
    row = f"{ob.data_ptr() >> 20 & 0xfffff:>9x} |"
    for j, hb in enumerate(oh_bufs):
        env._obs_ring, env._onehot = [ob], hb
        env._bind_outputs()
        row += f" {rate():7.1f}"
    print(row, flush=True)
# the same with the one-hot output switched off (observation alone) and the observation alone per buffer
print("observation stream alone (plain step, no planes / reward), per observation buffer:")
plain = VecTilerSliderEnv.from_arrays(cfg["size"], env._blk, env._init, env._tgt, multi_color=True, max_steps=2**30, auto_reset=True)
plain.reset()
row = ""
for ob in obs_bufs:
    plain._obs_ring = [ob]
    plain._bind_outputs()
    for i in range(50):
        plain.step_async(act[i & 3])
    e0.record()
    for i in range(30):
        plain.step_async(act[i & 3])
    e1.record(); torch.cuda.synchronize()
    row += f" {e0.elapsed_time(e1) / 30 * 1e3:7.1f}"
print(row)
