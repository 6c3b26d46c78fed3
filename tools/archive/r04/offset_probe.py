#!/usr/bin/env python3
"""Round 4 experiment: does the PHASE of the output buffers inside physically contiguous memory matter?  cfg2's two streams
(observation, one-hot planes) are placed at byte offsets inside over-allocated contiguous buffers (the observation at `a`,
the planes at `b`) and the step is timed for a grid of (a, b)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv
from tiler_slider_amd.vec_env import _contiguous_zeros

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
n = cfg["boards"]
env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=bench.LEVEL_SEED, multi_color=True,
                               max_steps=2**30, auto_reset=True, with_reward=cfg["reward"], with_onehot=cfg["onehot"])
env.reset()
act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
bps = bench.algorithmic_bytes_per_board_step(cfg["size"], cfg["tiles"], cfg["onehot"], cfg["reward"])


def rate(reps=30, warm=40):
    for i in range(warm):
        env.step_async(act[i & 3])
    e0.record()
    for i in range(reps):
        env.step_async(act[i & 3])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"as allocated: {rate():.1f} us")
slack = 64 << 20
obs_bytes = env._obs.numel() * 4
big_obs = _contiguous_zeros((obs_bytes + slack,), torch.uint8, env.device)
big_oh = _contiguous_zeros((env._onehot.numel() * 4 + slack,), torch.uint8, env.device) if env._onehot is not None else None
OFFS = (0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 3 << 20, 16 << 20, 33 << 20)
print("rows: observation offset; columns: one-hot offset " + str(OFFS if big_oh is not None else "(no planes)"))
for a in OFFS:
    row = f"{a:9d} |"
    env._obs_ring = [big_obs[a:a + obs_bytes].view(torch.float32).view(env._obs.shape)]
    for b in (OFFS if big_oh is not None else (0,)):
        if big_oh is not None:
            env._onehot = big_oh[b:b + env._onehot.numel() * 4].view(torch.float32).view(env._onehot.shape)
        env._bind_outputs()
        us = rate()
        row += f" {us:6.1f}"
    print(row + f"   best {bps * n / min(float(x) for x in row.split('|')[1].split()) / 1e3 / 8000:.3f}", flush=True)
