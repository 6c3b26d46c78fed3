#!/usr/bin/env python3
"""Development tool: time every entry point of the C-ABI at a bench config (default: the headline batch cfg1)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv

cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
cfg = bench.CONFIGS[cfgname]
n = cfg["boards"]
env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=1,
                               multi_color=True, max_steps=2**30, auto_reset=True)
envx = VecTilerSliderEnv.from_arrays(cfg["size"], env._blk, env._init, env._tgt, multi_color=True, max_steps=2**30,
                                     auto_reset=True, with_reward=True, with_valid_moves=True)
env.reset(); envx.reset()
act = torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
obs_buf = torch.empty_like(env._obs)
oh_bytes = n * env.onehot_channels * env.size ** 2 * 4
oh_buf = torch.empty((n, env.onehot_channels, env.size, env.size), dtype=torch.float32, device=env.device) if oh_bytes < 4e9 else None
r_buf = torch.empty(n, dtype=torch.int32, device=env.device)


def t(fn, reps=20):
    ts = []
    for r in range(5):
        fn(); torch.cuda.synchronize(); e0.record()
        for i in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)


print(f"{cfgname}: {n} boards")
for name, fn in (("ts_step (flags+obs)", lambda: env.step_async(act)),
                 ("ts_step + reward + legality mask", lambda: envx.step_async(act)),
                 ("ts_reset (+obs)", env.reset),
                 ("ts_encode", lambda: env.encode(obs_buf)),
                 ("ts_valid_moves", env.get_valid_moves),
                 ("ts_is_won", env.is_won),
                 ("ts_reward", lambda: env.reward(r_buf)),
                 ("ts_encode_onehot", (lambda: env.encode_onehot(oh_buf)) if oh_buf is not None else None)):
    if fn is not None:
        print(f"  {name:34s} {t(fn):8.2f} us", flush=True)
