#!/usr/bin/env python3
"""Round 4 experiment: cfg2 environments are built (physically contiguous outputs) until one of the FAST class (< 116 us) and one
of the SLOW class (> 118 us) exist; then their buffers are crossed: whose observation buffer, one-hot buffer, state arrays make
the step fast?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi
from tiler_slider_amd.vec_env import _ptr

cfg = bench.CONFIGS["cfg2"]
n = cfg["boards"]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda") for _ in range(4)]


def rate(env, reps=30, warm=50):
    for i in range(warm):
        env.step_async(act[i & 3])
    e0.record()
    for i in range(reps):
        env.step_async(act[i & 3])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def make():
    env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=bench.LEVEL_SEED, multi_color=True,
                                   max_steps=2**30, auto_reset=True, with_reward=True, with_onehot=True)
    env.reset()
    return env


fast = slow = None
keep = []
for t in range(40):
    env = make()
    us = rate(env)
    print(f"env {t}: {us:6.1f} us  obs {env._obs.data_ptr():#x} onehot {env._onehot.data_ptr():#x}", flush=True)
    if us < 116 and fast is None:
        fast = env
    elif us > 118 and slow is None:
        slow = env
    else:
        keep.append(env) if len(keep) < 3 else None  # a few stay alive so that later allocations land elsewhere
    if fast is not None and slow is not None:
        break
if fast is None or slow is None:
    print("only one class turned up")
    sys.exit(0)
print(f"fast {rate(fast):.1f}  slow {rate(slow):.1f}")
STATE = ["_pos", "_init", "_tgt", "_blk", "_step_count", "_done", "_flags", "_reward"]


def bind(env, obs, oh, state_from):
    env._obs_ring, env._onehot = [obs], oh
    for nm in STATE:
        setattr(env, nm, getattr(state_from, "_orig" + nm))
    env._state = _cabi.State(_ptr(env._pos), _ptr(env._init), _ptr(env._tgt), _ptr(env._blk), _ptr(env._step_count), _ptr(env._done), None)
    env._bind_outputs()


for e in (fast, slow):
    for nm in STATE:
        setattr(e, "_orig" + nm, getattr(e, nm))
    e._orig_obs, e._orig_oh = e._obs, e._onehot
F, S = fast, slow
for label, obs, oh, st in (("obs F, planes F, state F", F._orig_obs, F._orig_oh, F), ("obs S, planes S, state S", S._orig_obs, S._orig_oh, S),
                           ("obs F, planes S, state S", F._orig_obs, S._orig_oh, S), ("obs S, planes F, state S", S._orig_obs, F._orig_oh, S),
                           ("obs F, planes F, state S", F._orig_obs, F._orig_oh, S), ("obs S, planes S, state F", S._orig_obs, S._orig_oh, F),
                           ("obs F, planes S, state F", F._orig_obs, S._orig_oh, F), ("obs S, planes F, state F", S._orig_obs, F._orig_oh, F)):
    bind(S, obs, oh, st)
    print(f"{label}: {rate(S):6.1f} us", flush=True)
