#!/usr/bin/env python3
"""Round 4 experiment: the small state arrays (pos, tgt, blk, step_count, done, flags, reward, actions: 1-4 MB each) all come
from torch's allocator at 2 MiB-aligned addresses, so the SAME board index has the SAME low address bits in every array.  Do
their relative offsets decide the fast / slow class of the out-of-cache step?  All arrays are re-placed inside one pool with
array k at offset k * stagger (+ its size, rounded up to 2 MiB) and the step is rated per stagger."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi
from tiler_slider_amd.vec_env import _ptr

cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = bench.CONFIGS[cfgname]
n = cfg["boards"]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def rate(env, act, reps=30, warm=50):
    for i in range(warm):
        env.step_async(act[i & 3])
    e0.record()
    for i in range(reps):
        env.step_async(act[i & 3])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


NAMES = ["_pos", "_init", "_tgt", "_blk", "_step_count", "_done", "_flags", "_reward"]
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=bench.LEVEL_SEED, multi_color=True,
                                   max_steps=2**30, auto_reset=True, with_reward=cfg["reward"], with_onehot=cfg["onehot"])
    env.reset()
    act0 = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
    row = f"env {trial}: as constructed {rate(env, act0):6.1f} us |"
    ptrs = " ".join(f"{nm[1:4]}@{(getattr(env, nm).data_ptr() >> 12) & 0xfffff:05x}" for nm in NAMES if getattr(env, nm) is not None)
    pool = torch.zeros(256 << 20, dtype=torch.uint8, device=env.device)
    keep = {nm: getattr(env, nm) for nm in NAMES}
    for stagger in (0, 128, 256, 512, 1024, 2048, 4096, 4224, 8320, 65664):
        off = 0
        for k, nm in enumerate(NAMES):
            t = keep[nm]
            if t is None:
                continue
            nbytes = t.numel() * t.element_size()
            start = off + k * stagger
            view = pool[start:start + nbytes].view(t.dtype).view(t.shape)
            view.copy_(t)
            setattr(env, nm, view)
            off += (nbytes + k * stagger + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        acts = []
        for i, a in enumerate(act0):
            start = off + (len(NAMES) + i) * stagger
            v = pool[start:start + n]
            v.copy_(a)
            acts.append(v)
            off += (n + (len(NAMES) + i) * stagger + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        env._state = _cabi.State(_ptr(env._pos), _ptr(env._init), _ptr(env._tgt), _ptr(env._blk), _ptr(env._step_count), _ptr(env._done),
                                 _ptr(env._lines) if env._lines is not None else None)
        env._bind_outputs()
        row += f" {stagger}:{rate(env, acts):6.1f}"
    print(row + "   [" + ptrs + "]", flush=True)
    del env, pool, keep
    torch.cuda.empty_cache()
