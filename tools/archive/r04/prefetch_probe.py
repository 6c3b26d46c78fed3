#!/usr/bin/env python3
"""Round 4: launches beyond ~1 GB lose half their rate although the HBM traffic per board is unchanged (PMC).  Is it the state
no longer being found in the Infinity Cache?  A pass that READS all state arrays right before every step (torch reductions),
timed with and without."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for S, T, K, mb in ((15, 32, 24, 700), (15, 32, 24, 1400), (16, 16, 24, 1400), (13, 3, 10, 2100), (8, 20, 10, 1400), (5, 6, 3, 1400)):
    n = (mb * 1_000_000 // (12 * S * S)) // 256 * 256
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
    state = [t.view(torch.int32) if t.numel() * t.element_size() % 4 == 0 else t for t in (env._pos.reshape(-1), env._tgt.reshape(-1), env._blk.reshape(-1), env._step_count, env._done)
             if t is not None]
    lines = getattr(env, "_lines", None)
    if lines is not None:
        state.append(lines.reshape(-1).view(torch.int32))
    state_mb = sum(t.numel() * t.element_size() for t in state) / 1e6

    def touch():
        for t in state:
            torch.sum(t)

    def rate(prefetch):
        ts = []
        for r in range(3):
            for i in range(3):
                env.step_async(act[i & 3])
            e0.record()
            for i in range(12):
                if prefetch:
                    touch()
                env.step_async(act[i & 3])
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 12 * 1e3)
        return statistics.median(ts)

    def rate_touch():
        e0.record()
        for i in range(12):
            touch()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 12 * 1e3

    for i in range(40):
        env.step_async(act[i & 3])
    a, b, c = rate(False), rate(True), rate_touch()
    print(f"{S}x{S} T={T} {mb} MB, state {state_mb:.0f} MB: step {a:.1f} us | read pass + step {b:.1f} us | read pass alone {c:.1f} us | step after a read pass ~{b - c:.1f} us", flush=True)
    del env
