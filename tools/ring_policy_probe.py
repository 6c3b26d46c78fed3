#!/usr/bin/env python3
"""Round 5: the step into a ring of two observation buffers beyond the cache - what the cache-share rules (edge stores, cached waves)
decide when they count the ring's bytes, against the same launch with the single-buffer decision forced.

    python tools/ring_policy_probe.py        (GPU box)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
L = _cabi.lib()


def rate(env, act, steps=40):
    ts = []
    for r in range(3):
        for i in range(6):
            env.step_async(act[i & 3])
        e0.record()
        for i in range(steps):
            env.step_async(act[i & 3])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e3)
    return statistics.median(ts)


for S, T, K, N in ((15, 32, 24, 262144), (4, 2, 2, 1 << 22), (4, 2, 2, 1 << 21), (5, 2, 3, 1703936), (6, 3, 4, 1 << 20), (8, 4, 8, 651008), (16, 16, 24, 162560), (12, 8, 16, 289280), (24, 30, 60, 72192)):
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
    row = []
    for bufs in (1, 2):
        env = VecTilerSliderEnv.random(N, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True, obs_candidates=0, obs_buffers=bufs)
        env.reset()
        act = [torch.randint(0, 4, (N,), dtype=torch.uint8, device="cuda") for _ in range(4)]
        for i in range(100):
            env.step_async(act[i & 3])
        d = _cabi.describe_launch(env._dims, _cabi.OP_STEP, _cabi.OUT_OBS | _cabi.OUT_FLAGS)
        us = rate(env, act)
        txt = f"ring of {bufs}: {us:.1f} us ({bps * N / us / 8e6:.3f}) edges {d['emit_edges']} cached_every {d['cached_every']}"
        if bufs == 2:
            ring = env._dims.ring_bytes
            env._dims.ring_bytes = 0  # the single-buffer decision (but keep the out-of-cache kernels: the launch is beyond the cache on its own)
            d1 = _cabi.describe_launch(env._dims, _cabi.OP_STEP, _cabi.OUT_OBS | _cabi.OUT_FLAGS)
            us1 = rate(env, act) if d1["out_of_cache"] else float("nan")
            env._dims.ring_bytes = ring
            txt += f"   | with the single-buffer rules (edges {d1['emit_edges']} cached_every {d1['cached_every']}): {us1:.1f} us   | edges forced none / first / last / both:"
            for e in (1, 2, 3, 4):
                env._dims.emit_edges = e
                txt += f" {rate(env, act, 25):.1f}"
            env._dims.emit_edges = 0
        row.append(txt)
        del env, act
        torch.cuda.empty_cache()
    print(f"{S}x{S}, {T} tiles, {N} boards ({bps * N / 1e6:.0f} MB per launch):  " + "   ".join(row), flush=True)
