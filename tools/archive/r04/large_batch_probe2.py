#!/usr/bin/env python3
"""Round 4: the multi-lane kernels and the any-tile-count path fall to 0.4 - 0.7 of the roofline beyond ~1 GB per launch
(profiles/r04_scaling_probe_before.log).  XCD piece x resident blocks at 1.4 / 2.1 GB."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
PIECES = (0, 1, 4, 64, 256, 2048)
print("S T MB | policy (frac) | per xcd_piece " + "/".join(str(p) for p in PIECES) + ": us at launch_hint 0, then best of hints -4 / +4 / +8 at the best piece")
for S, T, K, mb in ((15, 32, 24, 1400), (15, 32, 24, 2100), (16, 16, 24, 1400), (12, 8, 16, 1400), (9, 4, 9, 1400), (24, 30, 60, 1400), (32, 4, 100, 2100), (8, 20, 10, 1400), (8, 12, 8, 2100),
                    (5, 6, 3, 1400), (6, 12, 4, 1400), (3, 1, 0, 2100), (13, 3, 10, 2100)):
    n = (mb * 1_000_000 // (12 * S * S)) // 256 * 256
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
    for i in range(60):
        env.step_async(act[i & 3])

    def rate():
        ts = []
        for r in range(3):
            for i in range(3):
                env.step_async(act[i & 3])
            e0.record()
            for i in range(12):
                env.step_async(act[i & 3])
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 12 * 1e3)
        return statistics.median(ts)

    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
    res = {}
    for p in PIECES:
        env._dims.xcd_piece = p
        res[p] = rate()
    best = min(res, key=res.get)
    env._dims.xcd_piece = best
    hs = {}
    for h in (-4, 4, 8):
        env._dims.launch_hint = h
        hs[h] = rate()
    env._dims.launch_hint = 0
    env._dims.xcd_piece = 0
    print(f"{S:2d} {T:2d} {mb:4d} | {res[0]:7.1f} ({bps * n / res[0] / 8e6:.3f}) | " + " ".join(f"{res[p]:7.1f}" for p in PIECES) + f" | piece {best}: " +
          " ".join(f"h{h:+d} {v:7.1f}" for h, v in hs.items()) + f" | best {bps * n / min(min(res.values()), min(hs.values())) / 8e6:.3f}", flush=True)
    del env, act
