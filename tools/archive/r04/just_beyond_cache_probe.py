#!/usr/bin/env python3
"""Round 4 experiment: batches whose observation stream is 260 - 700 MB (1 - 2.7 x the Infinity Cache), plain step of the
one-lane-per-board kernel: all-nontemporal (shipped) against every N-th wave writing with the cached (agent-scope) stores
(ts_tuning(9, N), experiment build), on the same buffers."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

L = _cabi.lib()
EVERY = [int(x) for x in os.environ.get("EVERY", "2,4,8,16").split(",")]
MB = [int(x) for x in os.environ.get("MB", "270,300,340,400,460,520,600,700").split(",")]
SHAPES = [(4, 2, 2), (5, 2, 3), (6, 3, 4), (7, 5, 6), (8, 4, 8), (3, 1, 0)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("   S    T   boards  out MB |  shipped (frac) | every " + " / ".join(str(e) for e in EVERY) + " | best vs shipped")
for S, T, K in SHAPES:
    for mb in MB:
        n = (mb * 1_000_000 // (12 * S * S)) // 256 * 256
        env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True)
        env.reset()
        act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
        for i in range(200):
            env.step_async(act[i & 3])

        def rate():
            ts = []
            for r in range(3):
                for i in range(5):
                    env.step_async(act[i & 3])
                e0.record()
                for i in range(30):
                    env.step_async(act[i & 3])
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 30 * 1e3)
            return statistics.median(ts)

        L.ts_tuning(9, 0)
        base = rate()
        row = []
        for ev in EVERY:
            L.ts_tuning(9, ev)
            row.append(rate())
        L.ts_tuning(9, 0)
        base = min(base, rate())
        bps = bench.algorithmic_bytes_per_board_step(S, T, False, False)
        print(f"{S:4d} {T:4d} {n:8d} {12 * S * S * n / 1e6:7.0f} | {base:7.1f} ({bps * n / base / 8e6:.3f}) | " + " ".join(f"{t:7.1f}" for t in row) +
              f" | {(min(row) / base - 1) * 100:+.1f} %", flush=True)
        del env
