#!/usr/bin/env python3
"""Round 4: the two ways of putting a share of a wave stream through the cache - write-back EDGE stores of every wave's chunk
(ts_dims.emit_edges: 1 = none, 2 = first, 3 = last, 4 = both; 0 = the policy) and every N-th WAVE with the cached stores
(ts_tuning(TS_TUNE_CACHED_EVERY)) - crossed, for the multi-lane kernels' shapes."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi
L = _cabi.lib()
SHAPES = [(15, 32, 24, 1 << 18), (14, 20, 20, 212_000), (12, 8, 16, 289_000), (9, 4, 9, 514_000), (10, 5, 10, 416_000), (16, 16, 24, 162_000), (8, 20, 10, 650_000), (8, 12, 8, 650_000)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
EDGES, EVERY = (0, 1, 2, 3, 4), (1, 16, 32)
print("rows: shape; columns: emit_edges " + " / ".join(str(e) for e in EDGES) + " for cached waves never | every 16th | every 32nd; us per step")
for S, T, K, n in SHAPES:
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
    for i in range(200):
        env.step_async(act[i & 3])
    row = f"{S:3d} {T:3d} {n:7d} |"
    for ev in EVERY:
        L.ts_tuning(_cabi.TUNE_CACHED_EVERY, ev)
        for ed in EDGES:
            env._dims.emit_edges = ed
            ts = []
            for r in range(3):
                for i in range(5):
                    env.step_async(act[i & 3])
                e0.record()
                for i in range(30):
                    env.step_async(act[i & 3])
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 30 * 1e3)
            row += f" {statistics.median(ts):6.1f}"
        row += " |"
    env._dims.emit_edges = 0
    L.ts_tuning(_cabi.TUNE_CACHED_EVERY, 0)
    print(row, flush=True)
    del env
