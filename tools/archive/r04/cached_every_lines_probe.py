#!/usr/bin/env python3
"""Round 4 experiment: every N-th wave of the kernels that deal a board over several lanes (k_lines, k_deal) writes its observation
with the cached stores (forced through ts_tuning(TS_TUNE_CACHED_EVERY, N)); all-nontemporal = 1."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi
L = _cabi.lib()
EVERY = [int(x) for x in os.environ.get("EVERY", "8,16,32").split(",")]
SHAPES = [(S, T, K, mb * 1_000_000 // (12 * S * S)) for mb in (500, 720) for (S, T, K) in
          ((17, 3, 20), (18, 4, 10), (20, 1, 1), (20, 10, 40), (22, 6, 40), (24, 4, 60), (24, 30, 60), (26, 6, 60), (28, 8, 60), (30, 16, 60), (32, 4, 100), (32, 32, 100), (32, 64, 100))]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("   S    T   boards  out MB | all nontemporal (frac) | every " + " / ".join(str(e) for e in EVERY) + " | best")
for S, T, K, n in SHAPES:
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

    L.ts_tuning(_cabi.TUNE_CACHED_EVERY, 1)
    base = rate()
    row = []
    for ev in EVERY:
        L.ts_tuning(_cabi.TUNE_CACHED_EVERY, ev)
        row.append(rate())
    L.ts_tuning(_cabi.TUNE_CACHED_EVERY, 1)
    base = min(base, rate())
    L.ts_tuning(_cabi.TUNE_CACHED_EVERY, 0)
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
    print(f"{S:4d} {T:4d} {n:8d} {12 * S * S * n / 1e6:7.0f} | {base:7.1f} ({bps * n / base / 8e6:.3f}) | " + " ".join(f"{t:7.1f}" for t in row) +
          f" | {(min(row) / base - 1) * 100:+.1f} %", flush=True)
    del env
