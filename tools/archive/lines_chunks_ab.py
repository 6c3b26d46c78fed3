#!/usr/bin/env python3
"""Round 4 experiment: k_lines beyond the Infinity Cache with one or two groups of boards per wave (ts_tuning(TS_TUNE_LINES_CHUNKS):
the second group's loads are in flight while the first group is written out), alternately on the same buffers."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

L = _cabi.lib()
SHAPES = [(15, 32, 24, 1 << 18), (15, 8, 24, 1 << 18), (14, 20, 20, 300_000), (12, 8, 16, 400_000), (9, 4, 9, 600_000), (10, 5, 10, 500_000), (11, 6, 8, 413_000),
          (13, 3, 10, 295_000), (16, 16, 24, 195_000), (16, 4, 24, 195_000), (20, 10, 40, 125_000), (24, 30, 60, 100_000), (32, 4, 100, 60_000), (32, 32, 100, 60_000)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
hints = [int(x) for x in os.environ.get("HINTS", "0").split(",")]
print("   S    T    K   boards |  one group | two groups at launch_hint " + " / ".join(f"{h:+d}" for h in hints))
for S, T, K, n in SHAPES:
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
    for i in range(200):
        env.step_async(act[i & 3])

    def rate():
        ts = []
        for r in range(4):
            for i in range(5):
                env.step_async(act[i & 3])
            e0.record()
            for i in range(30):
                env.step_async(act[i & 3])
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 30 * 1e3)
        return statistics.median(ts)

    L.ts_tuning(_cabi.TUNE_LINES_CHUNKS, 1)
    one = rate()
    L.ts_tuning(_cabi.TUNE_LINES_CHUNKS, 2)
    two = []
    for h in hints:
        env._dims.launch_hint = h
        two.append(rate())
    env._dims.launch_hint = 0
    L.ts_tuning(_cabi.TUNE_LINES_CHUNKS, 1)
    again = rate()
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
    print(f"{S:4d} {T:4d} {K:4d} {n:8d} | {one:6.1f} {again:6.1f} ({bps * n / min(one, again) / 8e6:.3f}) | " + " ".join(f"{t:6.1f}" for t in two) +
          f"   best {(min(two) / min(one, again) - 1) * 100:+.1f} %", flush=True)
    del env
