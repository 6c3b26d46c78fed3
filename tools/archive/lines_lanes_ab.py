#!/usr/bin/env python3
"""Development tool: k_lines with 4 / 8 / 16 lanes per board (ts_tuning TS_TUNE_LINES_LANES) over sparse and dense
large-board shapes, ~500 MB of observation per launch, launch_hint -1 / 0 / +1 (resident blocks per CU)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from tiler_slider_amd import VecTilerSliderEnv, _cabi  # noqa: E402

L = _cabi.lib()
SHAPES = [(9, 1, 9), (9, 4, 9), (9, 8, 9), (10, 5, 10), (12, 4, 16), (12, 8, 16), (12, 16, 16), (14, 20, 20), (16, 4, 24), (16, 16, 24),
          (20, 1, 1), (20, 10, 40), (24, 4, 60), (24, 30, 60), (32, 4, 100), (32, 32, 100), (15, 32, 24)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
HINTS = tuple(int(x) for x in os.environ.get("TS_AB_HINTS", "-1,0,1").split(","))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("   S    T   boards   lanes: us/step at launch_hint -1 / 0 / +1 (frac of 8 TB/s at the best)")
for S, T, K in SHAPES:
    n = max(4096, (500_000_000 // (12 * S * S)) // 256 * 256)
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True, placement_trials=0)
    env.reset()
    act = torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device)
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
    row = f"{S:4d} {T:4d} {n:8d}  "
    combos = [(16, 0), (8, 0), (4, 0)] if not os.environ.get("TS_AB_BPW") else [(16, 4), (16, 2), (16, 1), (8, 4), (8, 2)]
    for lanes, bpw in combos:
        L.ts_tuning(_cabi.TUNE_LINES_LANES, lanes)
        L.ts_tuning(_cabi.TUNE_LINES_BPW, bpw)
        res = []
        for hint in HINTS:
            env._dims.launch_hint = hint
            ts = []
            for r in range(3):
                for i in range(3):
                    env.step_async(act)
                e0.record()
                for i in range(20):
                    env.step_async(act)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 20 * 1e3)
            res.append(statistics.median(ts))
        best = min(res)
        row += f" | {lanes:2d}" + (f"x{bpw}" if bpw else "") + ": " + " / ".join(f"{x:6.1f}" for x in res) + f" ({bps * n / best / 1e3 / 8000:.3f})"
    env._dims.launch_hint = 0
    L.ts_tuning(_cabi.TUNE_LINES_LANES, 0)
    L.ts_tuning(_cabi.TUNE_LINES_BPW, 0)
    print(row, flush=True)
    del env
