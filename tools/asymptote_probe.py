#!/usr/bin/env python3
"""Round 5: where between 2 and 4 GB per launch the rate of a shape falls, and which per-call policy field moves it.

    python tools/asymptote_probe.py [S,T,K ...]      (GPU box; default: cfg1's, cfg4's and two more shapes)

Per shape and size: us per step (HIP events, median of 3 x 8 steps), fraction of 8 TB/s, the launch description and where the
observation buffer was allocated; then the same launch with the per-call policy fields varied (edge stores, block mapping,
resident blocks)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
MB = [int(x) for x in os.environ.get("MB", "2100,2500,2900,3400").split(",")]
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(4, 2, 2), (15, 32, 24), (9, 4, 9), (13, 3, 10)]


def rate(env, act, steps=8):
    ts = []
    for r in range(3):
        for i in range(2):
            env.step_async(act[i & 1])
        e0.record()
        for i in range(steps):
            env.step_async(act[i & 1])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e3)
    return statistics.median(ts)


for S, T, K in shapes:
    for mb in MB:
        n = (mb * 1_000_000 // (12 * S * S)) // 256 * 256
        env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True, obs_candidates=0)
        env.reset()
        act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(2)]
        for i in range(20):
            env.step_async(act[i & 1])
        bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
        d = _cabi.describe_launch(env._dims, _cabi.OP_STEP, _cabi.OUT_OBS | _cabi.OUT_FLAGS)
        us = rate(env, act)
        print(f"{S}x{S} T={T} {mb} MB ({n} boards): {us:8.1f} us  frac {bps * n / us / 8e6:.3f}   {d['name']} bpw {d['boards_per_wave']} blocks/CU {d['blocks_per_cu']} "
              f"edges {d['emit_edges']} piece {d['xcd_piece']} cached_every {d['cached_every']}  memory {[m['memory'] for m in env.output_memory_report]}", flush=True)
        if mb == MB[-1] or mb == MB[0]:
            for field, values in (("emit_edges", (1, 2, 3, 4)), ("xcd_piece", (1, 8, 16, 64, 256)), ("launch_hint", (-4, -2, 2, 4, 8)), ("lines_lanes", (4, 8, 16, 32))):
                out = []
                for v in values:
                    setattr(env._dims, field, v)
                    out.append(f"{v}: {rate(env, act, 5):.1f}")
                    setattr(env._dims, field, 0)
                print(f"      {field:12s} " + "   ".join(out), flush=True)
            if S <= 8:  # boards per wave of the one-lane-per-board kernel (ts_tuning: process-wide)
                out = []
                for v in (16, 32, 64):
                    before = _cabi.lib().ts_tuning(_cabi.TUNE_SMALL_BPW, v)
                    out.append(f"{v}: {rate(env, act, 5):.1f}")
                    _cabi.lib().ts_tuning(_cabi.TUNE_SMALL_BPW, before)
                print(f"      {'small_bpw':12s} " + "   ".join(out), flush=True)
        del env, act
        torch.cuda.empty_cache()
