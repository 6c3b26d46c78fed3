#!/usr/bin/env python3
"""Round 5: cfg2 (two streams) with two- / four-wave blocks forced (ts_tuning(TS_TUNE_SMALL_WAVES)): 118.2 -> 116.8 us at best on slow-class
observation buffers - not made a rule (profiles/r05_cfg2_waves_probe.log).    python tools/cfg2_waves_probe.py    (GPU box)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
L = _cabi.lib()
def rate(env, act, steps=60):
    ts = []
    for r in range(5):
        for i in range(6): env.step_async(act[i & 3])
        e0.record()
        for i in range(steps): env.step_async(act[i & 3])
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e3)
    return statistics.median(ts)
N = 1 << 20
act = [torch.randint(0, 4, (N,), dtype=torch.uint8, device="cuda") for _ in range(4)]
for cand in (None, 0):
    env = VecTilerSliderEnv.random(N, size=5, num_tiles=2, num_obstacles=3, seed=3, multi_color=True, max_steps=2**30, auto_reset=True, with_onehot=True, with_reward=True, obs_candidates=cand)
    env.reset()
    for i in range(200): env.step_async(act[i & 3])
    print("obs_candidates", cand, "placement", env.observation_placement_report, flush=True)
    print("  policy:", round(rate(env, act), 2), flush=True)
    for w in (2, 4):
        before = L.ts_tuning(_cabi.TUNE_SMALL_WAVES, w)
        out = []
        for piece in (4, 8, 16, 32):
            env._dims.xcd_piece = piece
            for hint in (0, 1, 2):
                env._dims.launch_hint = hint
                out.append(f"p{piece} h{hint}: {rate(env, act, 40):.1f}")
        env._dims.xcd_piece = env._dims.launch_hint = 0
        L.ts_tuning(_cabi.TUNE_SMALL_WAVES, before)
        print(f"  {w} waves:", "  ".join(out), flush=True)
    print("  policy again:", round(rate(env, act), 2), flush=True)
    del env
    torch.cuda.empty_cache()
