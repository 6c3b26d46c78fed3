#!/usr/bin/env python3
"""Round 5: boards per wave x resident blocks of k_lines for the weakest shapes of the 500 MB sweep (9x9, 10x10, 15x15 / 32 tiles ...).

    python tools/lines_bpw_probe.py [S,T,K,N ...]        (GPU box)"""
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
        for i in range(5):
            env.step_async(act[i & 3])
        e0.record()
        for i in range(steps):
            env.step_async(act[i & 3])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e3)
    return statistics.median(ts)


for spec in sys.argv[1:] or ["9,4,9,524288", "10,5,10,425984", "9,1,3,524288", "9,8,9,524288"]:
    S, T, K, N = (int(x) for x in spec.split(","))
    env = VecTilerSliderEnv.random(N, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True, obs_candidates=0)
    env.reset()
    act = [torch.randint(0, 4, (N,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    for i in range(100):
        env.step_async(act[i & 3])
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
    d = _cabi.describe_launch(env._dims, _cabi.OP_STEP, _cabi.OUT_OBS | _cabi.OUT_FLAGS)
    base = rate(env, act)
    print(f"{S}x{S}, {T} tiles, {N} boards, {bps * N / 1e6:.0f} MB: policy {base:.1f} us ({bps * N / base / 8e6:.3f})  {d['name']} bpw {d['boards_per_wave']} blocks/CU {d['blocks_per_cu']} edges {d['emit_edges']} cached_every {d['cached_every']}", flush=True)
    for lanes in (4, 8, 16):
        env._dims.lines_lanes = lanes
        for bpw in [b for b in (2, 3, 4, 6, 8, 12, 16) if b <= 64 // lanes]:
            before = L.ts_tuning(_cabi.TUNE_LINES_BPW, bpw)
            dd = _cabi.describe_launch(env._dims, _cabi.OP_STEP, _cabi.OUT_OBS | _cabi.OUT_FLAGS)
            out = []
            for hint in (-4, 0, 4, 8):
                env._dims.launch_hint = hint
                us = rate(env, act, 25)
                out.append(f"hint {hint:+d}: {us:.1f}")
            env._dims.launch_hint = 0
            L.ts_tuning(_cabi.TUNE_LINES_BPW, before)
            print(f"   lanes {lanes:2d} bpw {bpw:2d} (got {dd['boards_per_wave']:2d}, blocks/CU {dd['blocks_per_cu']:2d}): " + "   ".join(out), flush=True)
    env._dims.lines_lanes = 0
    del env, act
    torch.cuda.empty_cache()
