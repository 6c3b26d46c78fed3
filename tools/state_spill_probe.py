#!/usr/bin/env python3
"""Round 5: 4x4 batches whose STATE no longer fits the Infinity Cache (15M boards and more): boards per wave x resident blocks.

    python tools/state_spill_probe.py [boards ...]       (GPU box)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def rate(env, act, steps=6):
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


L = _cabi.lib()
for n in [int(x) for x in sys.argv[1:]] or [12 << 20, 14 << 20, 16 << 20, 20 << 20, 24 << 20]:
    env = VecTilerSliderEnv.random(n, size=4, num_tiles=2, num_obstacles=2, seed=3, multi_color=True, max_steps=2**30, auto_reset=True, obs_candidates=0)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(2)]
    for i in range(10):
        env.step_async(act[i & 1])
    bps = bench.algorithmic_bytes_per_board_step(4, 2, False, False)
    base = rate(env, act)
    d = _cabi.describe_launch(env._dims, _cabi.OP_STEP, _cabi.OUT_OBS | _cabi.OUT_FLAGS)
    print(f"{n} boards ({bps * n / 1e9:.2f} GB per launch, state {15 * n / 2**20:.0f} MiB): policy {base:.1f} us (frac {bps * n / base / 8e6:.3f})  bpw {d['boards_per_wave']} blocks/CU {d['blocks_per_cu']} piece {d['xcd_piece']}", flush=True)
    for bpw in (32, 64):
        before = L.ts_tuning(_cabi.TUNE_SMALL_BPW, bpw)
        out = []
        for hint in (0, 2, 4, 6, 8):
            env._dims.launch_hint = hint
            for piece in (0, 1):
                env._dims.xcd_piece = piece
                us = rate(env, act)
                out.append(f"hint {hint}{' eighths' if piece else ''}: {us:.1f} ({bps * n / us / 8e6:.3f})")
        env._dims.launch_hint = env._dims.xcd_piece = 0
        L.ts_tuning(_cabi.TUNE_SMALL_BPW, before)
        print(f"   bpw {bpw}: " + "   ".join(out), flush=True)
    del env, act
    torch.cuda.empty_cache()
