#!/usr/bin/env python3
"""Round 5: waves per block of k_small beyond the cache (ts_tuning(TS_TUNE_SMALL_WAVES)) x block mapping x resident blocks, by shape -
and of k_lines with 4 / 8 / 32 lanes per board (TS_TUNE_LINES_WAVES) for the shapes the 16-lane probe left out.

    python tools/small_waves_probe.py        (GPU box)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
L = _cabi.lib()
HINTS = (-2, 0, 2)


def rate(env, act, steps=25):
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


# (S, T, K, N, onehot)
SHAPES = [(5, 2, 3, 1 << 20, True), (4, 2, 2, 1 << 22, False), (5, 2, 3, 1703936, False), (6, 3, 4, 1 << 20, False), (7, 5, 6, 850176, False), (8, 4, 8, 651008, False),
          (3, 1, 0, 4 << 20, False), (5, 6, 3, 1703936, False), (9, 4, 9, 514304, False), (10, 5, 10, 416512, False), (12, 8, 16, 289280, False), (24, 30, 60, 72192, False),
          (20, 10, 40, 103936, False), (28, 8, 60, 53248, False)]
if len(sys.argv) > 1:  # S,T,K,N[,onehot] ...
    SHAPES = [tuple(int(x) for x in a.split(",")[:4]) + (a.count(",") > 3,) for a in sys.argv[1:]]
for S, T, K, N, oh in SHAPES:
    env = VecTilerSliderEnv.random(N, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True, obs_candidates=0,
                                   with_onehot=oh, with_reward=oh)
    env.reset()
    act = [torch.randint(0, 4, (N,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    for i in range(100):
        env.step_async(act[i & 3])
    bps = bench.algorithmic_bytes_per_board_step(S, T, oh, oh) + (T * 2 if S > 16 else 0)
    outs = _cabi.OUT_OBS | _cabi.OUT_FLAGS | ((_cabi.OUT_ONEHOT | _cabi.OUT_REWARD) if oh else 0)
    d = _cabi.describe_launch(env._dims, _cabi.OP_STEP, outs)
    base = rate(env, act)
    print(f"{S}x{S}, {T} tiles, {N} boards{' + one-hot + reward' if oh else ''}, {bps * N / 1e6:.0f} MB: policy {base:.1f} us ({bps * N / base / 8e6:.3f})  {d['name']} bpw {d['boards_per_wave']} "
          f"waves/block {d['waves_per_block']} blocks/CU {d['blocks_per_cu']} piece {d['xcd_piece']}", flush=True)
    knob = _cabi.TUNE_SMALL_WAVES if S <= 8 else _cabi.TUNE_LINES_WAVES
    p0 = max(d["xcd_piece"], 4)
    for w in (2, 4):
        before = L.ts_tuning(knob, w)
        best = (1e9, None)
        rows = []
        for piece in sorted({max(p0 // w, 2), max(p0 // w * 3 // 2, 3), p0 // w * 2, p0, p0 * 2}):
            env._dims.xcd_piece = piece
            out = []
            for hint in HINTS:
                env._dims.launch_hint = hint
                us = rate(env, act, 20)
                best = min(best, (us, (piece, hint)))
                out.append(f"{us:.1f}")
            rows.append(f"piece {piece}: " + " / ".join(out))
        env._dims.launch_hint = env._dims.xcd_piece = 0
        L.ts_tuning(knob, before)
        print(f"   {w} waves per block (hint {' / '.join(map(str, HINTS))}): " + "   ".join(rows) + f"   best {best[0]:.1f} ({bps * N / best[0] / 8e6:.3f}) at piece {best[1][0]} hint {best[1][1]}", flush=True)
    del env, act
    torch.cuda.empty_cache()
