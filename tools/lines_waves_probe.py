#!/usr/bin/env python3
"""Round 5: waves per block of k_lines beyond the cache x resident blocks x block mapping, by shape (ts_tuning(TS_TUNE_LINES_WAVES)).

    python tools/lines_waves_probe.py [S,T,K,N ...]        (GPU box)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
L = _cabi.lib()
PIECES = [int(x) for x in os.environ.get("PIECES", "4,8,12,16,24").split(",")]
HINTS = [int(x) for x in os.environ.get("HINTS", "-2,0,2").split(",")]


def rate(env, act, steps=30):
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


DEFAULT = ["15,32,24,262144", "15,32,24,131072", "15,32,24,524288", "14,20,20,212480", "16,16,24,162560", "15,8,24,262144", "12,8,16,289280", "13,3,10,246528",
           "11,6,8,344320", "16,40,30,162560", "16,64,20,162560", "15,24,24,185088", "20,10,40,103936", "24,30,60,72192", "32,32,100,40448"]
for spec in sys.argv[1:] or DEFAULT:
    S, T, K, N = (int(x) for x in spec.split(","))
    env = VecTilerSliderEnv.random(N, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True, obs_candidates=0)
    env.reset()
    act = [torch.randint(0, 4, (N,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    for i in range(100):
        env.step_async(act[i & 3])
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
    d = _cabi.describe_launch(env._dims, _cabi.OP_STEP, _cabi.OUT_OBS | _cabi.OUT_FLAGS)
    base = rate(env, act)
    print(f"{S}x{S}, {T} tiles, {N} boards, {bps * N / 1e6:.0f} MB: policy {base:.1f} us ({bps * N / base / 8e6:.3f})  {d['name']} bpw {d['boards_per_wave']} waves/block {d['waves_per_block']} "
          f"blocks/CU {d['blocks_per_cu']} piece {d['xcd_piece']}", flush=True)
    for w in (1, 2, 4):
        before = L.ts_tuning(_cabi.TUNE_LINES_WAVES, w)
        best = (1e9, None)
        rows = []
        for piece in PIECES:
            env._dims.xcd_piece = piece
            out = []
            for hint in HINTS:
                env._dims.launch_hint = hint
                us = rate(env, act, 20)
                best = min(best, (us, (piece, hint)))
                out.append(f"{us:.1f}")
            rows.append(f"piece {piece}: " + " / ".join(out))
        env._dims.launch_hint = env._dims.xcd_piece = 0
        L.ts_tuning(_cabi.TUNE_LINES_WAVES, before)
        print(f"   {w} waves per block (hint {' / '.join(map(str, HINTS))}): " + "   ".join(rows) + f"   best {best[0]:.1f} ({bps * N / best[0] / 8e6:.3f}) at piece {best[1][0]} hint {best[1][1]}", flush=True)
    del env, act
    torch.cuda.empty_cache()
