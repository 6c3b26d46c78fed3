#!/usr/bin/env python3
"""Round 4: the write-front mapping of out-of-cache launches on PHYSICALLY CONTIGUOUS output buffers
(hipExtMallocWithFlags(hipDeviceMallocContiguous): the same physical layout in every process, so an A/B here is an
experiment and not an allocation lottery).  Full grid over
    xcd_piece (ts_dims)            1 = one contiguous eighth of the batch per XCD, P = pieces of P one-wave blocks
    skew (ts_tuning, experiment)   XCD x starts x * skew blocks into its eighth / piece
    launch_hint (ts_dims)          resident blocks per CU relative to the policy
then, at the best cells, emit_edges and the block order inside a piece.
usage: contig_sweep.py [config | S,T,K,N[,onehot]] ...     env: TS_SWEEP_MEM=torch for the caching allocator instead"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from tiler_slider_amd import VecTilerSliderEnv, _cabi  # noqa: E402
from tiler_slider_amd.vec_env import _ContiguousBuffer  # noqa: E402

L = _cabi.lib()
MEM = os.environ.get("TS_SWEEP_MEM", "contiguous")
PIECES = tuple(int(x) for x in os.environ.get("TS_SWEEP_PIECES", "1,8,16,32,64,128,256").split(","))
SKEWS = tuple(int(x) for x in os.environ.get("TS_SWEEP_SKEWS", "0,1,2,3,5,7,11").split(","))
HINTS = tuple(int(x) for x in os.environ.get("TS_SWEEP_HINTS", "-4,-2,0,2,4,8").split(","))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def rate(env, act, reps=20, rounds=3):
    ts = []
    for r in range(rounds):
        for i in range(3):
            env.step_async(act[i & 3])
        e0.record()
        for i in range(reps):
            env.step_async(act[i & 3])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)


for spec in sys.argv[1:] or ["cfg2", "cfg4"]:
    if "," in spec:
        v = [int(x) for x in spec.split(",")]
        cfg = dict(size=v[0], tiles=v[1], obstacles=v[2], boards=v[3], onehot=len(v) > 4 and bool(v[4]), reward=len(v) > 4 and bool(v[4]))
    else:
        cfg = bench.CONFIGS[spec]
    n = cfg["boards"]
    bps = bench.algorithmic_bytes_per_board_step(cfg["size"], cfg["tiles"], cfg["onehot"], cfg["reward"])
    env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=bench.LEVEL_SEED,
                                   multi_color=True, max_steps=2**30, auto_reset=True, with_reward=cfg["reward"], with_onehot=cfg["onehot"],
                                   placement_trials=0, output_memory=MEM)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
    frac = lambda us: bps * n / us / 1e3 / 8000
    d = env._dims
    base = rate(env, act)
    print(f"\n=== {spec}: {n} boards, {bps * n / 1e6:.0f} MB per launch, output_memory={MEM}; library policy: {base:.1f} us ({frac(base):.3f})", flush=True)
    grid = {}
    for hint in HINTS:
        print(f"  launch_hint {hint:+d}: us per step, rows = xcd_piece {PIECES}, columns = skew {SKEWS}")
        for piece in PIECES:
            row = f"    piece {piece:4d}:"
            for skew in SKEWS:
                if piece > 1 and skew >= piece:
                    row += "      - "
                    continue
                d.launch_hint, d.xcd_piece = hint, piece
                L.ts_tuning(_cabi.TUNE_XCD_SKEW, skew if piece > 1 else skew * 97)  # eighths: x * 97 * skew blocks, a longer reach
                us = rate(env, act, reps=12, rounds=2)
                grid[(hint, piece, skew)] = us
                row += f" {us:7.1f}"
            print(row, flush=True)
    L.ts_tuning(_cabi.TUNE_XCD_SKEW, 0)
    top = sorted(grid, key=grid.__getitem__)[:5]
    print("  best cells (hint, piece, skew): " + "  ".join(f"{k} {grid[k]:.1f} ({frac(grid[k]):.3f})" for k in top))
    for hint, piece, skew in top[:3]:
        d.launch_hint, d.xcd_piece = hint, piece
        L.ts_tuning(_cabi.TUNE_XCD_SKEW, skew if piece > 1 else skew * 97)
        row = f"  at {(hint, piece, skew)}: confirm {rate(env, act):.1f};  emit_edges 1..4:"
        for e in (1, 2, 3, 4):
            d.emit_edges = e
            row += f" {rate(env, act, reps=12, rounds=2):.1f}"
        d.emit_edges = 0
        row += ";  order 1 (bit-reversed), 2 (descending):"
        for o in (1, 2):
            L.ts_tuning(_cabi.TUNE_XCD_ORDER, o)
            row += f" {rate(env, act, reps=12, rounds=2):.1f}"
        L.ts_tuning(_cabi.TUNE_XCD_ORDER, 0)
        if cfg["size"] > 8:
            row += ";  lines_lanes 8, 16:"
            for ln in (8, 16):
                d.lines_lanes = ln
                row += f" {rate(env, act, reps=12, rounds=2):.1f}"
            d.lines_lanes = 0
        print(row, flush=True)
    L.ts_tuning(_cabi.TUNE_XCD_SKEW, 0)
    d.launch_hint = d.xcd_piece = 0
    print(f"  library policy again: {rate(env, act):.1f}", flush=True)
    del env, act
    torch.cuda.empty_cache()
