#!/usr/bin/env python3
"""Development tool: run every build/variants/*.so on one bench config and compare each with the
CPU oracle (not only with each other, as variant_bench.py does)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from oracle import binding as orc  # noqa: E402
from tiler_slider_amd import VecTilerSliderEnv, _cabi  # noqa: E402

# usage: check_variants_vs_oracle.py <config | S,T,K> <boards> [variant,variant,...]
# TS_SHOW_DIFF=1: print the first wrong tiles / floats;  TS_WS_ANALYZE=1: the wrong-slide statistics of round 3 (8x8 only)


def _dest(p, blk, d):
    """slide destination of a lone tile on an 8x8 board (cell ids, python ints) and its lane / side masks"""
    r, c = divmod(p, 8)
    lane = (0x0101010101010101 << c) if d < 2 else (0xff << (8 * r))
    below, st = (1 << p) - 1, (8 if d < 2 else 1)
    side = lane & (below if d in (0, 2) else ~(below | (1 << p)) & (2**64 - 1))
    bb = blk & side
    if d in (0, 2):
        return (bb.bit_length() - 1 + st) if bb else (c if d < 2 else 8 * r)
    return ((bb & -bb).bit_length() - 1 - st) if bb else (56 + c if d < 2 else 8 * r + 7)


def _slide_with_lane(p, occ, blk, d, lane):
    """ts_core.h slide_cell with the lane mask given (the hypothesis under test supplies a wrong one)"""
    r, c = divmod(p, 8)
    st, neg = (8 if d < 2 else 1), d in (0, 2)
    below = (1 << p) - 1
    side = lane & (below if neg else ~(below | (1 << p)) & (2**64 - 1))
    bb = blk & side
    if neg:
        dest = (bb.bit_length() - 1 + st) if bb else (c if d < 2 else 8 * r)
        span = side & ~((1 << dest) - 1)
    else:
        dest = ((bb & -bb).bit_length() - 1 - st) if bb else (56 + c if d < 2 else 8 * r + 7)
        span = side & (((2 << dest) - 1) if dest >= 0 else 0)  # a wrong lane mask can put an "obstacle" in front of the board
    ahead = bin(occ & span).count("1")
    return dest + st * ahead if neg else dest - st * ahead


def wrong_slide_stats(name, init, blk, act, got, want):
    """Which tiles are wrong, where, and do they follow from a STALE lane mask (the span mask the previous tile's
    iteration left in the same register pair)?"""
    badb = np.flatnonzero((got != want).any(axis=0))
    if not len(badb):
        print(f"{name}: wrong-slide statistics: no wrong positions", flush=True)
        return
    waves, tiles_per_wave, horiz, match_a, match_below, total = {}, {}, 0, 0, 0, 0
    multi = 0
    for b in badb.tolist():
        ts_ = np.flatnonzero(got[:, b] != want[:, b]).tolist()
        multi += len(ts_) > 1
        d = int(act[b])
        waves.setdefault(b // 64, []).append(b % 64)
        tiles_per_wave.setdefault(b // 64, set()).update(ts_)
        horiz += d >= 2
        bl = int(blk[0, b]) | (int(blk[1, b]) << 32)
        cells = [int(x) for x in init[:, b]]
        occ = 0
        for x in cells:
            occ |= 1 << x
        t = ts_[0]  # the first wrong tile: later ones may be consequences (the observation only; positions are independent)
        total += 1
        if t > 0:
            dp = _dest(cells[t - 1], bl, d)
            stale = ((2 << dp) - 1) if d in (1, 3) else ((2**64 - 1) << dp) & (2**64 - 1)
            match_a += _slide_with_lane(cells[t], occ, bl, d, stale) == int(got[t, b])
        match_below += _slide_with_lane(cells[t], occ, bl, d, (1 << cells[t]) - 1) == int(got[t, b])
    one_index = sum(1 for w in tiles_per_wave.values() if len(w) == 1)
    hist = np.bincount([t for w in tiles_per_wave.values() for t in w], minlength=init.shape[0])
    print(f"{name}: wrong-slide statistics over {len(badb)} wrong boards in {len(waves)} waves "
          f"(of {got.shape[1] // 64}; first wave {min(waves)}, last {max(waves)}):", flush=True)
    print(f"    horizontal moves: {horiz} of {len(badb)} (vertical: {len(badb) - horiz});  boards with more than one wrong tile: {multi}", flush=True)
    print(f"    waves whose wrong boards all share ONE tile index: {one_index} of {len(waves)};  tile-index histogram over waves: {hist.tolist()}", flush=True)
    print(f"    wrong result == slide with the PREVIOUS tile's span mask as lane mask (stale v[20:21]): {match_a} of {total}", flush=True)
    print(f"    wrong result == slide with lane mask (1 << p) - 1 (the register's NEXT value): {match_below} of {total}", flush=True)
    lanes_per_wave = np.array([len(v) for v in waves.values()])
    print(f"    wrong lanes per affected wave: min {lanes_per_wave.min()} median {int(np.median(lanes_per_wave))} max {lanes_per_wave.max()}", flush=True)
    wv = np.array(sorted(waves))
    print(f"    affected waves by wave-in-block: {np.bincount(wv % 4, minlength=4).tolist()};  by block mod 8: {np.bincount((wv // 4) % 8, minlength=8).tolist()}", flush=True)

cfgname, n = sys.argv[1], int(sys.argv[2])
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
if "," in cfgname:
    S_, T_, K_ = (int(x) for x in cfgname.split(","))
    cfg = dict(size=S_, tiles=T_, obstacles=K_)
else:
    cfg = bench.CONFIGS[cfgname]
VDIR = os.path.join(ROOT, "build", "variants")
blk, init, tgt = orc.generate(cfg["size"], cfg["tiles"], cfg["tiles"], cfg["obstacles"], n, seed=bench.LEVEL_SEED)
acts = [orc.fill_actions(n, seed=bench.ACTION_SEED, step_index=i) for i in range(3)]
ref = orc.OracleBatch(cfg["size"], True, 2**30, blk, init, tgt)
ref.reset()
wants = [ref.step(a, mode=orc.MODE_AUTORESET) for a in acts]
for name in json.load(open(os.path.join(VDIR, "manifest.json"))):
    if only and name not in only:
        continue
    L = C.CDLL(os.path.join(VDIR, f"{name}.so"))
    L.ts_step.argtypes = [C.POINTER(_cabi.Dims), C.POINTER(_cabi.State), C.c_void_p, C.c_uint32, C.POINTER(_cabi.StepOut), C.c_void_p]
    env = VecTilerSliderEnv.from_arrays(cfg["size"], blk, init, tgt, multi_color=True, max_steps=2**30, auto_reset=True)
    env.reset()
    st = torch.cuda.current_stream().cuda_stream
    for i, a in enumerate(acts):
        t = torch.from_numpy(a).cuda()
        rc = L.ts_step(C.byref(env._dims), C.byref(env._state), t.data_ptr(), env._mode, C.byref(env._out), st)
        if rc != 0:
            L.ts_last_hip_error.restype = C.c_int32
            print(f"{name}: step {i}: ts_step returned {rc} (hipError {L.ts_last_hip_error()}): launch refused", flush=True)
            break
        torch.cuda.synchronize()
        obs = env._obs.cpu().numpy()
        bad = np.flatnonzero((obs != wants[i]["obs"]).reshape(n, -1).any(axis=1))
        badf = np.flatnonzero(env._flags.cpu().numpy() != wants[i]["flags"])
        if i == 0 and os.environ.get("TS_WS_ANALYZE") and cfg["size"] == 8:
            ref0 = orc.OracleBatch(8, True, 2**30, blk, init, tgt)
            ref0.reset()
            ref0.step(acts[0], mode=orc.MODE_AUTORESET)
            wrong_slide_stats(name, init, blk, acts[0], env._pos.cpu().numpy(), ref0.pos)
            if os.environ.get("TS_WS_DUMP"):  # raw material for offline hypothesis tests: the wrong boards of this variant
                gotp = env._pos.cpu().numpy()
                wb = np.flatnonzero((gotp != ref0.pos).any(axis=0))
                np.savez_compressed(os.path.join(os.environ["TS_WS_DUMP"], f"ws_dump_{name}.npz"), boards=wb, blk=blk[:, wb], init=init[:, wb],
                                    act=acts[0][wb], got=gotp[:, wb], want=ref0.pos[:, wb])
        if i == 0 and os.environ.get("TS_SHOW_DIFF"):
            ref0 = orc.OracleBatch(cfg["size"], True, 2**30, blk, init, tgt)
            ref0.reset()
            ref0.step(acts[0], mode=orc.MODE_AUTORESET)
            badp = np.flatnonzero((env._pos.cpu().numpy() != ref0.pos).any(axis=0))
            print(f"{name}: step 0: {len(badp)} boards with wrong positions in HBM (first {badp[:8].tolist()})", flush=True)
            gotp = env._pos.cpu().numpy()
            S_ = cfg["size"]
            for b in badp[:12].tolist():
                ts_ = np.flatnonzero(gotp[:, b] != ref0.pos[:, b]).tolist()
                for t_ in ts_:
                    # hypothesis A: the tile was slid twice (a second wave processed the same board)
                    twice = orc.OracleBatch(S_, True, 2**30, blk[:, b:b + 1].copy(), ref0.pos[:, b:b + 1].copy(), tgt[:, b:b + 1].copy())
                    twice.reset()
                    twice.step(acts[0][b:b + 1].copy(), mode=orc.MODE_AUTORESET)
                    print(f"    board {b} action {acts[0][b]} tile {t_}: start {init[t_, b]} want {ref0.pos[t_, b]} got {gotp[t_, b]}"
                          f"  (slid twice would be {twice.pos[t_, 0]}; start cells of the board: {sorted(init[:, b].tolist())})", flush=True)
        print(f"{name}: step {i}: {len(bad)} boards with wrong obs (first {bad[:8].tolist()}), {len(badf)} wrong flags", flush=True)
        if len(bad) and i == 0 and os.environ.get("TS_SHOW_DIFF"):
            for b in bad[:3].tolist() + bad[-2:].tolist():
                got, want = obs[b].reshape(-1), wants[i]["obs"][b].reshape(-1)
                idx = np.flatnonzero(got != want)
                print(f"    board {b} (wave {b // 64}, lane {b % 64}): {len(idx)} floats differ; (flat index: got/want) "
                      + ", ".join(f"{k}: {got[k]:g}/{want[k]:g}" for k in idx[:24].tolist()), flush=True)
            lanes = np.bincount(bad % 64, minlength=64)
            waves_in_block = np.bincount((bad // 64) % 4, minlength=4)
            print(f"    wrong boards by lane: {lanes.tolist()}", flush=True)
            print(f"    wrong boards by wave-in-block: {waves_in_block.tolist()}; by block mod 8: {np.bincount((bad // 256) % 8, minlength=8).tolist()}", flush=True)
