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
