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
        print(f"{name}: step {i}: {len(bad)} boards with wrong obs (first {bad[:8].tolist()}), {len(badf)} wrong flags", flush=True)
