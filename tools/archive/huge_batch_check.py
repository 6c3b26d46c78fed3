#!/usr/bin/env python3
"""Development tool: one batch whose observation buffer is larger than 4 GiB (64-bit indexing
check): N boards of a shape, a reset and two steps, everything compared with the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import binding as orc
from tiler_slider_amd import VecTilerSliderEnv
import bench

S, T, K, N = (int(x) for x in sys.argv[1:5])
orc.lib().tso_set_num_threads(bench.host_cpu_share())
blk, init, tgt = orc.generate(S, T, T, K, N, seed=9)
ref = orc.OracleBatch(S, True, 2**30, blk, init, tgt)
env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=True, max_steps=2**30, auto_reset=True)
print(f"{N} boards {S}x{S}: obs buffer {env._obs.numel() * 4 / 2**30:.2f} GiB", flush=True)
ok = np.array_equal(env.reset().cpu().numpy(), ref.reset())
for i in range(2):
    a = orc.fill_actions(N, seed=3, step_index=i)
    obs, done, info = env.step(torch.from_numpy(a))
    w = ref.step(a, mode=orc.MODE_AUTORESET)
    ok &= np.array_equal(obs.cpu().numpy(), w["obs"]) and np.array_equal(info["flags"].cpu().numpy(), w["flags"])
    ok &= np.array_equal(env.positions.cpu().numpy().astype(np.int64), ref.pos.astype(np.int64))
    print(f"step {i}: {'ok' if ok else 'MISMATCH'}", flush=True)
sys.exit(0 if ok else 1)
