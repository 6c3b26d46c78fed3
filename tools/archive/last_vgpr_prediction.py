#!/usr/bin/env python3
"""Round 3, wrong-slide study: the scanner (tools/scan_last_vgpr.py) predicts which kernels of a build can slide a tile
past its row.  This runs one predicted shape against the oracle (default: 8x8, 5 tiles, plain outputs, cache-resident
batch = k_small<8, 5, false, false>, whose row-mask shift reads v47 of 48 allocated VGPRs in the round-2 build).

    python tools/last_vgpr_prediction.py [S T K N]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import binding as orc  # noqa: E402
from tiler_slider_amd import VecTilerSliderEnv  # noqa: E402

S, T, K, N = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 5, 10, 300_000)
blk, init, tgt = orc.generate(S, T, T, K, N, seed=0x715311DE)
ref = orc.OracleBatch(S, True, 2**30, blk, init, tgt)
env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=True, max_steps=2**30, auto_reset=True)
ok = np.array_equal(env.reset().cpu().numpy(), ref.reset())
print(f"{S}x{S}, {T} tiles, {N} boards, plain outputs: reset {'ok' if ok else 'MISMATCH'}", flush=True)
for i in range(3):
    start = ref.pos.copy()
    a = orc.fill_actions(N, seed=0xAC710005, step_index=i)
    obs, done, info = env.step(torch.from_numpy(a))
    ref.step(a, mode=orc.MODE_AUTORESET)
    got = env.positions.cpu().numpy()
    bad = np.flatnonzero((got != ref.pos).any(axis=0))
    horiz = int((a[bad] >= 2).sum())
    print(f"step {i}: {len(bad)} boards with wrong positions ({horiz} of them moved horizontally, {len(bad) - horiz} vertically); "
          f"waves affected: {len(set((bad // 64).tolist()))}", flush=True)
    if len(bad):
        env._pos.copy_(torch.from_numpy(ref.pos).to(env._pos.device))  # continue from the oracle's state
sys.exit(0)
