#!/usr/bin/env python3
"""Development tool: long differential run.  N boards stepped K times on the GPU and by the CPU
oracle with the same counter-based action stream; per-step flag checksums and the final state
must agree.  Usage: python tools/soak.py cfg1 300 [boards] [plain]   (plain: no optional outputs, i.e. the
k_small<EXTRAS = false> / k_lines kernels instead of the EXTRAS ones)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from oracle import binding as orc  # noqa: E402
from tiler_slider_amd import VecTilerSliderEnv  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1]]
K = int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else cfg["boards"]
plain = len(sys.argv) > 4 and sys.argv[4] == "plain"
orc.lib().tso_set_num_threads(bench.host_cpu_share())
blk, init, tgt = orc.generate(cfg["size"], cfg["tiles"], cfg["tiles"], cfg["obstacles"], n, seed=bench.LEVEL_SEED)
ref = orc.OracleBatch(cfg["size"], True, 37, blk, init, tgt)  # short episodes: wins, timeouts, autoresets all occur
extras = {} if plain else dict(with_reward=True, with_valid_moves=True, with_onehot=cfg["onehot"])
env = VecTilerSliderEnv.from_arrays(cfg["size"], blk, init, tgt, multi_color=True, max_steps=37, auto_reset=True, **extras)
env.reset()
ref.reset()
t0 = time.time()
bad = 0
for i in range(K):
    a = orc.fill_actions(n, seed=bench.ACTION_SEED, step_index=i)
    env.step_async(torch.from_numpy(a).cuda())
    full = i % 50 == 49
    want = ref.step(a, mode=orc.MODE_AUTORESET, obs=full, reward=full and not plain, valid=full and not plain,
                    onehot=full and cfg["onehot"] and not plain)
    got = env._flags.cpu().numpy()
    if not np.array_equal(got, want["flags"]):
        bad += 1
        print(f"step {i}: flags differ on {int((got != want['flags']).sum())} boards", flush=True)
    if i % 50 == 49:
        same = np.array_equal(env._obs.cpu().numpy(), want["obs"]) and np.array_equal(env.positions.cpu().numpy(), ref.pos)
        if not plain:
            same &= np.array_equal(env._reward.cpu().numpy(), want["reward"]) and np.array_equal(env._valid.cpu().numpy(), want["valid"])
        if cfg["onehot"] and not plain:
            same &= np.array_equal(env._onehot.cpu().numpy(), want["onehot"])
        bad += not same
        print(f"step {i + 1}/{K}: obs+pos+reward+valid{'+onehot' if cfg['onehot'] else ''} {'equal' if same else 'DIFFER'}; wins so far flagged this step: "
              f"{int((want['flags'] & 4 != 0).sum())}, autoresets: {int((want['flags'] & 32 != 0).sum())}  "
              f"[{time.time() - t0:.0f} s]", flush=True)
ok = bad == 0 and np.array_equal(env.positions.cpu().numpy(), ref.pos) and np.array_equal(env.step_count.cpu().numpy(), ref.step_count) \
    and np.array_equal(env._done.cpu().numpy(), ref.done)
print(f"soak {sys.argv[1]} boards={n} steps={K}: {'OK' if ok else 'MISMATCH'}")
sys.exit(0 if ok else 1)
