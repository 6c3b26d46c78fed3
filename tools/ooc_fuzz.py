#!/usr/bin/env python3
"""Development tool: every kernel family at a batch just beyond the Infinity Cache (where the out-of-cache launch
policies apply: nontemporal stores, one-wave blocks, bounded residency, half waves), against the oracle: reset + 3 steps,
plain and with the optional outputs; then an `onehot` leg — the fused step + one-hot + reward launch (cfg2's kernel) with
~300 MB of planes per step, plane for plane against the oracle.  Usage: python tools/ooc_fuzz.py [--onehot-only]"""
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

orc.lib().tso_set_num_threads(bench.host_cpu_share())
rng = np.random.default_rng(7)
cases = []
for S in range(1, 9):
    C = S * S
    for T in sorted({1, min(2, C), min(3, C), min(5, C), min(8, C), min(12, C)}):
        cases.append((S, T, min(max(0, C - 2 * T) // 3, 6)))
for S, T in ((9, 4), (10, 17), (12, 8), (15, 32), (16, 40), (17, 3), (20, 33), (24, 30), (32, 64),
             # round 4: tiles dealt over 4 / 8 lanes (7x7, 8x8), 8 lanes up to 13x13, 32 lanes per board, one board per wave
             (7, 20), (7, 40), (8, 20), (8, 28), (8, 40), (11, 6), (13, 16), (20, 4), (26, 100), (28, 8), (30, 16), (32, 4)):
    cases.append((S, T, S))
bad = 0
t0 = time.time()
for S, T, K in ([] if "--onehot-only" in sys.argv else cases):
    C = S * S
    n = (300_000_000 // (12 * C)) + int(rng.integers(1, 200))  # ~300 MB of observation: nontemporal path, ragged N
    mc = bool(rng.integers(0, 2))
    if K + 2 * T <= C:
        blk, init, tgt = orc.generate(S, T, T, K, n, seed=100 + S * 31 + T)
    else:
        blk, init, _ = orc.generate(S, T, 0, K, n, seed=100 + S * 31 + T)
        _, _, tgt = orc.generate(S, 0, T, 0, n, seed=200 + S * 31 + T)
    for extras in (False, True):
        ref = orc.OracleBatch(S, mc, 5, blk, init, tgt)
        env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=5, auto_reset=True,
                                            with_reward=extras, with_valid_moves=extras)
        ok = np.array_equal(env.reset().cpu().numpy(), ref.reset())
        for i in range(3):
            a = orc.fill_actions(n, seed=9, step_index=i)
            obs, done, info = env.step(torch.from_numpy(a))
            w = ref.step(a, mode=orc.MODE_AUTORESET, reward=extras, valid=extras)
            ok &= np.array_equal(obs.cpu().numpy(), w["obs"]) and np.array_equal(info["flags"].cpu().numpy(), w["flags"])
            ok &= np.array_equal(env.positions.cpu().numpy().astype(np.int64), ref.pos.astype(np.int64))
            if extras:
                ok &= np.array_equal(info["reward"].cpu().numpy(), w["reward"]) and np.array_equal(env._valid.cpu().numpy(), w["valid"])
        bad += not ok
        print(f"S={S:2d} T={T:3d} K={K:2d} mc={int(mc)} N={n:9d} extras={int(extras)}: {'ok' if ok else 'MISMATCH'}  [{time.time() - t0:.0f} s]", flush=True)
        del env, ref
# one-hot leg: batch sized by the PLANES (Ch * C floats per board), so that the one-hot stream is the out-of-cache one
for S, T, K in cases:
    C = S * S
    mc = bool(rng.integers(0, 2))
    Ch = 1 + 2 * T if mc else 3
    n = (300_000_000 // (4 * C * Ch)) + int(rng.integers(1, 200))
    if K + 2 * T <= C:
        blk, init, tgt = orc.generate(S, T, T, K, n, seed=300 + S * 31 + T)
    else:
        blk, init, _ = orc.generate(S, T, 0, K, n, seed=300 + S * 31 + T)
        _, _, tgt = orc.generate(S, 0, T, 0, n, seed=400 + S * 31 + T)
    ref = orc.OracleBatch(S, mc, 5, blk, init, tgt)
    env = VecTilerSliderEnv.from_arrays(S, blk, init, tgt, multi_color=mc, max_steps=5, auto_reset=True,
                                        with_reward=True, with_onehot=True)
    ok = np.array_equal(env.reset().cpu().numpy(), ref.reset())
    for i in range(3):
        a = orc.fill_actions(n, seed=10, step_index=i)
        obs, done, info = env.step(torch.from_numpy(a))
        w = ref.step(a, mode=orc.MODE_AUTORESET, reward=True, onehot=True)
        ok &= np.array_equal(obs.cpu().numpy(), w["obs"]) and np.array_equal(info["flags"].cpu().numpy(), w["flags"])
        ok &= np.array_equal(info["reward"].cpu().numpy(), w["reward"]) and np.array_equal(info["onehot"].cpu().numpy(), w["onehot"])
        del w
    bad += not ok
    print(f"S={S:2d} T={T:3d} K={K:2d} mc={int(mc)} N={n:9d} onehot Ch={Ch:3d}: {'ok' if ok else 'MISMATCH'}  [{time.time() - t0:.0f} s]", flush=True)
    del env, ref
print("ooc fuzz:", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(0 if bad == 0 else 1)
