#!/usr/bin/env python3
"""Times the REFERENCE's own Python step() loop on one core of this container (the reference cannot travel to the
GPU box, so these numbers are carried into bench.py / DESIGN.md as recorded constants).  Loads the reference the way
tests/golden/make_env_golden.py does.  Usage: python tools/time_reference_python.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from make_env_golden import load_reference_environment  # noqa: E402

Env, Factory, GameState = load_reference_environment()
moves = [GameState.Move.from_int(i) for i in range(4)]
for name, S, T, K in (("cfg0 3x3 T1 K0", 3, 1, 0), ("cfg1 4x4 T2 K2", 4, 2, 2), ("cfg2 5x5 T2 K3", 5, 2, 3), ("cfg4 15x15 T32 K24", 15, 32, 24)):
    envs = []
    for seed in range(64):
        f = Factory.create_simple_env(size=S, num_tiles=T, num_obstacles=K, seed=seed)
        e = Env(size=S, blocked_locations=f.blocked_locations, initial_locations=f.initial_locations,
                target_locations=f.target_locations, multi_color=True, max_steps=2**30)
        e.reset()
        envs.append(e)
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 4, size=4096)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        for e in envs:
            for a in acts[:64]:
                if e.done:
                    e.reset()
                e.step(moves[a])
            n += 64
    dt = time.perf_counter() - t0
    print(f"{name}: {n / dt:10.0f} env-steps/s on one core ({n} steps in {dt:.1f} s)", flush=True)
