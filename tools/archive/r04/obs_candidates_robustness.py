#!/usr/bin/env python3
"""Round 4: does the default observation-buffer search (obs_candidates, spaced candidates) find the fast class wherever the
process's allocations happen to stand?  cfg2 environments are constructed one after another in ONE process with junk
allocations of varying size kept alive in between (as a trainer's own buffers would be); per environment the search report and
the step time over 200 launches."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiler_slider_amd import VecTilerSliderEnv
from tiler_slider_amd.vec_env import _ContiguousBuffer

dev = torch.device("cuda", 0)
rng = random.Random(int(os.environ.get("SEED", "5")))
junk = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 1 << 20
act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev) for _ in range(4)]
cand = os.environ.get("CANDIDATES")
print(f"# candidates {cand or 'default'}, seed {os.environ.get('SEED', '5')}")
import time
for trial in range(int(os.environ.get("TRIALS", "10"))):
    mib = rng.choice([0, 200, 700, 1500, 2600, 4100, 6000])
    if mib:
        junk.append(_ContiguousBuffer(mib << 20, dev) if trial & 1 else torch.empty(mib << 20, dtype=torch.uint8, device=dev))
    t0 = time.perf_counter()
    env = VecTilerSliderEnv.random(n, size=5, num_tiles=2, num_obstacles=3, seed=3, multi_color=True, max_steps=2**30, auto_reset=True,
                                   with_onehot=True, with_reward=True, obs_candidates=None if cand is None else int(cand))
    torch.cuda.synchronize()
    built = time.perf_counter() - t0
    env.reset()
    for i in range(300):
        env.step_async(act[i & 3])
    e0.record()
    for i in range(200):
        env.step_async(act[i & 3])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    rep = env.observation_placement_report
    print(f"trial {trial}: junk +{mib:5d} MiB | {us:6.1f} us/step (frac {866123776 / us / 8e6:.3f}) | built in {built * 1e3:.0f} ms | search {[r['us_per_step'] for r in rep] if isinstance(rep, list) else rep}", flush=True)
    del env
    torch.cuda.empty_cache()
