#!/usr/bin/env python3
"""Development tool: are "slow placements" a TLB effect?  Builds several environments of one config
in one process (different allocations), times each with HIP events, then steps each a fixed number
of times so that a rocprofv3 --pmc run of this script can attribute per-dispatch counters to each
environment by dispatch order (see tools/r02_tlb.sh).

    python tools/placement_tlb_probe.py cfg4 6
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from tiler_slider_amd import VecTilerSliderEnv, _cabi  # noqa: E402

cfgname, trials = sys.argv[1], int(sys.argv[2])
cfg = dict(bench.CONFIGS[cfgname]) if cfgname in bench.CONFIGS else None
if cfg is None:
    S, T, K, N = (int(x) for x in cfgname.split(","))
    cfg = dict(size=S, tiles=T, obstacles=K, boards=N)
n = cfg["boards"]
dev = torch.device("cuda", 0)
act = torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev)
envs = []
for k in range(trials):
    env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=3,
                                   multi_color=True, max_steps=2**30, device=dev, auto_reset=True)
    env.reset()
    envs.append(env)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for k, env in enumerate(envs):
    ts = []
    for _ in range(3):
        env.step_async(act)
        e0.record()
        for i in range(10):   # exactly 11 launches per repetition, 33 per environment
            env.step_async(act)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    print(f"env {k}: {statistics.median(ts):8.2f} us  obs@{env._obs.data_ptr():#x}", flush=True)
