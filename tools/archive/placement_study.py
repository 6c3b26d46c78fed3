#!/usr/bin/env python3
"""Development tool: does the step time depend on WHERE the buffers were allocated?

Round 1 quoted "box-to-box variance up to 20 %" for cfg2 / cfg4.  Round 2's first GPU session
showed the spread between PROCESSES on one box (bench.py cfg4: 130.9 / 144.8 / 129.7 us in three
consecutive runs, identical code; three kernel versions A/B'd inside one process: equal to 0.3 %).
This script allocates the environment several times inside one process, keeping the earlier ones
alive so that every trial gets different addresses, and times (a) the step kernel, (b) a plain
torch fill of the same observation buffer, printing the device pointers next to the times.

    python tools/placement_study.py cfg4 [trials]
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from tiler_slider_amd import VecTilerSliderEnv, _cabi  # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = bench.CONFIGS[cfgname]
n = cfg["boards"]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
ring = []
for i in range(16):
    t = torch.empty(n, dtype=torch.uint8, device=dev)
    _cabi.check(_cabi.lib().ts_fill_actions(n, bench.ACTION_SEED, 0, i, t.data_ptr(), stream), "fill")
    ring.append(t)


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts)


def one(tag):
    env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"],
                                   seed=bench.LEVEL_SEED, multi_color=True, max_steps=2**30, device=dev, auto_reset=True,
                                   with_reward=cfg["reward"], with_onehot=cfg["onehot"])
    env.reset()
    for i in range(20):
        env.step_async(ring[i & 15])
    us = timed(lambda i: env.step_async(ring[i & 15]), 200)
    fill = timed(lambda i: env._obs.fill_(1.5), 50)
    oh = env._onehot.data_ptr() if env._onehot is not None else 0
    print(f"{tag}: step {us:8.2f} us   fill(obs) {fill:8.2f} us   obs@{env._obs.data_ptr():#x} (mod 2MiB {env._obs.data_ptr() % (2 << 20):#x})"
          f"  onehot@{oh:#x}  pos@{env._pos.data_ptr():#x}", flush=True)
    return env


print(f"{cfgname}: {n} boards; every trial keeps its buffers alive", flush=True)
keep = [one(f"trial {k}") for k in range(trials)]
print("re-timing the first and the last environment (same buffers, later in the process)", flush=True)
for k in (0, trials - 1):
    env = keep[k]
    us = timed(lambda i: env.step_async(ring[i & 15]), 200)
    print(f"trial {k} again: step {us:8.2f} us", flush=True)
del keep
torch.cuda.empty_cache()
print("after freeing everything (allocator cache emptied):", flush=True)
for k in range(2):
    one(f"fresh {k}")
