#!/usr/bin/env python3
"""Round 4: where do the ~65 us between the HIP-event time and the wall clock of bench.py's 20-step timed region go?"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi

dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["cfg1"]
n = cfg["boards"]
env = VecTilerSliderEnv.random(n, size=4, num_tiles=2, num_obstacles=2, seed=bench.LEVEL_SEED, multi_color=True, max_steps=2**30, auto_reset=True)
env.reset()
ring = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev) for _ in range(16)]
for i in range(3000):
    env.step_async(ring[i & 15])
torch.cuda.synchronize()
pc = time.perf_counter
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def med(f, reps=30):
    return statistics.median(f() for _ in range(reps)) * 1e6


def t_sync_idle():
    torch.cuda.synchronize(dev); t = pc(); torch.cuda.synchronize(dev); return pc() - t
def t_record():
    torch.cuda.synchronize(dev); t = pc(); ev0.record(); return pc() - t
def t_launch():
    torch.cuda.synchronize(dev); t = pc(); env.step_async(ring[0]); return pc() - t
def t_one_step_wall(wait):
    torch.cuda.synchronize(dev); t = pc(); env.step_async(ring[0]); ev1.record()
    if wait == "spin":
        while not ev1.query(): pass
    torch.cuda.synchronize(dev); return pc() - t
def t_k_steps(k, wait):
    torch.cuda.synchronize(dev); t = pc(); ev0.record()
    for i in range(k): env.step_async(ring[i & 15])
    ev1.record()
    if wait == "spin":
        while not ev1.query(): pass
    torch.cuda.synchronize(dev); w = pc() - t
    return w, ev0.elapsed_time(ev1) * 1e-3

print(f"synchronize on an idle device      {med(t_sync_idle):7.1f} us")
print(f"event record (host time)           {med(t_record):7.1f} us")
print(f"one step_async (host time)         {med(t_launch):7.1f} us")
print(f"one step, launch -> synchronize    {med(lambda: t_one_step_wall('block')):7.1f} us (blocking)   {med(lambda: t_one_step_wall('spin')):7.1f} us (spin on the event first)")
for k in (1, 5, 20, 100):
    for wait in ("block", "spin"):
        rs = [t_k_steps(k, wait) for _ in range(15)]
        w = statistics.median(r[0] for r in rs) * 1e6; d = statistics.median(r[1] for r in rs) * 1e6
        print(f"{k:4d} steps ({wait:5s}): wall {w:8.1f} us  events {d:8.1f} us  difference {w - d:6.1f} us  wall per step {w / k:6.2f}")
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        for i in range(20): env.step_async(ring[i & 15])
def t_graph():
    torch.cuda.synchronize(dev); t = pc(); g.replay(); torch.cuda.synchronize(dev); return pc() - t
print(f"  20 steps as one hipGraph replay: wall {med(t_graph, 15):8.1f} us")
