#!/usr/bin/env python3
"""Experiment: a 32 us launch spends ~2 us before its first observation store (state loads + transition of the first
round of waves) and as long draining.  Do P part-size environments stepped on P streams hide each other's ramps?
Eager launches from Python become launch-bound from two parts on, so every variant is also replayed from a hipGraph
(20 steps per graph, fork / join per step).

    python tools/two_stream_split.py [boards] [size] [tiles] [obstacles]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tiler_slider_amd import VecTilerSliderEnv, _cabi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T = int(sys.argv[3]) if len(sys.argv) > 3 else 2
K = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device("cuda", 0)
L = _cabi.lib()
L.ts_tuning(0, 0)
STEPS = 20


def build(parts):
    envs, acts, streams = [], [], []
    for p in range(parts):
        e = VecTilerSliderEnv.random(n // parts, size=S, num_tiles=T, num_obstacles=K, seed=0x715311DE, multi_color=True,
                                     max_steps=2**30, auto_reset=True, device=dev, board_offset=p * (n // parts))
        e.reset()
        envs.append(e)
        acts.append([torch.randint(0, 4, (n // parts,), dtype=torch.uint8, device=dev) for _ in range(4)])
        streams.append(torch.cuda.Stream(dev))
    return envs, acts, streams


def issue(envs, acts, streams, main):
    """STEPS steps of every part: part p runs on stream p, forked from and joined into `main` once per sequence."""
    fork = torch.cuda.Event()
    fork.record(main)
    for e, a, s in zip(envs, acts, streams):
        s.wait_event(fork)
        with torch.cuda.stream(s):
            for i in range(STEPS):
                e.step_async(a[i & 3])
        main.wait_stream(s)


for parts in (1, 2, 3, 4, 8):
    if n % parts:
        continue
    envs, acts, streams = build(parts)
    torch.cuda.synchronize()
    cap = torch.cuda.Stream(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(graph, stream=cap):
            issue(envs, acts, streams, cap)
    res = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for r in range(10):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / (10 * STEPS))
    print(f"{n} {S}x{S} boards as {parts} part(s) on {parts} stream(s), hipGraph replay: {sorted(res)[2]:7.2f} us per step of all boards", flush=True)
    del graph, envs, acts, streams
L.ts_tuning(0, 1048576)
