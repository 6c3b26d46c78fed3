#!/usr/bin/env python3
"""Development tool: small batches are launch-bound from Python; hipGraph replay of K steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiler_slider_amd import VecTilerSliderEnv

for n in (1024, 4096, 16384, 65536, 262144):
    env = VecTilerSliderEnv.random(n, size=4, num_tiles=2, num_obstacles=2, seed=1, multi_color=True, max_steps=2**30,
                                   auto_reset=True)
    env.reset()
    acts = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(64)]
    K = 64 * 20
    for i in range(64):
        env.step_async(acts[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        env.step_async(acts[i & 63])
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / K * 1e6
    g = env.capture_steps(acts)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(20):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / K * 1e6
    print(f"{n:7d} boards: eager {eager:6.2f} us/step ({n / eager * 1e6:.2e} steps/s)   hipGraph {graph:6.2f} us/step "
          f"({n / graph * 1e6:.2e} steps/s)", flush=True)
