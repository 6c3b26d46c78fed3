#!/usr/bin/env python3
"""Round 5: cfg4's launch on torch-allocator memory against physically contiguous memory, by block mapping (xcd_piece) - several
allocations in one process (earlier ones stay alive, so each lands elsewhere).

    python tools/memory_piece_probe.py [cfg4|S,T,K,N] [allocations]        (GPU box)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
specs = [a for a in sys.argv[1:] if not a.isdigit()] or ["cfg4"]
allocs = next((int(a) for a in sys.argv[1:] if a.isdigit()), 6)
PIECES = [int(x) for x in os.environ.get("PIECES", "0,1,16,32,64,128,256,512").split(",")]
HINTS = [int(x) for x in os.environ.get("HINTS", "0,2").split(",")]


def rate(env, act, steps=40):
    ts = []
    for r in range(3):
        for i in range(5):
            env.step_async(act[i & 3])
        e0.record()
        for i in range(steps):
            env.step_async(act[i & 3])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e3)
    return statistics.median(ts)


for spec in specs:
    if spec in bench.CONFIGS:
        c = bench.CONFIGS[spec]
        S, T, K, N = c["size"], c["tiles"], c["obstacles"], c["boards"]
    else:
        S, T, K, N = (int(x) for x in spec.split(","))
    act = [torch.randint(0, 4, (N,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False) + (T * 2 if S > 16 else 0)
    print(f"{S}x{S}, {T} tiles, {N} boards, {bps * N / 1e6:.0f} MB per launch; us per step by xcd_piece (launch_hint {' / '.join(map(str, HINTS))}); 0.90 of 8 TB/s = {bps * N / 7.2e6:.1f} us")
    keep = []
    for mem in ("contiguous", "torch"):
        for k in range(allocs):
            env = VecTilerSliderEnv.random(N, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True,
                                           output_memory=mem, obs_candidates=0)
            env.reset()
            for i in range(100):
                env.step_async(act[i & 3])
            keep.append(env)
            out = []
            for piece in PIECES:
                env._dims.xcd_piece = piece
                vals = []
                for h in HINTS:
                    env._dims.launch_hint = h
                    vals.append(f"{rate(env, act):.1f}")
                env._dims.launch_hint = 0
                out.append(f"{'policy' if piece == 0 else 'eighths' if piece == 1 else piece}: {' / '.join(vals)}")
            env._dims.xcd_piece = 0
            print(f"  {mem:10s} allocation {k}: " + "   ".join(out), flush=True)
        keep.clear()
        torch.cuda.empty_cache()
    del act
