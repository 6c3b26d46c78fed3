#!/usr/bin/env python3
"""Round 4: 7x7 / 8x8 batches of 1.0 - 1.7 GB of observation fall from 0.93 - 0.94 of the roofline to 0.65 - 0.75
(profiles/r04_cached_every_nth_wave_large.log).  Which knob brings them back: boards per wave, XCD piece, resident blocks?"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi
L = _cabi.lib()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
mem = os.environ.get("MEM", "contiguous")
print(f"output_memory={mem}")
DEFAULT = '7,5,6,1000;7,5,6,1300;7,5,6,1700;8,4,8,1300;8,4,8,1700;6,3,4,1700;3,1,0,1700'
for S, T, K, mb in [tuple(int(v) for v in x.split(',')) for x in os.environ.get('CASES', DEFAULT).split(';')]:
    n = (mb * 1_000_000 // (12 * S * S)) // 256 * 256
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True, output_memory=mem)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
    for i in range(100):
        env.step_async(act[i & 3])

    def rate():
        ts = []
        for r in range(3):
            for i in range(3):
                env.step_async(act[i & 3])
            e0.record()
            for i in range(20):
                env.step_async(act[i & 3])
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        return statistics.median(ts)

    bps = bench.algorithmic_bytes_per_board_step(S, T, False, False)
    base = rate()
    row = f"{S}x{S} {n:8d} boards {mb:5d} MB | policy {base:6.1f} ({bps * n / base / 8e6:.3f}) |"
    for bpw in (16, 32, 64):
        L.ts_tuning(_cabi.TUNE_SMALL_BPW, bpw)
        row += f" bpw {bpw}: {rate():6.1f}"
        for piece in (1, 16, 256):
            env._dims.xcd_piece = piece
            row += f" p{piece}: {rate():6.1f}"
        env._dims.xcd_piece = 0
        for h in (-4, 4):
            env._dims.launch_hint = h
            row += f" h{h:+d}: {rate():6.1f}"
        env._dims.launch_hint = 0
        row += " |"
    L.ts_tuning(_cabi.TUNE_SMALL_BPW, 0)
    print(row, flush=True)
    del env
