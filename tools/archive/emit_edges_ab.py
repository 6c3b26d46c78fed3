#!/usr/bin/env python3
"""Development tool (round 3): out-of-cache launches with the first / last store instruction of every wave's chunk as
write-back stores (ts_tuning TS_TUNE_EMIT_EDGES 0..3) x resident blocks per CU (launch_hint), in one process.
usage: emit_edges_ab.py [config | S,T,K,N[,onehot]] ..."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from tiler_slider_amd import VecTilerSliderEnv, _cabi  # noqa: E402

L = _cabi.lib()
HINTS = tuple(int(x) for x in os.environ.get("TS_AB_HINTS", "0,1,2,3,4,6").split(","))
EDGES = tuple(int(x) for x in os.environ.get("TS_AB_EDGES", "0,1,2,3").split(","))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for spec in sys.argv[1:] or ["cfg4", "cfg2"]:
    if "," in spec:
        v = [int(x) for x in spec.split(",")]
        cfg = dict(size=v[0], tiles=v[1], obstacles=v[2], boards=v[3], onehot=len(v) > 4 and bool(v[4]), reward=len(v) > 4 and bool(v[4]))
    else:
        cfg = bench.CONFIGS[spec]
    n = cfg["boards"]
    env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=bench.LEVEL_SEED,
                                   multi_color=True, max_steps=2**30, auto_reset=True, with_reward=cfg["reward"], with_onehot=cfg["onehot"], placement_trials=0)
    env.reset()
    act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
    bps = bench.algorithmic_bytes_per_board_step(cfg["size"], cfg["tiles"], cfg["onehot"], cfg["reward"])
    print(f"{spec}: {n} boards, {bps * n / 1e6:.0f} MB per launch; us per step (frac of 8 TB/s) by launch_hint {HINTS}")
    for edges in EDGES:
        L.ts_tuning(_cabi.TUNE_EMIT_EDGES, edges)
        row = f"  edges={edges}: "
        for hint in HINTS:
            env._dims.launch_hint = hint
            ts = []
            for r in range(3):
                for i in range(3):
                    env.step_async(act[i & 3])
                e0.record()
                for i in range(30):
                    env.step_async(act[i & 3])
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 30 * 1e3)
            us = statistics.median(ts)
            row += f" {us:7.1f} ({bps * n / us / 1e3 / 8000:.3f})"
        print(row, flush=True)
    L.ts_tuning(_cabi.TUNE_EMIT_EDGES, 4)
    del env
    torch.cuda.empty_cache()
