#!/usr/bin/env python3
"""Target of a rocprofv3 --pmc pass: 30 steps of one shape at one batch size.  argv: S T K boards [cached_every [emit_edges]]
(ts_tuning(TS_TUNE_CACHED_EVERY) / ts_dims.emit_edges; 0 = the policy)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiler_slider_amd import VecTilerSliderEnv
S, T, K, n = (int(x) for x in sys.argv[1:5])
env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, auto_reset=True)
if len(sys.argv) > 5:
    from tiler_slider_amd import _cabi
    _cabi.lib().ts_tuning(_cabi.TUNE_CACHED_EVERY, int(sys.argv[5]))
if len(sys.argv) > 6:
    env._dims.emit_edges = int(sys.argv[6])
env.reset()
act = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=env.device) for _ in range(4)]
for i in range(30):
    env.step_async(act[i & 3])
torch.cuda.synchronize()
