#!/usr/bin/env python3
"""Development tool: the uint8-observation environment (obs_dtype="uint8", 48 B per 4x4 board) at large batches - store
policy of its byte stream beyond the Infinity Cache (round 3: nontemporal like the float32 stream; round 2 always stored at
agent scope) and one vs two boards per lane."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
from tiler_slider_amd import VecTilerSliderEnv, _cabi  # noqa: E402

L = _cabi.lib()
dev = torch.device("cuda", 0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("boards      MB of uint8 obs |  policy (nontemporal beyond 256 MiB)  |  agent scope forced (round 2)  |  nontemporal forced")
for n in (1 << 20, 1 << 22, 6_000_000, 1 << 23, 1 << 24):
    env = VecTilerSliderEnv.random(n, size=4, num_tiles=2, num_obstacles=2, seed=1, multi_color=True, max_steps=2**30, auto_reset=True,
                                   device=dev, obs_dtype="uint8", placement_trials=0)
    env.reset()
    acts = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev) for _ in range(4)]
    row = f"{n:9d} {n * 48 / 1e6:8.0f}        |"
    for thr in (-1, 1 << 60, 0):
        before = L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, thr) if thr >= 0 else None
        ts = []
        for r in range(5):
            for i in range(10):
                env.step_async(acts[i & 3])
            e0.record()
            for i in range(100):
                env.step_async(acts[i & 3])
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 10)
        if before is not None:
            L.ts_tuning(_cabi.TUNE_NT_THRESHOLD_BYTES, before)
        row += f"   {sorted(ts)[2]:8.2f} us              |"
    print(row, flush=True)
    del env
    torch.cuda.empty_cache()
