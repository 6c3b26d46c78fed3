import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from tiler_slider_amd import VecTilerSliderEnv, _cabi
L = _cabi.lib()
dev = torch.device("cuda", 0)
for n in (1 << 18, 1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 24):
    for kw, tag in ((dict(obs_dtype="uint8"), "u8 obs"), (dict(obs_dtype="uint8", with_reward=True, with_valid_moves=True), "u8 obs + reward + mask")):
        env = VecTilerSliderEnv.random(n, size=4, num_tiles=2, num_obstacles=2, seed=1, multi_color=True, max_steps=2**30, auto_reset=True, device=dev, **kw)
        env.reset()
        acts = [torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev) for _ in range(4)]
        out = []
        for knob in (2**62, 0):
            L.ts_tuning(0, knob)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for r in range(5):
                for i in range(10): env.step_async(acts[i & 3])
                e0.record()
                for i in range(200): env.step_async(acts[i & 3])
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 5)
            out.append(sorted(ts)[2])
        L.ts_tuning(0, 1048576)
        print(f"{n:8d} boards, {tag:24s}: k_small {out[0]:6.2f} us   k_multi {out[1]:6.2f} us", flush=True)
