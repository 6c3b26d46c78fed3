#!/usr/bin/env python3
"""Round 5: A/B of k_state builds (one board per lane, launches without an image output above 8x8) - development tool.

    python tools/variant_bench.py build base: b8:-DTS_STATE_BATCH=8 b32:-DTS_STATE_BATCH=32      (CPU box)
    python tools/state_only_ab.py [--shapes 15,32,24,262144 ...]                                   (GPU box)

For every shape: ts_is_won, ts_valid_moves, ts_valid_moves4, ts_reward and the step of an environment without observation
(flags only; flags + reward + legality mask), each timed with HIP events over 100 launches, the variants interleaved in rounds
in ONE process; every variant's outputs are compared with the first variant's before timing."""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "build", "variants")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="*", default=["15,32,24,262144", "15,32,24,1048576", "9,4,9,1048576", "20,6,30,262144", "32,64,100,65536"])
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--only")
    a = ap.parse_args()
    import torch
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    manifest = json.load(open(os.path.join(VDIR, "manifest.json")))
    names = [n for n in manifest if not a.only or n in a.only.split(",")]
    P, DP, SP = C.c_void_p, C.POINTER(_cabi.Dims), C.POINTER(_cabi.State)
    libs = {}
    for name in names:
        L = C.CDLL(os.path.join(VDIR, f"{name}.so"))
        for fn in ("ts_valid_moves", "ts_valid_moves4", "ts_is_won", "ts_reward"):
            getattr(L, fn).argtypes, getattr(L, fn).restype = [DP, SP, P, P], C.c_int32
        L.ts_step.argtypes, L.ts_step.restype = [DP, SP, P, C.c_uint32, C.POINTER(_cabi.StepOut), P], C.c_int32
        libs[name] = L
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for shape in a.shapes:
        S, T, K, N = (int(x) for x in shape.split(","))
        bare = VecTilerSliderEnv.random(N, size=S, num_tiles=T, num_obstacles=K, seed=3, multi_color=True, max_steps=2**30, device=dev,
                                        auto_reset=True, obs_dtype=None)
        full = VecTilerSliderEnv.from_arrays(S, bare._blk, bare._init, bare._tgt, multi_color=True, max_steps=2**30, device=dev, auto_reset=True,
                                             obs_dtype=None, with_reward=True, with_valid_moves=True)
        bare.reset(), full.reset()
        acts = [torch.randint(0, 4, (N,), dtype=torch.uint8, device=dev) for _ in range(4)]
        m1, m4 = torch.empty(N, dtype=torch.uint8, device=dev), torch.empty((N, 4), dtype=torch.uint8, device=dev)
        w1, rw = torch.empty(N, dtype=torch.uint8, device=dev), torch.empty(N, dtype=torch.int32, device=dev)
        d, st = bare._dims, bare._state

        def ops(L):
            return [("ts_is_won", lambda i: L.ts_is_won(C.byref(d), C.byref(st), w1.data_ptr(), stream)),
                    ("ts_valid_moves", lambda i: L.ts_valid_moves(C.byref(d), C.byref(st), m1.data_ptr(), stream)),
                    ("ts_valid_moves4", lambda i: L.ts_valid_moves4(C.byref(d), C.byref(st), m4.data_ptr(), stream)),
                    ("ts_reward", lambda i: L.ts_reward(C.byref(d), C.byref(st), rw.data_ptr(), stream)),
                    ("step (flags)", lambda i: L.ts_step(C.byref(bare._dims), C.byref(bare._state), acts[i & 3].data_ptr(), bare._mode, C.byref(bare._out), stream)),
                    ("step (flags, reward, valid, valid4)", lambda i: L.ts_step(C.byref(full._dims), C.byref(full._state), acts[i & 3].data_ptr(), full._mode, C.byref(full._out), stream))]

        # equality of every variant with the first: five steps from the same start, then the entry points
        snap = None
        for name in names:
            bare.reset(), full.reset()
            o = ops(libs[name])
            for i in range(5):
                assert o[4][1](i) == 0 and o[5][1](i) == 0
            for k in range(4):
                assert o[k][1](0) == 0
            torch.cuda.synchronize()
            cur = [t.clone() for t in (bare._pos, bare._flags, bare._step_count, bare._done, full._pos, full._flags, full._reward, full._valid, m1, m4, w1, rw)
                   if t is not None]
            if snap is None:
                snap = cur
            else:
                assert all(torch.equal(x, y) for x, y in zip(snap, cur)), f"variant {name} differs from {names[0]} at {shape}"
        times = {name: {} for name in names}
        for r in range(a.rounds + 1):
            for name in names:
                for op, fn in ops(libs[name]):
                    for i in range(10):
                        fn(i)
                    e0.record()
                    for i in range(a.reps):
                        fn(i)
                    e1.record()
                    torch.cuda.synchronize()
                    if r:
                        times[name].setdefault(op, []).append(e0.elapsed_time(e1) * 1e3 / a.reps)
        print(f"{S}x{S}, {T} tiles, {K} obstacles, {N} boards (us per launch, median of {a.rounds} rounds x {a.reps} launches)")
        for op in times[names[0]]:
            print(f"  {op:38s}" + "".join(f"  {name} {statistics.median(times[name][op]):7.2f}" for name in names))
        del bare, full
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
