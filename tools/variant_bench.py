#!/usr/bin/env python3
"""A/B harness for kernel tunables (development tool, not part of the product).

  build (CPU box, ships to the GPU box with the snapshot):
      python tools/variant_bench.py build base: nt:-DTS_NT_STORE=1 u12:-DTS_EMIT_UNROLL=12
  run (GPU box): interleaved rounds in ONE process, per the guide's methodology rule 24
      python tools/variant_bench.py run --config cfg1 --rounds 12 --steps 100

Each variant is the shipped source compiled with extra -D flags into build/variants/<name>.so.
Every variant's first step is checked bit-for-bit against the first variant's before timing.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "build", "variants")
SRC = os.path.join(ROOT, "tiler_slider_amd", "csrc", "ts_kernels.hip")


def build(specs):
    """Every variant goes through the product's own guarded pipeline (tiler_slider_amd._cabi.compile_guarded: scan the
    unpadded object for reads of the last allocated VGPR, pad, scan again), so that no A/B runs on a build the product
    would never ship - and none can hit the gfx950 wrong-slide hazard (profiles/r03_wrong_slide_isa.md)."""
    from concurrent.futures import ThreadPoolExecutor
    from tiler_slider_amd import _cabi
    os.makedirs(VDIR, exist_ok=True)
    manifest = {}

    def one(spec):
        name, _, flags = spec.partition(":")
        flags = [f for f in flags.split(",") if f]
        report = _cabi.compile_guarded(SRC, os.path.join(VDIR, f"{name}.so"), defines=flags, work=os.path.join(VDIR, "work", name))
        return name, flags, report["padded"]

    with ThreadPoolExecutor(max_workers=3) as pool:
        for name, flags, padded in pool.map(one, specs):
            manifest[name] = flags
            print(f"  {name}: {flags} (VGPR guard padded {padded} kernels)")
    json.dump(manifest, open(os.path.join(VDIR, "manifest.json"), "w"), indent=1)
    print("built", list(manifest))


def run(a):
    import torch
    import bench
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    manifest = json.load(open(os.path.join(VDIR, "manifest.json")))
    names = [n for n in manifest if not a.only or n in a.only.split(",")]
    if a.shape:
        S_, T_, K_, N_ = (int(x) for x in a.shape.split(","))
        bench.CONFIGS[a.config] = dict(size=S_, tiles=T_, obstacles=K_, boards=N_, onehot=a.onehot, reward=a.onehot)
    cfg = dict(bench.CONFIGS[a.config])
    if a.boards:
        cfg["boards"] = a.boards
    n = cfg["boards"]
    dev = torch.device("cuda", 0)
    keep = []  # --placement K: the K-th allocation of the environment (earlier ones stay alive), since
    for _ in range(a.placement + 1):  # the step time depends on where the buffers landed (tools/placement_study.py)
        env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"],
                                       seed=bench.LEVEL_SEED, multi_color=True, max_steps=2**30, device=dev,
                                       auto_reset=True, with_reward=cfg["reward"], with_onehot=cfg["onehot"], placement_trials=0, obs_candidates=a.obs_candidates)
        keep.append(env)
    env.reset()
    env._dims.launch_hint, env._dims.xcd_piece, env._dims.emit_edges = a.hint, a.piece, a.edges
    stream = torch.cuda.current_stream(dev).cuda_stream
    ring = []
    for i in range(16):
        t = torch.empty(n, dtype=torch.uint8, device=dev)
        _cabi.check(_cabi.lib().ts_fill_actions(n, bench.ACTION_SEED, 0, i, t.data_ptr(), stream), "fill")
        ring.append(t)
    if a.pick:  # --pick slowest|fastest: rate every allocation with the shipped library first, then A/B on that one
        rated = []
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for k, e in enumerate(keep):
            e.reset()
            for i in range(10):
                e.step_async(ring[i & 15])
            ev0.record()
            for i in range(100):
                e.step_async(ring[i & 15])
            ev1.record()
            torch.cuda.synchronize()
            rated.append((ev0.elapsed_time(ev1) * 10, k))
        print("allocations (us per step, shipped library):", " ".join(f"{k}:{us:.1f}" for us, k in rated))
        us, k = min(rated) if a.pick == "fastest" else max(rated)
        env = keep[k]
        print(f"using allocation {k} ({a.pick}, {us:.1f} us)")
        env.reset()
    libs = {}
    for name in names:
        L = C.CDLL(os.path.join(VDIR, f"{name}.so"))
        L.ts_step.argtypes = [C.POINTER(_cabi.Dims), C.POINTER(_cabi.State), C.c_void_p, C.c_uint32,
                              C.POINTER(_cabi.StepOut), C.c_void_p]
        L.ts_step.restype = C.c_int32
        libs[name] = L
        for kv in (a.tuning.split(",") if a.tuning else []):  # ts_tuning knobs (include/tiler_slider.h), e.g. 8=16: k_small quarter waves
            key, _, val = kv.partition("=")
            L.ts_tuning.argtypes, L.ts_tuning.restype = [C.c_int32, C.c_int64], C.c_int64
            L.ts_tuning(int(key), int(val))

    def step(L, i):
        rc = L.ts_step(C.byref(env._dims), C.byref(env._state), ring[i & 15].data_ptr(), env._mode, C.byref(env._out), stream)
        assert rc == 0, rc

    # correctness of every variant against the first, from the same starting state
    snap = None
    for name in names:
        env.reset()
        for i in range(3):
            step(libs[name], i)
        torch.cuda.synchronize()
        cur = [env._pos.clone(), env._flags.clone(), env._obs.clone(), env._step_count.clone()]
        if env._onehot is not None:
            cur.append(env._onehot.clone())
        if snap is None:
            snap = cur
        else:
            same = all(torch.equal(x, y) for x, y in zip(snap, cur))
            assert same or a.no_check, f"variant {name} differs from {names[0]}"
            if not same:
                print(f"  (variant {name} differs from {names[0]}: ablation)")
    times = {name: [] for name in names}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(a.rounds + 1):
        for name in names:
            L = libs[name]
            for i in range(5):
                step(L, i)
            e0.record()
            for i in range(a.steps):
                step(L, i)
            e1.record()
            torch.cuda.synchronize()
            if r:  # round 0 is warm-up
                times[name].append(e0.elapsed_time(e1) * 1e3 / a.steps)
    bps = bench.algorithmic_bytes_per_board_step(cfg["size"], cfg["tiles"], cfg["onehot"], cfg["reward"])
    res = {}
    print(f"{a.config}: {n} boards, {bps} B/board-step, {a.rounds} rounds x {a.steps} steps, launch_hint {a.hint}, xcd_piece {a.piece}, emit_edges {a.edges}")
    for name in names:
        med, mn = statistics.median(times[name]), min(times[name])
        res[name] = {"flags": manifest[name], "median_us": med, "min_us": mn, "GBps_median": bps * n / med / 1e3}
        print(f"  {name:24s} median {med:8.2f} us  min {mn:8.2f} us  {bps * n / med / 1e3:8.1f} GB/s  {manifest[name]}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"variants_{a.config}{a.tag}.json"), "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    b = sub.add_parser("build")
    b.add_argument("specs", nargs="+")
    r = sub.add_parser("run")
    r.add_argument("--config", default="cfg1")
    r.add_argument("--boards", type=int)
    r.add_argument("--rounds", type=int, default=10)
    r.add_argument("--steps", type=int, default=100)
    r.add_argument("--only")
    r.add_argument("--shape", help="S,T,K,N: ad-hoc shape registered under --config's name")
    r.add_argument("--no-check", action="store_true")
    r.add_argument("--hint", type=int, default=0, help="ts_dims.launch_hint for every variant")
    r.add_argument("--piece", type=int, default=0, help="ts_dims.xcd_piece for every variant")
    r.add_argument("--edges", type=int, default=0, help="ts_dims.emit_edges for every variant")
    r.add_argument("--tag", default="")
    r.add_argument("--obs-candidates", type=int, default=0, help="VecTilerSliderEnv(obs_candidates=...)")
    r.add_argument("--onehot", action="store_true", help="with --shape: one-hot planes and reward too")
    r.add_argument("--tuning", help="ts_tuning settings for every variant: key=value,key=value")
    r.add_argument("--placement", type=int, default=0, help="use the K-th allocation of the environment")
    r.add_argument("--pick", choices=["slowest", "fastest"], help="with --placement K: rate the K+1 allocations, use that one")
    args = ap.parse_args()
    build(args.specs) if args.cmd == "build" else run(args)
