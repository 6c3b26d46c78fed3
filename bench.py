#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched Tiler-Slider step() on MI355X.

    python bench.py --gpus 1 --steps 500 --warmup 50
    python bench.py --gpus 8                      # starts its own 8 ranks (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one ts_step launch over every board of the rank's shard (slide, flags, counters,
done latch, float32 observation).  Workload = BASELINE.json configs[1]: 1,048,576 concurrent
4x4 boards per GPU (T=2 tiles, K=2 obstacles, multi_color), synthetic random levels and
actions, inputs resident in HBM before the timed region, autoreset so every board stays live.
Boards shard across ranks with no data-path collective (weak scaling); the RCCL all-gather
that hands observations to a single learner is timed separately and reported under
"allgather" (serial, and overlapped with the next step on double-buffered observations).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LEVEL_SEED = 0x715311DE
ACTION_SEED = 0xAC710005
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
COPY_CEILING_GBS = 6290.0  # measured float4 copy ceiling (/opt/skills/guides/MI355X_MICROARCH.md, chip-level table)
INFINITY_CACHE_BYTES = 256 << 20

CONFIGS = {
    # name: (size, tiles, obstacles, boards per GPU, extras)
    "cfg1": dict(size=4, tiles=2, obstacles=2, boards=1 << 20, onehot=False, reward=False),
    "cfg2": dict(size=5, tiles=2, obstacles=3, boards=1 << 20, onehot=True, reward=True),
    "cfg4": dict(size=15, tiles=32, obstacles=24, boards=1 << 18, onehot=False, reward=False),
    # not in BASELINE.json: exercises the any-tile-count path of the one-lane-per-board kernel
    "s8t20": dict(size=8, tiles=20, obstacles=10, boards=1 << 19, onehot=False, reward=False),
    "s9t4": dict(size=9, tiles=4, obstacles=9, boards=1 << 19, onehot=False, reward=False),
    "s12t8": dict(size=12, tiles=8, obstacles=16, boards=1 << 18, onehot=False, reward=False),
    "s32t64": dict(size=32, tiles=64, obstacles=100, boards=1 << 15, onehot=False, reward=False),
    # round 4's launch forms beyond the cache (tools/soak.py): quarter waves, one board per wave on 32 lanes
    "s7t5": dict(size=7, tiles=5, obstacles=6, boards=1 << 20, onehot=False, reward=False),
    "s8t4": dict(size=8, tiles=4, obstacles=8, boards=1 << 19, onehot=False, reward=False),
    "s28t8": dict(size=28, tiles=8, obstacles=60, boards=1 << 16, onehot=False, reward=False),
}


def algorithmic_bytes_per_board_step(size, tiles, onehot, reward, multi_color=True):
    """read pos+tgt+blk+step+done+act, write pos+step+done+flags+obs."""
    C = size * size
    blk = 2 if C <= 16 else 4 * ((C + 31) // 32)
    read = tiles + tiles + blk + 4 + 1 + 1
    write = tiles + 4 + 1 + 1 + 12 * C
    if onehot:
        write += 4 * C * ((1 + 2 * tiles) if multi_color else 3)
    if reward:
        write += 4
    return read + write


def launch_description(env, outputs=None):
    """What ts_step launches for this environment, from the library itself (ts_describe_launch: the code path the launch takes)."""
    from tiler_slider_amd import _cabi
    if outputs is None:
        outputs = ((_cabi.OUT_OBS if env.obs_dtype is not None and env._obs.dtype.is_floating_point else 0)
                   | (_cabi.OUT_OBS_U8 if env.obs_dtype is not None and not env._obs.dtype.is_floating_point else 0)
                   | (_cabi.OUT_REWARD if env._reward is not None else 0) | (_cabi.OUT_ONEHOT if env._onehot is not None else 0)
                   | (_cabi.OUT_VALID | _cabi.OUT_VALID4 if env._valid is not None else 0))
    return _cabi.describe_launch(env._dims, _cabi.OP_STEP, outputs)


def pmc_traffic(config, boards):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic_pmc.json; counters cannot be read from inside this process).  None when no
    pass exists for this config / batch size."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic_pmc.json"))).get(config)
    except (OSError, ValueError):
        return None
    if not rec or rec.get("boards") != boards:
        return None
    return rec


def host_cpu_share():
    """Cores this process may actually use: the cgroup CPU quota if there is one, else the
    scheduler affinity (a GPU box hands each GPU's job a share of the host, not all of it)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cfg, budget_s=10.0):
    """The CPU oracle (a C port of the reference algorithm, OpenMP over boards) on the SAME
    synthetic boards and action stream, on this box's host cores.  Reported, never the target."""
    from oracle import binding as orc
    orc.lib().tso_set_num_threads(host_cpu_share())
    n = min(cfg["boards"], 1 << 20)
    blk, init, tgt = orc.generate(cfg["size"], cfg["tiles"], cfg["tiles"], cfg["obstacles"], n, seed=LEVEL_SEED)
    env = orc.OracleBatch(cfg["size"], True, 2**30, blk, init, tgt)
    env.reset()
    acts = [orc.fill_actions(n, seed=ACTION_SEED, step_index=i) for i in range(8)]
    kw = dict(mode=orc.MODE_AUTORESET, reward=cfg["reward"], onehot=cfg["onehot"])
    t0 = time.perf_counter()
    env.step(acts[0], **kw)
    env.step(acts[1], **kw)
    per_step = (time.perf_counter() - t0) / 2
    steps = max(3, min(400, int(budget_s / max(per_step, 1e-6))))
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(acts[i & 7], **kw)
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": int(orc.lib().tso_num_threads()),
            "kind": "port",
            "sample": f"{n} boards x {steps} steps of the same workload (oracle/ts_oracle.c, OpenMP), {dt:.1f} s",
            "reference_python_recorded": reference_python_recorded(cfg)}


# The reference's own Python TilerSliderEnv.step loop on ONE core (it is single-threaded) of the build container
# (Xeon 2.1 GHz, py3.10, numpy 2.2), tools/time_reference_python.py; the reference cannot travel to the GPU box, so
# these are recorded constants, keyed by (size, tiles, obstacles).
REFERENCE_PYTHON_STEPS_PER_S = {(3, 1, 0): 1.03e5, (4, 2, 2): 1.22e5, (5, 2, 3): 1.05e5, (15, 32, 24): 1.45e4}


def reference_python_recorded(cfg):
    v = REFERENCE_PYTHON_STEPS_PER_S.get((cfg["size"], cfg["tiles"], cfg["obstacles"]))
    return {"value": v, "unit": "env-steps/s", "cores": 1,
            "note": "reference TilerSliderEnv.step loop of this board shape, timed in the build container "
                    "(tools/time_reference_python.py); the reference cannot travel to the GPU box"}


def spawn_ranks(n):
    """`python -m torch.distributed.run --nproc-per-node n bench.py <same arguments>` as a child
    process; returns its exit code.  Fails with a message (no hang) when the box has fewer GPUs."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} needs {n} GPUs on this node, found {have}", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


_STDOUT_FD = None


def main():
    global _STDOUT_FD
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg1")
    ap.add_argument("--boards", type=int, default=None, help="boards per GPU (default: the config's)")
    ap.add_argument("--graph", action="store_true", help="replay the timed steps from one hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--placement-trials", type=int, default=0,
                    help="VecTilerSliderEnv(placement_trials=...) for batches beyond the Infinity Cache: 0 = the library's static "
                         "launch policy (the class default), 1 = launch policy rated at construction on the first allocation, "
                         "k > 1 = also up to k candidate allocations of the output buffers (reported in config)")
    ap.add_argument("--policy", default=None,
                    help="launch_hint,emit_edges,lines_lanes,xcd_piece: fix the per-call launch policy of ts_dims (use with --placement-trials 0 "
                         "to profile exactly the launches a tuned run settled on)")
    ap.add_argument("--output-memory", choices=["torch", "contiguous"], default="contiguous",
                    help="VecTilerSliderEnv(output_memory=...): physically contiguous output buffers beyond the Infinity Cache, or torch's allocator")
    ap.add_argument("--obs-candidates", type=int, default=None,
                    help="VecTilerSliderEnv(obs_candidates=...): the fastest of up to k candidate observation buffers (environments "
                         "with one-hot planes run at one of two speeds by where the observation buffer lies, even in physically "
                         "contiguous memory; ~2 ms per candidate at construction).  Default: the class default (16 for such "
                         "environments - at most 4 GiB of candidates in all: 13 at cfg2 -, else 0); 0 = the first allocation")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the two-halves-on-two-streams figure")
    ap.add_argument("--no-sibling", action="store_true", help="skip the out-of-cache sibling of a cache-resident config")
    ap.add_argument("--clock-warmup-ms", type=float, default=150.0,
                    help="untimed stepping before the warm-up steps, so that the timed region runs at steady-state clocks (0 = off)")
    ap.add_argument("--no-entry-points", action="store_true", help="skip reset / scramble / stand-alone entry-point timings")
    ap.add_argument("--no-learner-side", action="store_true", help="skip the cfg3 learner-side timings (8,388,608 gathered boards)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short timings of the other single-GPU BASELINE configs (cfg2, cfg4) that a cfg1 run appends")
    ap.add_argument("--compact-u8", action="store_true",
                    help="also time the opt-in uint8-observation variant (reported as compact_u8_obs; off by "
                         "default so that a profile of the default run contains only the headline launches)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: take the multi-rank code path (RCCL group, max-over-ranks reduction, "
                         "all-gather timings) even with one rank; launch under torch.distributed.run")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves.  This process has not
        # touched the GPU (torch.cuda.device_count() does not initialise it on ROCm); the ranks
        # are fresh children, this process only relays their output and exit code.
        raise SystemExit(spawn_ranks(args.gpus))

    sys.stdout.flush()
    _STDOUT_FD = os.dup(1)  # restored for the one JSON line; until then fd 1 is stderr
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from tiler_slider_amd import VecTilerSliderEnv, _cabi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: they must agree")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    multi = world > 1 or args.force_dist
    if multi:
        dist.init_process_group("nccl", device_id=device)

    cfg = dict(CONFIGS[args.config])
    if args.boards:
        cfg["boards"] = args.boards
    n = cfg["boards"]
    env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"],
                                   seed=LEVEL_SEED, multi_color=True, max_steps=2**30, board_offset=rank * n,
                                   device=device, auto_reset=True, with_reward=cfg["reward"],
                                   with_onehot=cfg["onehot"], placement_trials=args.placement_trials, output_memory=args.output_memory,
                                   obs_candidates=args.obs_candidates)
    if args.policy:
        env._dims.launch_hint, env._dims.emit_edges, env._dims.lines_lanes, env._dims.xcd_piece = (int(x) for x in args.policy.split(","))
    env.reset()
    ring = []
    L = _cabi.lib()
    stream = torch.cuda.current_stream(device).cuda_stream
    for i in range(16):
        a = torch.empty(n, dtype=torch.uint8, device=device)
        _cabi.check(L.ts_fill_actions(n, ACTION_SEED, rank * n, i, a.data_ptr(), stream), "ts_fill_actions")
        ring.append(a)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize(device)

    # Clock warm-up (untimed, reported in config.clock_warmup): after the host-side set-up the GPU sits at idle clocks, and a
    # handful of warm-up steps (the driver's 5 are 0.15 ms of work) do not bring them up - the first timed steps then run
    # ~5 % slow (cfg1: 31.6 us per step in a 20-step window against 30.0 in steady state).  The same step launches as the
    # timed region, on the same boards (autoreset keeps them statistically stationary), until ~args.clock_warmup_ms have passed.
    clock_warmup = {"ms": 0.0, "steps": 0}
    if args.clock_warmup_ms > 0:
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e3 < args.clock_warmup_ms:
            for i in range(64):
                env.step_async(ring[i & 15])
            torch.cuda.synchronize(device)
            clock_warmup["steps"] += 64
        clock_warmup["ms"] = (time.perf_counter() - t_w) * 1e3

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def run_steps(k, graph=None):
        """k step launches between two HIP events on the launch stream, then the synchronize of the contract."""
        ev0.record()
        if graph is not None:
            graph.replay()
        else:
            for i in range(k):
                env.step_async(ring[i & 15])
        ev1.record()
        torch.cuda.synchronize(device)

    # the W warm-up steps run through the SAME host code as the timed region (events, launches, synchronize): executed for the
    # first time, that path costs ~40 us more than the second time - 6 % of a 20-step window (profiles/r04_timed_region_probe.log)
    if args.warmup > 0:
        run_steps(args.warmup)
    graph = None
    if args.graph:
        torch.cuda.synchronize(device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(args.steps):
                env.step_async(ring[i & 15])
    # timed region: barrier + synchronize, K steps, synchronize (clock stops when THIS rank's K
    # steps have finished) + barrier; the slowest rank's time is taken below (MAX over ranks)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, graph)
    wall = time.perf_counter() - t0
    barrier()
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the stream the kernels were launched on

    tm = torch.tensor([wall, dev_ms], dtype=torch.float64, device=device)
    rccl_ranks = None
    if multi:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ones = torch.ones(1, dtype=torch.int32, device=device)  # how many ranks the collective library itself saw
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl_ranks = int(ones[0])
    wall, dev_ms = float(tm[0]), float(tm[1])
    total_boards = n * world
    value = total_boards * args.steps / wall

    # the same loop through the public step() (argument staging + lazy info object, no sync):
    # what a Python learner pays per call on top of the kernel
    api = None
    if world == 1:
        k = min(args.steps, 200)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for i in range(k):
            env.step(ring[i & 15])
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t1
        api = {"value": n * k / dt, "unit": "env-steps/s", "us_per_call": dt / k * 1e6, "steps": k}

    # opt-in compact observation (uint8 instead of float32): reported separately, never mixed with
    # the headline figure, which is the reference's float32 format
    compact = None
    if args.compact_u8 and world == 1 and not cfg["onehot"]:
        env8 = VecTilerSliderEnv.from_arrays(cfg["size"], env._blk, env._init, env._tgt, multi_color=True,
                                             max_steps=2**30, device=device, auto_reset=True, obs_dtype="uint8")
        env8.reset()
        for i in range(10):
            env8.step_async(ring[i & 15])
        k = min(args.steps, 200)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(k):
            env8.step_async(ring[i & 15])
        e1.record()
        torch.cuda.synchronize(device)
        us = e0.elapsed_time(e1) * 1e3 / k
        b8 = algorithmic_bytes_per_board_step(cfg["size"], cfg["tiles"], False, False) - 9 * cfg["size"] ** 2
        compact = {"value": n / us * 1e6, "unit": "env-steps/s", "kernel_us": us, "obs": "uint8 [N,S,S,3]",
                   "algorithmic_bytes_per_board_step": b8, "achieved_GBps": b8 * n / us / 1e3}
        del env8

    # the same kernel on a batch whose working set cannot stay in the Infinity Cache: the
    # HBM-bound sibling of a cache-resident headline figure
    sibling = None
    bps_cfg = algorithmic_bytes_per_board_step(cfg["size"], cfg["tiles"], cfg["onehot"], cfg["reward"])
    if world == 1 and bps_cfg * n < INFINITY_CACHE_BYTES and not args.no_sibling:
        n_big = 4 * n
        big = VecTilerSliderEnv.random(n_big, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"],
                                       seed=LEVEL_SEED, multi_color=True, max_steps=2**30, device=device, auto_reset=True,
                                       with_reward=cfg["reward"], with_onehot=cfg["onehot"],
                                       placement_trials=args.placement_trials, output_memory=args.output_memory)
        big.reset()
        acts = []
        for i in range(4):
            a = torch.empty(n_big, dtype=torch.uint8, device=device)
            _cabi.check(L.ts_fill_actions(n_big, ACTION_SEED, 0, i, a.data_ptr(), stream), "ts_fill_actions")
            acts.append(a)
        for i in range(10):
            big.step_async(acts[i & 3])
        k = 100
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(k):
            big.step_async(acts[i & 3])
        e1.record()
        torch.cuda.synchronize(device)
        us = e0.elapsed_time(e1) * 1e3 / k
        gbs = bps_cfg * n_big / us / 1e3
        sibling = {"boards": n_big, "kernel_us": us, "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                   "frac_of_copy_ceiling": gbs / COPY_CEILING_GBS, "algorithmic_bytes_per_launch": bps_cfg * n_big,
                   "value": n_big / us * 1e6, "value_unit": "env-steps/s", "placement": big.placement_report}
        del big, acts
        torch.cuda.empty_cache()

    # the same config at >= 3 GB per launch: the figure the Infinity Cache cannot help (roofline.hbm_asymptote)
    asymptote = None
    if world == 1 and args.config in ASYMPTOTE_BOARDS and not args.boards and not args.no_sibling:
        asymptote = time_asymptote(args.config, device, L, stream, args.output_memory)

    # cfg3's building blocks on one GPU: (a) the step into a ring of TWO observation buffers - what the overlapped hand-off of
    # float32 / uint8 observations needs (obs_buffers=2; the ring's bytes, not the launch's, decide cache residency:
    # ts_dims.ring_bytes) - and (b) the actor step that keeps NO observation (compact hand-off: the learner re-encodes)
    double_buffered = actor = None
    if world == 1 and not args.no_sibling and not cfg["onehot"]:
        kw_twin = dict(multi_color=True, max_steps=2**30, device=device, auto_reset=True, with_reward=cfg["reward"])
        e2 = VecTilerSliderEnv.from_arrays(cfg["size"], env._blk, env._init, env._tgt, obs_buffers=2, output_memory=args.output_memory, **kw_twin)
        e2.reset()
        us2 = time_env_steps(e2, ring, min(args.steps, 200), warm=20)
        d2 = launch_description(e2)
        double_buffered = {"obs_buffers": 2, "kernel": d2["name"], "kernel_us": us2, "frac": bps_cfg * n / us2 / 1e3 / HBM_PEAK_GBS,
                           "achieved_GBps": bps_cfg * n / us2 / 1e3, "value": n / us2 * 1e6, "unit": "env-steps/s",
                           "ring_bytes": e2._dims.ring_bytes, "out_of_cache": d2["out_of_cache"],
                           "note": "ts_step into a ring of two observation buffers (what an overlapped gather needs); classified by the ring's bytes"}
        del e2
        ea = VecTilerSliderEnv.from_arrays(cfg["size"], env._blk, env._init, env._tgt, obs_dtype=None, **kw_twin)
        ea.reset()
        usa = time_env_steps(ea, ring, min(args.steps, 200), warm=20)
        C_ = cfg["size"] ** 2
        bpa = bps_cfg - 12 * C_
        actor = {"kernel": launch_description(ea)["name"], "kernel_us": usa, "value": n / usa * 1e6, "unit": "env-steps/s",
                 "algorithmic_bytes_per_board_step": bpa, "achieved_GBps": bpa * n / usa / 1e3, "frac": bpa * n / usa / 1e3 / HBM_PEAK_GBS,
                 "note": "VecTilerSliderEnv(obs_dtype=None): ts_step_out.obs = NULL - the actor ranks of the compact hand-off; "
                         "launch-bound (a few MB of state), so frac is not the yardstick"}
        del ea
        torch.cuda.empty_cache()

    # The other single-GPU configs of BASELINE.json (cfg2: 5x5 + one-hot + reward; cfg4: 15x15 / 32 tiles - the genuinely
    # HBM-bound ones), a few hundred ms each, so that the driver's one default run records all three: with the library's static
    # launch policy (placement_trials = 0), with the class default (launch policy rated at construction on the first
    # allocation) and with the construction-time choice among candidate buffers (the headline's own --placement-trials).
    others = None
    if world == 1 and args.config == "cfg1" and not args.boards and not args.no_other_configs:
        others = {}
        for name in ("cfg2", "cfg4"):
            others[name] = time_config(name, args.placement_trials, 100, device, L, stream)
            torch.cuda.empty_cache()

    # two halves of the batch on two streams (tiler_slider_amd.pipelined): what a double-buffered actor loop gets when
    # one half steps while it works on the other - the ramps of consecutive launches overlap.  Not the headline: the
    # synchronous step() of the reference API joins all boards every step.
    pipelined = None
    if world == 1 and not args.no_pipelined and n % 2 == 0:
        from tiler_slider_amd import PipelinedTilerSliderEnv
        pe = PipelinedTilerSliderEnv(n, parts=2, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"],
                                     seed=LEVEL_SEED, multi_color=True, max_steps=2**30, device=device, auto_reset=True,
                                     with_reward=cfg["reward"], with_onehot=cfg["onehot"], placement_trials=args.placement_trials)
        pe.reset()
        pe.wait()
        half = n // 2
        pacts = [[ring[i][p * half:(p + 1) * half].contiguous() for i in range(4)] for p in range(2)]
        k = min(args.steps, 200)
        # the action buffers exist already: launch straight on the parts' streams (no per-step join with this stream)
        def run(steps):
            for i in range(steps):
                for p in range(2):
                    pe.step_part_async(p, pacts[p][i & 3], actions_ready=True)
        run(10)
        pe.wait()
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s_ in pe.streams:
            s_.wait_event(e0)
        t0 = time.perf_counter()
        run(k)
        pe.wait()
        e1.record()
        torch.cuda.synchronize(device)
        us = e0.elapsed_time(e1) * 1e3 / k
        pipelined = {"parts": 2, "value": n / us * 1e6, "unit": "env-steps/s", "us_per_step_of_all_boards": us,
                     "us_per_step_wall": (time.perf_counter() - t0) * 1e6 / k, "steps": k,
                     "note": "two halves on two streams, no join between steps (tiler_slider_amd.pipelined)"}
        pe.close()
        del pe, pacts
        torch.cuda.empty_cache()

    # reset() and the "scramble" (the reference's own seed -> level map on the device, ts_generate_mt19937) of this batch
    extras_us = None
    if world == 1 and not args.no_entry_points:
        extras_us = time_entry_points(env, cfg, device, L, stream)

    # cfg3 (8 x 1,048,576 boards, one learner): what the learner GPU does per step after a compact / uint8 all-gather, at the
    # FULL gathered size - one ts_encode over 8,388,608 boards and one ts_expand_u8 over their 403 MB of bytes (1.6 GB out)
    learner = None
    if world == 1 and args.config == "cfg1" and not args.boards and not args.no_learner_side:
        learner = time_learner_side(cfg, 8 * n, device, L, stream)
        torch.cuda.empty_cache()

    gather = None
    if multi and not args.no_gather:
        gather = time_gathers(env, ring, world, n, dist, torch, device, min(args.steps, 20))

    if rank == 0:
        bps = algorithmic_bytes_per_board_step(cfg["size"], cfg["tiles"], cfg["onehot"], cfg["reward"])
        kern_s = dev_ms / 1e3 / args.steps
        achieved = bps * n / kern_s / 1e9
        achieved_wall = bps * n / (wall / args.steps) / 1e9
        cache_resident = bps * n < INFINITY_CACHE_BYTES
        line = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.config}: {n:,} concurrent {cfg['size']}x{cfg['size']} boards per GPU, "
                                   f"T={cfg['tiles']} tiles, K={cfg['obstacles']} obstacles, multi_color, "
                                   f"random actions, autoreset"
                                   + (", + one-hot + Manhattan reward" if cfg["onehot"] else ""),
                       "boards_per_gpu": n, "total_boards": total_boards, "obs": "float32 [N,S,S,3]",
                       "launch": "hipGraph" if graph is not None else "eager", "clock_warmup": clock_warmup,
                       "clock_note": "steady-state-clock figure since round 4 (untimed clock warm-up before the W warm-up steps); rounds 1-3 "
                                     "timed the first steps after idle, ~5 % slower: not comparable one-to-one",
                       # construction-time choice among candidate allocations of the output buffers (outside the
                       # timed region; only for batches beyond the Infinity Cache): VecTilerSliderEnv docstring
                       "placement_trials": args.placement_trials, "placement": env.placement_report, "output_memory": args.output_memory,
                       "obs_candidates": args.obs_candidates, "observation_placement": env.observation_placement_report,
                       "launch_policy": {"launch_hint": env._dims.launch_hint, "emit_edges": env._dims.emit_edges,
                                         "lines_lanes": env._dims.lines_lanes, "xcd_piece": env._dims.xcd_piece},
                       "parallelism": f"boards sharded over {world} GPU(s), no data-path collective",
                       "level_seed": hex(LEVEL_SEED), "action_seed": hex(ACTION_SEED)},
            # `bound`: "hbm" only where the 256 MiB Infinity Cache cannot help (>= 3 GB per launch).  A launch that fits the
            # cache runs at the on-die cache / fabric write rate; one of 1 .. 12 x the cache has part of its stream absorbed
            # between launches (algorithmic bytes / time is then an UPPER bound on HBM traffic).  `peak` is the HBM3E spec peak
            # in every case; the HBM-bound figure of the same kernel and shape is `hbm_asymptote`.
            "roofline": {"bound": "hbm" if bps * n >= 12 * INFINITY_CACHE_BYTES else ("infinity_cache" if cache_resident else "hbm+infinity_cache"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "working_set_over_infinity_cache": bps * n / INFINITY_CACHE_BYTES,
                         # the same from the wall clock of the timed region (host launch gaps included);
                         # `frac` is from HIP events on the launch stream
                         "frac_wall": achieved_wall / HBM_PEAK_GBS,
                         "frac_of_copy_ceiling": achieved / COPY_CEILING_GBS, "copy_ceiling": COPY_CEILING_GBS,
                         "kernel": launch_description(env)["name"], "launch": launch_description(env),
                         "kernel_us": kern_s * 1e6, "algorithmic_bytes_per_board_step": bps,
                         "algorithmic_bytes_per_launch": bps * n,
                         # a launch that moves less than the 256 MiB Infinity Cache runs at the on-die
                         # cache / fabric write rate, not at the HBM rate: see "hbm_asymptote"
                         "cache_resident": cache_resident},
        }
        rec = pmc_traffic(args.config if n == CONFIGS[args.config]["boards"] else args.config + "_sibling_4m", n)
        if rec is not None:
            line["roofline"]["traffic"] = rec["write_bytes"] + rec["fetch_bytes_x2"]
            line["roofline"]["traffic_source"] = ("rocprofv3 --pmc WRITE_SIZE + 2 x FETCH_SIZE (gfx950 wide-read "
                                                  "correction; raw FETCH_SIZE %d B), separate passes, recorded in "
                                                  "profiles/traffic_pmc.json" % rec["fetch_bytes_raw"])
        if sibling is not None:
            sibling["note"] = "the same kernel family at 4 x the boards: %.1f x the Infinity Cache - still cache-assisted, see hbm_asymptote" % (
                sibling["algorithmic_bytes_per_launch"] / INFINITY_CACHE_BYTES)
            line["roofline"]["sibling_3p4x_infinity_cache" if args.config == "cfg1" else "sibling_4x_boards"] = sibling
        if asymptote is not None:
            line["roofline"]["hbm_asymptote"] = asymptote
        if double_buffered is not None:
            line["double_buffered"] = double_buffered
        if actor is not None:
            line["actor_without_observation"] = actor
        if others is not None:
            line["other_configs"] = others
        if api is not None:
            line["python_step_api"] = api
        if pipelined is not None:
            line["pipelined_halves"] = pipelined
        if compact is not None:
            line["compact_u8_obs"] = compact
        if gather is not None:
            line["allgather"] = gather
        if rccl_ranks is not None:
            line["rccl_ranks"] = rccl_ranks  # all_reduce(SUM) of one 1 per rank over the "nccl" (= RCCL) group
        if extras_us is not None:
            line["entry_points"] = extras_us
        if learner is not None:
            line["cfg3_learner_side"] = learner
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        # libraries write to the process's stdout on their own (RCCL prints a version banner when the first
        # communicator is created): everything up to here went to stderr, the JSON line alone goes to stdout
        sys.stdout.flush()
        os.dup2(_STDOUT_FD, 1)
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def _event_us(torch, fn, reps, rounds=3):
    import statistics
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(rounds):
        fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts)


def time_entry_points(env, cfg, device, L, stream):
    """HIP-event times (us per call, C-ABI launches only, outputs preallocated) of reset(), the scramble and the stand-alone
    entry points on the bench batch, each with its own algorithmic bytes and fraction of the HBM peak."""
    import ctypes as C
    import torch
    from tiler_slider_amd import _cabi
    n, S, T = env.num_envs, env.size, env.n_tiles
    Cc = S * S
    blk_b = 2 if Cc <= 16 else 4 * ((Cc + 31) // 32)
    state_read = 2 * T + blk_b
    d, st = env._dims, env._state
    obs = torch.empty_like(env._obs)
    m1, m4 = torch.empty(n, dtype=torch.uint8, device=device), torch.empty((n, 4), dtype=torch.uint8, device=device)
    rw = torch.empty(n, dtype=torch.int32, device=device)
    seeds = torch.arange(n, dtype=torch.int64, device=device).to(torch.int32)
    lv = [torch.zeros_like(env._blk), torch.zeros_like(env._init), torch.zeros_like(env._tgt)]
    gen_st = _cabi.State(None, lv[1].data_ptr(), lv[2].data_ptr(), lv[0].data_ptr(), None, None, None)
    saved = (env._pos.clone(), env._step_count.clone(), env._done.clone())
    calls = {
        # name: (callable, algorithmic bytes per board)
        "ts_reset": (lambda: L.ts_reset(C.byref(d), C.byref(st), obs.data_ptr(), stream), T + blk_b + T + T + 4 + 1 + 12 * Cc),
        "ts_encode": (lambda: L.ts_encode(C.byref(d), C.byref(st), obs.data_ptr(), stream), state_read + 12 * Cc),
        "ts_valid_moves": (lambda: L.ts_valid_moves(C.byref(d), C.byref(st), m1.data_ptr(), stream), T + blk_b + 1),
        "ts_valid_moves4": (lambda: L.ts_valid_moves4(C.byref(d), C.byref(st), m4.data_ptr(), stream), T + blk_b + 4),
        "ts_is_won": (lambda: L.ts_is_won(C.byref(d), C.byref(st), m1.data_ptr(), stream), 2 * T + 1),
        "ts_reward": (lambda: L.ts_reward(C.byref(d), C.byref(st), rw.data_ptr(), stream), 2 * T + 4),
        "ts_generate_mt19937": (lambda: L.ts_generate_mt19937(C.byref(d), C.byref(gen_st), seeds.data_ptr(), cfg["obstacles"], stream),
                                4 + 2 * T + blk_b),
    }
    out = {}
    for name, (fn, bpb) in calls.items():
        us = _event_us(torch, fn, 3 if name == "ts_generate_mt19937" else 20)
        out[name] = {"us": us, "algorithmic_bytes_per_board": bpb, "achieved_GBps": bpb * n / us / 1e3,
                     "frac": bpb * n / us / 1e3 / HBM_PEAK_GBS}
    t0 = time.perf_counter()
    for _ in range(20):
        env.get_valid_moves()
    torch.cuda.synchronize(device)
    out["get_valid_moves_python"] = {"us": (time.perf_counter() - t0) * 1e6 / 20,
                                     "note": "VecTilerSliderEnv.get_valid_moves(): allocation + one ts_valid_moves4 launch + bool view, wall clock"}
    env._pos.copy_(saved[0]), env._step_count.copy_(saved[1]), env._done.copy_(saved[2])
    out["reset_us"], out["scramble_us"] = out["ts_reset"]["us"], out["ts_generate_mt19937"]["us"]
    out["boards"] = n
    return out


def time_learner_side(cfg, n_all, device, L, stream):
    """One ts_encode over all gathered boards (after the compact all-gather) and one ts_expand_u8 over all gathered byte
    observations (after the uint8 all-gather), at the full cfg3 size, on one GPU."""
    import ctypes as C
    import torch
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    S, T = cfg["size"], cfg["tiles"]
    Cc = S * S
    env = VecTilerSliderEnv.random(n_all, size=S, num_tiles=T, num_obstacles=cfg["obstacles"], seed=LEVEL_SEED, multi_color=True,
                                   max_steps=2**30, device=device, placement_trials=0, obs_dtype="uint8")
    u8 = env.reset()                                        # uint8 [n_all, S, S, 3]: what the uint8 all-gather delivers
    from tiler_slider_amd.vec_env import _contiguous_zeros
    out = _contiguous_zeros((n_all, S, S, 3), torch.float32, device)  # the learner's buffer: physically contiguous, as the environments' own
    if out is None:
        out = torch.empty((n_all, S, S, 3), dtype=torch.float32, device=device)
    d, st = env._dims, env._state
    enc_us = _event_us(torch, lambda: L.ts_encode(C.byref(d), C.byref(st), out.data_ptr(), stream), 10)
    exp_us = _event_us(torch, lambda: L.ts_expand_u8(u8.data_ptr(), out.data_ptr(), u8.numel(), stream), 10)
    ok = None
    if n_all <= (1 << 23):  # the two hand-offs deliver the same tensor: ts_encode(cells) == float(ts_encode_u8 bytes)
        L.ts_encode(C.byref(d), C.byref(st), out.data_ptr(), stream)
        ok = bool(torch.equal(out, u8.to(torch.float32)))
    enc_b = (2 * T + (2 if Cc <= 16 else 4 * ((Cc + 31) // 32)) + 12 * Cc) * n_all
    exp_b = 15 * Cc * n_all
    res = {"boards": n_all, "encode_us": enc_us, "expand_us": exp_us,
           "encode": {"algorithmic_bytes": enc_b, "achieved_GBps": enc_b / enc_us / 1e3, "frac": enc_b / enc_us / 1e3 / HBM_PEAK_GBS,
                      "frac_of_copy_ceiling": enc_b / enc_us / 1e3 / COPY_CEILING_GBS},
           "expand": {"algorithmic_bytes": exp_b, "achieved_GBps": exp_b / exp_us / 1e3, "frac": exp_b / exp_us / 1e3 / HBM_PEAK_GBS,
                      "frac_of_copy_ceiling": exp_b / exp_us / 1e3 / COPY_CEILING_GBS,  # a mixed read / write stream: the copy ceiling is its yardstick
                      },
           "frac": min(enc_b / enc_us, exp_b / exp_us) / 1e3 / HBM_PEAK_GBS, "expand_equals_encode": ok,
           "note": "learner GPU of cfg3: ts_encode over 8 x 1,048,576 gathered boards / ts_expand_u8 over their byte observations"}
    del env, out, u8
    return res


ASYMPTOTE_BOARDS = {"cfg1": 1 << 24, "cfg2": 1 << 22, "cfg4": (1 << 20) + (1 << 18)}  # >= 3 GB of large outputs per launch
STATE_IN_CACHE_BOARDS = {"cfg1": 10 << 20, "cfg2": 5 << 19, "cfg4": 3 << 18}              # ~2.2 GB per launch: 8 x the Infinity Cache


def time_env_steps(env, acts, steps, warm=10):
    """us per ts_step (HIP events on the launch stream) of an environment that is already reset."""
    import torch
    m = len(acts) - 1
    for i in range(warm):
        env.step_async(acts[i & m])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        env.step_async(acts[i & m])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps


def action_buffers(n, count, device, L, stream, offset=0):
    import torch
    from tiler_slider_amd import _cabi
    acts = []
    for i in range(count):
        a = torch.empty(n, dtype=torch.uint8, device=device)
        _cabi.check(L.ts_fill_actions(n, ACTION_SEED, offset, i, a.data_ptr(), stream), "ts_fill_actions")
        acts.append(a)
    return acts


def time_asymptote(name, device, L, stream, output_memory="contiguous", boards=None):
    """The same shape and outputs at >= 3 GB of large outputs per launch: twelve times the 256 MiB Infinity Cache and more, where
    what the cache can absorb between two launches is a few per cent of the stream - the HBM-bound figure of this config's kernel
    (every BASELINE config itself lies between 0.9 and 3.4 x the cache, where a third of the stream can be absorbed).
    At that size the STATE a step re-reads (cells, targets, obstacles, counters, line tables: 14 .. 230 B per board) no longer fits
    the cache either, and that - not the output stream - is what the rate falls on (profiles/r05_asymptote_probe.log: cfg1's shape
    0.93 / 0.90 / 0.89 of 8 TB/s at 2.1 / 2.5 / 2.9 GB per launch, 0.67 at 3.4 GB, where its state passes 256 MiB; cfg4's 0.84 / 0.82
    at 2.1 / 2.5 GB, 0.63 / 0.47 at 2.9 / 3.4 GB).  `with_state_in_cache` is the same measurement at ~2.2 GB per launch (8 x the cache),
    the largest round size at which the state still fits."""
    import torch
    from tiler_slider_amd import VecTilerSliderEnv
    cfg = CONFIGS[name]
    n = boards or ASYMPTOTE_BOARDS[name]
    S, T = cfg["size"], cfg["tiles"]
    bps = algorithmic_bytes_per_board_step(S, T, cfg["onehot"], cfg["reward"])
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=cfg["obstacles"], seed=LEVEL_SEED,
                                   multi_color=True, max_steps=2**30, device=device, auto_reset=True, with_reward=cfg["reward"],
                                   with_onehot=cfg["onehot"], output_memory=output_memory, obs_candidates=0)
    env.reset()
    us = time_env_steps(env, action_buffers(n, 2, device, L, stream), 10, warm=3)
    desc = launch_description(env)
    gbs = bps * n / us / 1e3
    large = 12 * S * S + (4 * S * S * (1 + 2 * T) + 4 if cfg["onehot"] else 0)  # observation (+ planes, reward): the streamed outputs
    state = bps - large + (0 if S <= 8 else (64 if S <= 16 else 256))            # + the half of the ts_prepare record a multi-colour launch reads
    res = {"boards": n, "kernel": desc["name"], "kernel_us": us, "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "frac_of_copy_ceiling": gbs / COPY_CEILING_GBS, "algorithmic_bytes_per_launch": bps * n,
           "times_infinity_cache": bps * n / INFINITY_CACHE_BYTES, "value": n / us * 1e6, "value_unit": "env-steps/s", "steps": 10,
           "state_bytes_per_launch": state * n, "state_fits_infinity_cache": state * n < INFINITY_CACHE_BYTES,
           "output_memory": [m["memory"] for m in env.output_memory_report],
           "launch": {k: desc[k] for k in ("boards_per_wave", "cached_every", "emit_edges", "xcd_piece", "blocks_per_cu", "blocks")}}
    del env
    torch.cuda.empty_cache()
    if boards is None and name in STATE_IN_CACHE_BOARDS:
        res["with_state_in_cache"] = time_asymptote(name, device, L, stream, output_memory, boards=STATE_IN_CACHE_BOARDS[name])
    return res


def time_config(name, trials, steps, device, L, stream):
    """Kernel time (HIP events on the launch stream) of one BASELINE config, first allocation and tuned."""
    import torch
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    cfg = CONFIGS[name]
    n = cfg["boards"]
    bps = algorithmic_bytes_per_board_step(cfg["size"], cfg["tiles"], cfg["onehot"], cfg["reward"])
    acts = []
    for i in range(4):
        a = torch.empty(n, dtype=torch.uint8, device=device)
        _cabi.check(L.ts_fill_actions(n, ACTION_SEED, 0, i, a.data_ptr(), stream), "ts_fill_actions")
        acts.append(a)
    out = {"workload": f"{name}: {n:,} concurrent {cfg['size']}x{cfg['size']} boards, T={cfg['tiles']}, K={cfg['obstacles']}, multi_color"
                       + (", + one-hot + Manhattan reward" if cfg["onehot"] else ""),
           "algorithmic_bytes_per_board_step": bps, "algorithmic_bytes_per_launch": bps * n,
           "steps": steps}
    # first_allocation_contiguous_memory: THE CLASS DEFAULT - output buffers in physically contiguous memory, the library's
    # static launch policy, nothing measured at construction (placement_trials = 0): the same in every process;
    # torch_allocator: the same policy on buffers from torch's caching allocator (one of two speeds, by allocation);
    # torch_allocator_rated: + launch policy rated at construction on those buffers; tuned: + candidate buffers (opt-in)
    # class_default: what VecTilerSliderEnv(...) does with no placement arguments - contiguous memory, static policy, and for
    # environments with one-hot planes (two large streams: the observation buffer's place decides between two speeds) the fastest
    # of up to 16 candidate observation buffers; first_allocation_contiguous_memory: the same with obs_candidates=0
    for key, k, mem, cand in (("class_default", 0, "contiguous", None), ("first_allocation_contiguous_memory", 0, "contiguous", 0),
                              ("torch_allocator", 0, "torch", 0), ("torch_allocator_rated", 1, "torch", 0), ("tuned", trials, "torch", 0)):
        if key == "tuned" and trials <= 1:
            continue
        env = VecTilerSliderEnv.random(n, size=cfg["size"], num_tiles=cfg["tiles"], num_obstacles=cfg["obstacles"], seed=LEVEL_SEED,
                                       multi_color=True, max_steps=2**30, device=device, auto_reset=True, with_reward=cfg["reward"],
                                       with_onehot=cfg["onehot"], placement_trials=k, output_memory=mem,
                                       obs_candidates=cand)
        env.reset()
        for i in range(50):
            env.step_async(acts[i & 3])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            env.step_async(acts[i & 3])
        e1.record()
        torch.cuda.synchronize(device)
        us = e0.elapsed_time(e1) * 1e3 / steps
        gbs = bps * n / us / 1e3
        out["kernel"] = launch_description(env)["name"]
        out[key] = {"placement_trials": k, "output_memory": mem, "kernel_us": us, "value": n / us * 1e6, "unit": "env-steps/s", "achieved": gbs,
                    "achieved_unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "frac_of_copy_ceiling": gbs / COPY_CEILING_GBS,
                    "launch_policy": {"launch_hint": env._dims.launch_hint, "emit_edges": env._dims.emit_edges, "lines_lanes": env._dims.lines_lanes, "xcd_piece": env._dims.xcd_piece},
                    "placement": env.placement_report, "observation_placement": env.observation_placement_report}
        del env
        torch.cuda.empty_cache()
    rec = pmc_traffic(name, n)
    if rec is not None:
        out["traffic"] = rec["write_bytes"] + rec["fetch_bytes_x2"]
    out["hbm_asymptote"] = time_asymptote(name, device, L, stream)
    return out


def time_gathers(env, ring, world, n, dist, torch, device, steps):
    """Step + hand-off to a single learner, three ways: (a) RCCL all-gather of the float32
    observations (what north_star names); (b) all-gather of the compact state (cell ids + flags, from actors
    that keep NO observation: obs_dtype=None) and re-encoding on the learner side with ts_encode; (c) all-gather of uint8
    observations and one ts_expand_u8 on the learner.  Every form delivers the step's flags too (done follows from them).
    Each is timed serially (step k, then gather k, on one stream) and overlapped: the environment cycles through two
    observation buffers, gather k runs on RCCL's stream while step k+1 writes the other buffer, and step k+2 waits for gather k;
    and each as an all-gather (every rank receives everything) and as a gather to rank 0 alone (`*_to_root`)."""
    from tiler_slider_amd import VecTilerSliderEnv
    from tiler_slider_amd.distributed import ObservationGatherer

    def twin(obs_dtype):
        e = VecTilerSliderEnv.from_arrays(env.size, env._blk, env._init, env._tgt, multi_color=env.multi_color,
                                          max_steps=env.max_steps, device=device, auto_reset=True, obs_dtype=obs_dtype,
                                          obs_buffers=1 if obs_dtype is None else 2)
        e.reset()
        return e

    def timed(loop):
        dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        loop()
        torch.cuda.synchronize(device)
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        dist.barrier()
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt[0])

    out = {}
    e32, eact, e8 = twin("float32"), twin(None), twin("uint8")
    forms = [("obs_f32", e32, lambda e, g, a: g.gather_observations(e._obs, async_op=a)),
             ("compact_state_then_encode", eact, lambda e, g, a: g.gather_compact_and_encode(async_op=a)),
             ("obs_u8_then_expand", e8, lambda e, g, a: g.gather_u8_and_expand(e._obs, async_op=a))]
    for root, suffix in ((None, ""), (0, "_to_root")):
        for name, e, fn in forms:
            g = ObservationGatherer(e, world, root=root)
            fn(e, g, False)  # warm-up (RCCL channel set-up)

            def serial():
                for i in range(steps):
                    e.step_async(ring[i & 15])
                    fn(e, g, False)

            def overlapped():
                prev = None
                for i in range(steps):
                    e.step_async(ring[i & 15])          # writes observation buffer i % 2
                    if prev is not None and not prev.two_phase:
                        prev.wait()                     # gather i-1 has read buffer (i-1) % 2 ...
                    elif prev is not None:
                        prev.receive()                  # compact form: message i-1 unpacked, the receive image is free ...
                    cur = fn(e, g, True)                # ... gather i starts behind step i, beside step i+1
                    if prev is not None and prev.two_phase:
                        prev.wait()                     # ... and beside the learner's ts_encode of step i-1 (all boards)
                    prev = cur
                prev.wait()

            ts, to = timed(serial), timed(overlapped)
            out[name + suffix] = {"value": n * world * steps / ts, "value_overlapped": n * world * steps / to, "unit": "env-steps/s",
                                  "steps": steps, "ms_per_step_serial": ts / steps * 1e3, "ms_per_step_overlapped": to / steps * 1e3,
                                  "bytes_per_rank_per_step": g.bytes_per_step[name], "delivers": "obs + flags (done)",
                                  "collective": "all_gather" if root is None else "gather to rank 0",
                                  "actor_step": ("no observation (obs_dtype=None)" if e is eact else
                                                 "ring of two observation buffers, %s forms" % ("out-of-cache" if e._dims.ring_bytes > INFINITY_CACHE_BYTES else "cache-resident"))}
            del g
            torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
