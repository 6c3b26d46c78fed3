#!/usr/bin/env python3
"""Round 4: kernel-level times of the stand-alone entry points (reset, encode, valid moves, is_won, reward, one-hot, scramble)
and of the fused step at a bench config, from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/entry_points_profile.py run cfg1     (GPU box)
    python3 tools/entry_points_profile.py summarize DIR cfg1 >> profiles/r04_entry_points.md                  (anywhere)

`run` calls every entry point REPS times through the C-ABI, the groups separated by 3 launches of a one-byte ts_fill_actions
(the marker); `summarize` cuts the trace at the markers and prints kernel, average us, the op's own algorithmic bytes and its
fraction of the 8 TB/s HBM peak."""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS = 25
WARM = 15


def ops_for(cfg, T, S, onehot_ch):
    Cc = S * S
    blk = 2 if Cc <= 16 else 4 * ((Cc + 31) // 32)
    lines = 0 if S <= 8 else (128 if S <= 16 else 512)  # the per-level line tables the large-board kernel reads
    state_rd = 2 * T + blk + lines
    return [  # name, algorithmic bytes per board
        ("ts_step (flags + float32 observation)", 2 * T + blk + 6 + T + 6 + 12 * Cc + lines),
        ("ts_step + reward + valid + valid4", 2 * T + blk + 6 + T + 6 + 12 * Cc + 4 + 1 + 4 + lines),
        ("ts_reset (+ observation)", T + blk + T + T + 5 + 12 * Cc + lines),
        ("ts_encode", state_rd + 12 * Cc),
        ("ts_encode_u8", state_rd + 3 * Cc),
        ("ts_valid_moves (bit mask)", T + blk + 1 + lines),
        ("ts_valid_moves4 (uint8 [N][4], the reference's shape)", T + blk + 4 + lines),
        ("ts_is_won", 2 * T + 1 + (lines if S > 8 else 0)),
        ("ts_reward", 2 * T + 4 + (lines if S > 8 else 0)),
        ("ts_encode_onehot", state_rd + 4 * Cc * onehot_ch),
        ("ts_generate_mt19937 (scramble)", 4 + 2 * T + blk),
    ]


def run(cfgname):
    import ctypes as C
    import torch
    import bench
    from tiler_slider_amd import VecTilerSliderEnv, _cabi
    cfg = bench.CONFIGS[cfgname]
    n, S, T = cfg["boards"], cfg["size"], cfg["tiles"]
    L = _cabi.lib()
    env = VecTilerSliderEnv.random(n, size=S, num_tiles=T, num_obstacles=cfg["obstacles"], seed=1, multi_color=True, max_steps=2**30, auto_reset=True)
    envx = VecTilerSliderEnv.from_arrays(S, env._blk, env._init, env._tgt, multi_color=True, max_steps=2**30, auto_reset=True, with_reward=True,
                                         with_valid_moves=True)
    env.reset(), envx.reset()
    dev = env.device
    stream = torch.cuda.current_stream(dev).cuda_stream
    act = torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev)
    obs, obs8 = torch.empty_like(env._obs), torch.empty(tuple(env._obs.shape), dtype=torch.uint8, device=dev)
    m1, m4 = torch.empty(n, dtype=torch.uint8, device=dev), torch.empty((n, 4), dtype=torch.uint8, device=dev)
    rw = torch.empty(n, dtype=torch.int32, device=dev)
    oh = torch.empty((n, env.onehot_channels, S, S), dtype=torch.float32, device=dev)
    seeds = torch.arange(n, dtype=torch.int64, device=dev).to(torch.int32)
    lv = [torch.zeros_like(env._blk), torch.zeros_like(env._init), torch.zeros_like(env._tgt)]
    gen_st = _cabi.State(None, lv[1].data_ptr(), lv[2].data_ptr(), lv[0].data_ptr(), None, None, None)
    d, st = env._dims, env._state
    one = torch.zeros(1, dtype=torch.uint8, device=dev)
    calls = [lambda: env.step_async(act), lambda: envx.step_async(act),
             lambda: L.ts_reset(C.byref(d), C.byref(st), obs.data_ptr(), stream),
             lambda: L.ts_encode(C.byref(d), C.byref(st), obs.data_ptr(), stream),
             lambda: L.ts_encode_u8(C.byref(d), C.byref(st), obs8.data_ptr(), stream),
             lambda: L.ts_valid_moves(C.byref(d), C.byref(st), m1.data_ptr(), stream),
             lambda: L.ts_valid_moves4(C.byref(d), C.byref(st), m4.data_ptr(), stream),
             lambda: L.ts_is_won(C.byref(d), C.byref(st), m1.data_ptr(), stream),
             lambda: L.ts_reward(C.byref(d), C.byref(st), rw.data_ptr(), stream),
             lambda: L.ts_encode_onehot(C.byref(d), C.byref(st), oh.data_ptr(), stream),
             lambda: L.ts_generate_mt19937(C.byref(d), C.byref(gen_st), seeds.data_ptr(), cfg["obstacles"], stream)]
    assert len(calls) == len(ops_for(cfg, T, S, env.onehot_channels))
    for _ in range(400):  # bring the clocks up before the first group (tens of ms of load)
        env.step_async(act)
    torch.cuda.synchronize()
    for fn in calls:
        for _ in range(3):
            L.ts_fill_actions(1, 7, 0, 0, one.data_ptr(), stream)  # marker
        for _ in range(WARM + REPS):  # the summary keeps the last REPS launches of a group (the first ones ramp the clocks up)
            fn()
        torch.cuda.synchronize()
    print("ran", cfgname, n, "boards, onehot channels", env.onehot_channels)


def summarize(directory, cfgname):
    import bench
    cfg = bench.CONFIGS[cfgname]
    n, S, T = cfg["boards"], cfg["size"], cfg["tiles"]
    trace = glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    groups, cur, markers = [], None, 0
    for r in rows:
        if "k_fill_actions" in r["Kernel_Name"] and r.get("Grid_Size_X", r.get("Grid_Size", "")) in ("256", "64", "1"):
            markers += 1
            if markers == 3:
                cur = []
                groups.append(cur)
                markers = 0
            continue
        markers = 0
        if cur is not None:
            cur.append(r)
    ops = ops_for(cfg, T, S, 1 + 2 * T)
    print(f"\n## {cfgname}: {n:,} boards of {S}x{S}, T = {T} (multi_color) - rocprofv3 --kernel-trace, {REPS} launches per entry point\n")
    print("| entry point | kernel | avg us | min us | algorithmic bytes per launch | GB/s | frac of 8 TB/s |")
    print("|---|---|---|---|---|---|---|")
    for (name, bpb), g in zip(ops, groups):
        g = g[-REPS:]
        if not g:
            continue
        us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in g]
        kern = (re.search(r"k_\w+(<[^>]*>)?", g[-1]["Kernel_Name"]) or [g[-1]["Kernel_Name"][:50]])[0]
        avg = sum(us) / len(us)
        print(f"| {name} | `{kern}` | {avg:.2f} | {min(us):.2f} | {bpb * n:,} | {bpb * n / avg / 1e3:.0f} | {bpb * n / avg / 1e3 / 8000:.3f} |")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        summarize(sys.argv[2], sys.argv[3])
