#!/usr/bin/env python3
"""Development tool (round 3): several fresh allocations of one out-of-cache environment in ONE process, a few steps on each
(eighths mapping, static policy), so that a rocprofv3 --pmc pass of this script has fast and slow buffers side by side:

    rocprofv3 --kernel-trace --pmc <counters> -d out -- python3 tools/lottery_pmc.py [S T K N]
    python3 tools/lottery_pmc.py summarize out      # per allocation: mean duration and counters per launch"""
import collections
import csv
import glob
import os
import sys

if len(sys.argv) > 1 and sys.argv[1] == "channels":
    # per-instance values of a raw (not _sum) counter: how evenly do the L2 channels share the traffic of a launch?
    root = sys.argv[2]
    trace = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
    cnt = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(trace)):
        if "k_small" in r["Kernel_Name"] or "k_lines" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    rows = collections.defaultdict(list)
    cols = None
    for r in csv.DictReader(open(cnt)):
        cols = cols or list(r.keys())
        if r["Dispatch_Id"] in dur:
            rows[(r["Dispatch_Id"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    print("columns:", cols)
    ids = sorted(dur, key=int)
    per = int(os.environ.get("LOTTERY_STEPS", "12")) + 1
    for a in range(len(ids) // per):
        i = ids[a * per + 6]
        for name in sorted({k[1] for k in rows}):
            v = rows[(i, name)]
            m = sum(v) / len(v)
            print(f"allocation {a} ({dur[i]:6.1f} us) {name}: {len(v)} instances, mean {m:.0f}, min {min(v):.0f}, max {max(v):.0f}, "
                  f"max/mean {max(v) / m:.3f}, rms dev {(sum((x - m) ** 2 for x in v) / len(v)) ** 0.5 / m:.4f}")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "summarize":
    root = sys.argv[2]
    trace = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
    cnt = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(trace)):
        if "k_small" in r["Kernel_Name"] or "k_lines" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    vals = collections.defaultdict(dict)
    for r in csv.DictReader(open(cnt)):
        if r["Dispatch_Id"] in dur:
            vals[r["Dispatch_Id"]][r["Counter_Name"]] = vals[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(dur, key=int)
    per = int(os.environ.get("LOTTERY_STEPS", "12")) + 1  # reset + steps per allocation
    names = sorted({k for v in vals.values() for k in v})
    print("allocation  launches  mean us   " + "  ".join(f"{n:>34s}" for n in names))
    for a in range(len(ids) // per):
        chunk = ids[a * per + 3:(a + 1) * per]  # skip the reset and two warm-up steps
        row = f"{a:10d} {len(chunk):9d} {sum(dur[i] for i in chunk) / len(chunk):8.1f}   "
        row += "  ".join(f"{sum(vals[i].get(n, 0.0) for i in chunk) / len(chunk):34.1f}" for n in names)
        print(row)
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tiler_slider_amd import VecTilerSliderEnv  # noqa: E402

S, T, K, N = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (4, 2, 2, 1 << 22)
steps = int(os.environ.get("LOTTERY_STEPS", "12"))
keep = []
act = None
for a in range(int(os.environ.get("LOTTERY_ALLOCS", "6"))):
    env = VecTilerSliderEnv.random(N, size=S, num_tiles=T, num_obstacles=K, seed=1, multi_color=True, max_steps=2**30, auto_reset=True,
                                   placement_trials=0)
    env._dims.xcd_piece = 1  # one contiguous eighth of the batch per XCD: the mapping that shows the two speeds most clearly
    keep.append(env)
    if act is None:
        act = torch.randint(0, 4, (N,), dtype=torch.uint8, device=env.device)
    env.reset()
    for i in range(steps):
        env.step_async(act)
    torch.cuda.synchronize()
