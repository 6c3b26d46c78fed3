#!/usr/bin/env python3
"""Markdown table of rocprofv3 --pmc passes laid out as <dir>/pmc_<shape>_<group>/**/*counter_collection.csv:
mean counter value per launch of the kernel of interest (k_lines), one column per shape, plus the kernel's average duration from
the kernel trace of the same passes.  argv: <dir> [kernel substring]"""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_lines"
vals = collections.defaultdict(lambda: collections.defaultdict(list))   # counter -> shape -> values
kern = {}
dur = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    m = re.match(r"pmc_([^_]+)_(.+)$", os.path.basename(d))
    if not m:
        continue
    shape = m.group(1)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                name = (re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"]) or [r["Kernel_Name"][:60]])[0]
                kern[shape] = f"{name} grid {r['Grid_Size']} wg {r['Workgroup_Size']} VGPR {r['VGPR_Count']} LDS {r['LDS_Block_Size']}"
                vals[r["Counter_Name"]][shape].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                dur[shape].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
shapes = sorted(kern)
print("# counters per launch (mean over the launches of each pass)\n")
for s in shapes:
    print(f"* `{s}`: `{kern[s]}`; duration under the counter passes {sum(dur[s]) / max(len(dur[s]), 1):.2f} us over {len(dur[s])} launches")
print("\n| counter | " + " | ".join(shapes) + " | ratio first / last |")
print("|---|" + "---|" * (len(shapes) + 1))
for c in sorted(vals):
    means = [sum(vals[c][s]) / len(vals[c][s]) if vals[c][s] else float("nan") for s in shapes]
    ratio = means[0] / means[-1] if len(means) > 1 and means[-1] else float("nan")
    print(f"| {c} | " + " | ".join(f"{m:,.0f}" for m in means) + f" | {ratio:.2f} |")
