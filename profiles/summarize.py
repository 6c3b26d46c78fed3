#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/..., scratch) into the small summaries kept
under profiles/ (tracked).  Usage:
    python profiles/summarize.py <tag> --stats <dir> [--pmc <dir> ...] [--bench <json>]
writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3's --stats table) and
profiles/<tag>_summary.md (per-kernel launch time; PMC counters per launch, with the gfx950
corrections of /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB,
WRITE_SIZE is exact for 16-B streaming stores, FETCH_SIZE reads 1/2 of a wide coalesced stream
and is uncalibrated for narrow loads — both raw and doubled values are shown)."""
import argparse
import collections
import csv
import glob
import json
import re
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--stats")
    ap.add_argument("--pmc", action="append", default=[])
    ap.add_argument("--bench")
    ap.add_argument("--kernel", default="k_small", help="substring of the kernel of interest")
    a = ap.parse_args()
    lines = [f"# {a.tag}", ""]
    if a.bench:
        b = json.loads(open(a.bench).read().strip().splitlines()[-1])
        lines += ["## bench.py line (same command, run under rocprofv3)", "", "```json", json.dumps(b, indent=1), "```", ""]
    if a.stats:
        f = glob.glob(os.path.join(a.stats, "**", "*kernel_stats.csv"), recursive=True)[0]
        shutil.copy(f, os.path.join(HERE, f"{a.tag}_kernel_stats.csv"))
        lines += ["## rocprofv3 --kernel-trace --stats", "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
        for r in csv.DictReader(open(f)):
            lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | "
                         f"{float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
        lines.append("")
        traces = glob.glob(os.path.join(a.stats, "**", "*kernel_trace.csv"), recursive=True)
        if traces:
            per = collections.defaultdict(list)
            for r in csv.DictReader(open(traces[0])):
                name = (re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"]) or [r["Kernel_Name"][:60]])[0]
                per[(name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""))].append(
                    (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            lines += ["## the same trace by (kernel, grid): one bench run launches the headline kernel at more than one batch size",
                      "", "| kernel | grid (threads) | block | LDS/block | launches | avg us | min us | max us |", "|---|---|---|---|---|---|---|---|"]
            for (name, grid, wg, lds), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
                lines.append(f"| `{name}` | {grid} | {wg} | {lds} | {len(v)} | {sum(v) / len(v):.2f} | {min(v):.2f} | {max(v):.2f} |")
            lines.append("")
    for d in a.pmc:
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if a.kernel in r["Kernel_Name"]:
                agg[((re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"]) or [r["Kernel_Name"][:60]])[0], r["Counter_Name"], r["Grid_Size"], r["VGPR_Count"],
                     r["LDS_Block_Size"])].append(float(r["Counter_Value"]))
        lines += [f"## rocprofv3 --pmc ({os.path.basename(d.rstrip('/'))})", "",
                  "| kernel | counter | launches | mean per launch | as bytes (KiB x 1024) | grid | VGPR | LDS/block |", "|---|---|---|---|---|---|---|---|"]
        for (k, c, grid, vg, lds), v in agg.items():
            m = sum(v) / len(v)
            extra = f"{m * 1024 / 1e6:.2f} MB" + (f" (x2 = {2 * m * 1024 / 1e6:.2f} MB if wide-read rule applies)" if c == "FETCH_SIZE" else "") \
                if c in ("FETCH_SIZE", "WRITE_SIZE") else ""
            lines.append(f"| `{k}` | {c} | {len(v)} | {m:.1f} | {extra} | {grid} | {vg} | {lds} |")
        lines.append("")
    open(os.path.join(HERE, f"{a.tag}_summary.md"), "w").write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
