#!/usr/bin/env python3
"""profiles/traffic_pmc.json from the rocprofv3 --pmc passes of tools/r05_pmc_traffic.sh (gpurun_out/r05_prof/pmc_<config>_<counter>):
mean WRITE_SIZE / FETCH_SIZE per launch of the dominant kernel (KiB -> bytes), which bench.py reports as roofline.traffic.
    python profiles/make_traffic_pmc.py gpurun_out/r05_prof r05"""
import collections, csv, glob, json, os, sys

HERE = os.path.dirname(os.path.abspath(__file__))
src, tag = sys.argv[1], sys.argv[2]
CFG = {"cfg1": ("k_multi<4, 2, false, 2>", 1048576, 212), "cfg2": ("k_small<5, 2, true, true>", 1048576, 826),
       "cfg4": ("k_lines<false, 16, 2, true, false>", 262144, 2840), "sib4m": ("k_small<4, 2, false, true>", 4194304, 212)}
out = {"_comment": f"HBM traffic per launch from the rocprofv3 PMC passes of round {tag[1:]} (one counter per pass: --pmc WRITE_SIZE / --pmc FETCH_SIZE; units KiB -> bytes x1024), "
                   "command: rocprofv3 --kernel-trace --pmc <C> -- python3 bench.py <config args> --no-cpu-baseline --no-pipelined --no-other-configs --no-entry-points "
                   "--no-learner-side --steps 30 --warmup 5 (tools/r05_pmc_traffic.sh; class defaults: physically contiguous output buffers, static launch policy). gfx950 corrections per "
                   "/opt/skills/guides/MI355X_MICROARCH.md section HBM: WRITE_SIZE is exact for 16-B streaming stores; FETCH_SIZE reads 1/2 of a wide coalesced stream and is "
                   "uncalibrated for narrow loads (the state loads here are 1-4 B per lane), so both the raw and the doubled value are kept and `traffic` uses the doubled one "
                   f"(upper bound). Sources: profiles/{tag}_{{cfg1,cfg2,cfg4,sib4m}}_summary.md."}
for name, (kernel, boards, bps) in CFG.items():
    rec = {"boards": boards, "kernel": kernel, "algorithmic_bytes": boards * bps}
    for counter, key in (("WRITE_SIZE", "write_bytes"), ("FETCH_SIZE", "fetch_bytes_raw")):
        f = glob.glob(os.path.join(src, f"pmc_{name}_{counter}", "**", "*counter_collection.csv"), recursive=True)[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if r["Counter_Name"] == counter and kernel.replace(" ", "") in r["Kernel_Name"].replace(" ", "")]
        rec[key] = int(round(sum(vals) / len(vals) * 1024))
        rec[key.replace("bytes", "launches").replace("_raw", "")] = len(vals)
    rec["fetch_bytes_x2"] = 2 * rec["fetch_bytes_raw"]
    rec["traffic_over_algorithmic"] = round((rec["write_bytes"] + rec["fetch_bytes_x2"]) / rec["algorithmic_bytes"], 4)
    out["cfg1_sibling_4m" if name == "sib4m" else name] = rec
json.dump(out, open(os.path.join(HERE, "traffic_pmc.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "_comment"}, indent=1))
