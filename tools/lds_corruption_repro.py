#!/usr/bin/env python3
"""Reproducer of round 1's "64 KiB LDS" corruption (development tool).

Round 2 found that the LDS size was never the cause (profiles/r02_lds_corruption_bisect.log,
profiles/r02_lds_probe.log).  What fails is one COMPILED FORM of k_small<8, 0, false>: the current
source with the single-pass observation block of commit 0690d71 pasted back, at -O2 / -O3
(-O1 of the same text is clean).  This script rebuilds that form from the current source
(it reproduced the failure when k_small still took the store policy as a run-time flag, commit 06962e4; later
source changes move the register allocation, so a clean run today does not mean the defect is gone):

    python tools/lds_corruption_repro.py build      # here: writes build/variants/cur_oldobs*.so
    TS_SHOW_DIFF=1 python tools/check_variants_vs_oracle.py 8,20,10 524288 cur_oldobs,cur_oldobs_O1,base   # GPU box
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tiler_slider_amd", "csrc", "ts_kernels.hip")
VDIR = os.path.join(ROOT, "build", "variants")

OLD_OBS_BLOCK = '''  if (a.obs) {
    for (int off = lane * 16; off < kImg; off += kWave * 16) *reinterpret_cast<uint4 *>(img + off) = make_uint4(0, 0, 0, 0);
    wave_sync();
    if (live) {
      unsigned char *my = img + lane * (3 * C);
      for (M m = blk; m; m &= m - 1) my[3 * ts::lsb(m)] = 1;
      if constexpr (TFIX > 0) {
#pragma unroll
        for (int t = 0; t < TFIX; ++t) my[3 * q[t] + 1] = (unsigned char)(mc ? t + 1 : 1);
#pragma unroll
        for (int t = 0; t < TFIX; ++t) my[3 * tg[t] + 2] = (unsigned char)(mc ? t + 1 : 1);
      } else {
        for (int t = 0; t < T; ++t) my[3 * st_np[t * kWave + lane] + 1] = (unsigned char)(mc ? t + 1 : 1);
        for (int j = 0; j < Tt; ++j) my[3 * st_tg[j * kWave + lane] + 2] = (unsigned char)(mc ? j + 1 : 1);
      }
    }
    wave_sync();
    emit_bytes_as_f32<NT>(img, a.obs + n0 * (3 * C), nb * 3 * C, lane);
  }

'''


def main():
    if len(sys.argv) < 2 or sys.argv[1] != "build":
        raise SystemExit(__doc__)
    cur = open(SRC).read()
    start = cur.index("  // kObsBoards boards per pass: all 64 up to 5x5; two passes of 32 from 6x6 on")
    end = cur.index("  // ---- build-defined one-hot planes [board][Ch][S][S] ----")
    text = cur[:start] + OLD_OBS_BLOCK + cur[end:]
    text = text.replace('"../../include/tiler_slider.h"', f'"{ROOT}/include/tiler_slider.h"')
    text = text.replace('"ts_core.h"', f'"{ROOT}/tiler_slider_amd/csrc/ts_core.h"')
    os.makedirs(VDIR, exist_ok=True)
    gen = os.path.join(ROOT, "build", "cur_oldobs.hip")
    open(gen, "w").write(text)
    mpath = os.path.join(VDIR, "manifest.json")
    manifest = json.load(open(mpath)) if os.path.exists(mpath) else {}
    procs = []
    for name, opt in (("cur_oldobs", "-O3"), ("cur_oldobs_O2", "-O2"), ("cur_oldobs_O1", "-O1")):
        cmd = ["hipcc", "--offload-arch=gfx950", opt, "-std=c++17", "-shared", "-fPIC", "-DTS_FORCE_OBS_BOARDS=64",
               "-o", os.path.join(VDIR, name + ".so"), gen]
        procs.append((name, subprocess.Popen(cmd)))
        manifest[name] = [f"current source + observation block of 0690d71, {opt}"]
    for name, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"{name} failed to compile")
    json.dump(manifest, open(mpath, "w"), indent=1)
    print("built", [n for n, _ in procs])


if __name__ == "__main__":
    main()
