#!/bin/bash
# Round 5, call 6 (GPU box): GPU suite + smoke of the build, the round's profiles (tools/r05_profile.sh), the asymptote probe again
# (single-edge cap; boards per wave, lanes per board, more resident blocks at 3.4 GB).
set -o pipefail
OUT=gpurun_out/r05_call06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1 || exit 1
bash tools/r05_profile.sh || exit 1
MB=2100,3400 timeout -k 10 500 python tools/asymptote_probe.py 4,2,2 15,32,24 13,3,10 9,4,9 > $OUT/asymptote_probe.log 2>&1 || { tail -30 $OUT/asymptote_probe.log; exit 1; }
grep -v amdgpu.ids $OUT/asymptote_probe.log
