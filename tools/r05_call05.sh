#!/bin/bash
# Round 5, call 5 (GPU box): k_state with the batch chosen by tile count (old / first rewrite / final), GPU suite, where the
# rate falls between 2 and 3.4 GB per launch (asymptote probe).
set -o pipefail
OUT=gpurun_out/r05_call05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 400 python tools/state_only_ab.py > $OUT/state_only_ab.log 2>&1 || { tail -30 $OUT/state_only_ab.log; exit 1; }
grep -v amdgpu.ids $OUT/state_only_ab.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
timeout -k 10 500 python tools/asymptote_probe.py > $OUT/asymptote_probe.log 2>&1 || { tail -30 $OUT/asymptote_probe.log; exit 1; }
grep -v amdgpu.ids $OUT/asymptote_probe.log
