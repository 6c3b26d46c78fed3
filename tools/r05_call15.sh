#!/bin/bash
# Round 5, call 15 (GPU box): GPU suite with the chunk rule for four-lane boards, then the 500 MB shape sweep on contiguous memory.
set -o pipefail
OUT=gpurun_out/r05_call15
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
timeout -k 10 600 python tools/shape_sweep.py > $OUT/shape_sweep.log 2>&1 || { tail -20 $OUT/shape_sweep.log; exit 1; }
grep -v amdgpu.ids $OUT/shape_sweep.log
