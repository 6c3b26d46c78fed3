#!/bin/bash
# Round 5, call 13 (GPU box): k_state with the two-buffer tile walk (old / previous / pipelined), then the GPU suite (with the 20M-board
# oracle replay of the launch form for state beyond the cache).
set -o pipefail
OUT=gpurun_out/r05_call13
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 400 python tools/state_only_ab.py --shapes 15,32,24,262144 15,32,24,1048576 9,4,9,1048576 16,64,20,131072 32,64,100,65536 > $OUT/state_only_ab.log 2>&1 || { tail -30 $OUT/state_only_ab.log; exit 1; }
grep -v amdgpu.ids $OUT/state_only_ab.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
