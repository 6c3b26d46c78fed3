#!/bin/bash
# Round 5, call 31 (GPU box): the extended smoke, the GPU suite, and the bench lines of the final build.
set -o pipefail
OUT=gpurun_out/r05_call31
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -2 || exit 1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench_driver_like.err || { tail -5 $OUT/bench_driver_like.err; exit 1; }
echo benches done
