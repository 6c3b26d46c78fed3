#!/bin/bash
set -e
mkdir -p gpurun_out/r04ad
rm -rf gpurun_out/r04_prof
python -m pytest tests -m gpu -x -q > gpurun_out/r04ad/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r04ad/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r04ad/pytest_gpu.log
bash tools/r04_profile.sh > gpurun_out/r04ad/profile.log 2>&1 || { tail -30 gpurun_out/r04ad/profile.log; exit 1; }
tail -8 gpurun_out/r04ad/profile.log
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04_prof/bench_default_steps20.json 2>> gpurun_out/r04_prof/bench_default.err
