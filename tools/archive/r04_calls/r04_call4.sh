#!/bin/bash
set -e
mkdir -p gpurun_out/r04d
python tools/mem_ab.py > gpurun_out/r04d/mem_ab.log 2>&1
cat gpurun_out/r04d/mem_ab.log
python -m pytest tests/test_mt19937_levels.py -m gpu -x -q > gpurun_out/r04d/pytest_mt.log 2>&1 || { tail -40 gpurun_out/r04d/pytest_mt.log; exit 1; }
tail -3 gpurun_out/r04d/pytest_mt.log
python tools/scramble_timing.py > gpurun_out/r04d/scramble.log 2>&1
cat gpurun_out/r04d/scramble.log
