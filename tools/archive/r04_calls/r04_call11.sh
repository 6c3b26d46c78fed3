#!/bin/bash
set -e
mkdir -p gpurun_out/r04l
python -m pytest tests -m gpu -x -q > gpurun_out/r04l/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r04l/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r04l/pytest_gpu.log
python tools/small_bpw_sweep.py > gpurun_out/r04l/small_bpw.log 2>&1
cat gpurun_out/r04l/small_bpw.log
python tools/shape_sweep.py > gpurun_out/r04l/shape_sweep.log 2>&1
cat gpurun_out/r04l/shape_sweep.log
TS_SWEEP_MEM=torch python tools/shape_sweep.py > gpurun_out/r04l/shape_sweep_torch.log 2>&1
cat gpurun_out/r04l/shape_sweep_torch.log
