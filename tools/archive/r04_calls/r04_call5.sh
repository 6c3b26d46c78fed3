#!/bin/bash
set -e
mkdir -p gpurun_out/r04e
python tools/residency_sweep.py > gpurun_out/r04e/residency_contiguous.log 2>&1
cat gpurun_out/r04e/residency_contiguous.log
TS_SWEEP_MEM=torch python tools/residency_sweep.py > gpurun_out/r04e/residency_torch.log 2>&1
cat gpurun_out/r04e/residency_torch.log
