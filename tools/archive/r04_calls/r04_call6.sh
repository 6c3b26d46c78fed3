#!/bin/bash
set -e
mkdir -p gpurun_out/r04f
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dealt_tiles or lanes_per_board" > gpurun_out/r04f/pytest_deal.log 2>&1 || { tail -40 gpurun_out/r04f/pytest_deal.log; exit 1; }
tail -3 gpurun_out/r04f/pytest_deal.log
python tools/deal_ab.py > gpurun_out/r04f/deal_ab.log 2>&1
cat gpurun_out/r04f/deal_ab.log
TS_SWEEP_HINTS=-4,-2,0,2,4 python tools/residency_sweep.py > gpurun_out/r04f/residency_contiguous_after.log 2>&1
cat gpurun_out/r04f/residency_contiguous_after.log
