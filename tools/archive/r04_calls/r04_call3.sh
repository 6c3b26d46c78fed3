#!/bin/bash
set -e
mkdir -p gpurun_out/r04c
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dealt_tiles" > gpurun_out/r04c/pytest_deal.log 2>&1 || { tail -40 gpurun_out/r04c/pytest_deal.log; exit 1; }
tail -3 gpurun_out/r04c/pytest_deal.log
python tools/deal_ab.py > gpurun_out/r04c/deal_ab.log 2>&1
cat gpurun_out/r04c/deal_ab.log
python tools/piece_by_shape.py > gpurun_out/r04c/piece_by_shape.log 2>&1
cat gpurun_out/r04c/piece_by_shape.log
