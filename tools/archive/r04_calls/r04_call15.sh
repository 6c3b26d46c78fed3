#!/bin/bash
set -e
mkdir -p gpurun_out/r04q
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dealt_tiles or large_batches or full_size" > gpurun_out/r04q/pytest.log 2>&1 || { tail -40 gpurun_out/r04q/pytest.log; exit 1; }
tail -3 gpurun_out/r04q/pytest.log
python tools/deal_ab.py > gpurun_out/r04q/deal_ab.log 2>&1
cat gpurun_out/r04q/deal_ab.log
