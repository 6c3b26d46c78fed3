#!/bin/bash
set -e
mkdir -p gpurun_out/r04ak
python -m pytest tests -m gpu -x -q > gpurun_out/r04ak/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r04ak/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r04ak/pytest_gpu.log
python tools/shape_sweep.py > gpurun_out/r04ak/shape_sweep.log 2>&1
cat gpurun_out/r04ak/shape_sweep.log
for k in 1 2; do timeout -k 10 300 python3 bench.py --config cfg4 --no-cpu-baseline --no-pipelined --no-other-configs --no-entry-points --no-learner-side > gpurun_out/r04ak/cfg4_$k.json 2>> gpurun_out/r04ak/err.log; python -c "
import json; x=json.loads(open('gpurun_out/r04ak/cfg4_$k.json').read().strip().splitlines()[-1]); print('cfg4 kernel_us %.2f frac %.4f' % (x['roofline']['kernel_us'], x['roofline']['frac']))"; done
