#!/bin/bash
set -o pipefail
OUT=gpurun_out/r02_s13
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_gpu.log || exit 1
timeout -k 10 600 python tools/shape_sweep.py 2>&1 | grep -v amdgpu.ids | tee $OUT/shape_sweep.log || exit 1
for cfg in s8t20; do timeout -k 10 200 python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg kernel_us', round(d['roofline']['kernel_us'],2), 'frac', round(d['roofline']['frac'],3))" | tee -a $OUT/misc.log; done
timeout -k 10 300 python tools/aux_ops_timing.py 2>&1 | grep -v amdgpu.ids | tee $OUT/aux_ops.log
