#!/bin/bash
set -o pipefail
OUT=gpurun_out/r02_s11c
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_gpu.log || exit 1
for cfg in cfg4 cfg2; do
for i in 1 2 3 4 5 6; do
  timeout -k 10 200 python bench.py --config $cfg --steps 300 --warmup 20 --no-cpu-baseline > $OUT/bench_${cfg}_run$i.json 2>/dev/null || exit 1
  python -c "import json; d=json.loads(open('$OUT/bench_${cfg}_run$i.json').read().strip().splitlines()[-1]); print('bench.py $cfg process $i: kernel_us', round(d['roofline']['kernel_us'],2), 'frac', round(d['roofline']['frac'],3))" | tee -a $OUT/bench_processes.log
done
done
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_cfg1.json 2>/dev/null || exit 1
python -c "import json; d=json.loads(open('$OUT/bench_cfg1.json').read().strip().splitlines()[-1]); print('cfg1', d['roofline']['kernel_us'], d['roofline']['hbm_sibling'])" | tee -a $OUT/bench_processes.log
