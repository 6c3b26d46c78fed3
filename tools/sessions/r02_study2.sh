#!/bin/bash
# Round-2 GPU session 2: (a) the kernel source of commit 0690d71 (parent of round 1's LDS fix) on the
# shape that failed then, once, against the oracle; (b) placement study: step time vs buffer address.
set -o pipefail
OUT=gpurun_out/r02_s2
mkdir -p $OUT
timeout -k 10 300 python tools/check_variants_vs_oracle.py 8,20,10 524288 base,old_0690d71,lds_old64 2>&1 | tee $OUT/lds_prefix_kernel.log || exit 1
timeout -k 10 300 python tools/placement_study.py cfg4 8 2>&1 | tee $OUT/placement_cfg4.log || exit 1
timeout -k 10 300 python tools/placement_study.py cfg2 6 2>&1 | tee $OUT/placement_cfg2.log || exit 1
timeout -k 10 300 python tools/placement_study.py cfg1 6 2>&1 | tee $OUT/placement_cfg1.log || exit 1
for i in 1 2 3 4 5; do
  timeout -k 10 200 python bench.py --config cfg4 --steps 300 --warmup 20 --no-cpu-baseline > $OUT/bench_cfg4_run$i.json 2>/dev/null || exit 1
  python -c "import json; d=json.loads(open('$OUT/bench_cfg4_run$i.json').read().strip().splitlines()[-1]); print('bench.py cfg4 process $i: kernel_us', round(d['roofline']['kernel_us'],2))" | tee -a $OUT/bench_cfg4_processes.log
done
