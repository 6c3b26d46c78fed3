#!/bin/bash
# Round-2 GPU session 1 (one lease): (a) dynamic-LDS probe, (b) the 64 KiB LDS corruption with the
# real kernel as -D variants, (c) bench.py vs variant_bench.py on the same box, burst vs sustained,
# with clocks logged, (d) cfg2 between commits 2184215 and 974f060 as variants, (e) cfg2 PMC passes.
set -o pipefail
OUT=gpurun_out/r02_s1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
smi() { rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" | head -12; }
echo "== lds probe" | tee $OUT/log.txt
timeout -k 10 120 python tools/lds_probe.py run 2>&1 | tee $OUT/lds_probe.log || exit 1
echo "== 64 KiB LDS variants vs oracle (8x8, T=20, 200k boards; then T=21 > 64 KiB)" | tee -a $OUT/log.txt
timeout -k 10 300 python tools/check_variants_vs_oracle.py 8,20,10 200000 base,lds_old64,lds_old64_attr,lds_old64_cap60 2>&1 | tee $OUT/lds_variants_t20.log || exit 1
timeout -k 10 300 python tools/check_variants_vs_oracle.py 8,21,10 200000 base,lds_big_attr,lds_big_noattr 2>&1 | tee $OUT/lds_variants_t21.log || exit 1
echo "== clocks idle" | tee -a $OUT/log.txt; smi | tee -a $OUT/log.txt
for cfg in cfg4 cfg2; do
  for steps in 50 500 3000; do
    echo "== bench.py $cfg steps=$steps" | tee -a $OUT/log.txt
    timeout -k 10 300 python bench.py --config $cfg --steps $steps --warmup 20 --no-cpu-baseline > $OUT/bench_${cfg}_${steps}.json 2>>$OUT/log.txt || exit 1
    python -c "import json,sys; d=json.loads(open('$OUT/bench_${cfg}_${steps}.json').read().strip().splitlines()[-1]); print('   kernel_us', round(d['roofline']['kernel_us'],2), 'ms_per_step', round(d['ms_per_step']*1e3,2))" | tee -a $OUT/log.txt
    smi | tee -a $OUT/log.txt
  done
  echo "== variant_bench $cfg (burst: 10 rounds x 50 steps)" | tee -a $OUT/log.txt
  timeout -k 10 300 python tools/variant_bench.py run --config $cfg --rounds 10 --steps 50 --only base,old_2184215,old_974f060 --tag _burst 2>&1 | tee -a $OUT/log.txt || exit 1
  echo "== variant_bench $cfg (sustained: 4 rounds x 1500 steps)" | tee -a $OUT/log.txt
  timeout -k 10 300 python tools/variant_bench.py run --config $cfg --rounds 4 --steps 1500 --only base,old_2184215,old_974f060 --tag _sustained 2>&1 | tee -a $OUT/log.txt || exit 1
  smi | tee -a $OUT/log.txt
done
echo "== cfg2 PMC passes" | tee -a $OUT/log.txt
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_cfg2_$c -- python3 bench.py --config cfg2 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/pmc_cfg2_$c.log 2>&1 || exit 1
done
echo "== done" | tee -a $OUT/log.txt
