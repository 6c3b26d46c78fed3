#!/bin/bash
# Round-2 GPU session 3: k_lines (table-driven large-board kernel) — parity first, then A/B against
# round 1's k_large (TS_USE_LINES=0) in one process, with the observation stores ablated, and with
# extra LDS per wave (fewer resident waves).
set -o pipefail
OUT=gpurun_out/r02_s3
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu.log || exit 1
V=base,nolines,pad2k,pad4k,pad8k
timeout -k 10 300 python tools/variant_bench.py run --config cfg4 --rounds 8 --steps 100 --only $V --tag _lines 2>&1 | tee $OUT/ab_cfg4.log || exit 1
timeout -k 10 300 python tools/variant_bench.py run --config cfg4 --rounds 6 --steps 100 --only ablate,ablate_nolines --no-check --tag _ablate 2>&1 | tee -a $OUT/ab_cfg4.log || exit 1
for shape in s9t4 s12t8 s32t64; do
  timeout -k 10 300 python tools/variant_bench.py run --config $shape --rounds 6 --steps 100 --only base,nolines --tag _lines 2>&1 | tee -a $OUT/ab_shapes.log || exit 1
done
for shape in "10,10,8,262144" "14,20,20,262144" "16,40,30,131072" "20,6,30,131072" "24,30,60,65536"; do
  timeout -k 10 300 python tools/variant_bench.py run --config s9t4 --shape $shape --rounds 5 --steps 60 --only base,nolines --tag _$shape 2>&1 | tee -a $OUT/ab_shapes.log || exit 1
done
