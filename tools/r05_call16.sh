#!/bin/bash
# Round 5, call 16 (GPU box): cfg4 with 2- and 4-wave blocks beyond the cache (the waves of a block share a CU's L1: the narrow state
# reads of four consecutive groups of boards), block mapping scaled accordingly.
set -o pipefail
OUT=gpurun_out/r05_call16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for piece in 0 2 4 8; do
  echo "== xcd_piece $piece"
  timeout -k 10 300 python tools/variant_bench.py run --config cfg4 --rounds 6 --steps 100 --piece $piece --tag _r05_waves_p$piece > $OUT/waves_p$piece.log 2>&1 || { tail -5 $OUT/waves_p$piece.log; exit 1; }
  grep -v amdgpu.ids $OUT/waves_p$piece.log
done
