#!/bin/bash
# Round 5, call 17 (GPU box): cfg4 with 2- and 4-wave blocks beyond the cache: resident blocks (launch_hint) x block mapping.
set -o pipefail
OUT=gpurun_out/r05_call17
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for piece in 4 6 8 12 16; do
  for hint in -2 0 1 2 4; do
    timeout -k 10 200 python tools/variant_bench.py run --config cfg4 --rounds 4 --steps 100 --piece $piece --hint $hint --only base,w2b11,w4b5 --tag _r05_waves2 > $OUT/tmp.log 2>&1 || { tail -5 $OUT/tmp.log; exit 1; }
    echo "piece $piece hint $hint: $(grep median $OUT/tmp.log | awk '{printf "%s %s   ", $1, $3}')" | tee -a $OUT/waves_hint_piece.log
  done
done
