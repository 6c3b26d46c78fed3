#!/bin/bash
# Round 5, call 25 (GPU box): differential runs of the new launch forms at full size against the oracle: every kernel family just
# beyond the cache (tools/ooc_fuzz.py), 300-step soaks of cfg4 (four-wave blocks of k_lines), of the 4M-board sibling of cfg1
# (four-wave blocks of k_small) and of cfg1, plain and with the optional outputs.
set -o pipefail
OUT=gpurun_out/r05_call25
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 700 python tools/ooc_fuzz.py > $OUT/ooc_fuzz.log 2>&1 || { tail -20 $OUT/ooc_fuzz.log; exit 1; }
tail -3 $OUT/ooc_fuzz.log
for spec in "cfg4 300" "cfg4 300 262144 plain" "cfg1 300 4194304 plain" "cfg1 300" "cfg2 200"; do
  timeout -k 10 600 python tools/soak.py $spec > $OUT/soak_$(echo $spec | tr ' ' '_').log 2>&1 || { tail -10 $OUT/soak_$(echo $spec | tr ' ' '_').log; exit 1; }
  echo "soak $spec: $(tail -1 $OUT/soak_$(echo $spec | tr ' ' '_').log)"
done
