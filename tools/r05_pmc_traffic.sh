#!/bin/bash
# Round 5 (GPU box): HBM traffic per launch of the dominant kernel of every bench config, one rocprofv3 --pmc pass per counter
# (WRITE_SIZE, FETCH_SIZE; kernel trace only beside them) -> gpurun_out/r05_prof/pmc_<config>_<counter>; profiles/make_traffic_pmc.py
# turns them into profiles/traffic_pmc.json, which bench.py reports as roofline.traffic.
set -o pipefail
OUT=gpurun_out/r05_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
COMMON="--no-cpu-baseline --no-pipelined --no-other-configs --no-entry-points --no-learner-side --no-sibling --steps 30 --warmup 5"
for spec in "cfg1:--config cfg1" "cfg2:--config cfg2" "cfg4:--config cfg4" "sib4m:--config cfg1 --boards 4194304"; do
  name=${spec%%:*}; args=${spec#*:}
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${name}_$c -- python3 bench.py $args $COMMON > $OUT/pmc_${name}_$c.log 2>&1 || { tail -5 $OUT/pmc_${name}_$c.log; exit 1; }
  done
  echo "$name done"
done
find $OUT -name "*agent_info.csv" -delete
