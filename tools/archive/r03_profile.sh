#!/bin/bash
# Round-3 profiles (GPU box): rocprofv3 --kernel-trace --stats of a bench command per config, then separate --pmc passes
# (WRITE_SIZE, FETCH_SIZE), into gpurun_out/r03_prof/ (profiles/summarize.py condenses them into profiles/r03_*).
# The profiled commands use --placement-trials 0 (the library's static launch policy: every launch of the dominant kernel in
# the process is one the bench line times - the class default rates ~100 launches of other policies at construction, which
# would pollute rocprofv3's per-kernel average); the unprofiled default run beside them shows the rated / tuned figures.
set -o pipefail
OUT=gpurun_out/r03_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
COMMON="--no-cpu-baseline --no-pipelined --no-other-configs --placement-trials 0"
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
for spec in "cfg1:--config cfg1 --no-sibling" "cfg2:--config cfg2" "cfg4:--config cfg4" "sib4m:--config cfg1 --boards 4194304"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 bench.py $args $COMMON > $OUT/bench_${name}_profiled.json 2> $OUT/stats_$name.err || { tail -5 $OUT/stats_$name.err; exit 1; }
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${name}_$c -- python3 bench.py $args $COMMON --steps 30 --warmup 5 > $OUT/pmc_${name}_$c.log 2>&1 || { tail -5 $OUT/pmc_${name}_$c.log; exit 1; }
  done
  echo "$name done"
done
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
