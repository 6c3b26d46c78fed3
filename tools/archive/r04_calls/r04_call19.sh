#!/bin/bash
set -e
OUT=gpurun_out/r04_prof2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
COMMON="--no-cpu-baseline --no-pipelined --no-other-configs --no-entry-points --no-learner-side"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg2_fast -- python3 bench.py --config cfg2 --obs-candidates 16 $COMMON > $OUT/bench_cfg2_fast_profiled.json 2> $OUT/stats_cfg2_fast.err || { tail -5 $OUT/stats_cfg2_fast.err; exit 1; }
for k in 1 2 3; do
  timeout -k 10 300 python3 bench.py --config cfg2 --obs-candidates 16 $COMMON > $OUT/fresh_cfg2_cand_$k.json 2>> $OUT/fresh.err
done
find $OUT -name "*agent_info.csv" -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_prof2/*.json')):
    x=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], "kernel_us %.2f frac %.4f"%(x['roofline']['kernel_us'],x['roofline']['frac']), x['config']['observation_placement'])
PY
python tools/period_probe.py 40 1024 > $OUT/period_probe_again.log 2>&1; cat $OUT/period_probe_again.log
