set -o pipefail
OUT=gpurun_out/r04_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
COMMON="--no-cpu-baseline --no-pipelined --no-other-configs --no-entry-points --no-learner-side"
rm -rf $OUT/stats_cfg2 $OUT/fresh_cfg2_*.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg2 -- python3 bench.py --config cfg2 $COMMON > $OUT/bench_cfg2_profiled.json 2> $OUT/stats_cfg2.err || { tail -5 $OUT/stats_cfg2.err; exit 1; }
for k in 1 2 3; do
  timeout -k 10 300 python3 bench.py --config cfg2 $COMMON > $OUT/fresh_cfg2_$k.json 2>> $OUT/fresh.err || { tail -5 $OUT/fresh.err; exit 1; }
done
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default_steps20.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
find $OUT -name "*agent_info.csv" -delete
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_prof/fresh_cfg2_*.json'))+['gpurun_out/r04_prof/bench_cfg2_profiled.json']:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['roofline']['kernel_us'],2), d['config'].get('observation_placement'))
for f in ['gpurun_out/r04_prof/bench_default.json','gpurun_out/r04_prof/bench_default_steps20.json']:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_us'])
PY
