set -e
mkdir -p gpurun_out
timeout -k 10 300 python tools/obs_candidates_robustness.py > gpurun_out/r04_obs_candidates_robustness.log 2>&1
SEED=11 timeout -k 10 300 python tools/obs_candidates_robustness.py >> gpurun_out/r04_obs_candidates_robustness.log 2>&1
cat gpurun_out/r04_obs_candidates_robustness.log
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r04_bench_spaced.json 2> gpurun_out/r04_bench_spaced.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_spaced.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'])
for k,v in d['other_configs'].items():
    if isinstance(v,dict) and 'class_default' in v: print(k, v['class_default']['kernel_us'], v['class_default']['frac'], v['class_default'].get('observation_placement'))
PY
