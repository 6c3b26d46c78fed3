set -e
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_final_steps20.json 2>> gpurun_out/r04_bench_final.err
python - <<'PY'
import json
for f in ('gpurun_out/r04_bench_final.json','gpurun_out/r04_bench_final_steps20.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['roofline']['frac'], d['ms_per_step'], d['roofline']['kernel_us'], d['cpu_baseline']['value'])
PY
