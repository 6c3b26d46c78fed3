set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu7.log 2>&1 || { tail -30 gpurun_out/r04_pytest_gpu7.log; exit 1; }
tail -2 gpurun_out/r04_pytest_gpu7.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_last.json 2> gpurun_out/r04_bench_last.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_last.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['ms_per_step'])
for k,v in d['other_configs'].items():
    if isinstance(v,dict):
        for kk,vv in v.items():
            if isinstance(vv,dict) and 'kernel_us' in vv: print(k,kk,round(vv['kernel_us'],2),round(vv['frac'],3))
PY
