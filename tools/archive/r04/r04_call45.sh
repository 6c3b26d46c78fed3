set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu3.log 2>&1 || { tail -30 gpurun_out/r04_pytest_gpu3.log; exit 1; }
tail -2 gpurun_out/r04_pytest_gpu3.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 900 python tools/ooc_fuzz.py > gpurun_out/r04_ooc_fuzz3.log 2>&1 || { tail -5 gpurun_out/r04_ooc_fuzz3.log; exit 1; }
grep -c " ok " gpurun_out/r04_ooc_fuzz3.log; grep -c MISMATCH gpurun_out/r04_ooc_fuzz3.log || true
timeout -k 10 600 python tools/shape_sweep.py > gpurun_out/r04_shape_sweep3.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_shape_sweep3.log
timeout -k 10 500 python bench.py > gpurun_out/r04_bench_ce.json 2> gpurun_out/r04_bench_ce.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_ce.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['ms_per_step'])
for k,v in d['other_configs'].items():
    if isinstance(v,dict):
        for kk,vv in v.items():
            if isinstance(vv,dict) and 'kernel_us' in vv: print(k,kk,round(vv['kernel_us'],2),round(vv['frac'],3),vv.get('observation_placement'))
print({k:(round(v['us'],2),round(v['frac'],3)) for k,v in d['entry_points'].items() if isinstance(v,dict) and 'frac' in v})
print(d.get('cfg3_learner_side',{}).get('encode_us'), d.get('cfg3_learner_side',{}).get('expand_us'))
PY
