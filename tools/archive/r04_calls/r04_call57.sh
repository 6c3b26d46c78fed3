set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu5.log 2>&1 || { tail -30 gpurun_out/r04_pytest_gpu5.log; exit 1; }
tail -2 gpurun_out/r04_pytest_gpu5.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1
MB=1000,1400,2100 timeout -k 10 1000 python tools/scaling_probe.py > gpurun_out/r04_scaling_probe_after2.log 2>&1 || { tail -20 gpurun_out/r04_scaling_probe_after2.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_scaling_probe_after2.log
timeout -k 10 600 python tools/shape_sweep.py > gpurun_out/r04_shape_sweep4.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_shape_sweep4.log | awk '{print $1, $2, $4, $6, $9}' | tr '\n' ';'
echo
timeout -k 10 500 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['ms_per_step'], d['roofline']['hbm_sibling']['kernel_us'])
for k,v in d['other_configs'].items():
    if isinstance(v,dict):
        for kk,vv in v.items():
            if isinstance(vv,dict) and 'kernel_us' in vv: print(k,kk,round(vv['kernel_us'],2),round(vv['frac'],3))
print(d.get('cfg3_learner_side',{}).get('encode_us'), d.get('cfg3_learner_side',{}).get('expand_us'))
PY
