#!/bin/bash
# Round 5, call 21 (GPU box): four-wave blocks for k_small up to 5x5: GPU suite, default bench line, 500 MB shape sweep, scaling probe.
set -o pipefail
OUT=gpurun_out/r05_call21
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_call21/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print('cfg1', d['value'], r['frac'], r['kernel_us'], 'sibling', r['sibling_3p4x_infinity_cache']['kernel_us'], r['sibling_3p4x_infinity_cache']['frac'])
a=r['hbm_asymptote']; print('asym', a['kernel_us'], a['frac'], a['with_state_in_cache']['frac'])
print('double', d['double_buffered']['kernel_us'], d['double_buffered']['frac'], d['double_buffered']['kernel'])
for k,v in d['other_configs'].items():
    for kk,vv in v.items():
        if isinstance(vv,dict) and 'kernel_us' in vv: print(k,kk,round(vv['kernel_us'],2),round(vv['frac'],3))
print(d['cfg3_learner_side']['encode_us'], d['cfg3_learner_side']['expand_us'])
PY
timeout -k 10 600 python tools/shape_sweep.py > $OUT/shape_sweep.log 2>&1 || { tail -20 $OUT/shape_sweep.log; exit 1; }
grep -v amdgpu.ids $OUT/shape_sweep.log | head -12
MB=700,1000,1400,2100 timeout -k 10 600 python tools/scaling_probe.py > $OUT/scaling_probe.log 2>&1 || { tail -20 $OUT/scaling_probe.log; exit 1; }
grep -v amdgpu.ids $OUT/scaling_probe.log | head -8
