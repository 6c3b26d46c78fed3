#!/bin/bash
# Round 5, call 19 (GPU box): four-wave blocks for the 16-lane form of k_lines: GPU suite, the bench line of cfg4, the 500 MB shape sweep,
# and the large-batch scaling of the shapes concerned.
set -o pipefail
OUT=gpurun_out/r05_call19
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
timeout -k 10 300 python bench.py --config cfg4 --no-cpu-baseline --no-pipelined --no-other-configs --no-learner-side > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err || { tail -20 $OUT/bench_cfg4.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_call19/bench_cfg4.json').read().strip().splitlines()[-1])
r=d['roofline']
print('cfg4', r['kernel_us'], r['frac'], r['kernel'], r['bound'], 'asymptote', r['hbm_asymptote']['kernel_us'], r['hbm_asymptote']['frac'], 'in-cache', r['hbm_asymptote']['with_state_in_cache']['frac'])
print('double_buffered', d.get('double_buffered',{}).get('kernel_us'), d.get('double_buffered',{}).get('frac'))
print({k:(round(v['us'],2),round(v['frac'],3)) for k,v in d['entry_points'].items() if isinstance(v,dict) and 'frac' in v})
PY
timeout -k 10 600 python tools/shape_sweep.py > $OUT/shape_sweep.log 2>&1 || { tail -20 $OUT/shape_sweep.log; exit 1; }
grep -v amdgpu.ids $OUT/shape_sweep.log
MB=700,1000,1400,2100 timeout -k 10 600 python tools/scaling_probe.py > $OUT/scaling_probe.log 2>&1 || { tail -20 $OUT/scaling_probe.log; exit 1; }
grep -v amdgpu.ids $OUT/scaling_probe.log
