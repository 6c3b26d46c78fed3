#!/bin/bash
set -e
mkdir -p gpurun_out/r04z
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "placement_trials" > gpurun_out/r04z/pytest.log 2>&1 || { tail -40 gpurun_out/r04z/pytest.log; exit 1; }
tail -2 gpurun_out/r04z/pytest.log
timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-pipelined --no-learner-side --no-entry-points > gpurun_out/r04z/bench.json 2> gpurun_out/r04z/bench.err
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r04z/bench.json').read().strip().splitlines()[-1])
for name,o in b['other_configs'].items():
    for k,v in o.items():
        if isinstance(v,dict): print(name,k,"us %.1f frac %.3f"%(v['kernel_us'],v['frac']), v.get('observation_placement'))
PY
