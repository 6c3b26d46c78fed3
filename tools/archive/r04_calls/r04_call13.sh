#!/bin/bash
set -e
mkdir -p gpurun_out/r04o
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python tools/learner_side_sweep.py > gpurun_out/r04o/learner_sweep.log 2>&1
cat gpurun_out/r04o/learner_sweep.log
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04o/bench_default_steps20.json 2> gpurun_out/r04o/bench.err
python -c "
import json; b=json.loads(open('gpurun_out/r04o/bench_default_steps20.json').read().strip().splitlines()[-1]); print('steps20: value %.4g ms %.5f kernel_us %.2f frac %.4f warm %s' % (b['value'], b['ms_per_step'], b['roofline']['kernel_us'], b['roofline']['frac'], b['config']['clock_warmup']))"
for c in cfg1 cfg2 cfg4; do
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04o/entry_$c -- python3 tools/entry_points_profile.py run $c > gpurun_out/r04o/entry_$c.log 2>&1 || { tail -5 gpurun_out/r04o/entry_$c.log; exit 1; }
done
find gpurun_out/r04o -name "*agent_info.csv" -delete
timeout 900 python tools/ooc_fuzz.py > gpurun_out/r04o/ooc_fuzz.log 2>&1 || { tail -20 gpurun_out/r04o/ooc_fuzz.log; exit 1; }
tail -3 gpurun_out/r04o/ooc_fuzz.log
