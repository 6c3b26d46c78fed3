#!/bin/bash
# Round 5, call 29 (GPU box): edge stores into an observation ring: GPU suite, ring probe with the shipped rule, cfg4 / cfg1 bench lines.
set -o pipefail
OUT=gpurun_out/r05_call29
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
timeout -k 10 800 python tools/ring_policy_probe.py > $OUT/ring_policy_probe.log 2>&1 || { tail -20 $OUT/ring_policy_probe.log; exit 1; }
grep -v amdgpu.ids $OUT/ring_policy_probe.log | cut -c1-200
timeout -k 10 300 python bench.py --config cfg4 --no-cpu-baseline --no-pipelined --no-other-configs --no-learner-side --no-entry-points > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err || { tail -20 $OUT/bench_cfg4.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_call29/bench_cfg4.json').read().strip().splitlines()[-1])
print('cfg4', d['roofline']['kernel_us'], d['roofline']['frac'], 'double_buffered', d['double_buffered']['kernel_us'], d['double_buffered']['frac'])
PY
