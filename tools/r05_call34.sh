#!/bin/bash
# Round 5, call 34 (GPU box): the hand-off with its collectives issued by a helper thread (an experiment that was measured and dropped before it was committed: this script is kept as the record of what ran; bench.py no longer prints the issue-thread figures it reads).
set -o pipefail
OUT=gpurun_out/r05_call34
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 400 python -m pytest tests/test_gpu_distributed.py -x -q > $OUT/pytest_gpu_dist.log 2>&1 || { tail -40 $OUT/pytest_gpu_dist.log; exit 1; }
tail -2 $OUT/pytest_gpu_dist.log
timeout -k 10 300 python tools/handoff_host_profile.py > $OUT/handoff_host_profile.log 2>&1 || { tail -30 $OUT/handoff_host_profile.log; exit 1; }
grep -v amdgpu.ids $OUT/handoff_host_profile.log | grep "wall" | cut -c1-140
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --steps 50 --warmup 10 --no-cpu-baseline --no-sibling --no-other-configs --no-entry-points --no-learner-side --no-pipelined > $OUT/bench_forced_dist_1rank_rccl.json 2> $OUT/bench_forced_dist.err || { tail -20 $OUT/bench_forced_dist.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_call34/bench_forced_dist_1rank_rccl.json').read().strip().splitlines()[-1])
for k,v in d['allgather'].items():
    print(k, 'serial', round(v['ms_per_step_serial'],4), 'overlapped', round(v['ms_per_step_overlapped'],4), 'overlapped + issue thread', round(v['ms_per_step_overlapped_issue_thread'],4))
PY
