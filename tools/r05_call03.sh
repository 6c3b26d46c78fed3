#!/bin/bash
# Round 5, call 3 (GPU box): k_state builds A/B (old / batch 8 / 16 / 32), the hand-off with one pack + one unpack launch
# (GPU suite, host-time profile of the hand-off loop, the one-rank RCCL rehearsal of bench.py).
set -o pipefail
OUT=gpurun_out/r05_call03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 400 python tools/state_only_ab.py > $OUT/state_only_ab.log 2>&1 || { tail -30 $OUT/state_only_ab.log; exit 1; }
grep -v amdgpu.ids $OUT/state_only_ab.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
timeout -k 10 300 python tools/handoff_host_profile.py > $OUT/handoff_host_profile.log 2>&1 || { tail -30 $OUT/handoff_host_profile.log; exit 1; }
grep -v amdgpu.ids $OUT/handoff_host_profile.log
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --steps 50 --warmup 10 --no-cpu-baseline --no-sibling --no-other-configs --no-entry-points --no-learner-side --no-pipelined > $OUT/bench_forced_dist_1rank_rccl.json 2> $OUT/bench_forced_dist.err || { tail -20 $OUT/bench_forced_dist.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_call03/bench_forced_dist_1rank_rccl.json').read().strip().splitlines()[-1])
print('rccl_ranks', d.get('rccl_ranks'))
for k,v in d['allgather'].items():
    print(k, 'serial', round(v['ms_per_step_serial'],4), 'overlapped', round(v['ms_per_step_overlapped'],4), v['bytes_per_rank_per_step'], v['actor_step'])
PY
