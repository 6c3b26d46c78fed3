#!/bin/bash
# Round 5, call 2 (GPU box): the GPU suite of the ABI v6 build, smoke, the default bench line, the one-rank RCCL rehearsal of the
# hand-off loop (serial vs overlapped, all-gather vs gather to root), the stand-alone entry points at cfg4 (k_state).
set -o pipefail
OUT=gpurun_out/r05_call02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1 || exit 1
timeout -k 10 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_call02/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print('cfg1', d['value'], r['frac'], d['ms_per_step'], r['kernel'], r['bound'])
print('asymptote', {k:r['hbm_asymptote'][k] for k in ('boards','kernel','kernel_us','frac')})
print('double_buffered', d.get('double_buffered'))
print('actor', d.get('actor_without_observation'))
for k,v in d['other_configs'].items():
    for kk,vv in v.items():
        if isinstance(vv,dict) and 'kernel_us' in vv: print(k,kk,round(vv['kernel_us'],2),round(vv['frac'],3),vv.get('observation_placement'))
print({k:(round(v['us'],2),round(v['frac'],3)) for k,v in d['entry_points'].items() if isinstance(v,dict) and 'frac' in v})
print(d.get('cfg3_learner_side',{}).get('encode_us'), d.get('cfg3_learner_side',{}).get('expand_us'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --steps 50 --warmup 10 --no-cpu-baseline --no-sibling --no-other-configs --no-entry-points --no-learner-side --no-pipelined > $OUT/bench_forced_dist_1rank_rccl.json 2> $OUT/bench_forced_dist.err || { tail -20 $OUT/bench_forced_dist.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_call02/bench_forced_dist_1rank_rccl.json').read().strip().splitlines()[-1])
print('rccl_ranks', d.get('rccl_ranks'))
for k,v in d['allgather'].items():
    print(k, 'serial', round(v['ms_per_step_serial'],4), 'overlapped', round(v['ms_per_step_overlapped'],4), v['bytes_per_rank_per_step'], v['actor_step'])
PY
timeout -k 10 300 python bench.py --config cfg4 --no-cpu-baseline --no-pipelined --no-other-configs --no-learner-side > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err || { tail -20 $OUT/bench_cfg4.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_call02/bench_cfg4.json').read().strip().splitlines()[-1])
r=d['roofline']
print('cfg4', r['kernel_us'], r['frac'], r['kernel'], r['bound'], 'asymptote', r['hbm_asymptote']['kernel_us'], r['hbm_asymptote']['frac'])
print('double_buffered', d.get('double_buffered')); print('actor', d.get('actor_without_observation'))
print({k:(round(v['us'],2),round(v['frac'],3)) for k,v in d['entry_points'].items() if isinstance(v,dict) and 'frac' in v})
PY
# cfg4 experiment (VERDICT next 5): multi-colour launches of k_lines touching 64 instead of 128 bytes of the per-level record
timeout -k 10 300 python tools/variant_bench.py run --config cfg4 --rounds 12 --steps 100 --tag _r05_record_half > $OUT/cfg4_record_half_ab.log 2>&1 || { tail -5 $OUT/cfg4_record_half_ab.log; exit 1; }
grep -v amdgpu.ids $OUT/cfg4_record_half_ab.log
