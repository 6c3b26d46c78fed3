#!/bin/bash
set -e
mkdir -p gpurun_out/r04ag
for k in 1 2 3; do
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 --no-other-configs --no-sibling --no-learner-side --no-cpu-baseline --no-pipelined --no-entry-points > gpurun_out/r04ag/bench_steps20_$k.json 2> gpurun_out/r04ag/bench.err
python -c "
import json; b=json.loads(open('gpurun_out/r04ag/bench_steps20_$k.json').read().strip().splitlines()[-1]); print('steps20: value %.4g ms %.5f kernel_us %.2f frac %.4f frac_wall %.4f' % (b['value'], b['ms_per_step'], b['roofline']['kernel_us'], b['roofline']['frac'], b['roofline']['frac_wall']))"
done
timeout -k 10 500 python3 bench.py > gpurun_out/r04ag/bench_default.json 2>> gpurun_out/r04ag/bench.err
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04ag/bench_default_steps20.json 2>> gpurun_out/r04ag/bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs --no-sibling --no-pipelined --no-learner-side > gpurun_out/r04ag/bench_forced_dist.json 2> gpurun_out/r04ag/bench_forced_dist.err || { tail -20 gpurun_out/r04ag/bench_forced_dist.err; exit 1; }
python -c "
import json
for f in ('bench_default','bench_default_steps20','bench_forced_dist'):
    b=json.loads(open('gpurun_out/r04ag/'+f+'.json').read().strip().splitlines()[-1]); print(f, 'value %.4g ms %.5f kernel_us %.2f frac %.4f' % (b['value'], b['ms_per_step'], b['roofline']['kernel_us'], b['roofline']['frac']), b.get('rccl_ranks'))"
