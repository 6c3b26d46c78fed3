#!/bin/bash
set -e
mkdir -p gpurun_out/r04n
rm -rf gpurun_out/r04_prof
bash tools/r04_profile.sh > gpurun_out/r04n/profile.log 2>&1 || { tail -30 gpurun_out/r04n/profile.log; exit 1; }
tail -8 gpurun_out/r04n/profile.log
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04_prof/bench_default_steps20.json 2>> gpurun_out/r04_prof/bench_default.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs --no-sibling --no-pipelined --no-learner-side > gpurun_out/r04_prof/bench_forced_dist.json 2> gpurun_out/r04_prof/bench_forced_dist.err || { tail -20 gpurun_out/r04_prof/bench_forced_dist.err; exit 1; }
tail -c 600 gpurun_out/r04_prof/bench_forced_dist.json
