#!/bin/bash
set -e
mkdir -p gpurun_out/r04t
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q > gpurun_out/r04t/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r04t/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r04t/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04t/smoke.log 2>&1 || { tail -20 gpurun_out/r04t/smoke.log; exit 1; }
tail -1 gpurun_out/r04t/smoke.log
timeout -k 10 500 python3 bench.py > gpurun_out/r04t/bench_default.json 2> gpurun_out/r04t/bench_default.err
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04t/bench_default_steps20.json 2>> gpurun_out/r04t/bench_default.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs --no-sibling --no-pipelined --no-learner-side > gpurun_out/r04t/bench_forced_dist.json 2> gpurun_out/r04t/bench_forced_dist.err || { tail -20 gpurun_out/r04t/bench_forced_dist.err; exit 1; }
python tools/shape_sweep.py > gpurun_out/r04t/shape_sweep.log 2>&1
cat gpurun_out/r04t/shape_sweep.log
