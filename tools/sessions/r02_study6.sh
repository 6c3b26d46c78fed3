#!/bin/bash
# Round-2 GPU session 6: full GPU suite, default bench line, multi-rank code path over RCCL with one rank.
set -o pipefail
OUT=gpurun_out/r02_s6
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu.log || exit 1
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
tail -c 3000 $OUT/bench_default.json
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --steps 100 --no-cpu-baseline > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err || { tail -20 $OUT/bench_forcedist.err; exit 1; }
tail -c 2500 $OUT/bench_forcedist.json
python bench.py --gpus 2 --steps 5; echo "bench --gpus 2 on a 1-GPU box: rc=$?" | tee $OUT/bench_gpus2_rc.log
