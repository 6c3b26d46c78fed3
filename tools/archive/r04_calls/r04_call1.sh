#!/bin/bash
# round 4, GPU call 1: new tests, contiguous-memory mapping sweep, entry-point and scramble baselines
set -e
mkdir -p gpurun_out/r04a
python -m pytest tests/test_levels_from_screenshots.py tests/test_reference_env_golden.py tests/test_gpu_two_boards_per_lane.py -m gpu -x -q > gpurun_out/r04a/pytest_new.log 2>&1 || { tail -30 gpurun_out/r04a/pytest_new.log; exit 1; }
tail -3 gpurun_out/r04a/pytest_new.log
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "random_boards_vs_oracle" > gpurun_out/r04a/pytest_random.log 2>&1 || { tail -30 gpurun_out/r04a/pytest_random.log; exit 1; }
tail -3 gpurun_out/r04a/pytest_random.log
python tools/contig_sweep.py cfg2 cfg4 4,2,2,4194304 > gpurun_out/r04a/contig_sweep.log 2>&1
tail -5 gpurun_out/r04a/contig_sweep.log
python tools/scramble_timing.py > gpurun_out/r04a/scramble.log 2>&1
cat gpurun_out/r04a/scramble.log
for c in cfg1 cfg2 cfg4; do python tools/aux_ops_timing.py $c >> gpurun_out/r04a/aux_ops.log 2>&1; done
cat gpurun_out/r04a/aux_ops.log
TS_SWEEP_MEM=torch python tools/contig_sweep.py cfg2 > gpurun_out/r04a/torch_sweep_cfg2.log 2>&1
tail -8 gpurun_out/r04a/torch_sweep_cfg2.log
