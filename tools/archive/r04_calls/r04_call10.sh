#!/bin/bash
set -e
mkdir -p gpurun_out/r04j
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q > gpurun_out/r04j/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r04j/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r04j/pytest_gpu.log
python tools/shape_sweep.py > gpurun_out/r04j/shape_sweep.log 2>&1
cat gpurun_out/r04j/shape_sweep.log
for c in cfg1 cfg2 cfg4; do
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04j/entry_$c -- python3 tools/entry_points_profile.py run $c > gpurun_out/r04j/entry_$c.log 2>&1 || { tail -5 gpurun_out/r04j/entry_$c.log; exit 1; }
done
find gpurun_out/r04j -name "*agent_info.csv" -delete
timeout -k 10 500 python3 bench.py > gpurun_out/r04j/bench_default.json 2> gpurun_out/r04j/bench_default.err
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04j/bench_default_steps20.json 2>> gpurun_out/r04j/bench_default.err
echo done
