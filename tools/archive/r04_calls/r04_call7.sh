#!/bin/bash
set -e
mkdir -p gpurun_out/r04g
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r04g/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r04g/pytest_gpu.log; exit 1; }
tail -12 gpurun_out/r04g/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04g/smoke.log 2>&1 || { tail -20 gpurun_out/r04g/smoke.log; exit 1; }
tail -2 gpurun_out/r04g/smoke.log
bash tools/r04_profile.sh > gpurun_out/r04g/profile.log 2>&1 || { tail -30 gpurun_out/r04g/profile.log; exit 1; }
tail -12 gpurun_out/r04g/profile.log
python tools/big_chunk_probe.py > gpurun_out/r04g/big_chunk_probe.log 2>&1 || true
cat gpurun_out/r04g/big_chunk_probe.log
