set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu3.log 2>&1 || { tail -30 gpurun_out/r04_pytest_gpu3.log; exit 1; }
tail -2 gpurun_out/r04_pytest_gpu3.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 900 python tools/ooc_fuzz.py > gpurun_out/r04_ooc_fuzz3.log 2>&1 || { tail -5 gpurun_out/r04_ooc_fuzz3.log; exit 1; }
grep -c " ok " gpurun_out/r04_ooc_fuzz3.log; grep -c MISMATCH gpurun_out/r04_ooc_fuzz3.log || true
timeout -k 10 600 python tools/shape_sweep.py > gpurun_out/r04_shape_sweep3.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_shape_sweep3.log
