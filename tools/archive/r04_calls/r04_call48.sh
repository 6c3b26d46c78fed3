set -e
mkdir -p gpurun_out
timeout -k 10 900 python tools/large_batch_probe.py > gpurun_out/r04_large_batch_probe_after.log 2>&1 || { tail -20 gpurun_out/r04_large_batch_probe_after.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_large_batch_probe_after.log | cut -c1-120
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu4.log 2>&1 || { tail -30 gpurun_out/r04_pytest_gpu4.log; exit 1; }
tail -2 gpurun_out/r04_pytest_gpu4.log
