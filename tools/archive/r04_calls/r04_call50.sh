set -e
mkdir -p gpurun_out
timeout -k 10 1000 python tools/large_batch_probe2.py > gpurun_out/r04_large_batch_probe2.log 2>&1 || { tail -20 gpurun_out/r04_large_batch_probe2.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_large_batch_probe2.log
