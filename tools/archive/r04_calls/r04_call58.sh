set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu5.log 2>&1 || { tail -30 gpurun_out/r04_pytest_gpu5.log; exit 1; }
tail -2 gpurun_out/r04_pytest_gpu5.log
MB=700,1000,1400,2100 timeout -k 10 1000 python tools/scaling_probe.py > gpurun_out/r04_scaling_probe_after2.log 2>&1 || { tail -20 gpurun_out/r04_scaling_probe_after2.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_scaling_probe_after2.log
