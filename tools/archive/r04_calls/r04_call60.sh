set -e
mkdir -p gpurun_out
timeout -k 10 600 python tools/nt_every_probe.py > gpurun_out/r04_nt_every_probe.log 2>&1 || { tail -20 gpurun_out/r04_nt_every_probe.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_nt_every_probe.log
