set -e
mkdir -p gpurun_out
timeout -k 10 600 python tools/cached_every_cfg2.py > gpurun_out/r04_cached_every_cfg2.log 2>&1 || { tail -20 gpurun_out/r04_cached_every_cfg2.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_cached_every_cfg2.log
