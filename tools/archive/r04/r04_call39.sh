set -e
mkdir -p gpurun_out
timeout -k 10 900 python tools/just_beyond_cache_probe.py > gpurun_out/r04_just_beyond_cache.log 2>&1 || { tail -20 gpurun_out/r04_just_beyond_cache.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_just_beyond_cache.log
