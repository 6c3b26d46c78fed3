set -e
mkdir -p gpurun_out
MB=800,1000,1300,1700 EVERY=12,16,24,32,64 timeout -k 10 900 python tools/just_beyond_cache_probe.py > gpurun_out/r04_just_beyond_cache2.log 2>&1 || { tail -20 gpurun_out/r04_just_beyond_cache2.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_just_beyond_cache2.log
