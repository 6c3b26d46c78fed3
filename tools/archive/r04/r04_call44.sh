set -e
mkdir -p gpurun_out
timeout -k 10 900 python tools/cached_every_lines_probe.py > gpurun_out/r04_cached_every_lines.log 2>&1 || { tail -20 gpurun_out/r04_cached_every_lines.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_cached_every_lines.log
