set -e
mkdir -p gpurun_out
timeout -k 10 900 python tools/edges_vs_cached_probe.py > gpurun_out/r04_edges_vs_cached.log 2>&1 || { tail -20 gpurun_out/r04_edges_vs_cached.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_edges_vs_cached.log
