set -e
mkdir -p gpurun_out
timeout -k 10 900 python tools/large_batch_edges_probe.py > gpurun_out/r04_large_batch_edges.log 2>&1 || { tail -20 gpurun_out/r04_large_batch_edges.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_large_batch_edges.log
