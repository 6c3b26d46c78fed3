set -e
mkdir -p gpurun_out
SHAPES="4,2,2;4,6,2;5,2,3;3,1,0;3,4,0;7,5,6;8,8,8;6,6,4;9,4,9;32,32,100" MB=1000,1400,2100 timeout -k 10 900 python tools/large_batch_edges_probe.py > gpurun_out/r04_large_batch_edges2.log 2>&1 || { tail -20 gpurun_out/r04_large_batch_edges2.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_large_batch_edges2.log
timeout -k 10 1000 python tools/scaling_probe.py > gpurun_out/r04_scaling_probe_after.log 2>&1 || { tail -20 gpurun_out/r04_scaling_probe_after.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_scaling_probe_after.log
