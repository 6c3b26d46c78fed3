set -e
mkdir -p gpurun_out
CASES="5,6,3,1400;5,6,3,2100;5,2,3,1400;6,6,4,1400;6,3,4,1400;4,6,2,2100;4,2,2,1400;3,1,0,2100;3,4,0,1400;3,4,0,2100;8,8,8,2100" timeout -k 10 900 python tools/large_batch_probe.py > gpurun_out/r04_large_batch_probe3.log 2>&1 || { tail -20 gpurun_out/r04_large_batch_probe3.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_large_batch_probe3.log | cut -c1-330
