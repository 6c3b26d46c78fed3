set -e
mkdir -p gpurun_out
HINTS=0,-2,2 timeout -k 10 600 python tools/lines_chunks_ab.py > gpurun_out/r04_lines_chunks_ab.log 2>&1 || { tail -20 gpurun_out/r04_lines_chunks_ab.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r04_lines_chunks_ab.log
