set -e
mkdir -p gpurun_out
timeout -k 10 500 python tools/piece_fine_sweep.py > gpurun_out/r04_piece_fine_sweep.log 2>&1
cat gpurun_out/r04_piece_fine_sweep.log
