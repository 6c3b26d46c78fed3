set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu6.log 2>&1 || { tail -30 gpurun_out/r04_pytest_gpu6.log; exit 1; }
tail -2 gpurun_out/r04_pytest_gpu6.log
