#!/bin/bash
set -e
mkdir -p gpurun_out/r04s
: > gpurun_out/r04s/soak.log
for spec in "cfg1 300" "cfg1 200 1048576 plain" "cfg2 120" "cfg4 120" "cfg4 80 262144 plain" "s8t20 150" "s8t20 100 524288 plain" "s7t5 120" "s7t5 80 1048576 plain" "s8t4 120 524288 plain" "s12t8 100" "s9t4 100" "s28t8 60" "s28t8 60 65536 plain" "s32t64 60"; do
  timeout -k 10 600 python tools/soak.py $spec > gpurun_out/r04s/one.log 2>&1 || { cat gpurun_out/r04s/one.log | tail -20; exit 1; }
  tail -1 gpurun_out/r04s/one.log | tee -a gpurun_out/r04s/soak.log
done
