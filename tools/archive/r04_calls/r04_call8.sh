#!/bin/bash
set -e
mkdir -p gpurun_out/r04h
python tools/bpw_sweep.py > gpurun_out/r04h/bpw_sweep.log 2>&1
cat gpurun_out/r04h/bpw_sweep.log
python tools/lanes_sweep.py > gpurun_out/r04h/lanes_sweep.log 2>&1
cat gpurun_out/r04h/lanes_sweep.log
