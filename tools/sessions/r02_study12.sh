#!/bin/bash
set -o pipefail
OUT=gpurun_out/r02_s12
mkdir -p $OUT
run() { timeout -k 10 400 python tools/variant_bench.py run "$@" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fine.log || exit 1; }
run --config cfg4 --rounds 6 --steps 60 --tag _fine
run --config cfg4 --rounds 6 --steps 60 --placement 2 --tag _fine2
run --config cfg2 --rounds 5 --steps 40 --tag _fine
run --config cfg1 --boards 4194304 --rounds 5 --steps 40 --tag _fine4m
