#!/bin/bash
# Round-2 GPU session 10: out-of-cache residency policy sweep (waves per block x blocks per CU) on every kernel
set -o pipefail
OUT=gpurun_out/r02_s10
mkdir -p $OUT
run() { timeout -k 10 400 python tools/variant_bench.py run "$@" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ooc_sweep.log || exit 1; }
run --config cfg4 --rounds 4 --steps 40 --tag _ooc
run --config cfg2 --rounds 4 --steps 40 --tag _ooc
run --config cfg1 --boards 4194304 --rounds 4 --steps 40 --tag _ooc4m
run --config cfg1 --boards 2097152 --rounds 4 --steps 40 --tag _ooc2m
run --config s12t8 --rounds 4 --steps 40 --tag _ooc
run --config s9t4 --rounds 4 --steps 40 --tag _ooc
run --config s32t64 --rounds 4 --steps 40 --tag _ooc
run --config s8t20 --rounds 4 --steps 40 --tag _ooc
run --config s9t4 --shape 16,40,30,131072 --rounds 4 --steps 40 --tag _ooc16
run --config s9t4 --shape 6,5,6,1048576 --rounds 4 --steps 40 --tag _ooc6
run --config s9t4 --shape 3,1,0,4194304 --rounds 4 --steps 40 --tag _ooc3
