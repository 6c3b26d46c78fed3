#!/bin/bash
# Round-2 GPU session 4: (a) bisect of round 1's LDS corruption on the pre-fix source (one run per
# variant, against the oracle); (b) which buffer's placement decides the step time.
set -o pipefail
OUT=gpurun_out/r02_s4
mkdir -p $OUT
timeout -k 10 400 python tools/check_variants_vs_oracle.py 8,20,10 524288 old_0690d71,o_wait,o_cap60,o_pad64,o_attr 2>&1 | grep -v amdgpu.ids | tee $OUT/lds_bisect.log || exit 1
timeout -k 10 400 python tools/placement_study2.py cfg4 10 2>&1 | grep -v amdgpu.ids | tee $OUT/placement2_cfg4.log || exit 1
timeout -k 10 400 python tools/placement_study2.py cfg2 6 2>&1 | grep -v amdgpu.ids | tee $OUT/placement2_cfg2.log || exit 1
