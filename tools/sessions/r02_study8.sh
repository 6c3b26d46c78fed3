#!/bin/bash
# Round-2 GPU session 8: XCD piece size and resident-wave sweeps of k_lines at cfg4 on several placements,
# the same piece sweep for the 4x4 kernel beyond the Infinity Cache.
set -o pipefail
OUT=gpurun_out/r02_s8
mkdir -p $OUT
V=base,nolines,noremap,p8,p32,p128,p512,pad8k,pad16k,pad32k
for pl in 0 1 2 3; do
  timeout -k 10 300 python tools/variant_bench.py run --config cfg4 --rounds 5 --steps 60 --only $V --placement $pl --tag _pl$pl 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_cfg4_pieces.log || exit 1
done
for pl in 0 1; do
  timeout -k 10 300 python tools/variant_bench.py run --config cfg1 --boards 4194304 --rounds 5 --steps 40 --only base,noremap,p8,p32,p128,p512 --placement $pl --tag _4m_pl$pl 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_cfg1_4m_pieces.log || exit 1
  timeout -k 10 300 python tools/variant_bench.py run --config cfg2 --rounds 5 --steps 40 --only base,noremap,p8,p32,p128,p512 --placement $pl --tag _pl$pl 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_cfg2_pieces.log || exit 1
done
