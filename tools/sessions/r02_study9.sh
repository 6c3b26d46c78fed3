#!/bin/bash
# Round-2 GPU session 9: resident waves per CU (waves per block x blocks per CU, forced with LDS padding)
set -o pipefail
OUT=gpurun_out/r02_s9
mkdir -p $OUT
V=base,w1b8,w2b4,w4b2,w8b1,w2b3,w2b5,w3b2,w5b2,w6b1,w2b6,w4b1
for pl in 0 1; do
  timeout -k 10 300 python tools/variant_bench.py run --config cfg4 --rounds 5 --steps 60 --only $V --placement $pl --tag _w_pl$pl 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_cfg4_waves.log || exit 1
done
for shape in s12t8 s32t64; do
  timeout -k 10 300 python tools/variant_bench.py run --config $shape --rounds 4 --steps 60 --only base,w2b4,w4b2,w2b5,w2b6 --tag _w 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_shapes_waves.log || exit 1
done
timeout -k 10 300 python tools/variant_bench.py run --config cfg1 --boards 4194304 --rounds 5 --steps 40 --only base,spad2k,spad4k,spad8k --tag _4m_s 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_small_waves.log || exit 1
timeout -k 10 300 python tools/variant_bench.py run --config cfg2 --rounds 5 --steps 40 --only base,spad2k,spad4k,spad8k --tag _s 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_small_waves.log || exit 1
timeout -k 10 300 python tools/variant_bench.py run --config cfg1 --rounds 5 --steps 100 --only base,spad2k,spad4k,spad8k --tag _s 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_small_waves.log || exit 1
timeout -k 10 300 python tools/variant_bench.py run --config s8t20 --rounds 4 --steps 60 --only base,spad2k,spad4k,spad8k --tag _s 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_small_waves.log || exit 1
