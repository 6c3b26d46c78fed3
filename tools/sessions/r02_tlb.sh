#!/bin/bash
set -o pipefail
OUT=gpurun_out/r02_s14
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for shape in cfg4 6,3,4,1048576; do
  tag=$(echo $shape | tr , _)
  timeout -k 10 300 python3 tools/placement_tlb_probe.py $shape 6 2>&1 | grep -v amdgpu.ids | tee $OUT/tlb_${tag}_plain.log || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum --output-format csv -d $OUT/tlb_$tag -- python3 tools/placement_tlb_probe.py $shape 6 > $OUT/tlb_${tag}_pmc.log 2>&1 || { tail -5 $OUT/tlb_${tag}_pmc.log; exit 1; }
  grep -v amdgpu.ids $OUT/tlb_${tag}_pmc.log | tail -8
done
find $OUT -name "*agent_info.csv" -delete
