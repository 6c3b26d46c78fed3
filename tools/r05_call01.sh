#!/bin/bash
# Round 5, call 1 (GPU box): (a) the observation-ring probe (VERDICT next 1c), (b) instruction / stall counters of cfg4's kernel
# k_lines<false,16,2,true,false> next to the 8-tile variant of the same board (VERDICT next 5), one rocprofv3 --pmc pass per
# counter group (kernel trace only beside the counters).  A pass that fails because the box does not know a counter is skipped;
# a pass that is killed at its limit ends the call.
set -o pipefail
OUT=gpurun_out/r05_call01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 400 python3 tools/ring_probe.py > $OUT/ring_probe.log 2>&1 || { tail -20 $OUT/ring_probe.log; exit 1; }
grep -v amdgpu.ids $OUT/ring_probe.log
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
GROUPS_=(
  "g1:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
  "g2:SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
  "g3:SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS"
  "g4:TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
  "g5:TCC_EA0_WRREQ_STALL_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
  "g6:GRBM_GUI_ACTIVE GRBM_COUNT"
)
for shape in "t32:15 32 24 262144" "t8:15 8 24 262144"; do
  sname=${shape%%:*}; sargs=${shape#*:}
  for g in "${GROUPS_[@]}"; do
    gname=${g%%:*}; ctrs=${g#*:}
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_${sname}_$gname -- python3 tools/pmc_scaling_target.py $sargs > $OUT/pmc_${sname}_$gname.log 2>&1
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $sname $gname killed at its limit"; exit 1; fi
    if [ $rc -ne 0 ]; then echo "pass $sname $gname failed ($rc), skipped"; tail -3 $OUT/pmc_${sname}_$gname.log; fi
  done
  echo "$sname done"
done
find $OUT -name "*agent_info.csv" -delete
python3 tools/summarize_counters.py $OUT > $OUT/counters_summary.md 2>&1 || true
cat $OUT/counters_summary.md
du -sh $OUT
