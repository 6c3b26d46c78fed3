#!/bin/bash
# Round 5, call 9 (GPU box): what the memory system does when the STATE no longer fits the Infinity Cache - rocprofv3 --pmc passes
# (one per counter group, kernel trace only beside them) of cfg4's shape at 262,144 / 786,432 / 1,310,720 boards and of 4x4 boards
# at 10M / 20M.
set -o pipefail
OUT=gpurun_out/r05_call09
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
GROUPS_=(
  "g1:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum"
  "g2:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
  "g3:TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum"
  "g4:FETCH_SIZE"
  "g5:WRITE_SIZE"
  "g6:TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
  "g7:SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
  "g8:TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum"
)
for shape in "l0256k:15 32 24 262144" "l0768k:15 32 24 786432" "l1280k:15 32 24 1310720" "s10m:4 2 2 10485760" "s20m:4 2 2 20971520"; do
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
python3 tools/summarize_counters.py $OUT k_lines > $OUT/counters_lines.md 2>&1 || true
python3 tools/summarize_counters.py $OUT k_small > $OUT/counters_small.md 2>&1 || true
cat $OUT/counters_lines.md $OUT/counters_small.md
du -sh $OUT
