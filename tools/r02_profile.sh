#!/bin/bash
# Round-2 profiles: rocprofv3 --kernel-trace --stats of the bench command per config, then separate --pmc passes
# (WRITE_SIZE, FETCH_SIZE, SQ instruction counters), all into gpurun_out/r02_prof/ (copied to profiles/ afterwards).
set -o pipefail
OUT=gpurun_out/r02_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for cfg in cfg1 cfg2 cfg4; do
  # the unprofiled run uses bench.py's defaults (placement_trials=6: candidate output buffers and launch hints are rated
  # at construction); the profiled runs take the first allocation and the library's policy (--placement-trials 1), so
  # that rocprofv3's per-kernel average is over the launches the bench line times and nothing else
  timeout -k 10 400 python3 bench.py --config $cfg --no-cpu-baseline > $OUT/bench_${cfg}_unprofiled.json 2>/dev/null || exit 1
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -- python3 bench.py --config $cfg --no-cpu-baseline --no-pipelined --placement-trials 1 > $OUT/bench_${cfg}_profiled.json 2> $OUT/stats_$cfg.err || { tail -5 $OUT/stats_$cfg.err; exit 1; }
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${cfg}_$c -- python3 bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-sibling --no-pipelined --placement-trials 1 > $OUT/pmc_${cfg}_$c.log 2>&1 || { tail -5 $OUT/pmc_${cfg}_$c.log; exit 1; }
  done
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_${cfg}_SQ -- python3 bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-sibling --no-pipelined --placement-trials 1 > $OUT/pmc_${cfg}_SQ.log 2>&1 || { tail -5 $OUT/pmc_${cfg}_SQ.log; exit 1; }
  echo "$cfg done"
done
# keep the merge small: drop the big per-dispatch traces of the PMC passes except the counter tables
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
