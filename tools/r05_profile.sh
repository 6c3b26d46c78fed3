#!/bin/bash
# Round-5 profiles (GPU box): the default bench run, then rocprofv3 --kernel-trace --stats per BASELINE config with the class
# defaults (same command as the bench line), the entry-point kernel trace at cfg4 (k_state) and cfg1, and the one-rank RCCL
# rehearsal of the hand-off.  PMC traffic passes: tools/r05_pmc_traffic.sh.
set -o pipefail
OUT=gpurun_out/r05_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
COMMON="--no-cpu-baseline --no-pipelined --no-other-configs --no-entry-points --no-learner-side"
timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
echo "default bench done"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench_driver_like.err || { tail -5 $OUT/bench_driver_like.err; exit 1; }
echo "driver-like bench done"
for spec in "cfg1:--config cfg1 --no-sibling" "cfg2:--config cfg2 --no-sibling" "cfg4:--config cfg4 --no-sibling"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 bench.py $args $COMMON > $OUT/bench_${name}_profiled.json 2> $OUT/stats_$name.err || { tail -5 $OUT/stats_$name.err; exit 1; }
  echo "$name done"
done
for c in cfg1 cfg4; do
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/entry_$c -- python3 tools/entry_points_profile.py run $c > $OUT/entry_$c.log 2>&1 || { tail -5 $OUT/entry_$c.log; exit 1; }
done
echo "entry points done"
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --steps 50 --warmup 10 --no-cpu-baseline --no-sibling --no-other-configs --no-entry-points --no-learner-side --no-pipelined > $OUT/bench_forced_dist_1rank_rccl.json 2> $OUT/bench_forced_dist.err || { tail -20 $OUT/bench_forced_dist.err; exit 1; }
timeout -k 10 300 python3 tools/handoff_host_profile.py > $OUT/handoff_host_profile.log 2>&1 || { tail -30 $OUT/handoff_host_profile.log; exit 1; }
find $OUT -name "*agent_info.csv" -delete
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
