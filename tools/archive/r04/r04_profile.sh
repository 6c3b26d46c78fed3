#!/bin/bash
# Round-4 profiles (GPU box): the default bench run, then rocprofv3 --kernel-trace --stats per config with the class defaults
# (physically contiguous output buffers, static launch policy: every launch of the dominant kernel in the process is one the
# bench line times - except cfg2's, whose constructor rates up to 16 candidate observation buffers with 11 launches each first),
# separate --pmc passes (WRITE_SIZE, FETCH_SIZE), cfg2 with the first allocation and on torch's allocator in fresh processes (the
# two speeds), three fresh processes of cfg2 / cfg4 (reproducibility), and the entry-point kernel traces.
set -o pipefail
OUT=gpurun_out/r04_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
COMMON="--no-cpu-baseline --no-pipelined --no-other-configs --no-entry-points --no-learner-side"
timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
echo "default bench done"
for spec in "cfg1:--config cfg1 --no-sibling" "cfg2:--config cfg2" "cfg4:--config cfg4" "sib4m:--config cfg1 --boards 4194304"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 bench.py $args $COMMON > $OUT/bench_${name}_profiled.json 2> $OUT/stats_$name.err || { tail -5 $OUT/stats_$name.err; exit 1; }
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${name}_$c -- python3 bench.py $args $COMMON --steps 30 --warmup 5 > $OUT/pmc_${name}_$c.log 2>&1 || { tail -5 $OUT/pmc_${name}_$c.log; exit 1; }
  done
  echo "$name done"
done
# cfg2 with the first observation buffer the runtime hands out (obs_candidates=0): the slow class in a fresh process
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg2_first -- python3 bench.py --config cfg2 --obs-candidates 0 $COMMON > $OUT/bench_cfg2_first_profiled.json 2> $OUT/stats_cfg2_first.err || { tail -5 $OUT/stats_cfg2_first.err; exit 1; }
# cfg2 on torch's caching allocator, four fresh processes under the profiler: whichever speeds they land on
for k in 1 2 3 4; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg2_torch$k -- python3 bench.py --config cfg2 --output-memory torch --obs-candidates 0 $COMMON > $OUT/bench_cfg2_torch${k}_profiled.json 2> $OUT/stats_cfg2_torch$k.err || { tail -5 $OUT/stats_cfg2_torch$k.err; exit 1; }
done
echo "torch allocator done"
# reproducibility: three fresh, unprofiled processes each
for k in 1 2 3; do
  for c in cfg2 cfg4; do
    timeout -k 10 300 python3 bench.py --config $c $COMMON > $OUT/fresh_${c}_$k.json 2>> $OUT/fresh.err || { tail -5 $OUT/fresh.err; exit 1; }
  done
done
echo "fresh processes done"
for c in cfg1 cfg2 cfg4; do
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/entry_$c -- python3 tools/entry_points_profile.py run $c > $OUT/entry_$c.log 2>&1 || { tail -5 $OUT/entry_$c.log; exit 1; }
done
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
