set -o pipefail
OUT=gpurun_out/r04_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for c in cfg1 cfg2 cfg4; do
  rm -rf $OUT/entry_$c
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/entry_$c -- python3 tools/entry_points_profile.py run $c > $OUT/entry_$c.log 2>&1 || { tail -5 $OUT/entry_$c.log; exit 1; }
done
find $OUT -name "*agent_info.csv" -delete
echo done
