set -e
mkdir -p gpurun_out
L=gpurun_out/r04_obs_candidates_robustness2.log
: > $L
for spec in "10 0" "10 1.2" "16 1,2,4,8" "16 0.3,5" "24 0"; do
  set -- $spec
  for seed in 5 11; do
    CANDIDATES=$1 GAPS=$2 SEED=$seed timeout -k 10 300 python tools/obs_candidates_robustness.py >> $L 2>&1
  done
done
grep -c "frac 0.95\|frac 0.96" $L || true
cat $L | cut -c1-150
