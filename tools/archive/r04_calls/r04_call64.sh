set -o pipefail
mkdir -p gpurun_out/r04_pmc_cached
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r04_pmc_cached
for spec in "6 3 4 1048576 1" "6 3 4 1048576 0" "6 3 4 1048576 8" "15 32 24 262144 1 1" "15 32 24 262144 1 4" "15 32 24 518400 1 4" "15 32 24 518400 1 1"; do
  tag=$(echo $spec | tr ' ' '_')
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${tag}_$c -- python3 tools/pmc_scaling_target.py $spec > $O/${tag}_$c.log 2>&1 || { tail -5 $O/${tag}_$c.log; exit 1; }
  done
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r04_pmc_cached/*_SIZE')):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    t = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    if not f: print(d, 'no csv'); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE') and ('k_lines' in r['Kernel_Name'] or 'k_small' in r['Kernel_Name']):
            acc[r['Kernel_Name'][:70]].append(float(r['Counter_Value']))
    for k, v in acc.items():
        v = v[5:] if len(v) > 10 else v
        print(d.split('/')[-1], k[28:], f"{sum(v) / len(v) * 1024 / 1e6:.1f} MB per launch ({len(v)} launches)")
PY
find gpurun_out/r04_pmc_cached -name "*agent_info.csv" -delete
