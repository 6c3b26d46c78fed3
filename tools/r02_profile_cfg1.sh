set -o pipefail
OUT=gpurun_out/r02_prof1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg1 -- python3 bench.py --config cfg1 --no-cpu-baseline --no-pipelined > $OUT/bench_cfg1_profiled.json 2> $OUT/stats_cfg1.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg1_graph -- python3 bench.py --config cfg1 --no-cpu-baseline --no-pipelined --graph > $OUT/bench_cfg1_graph_profiled.json 2> $OUT/stats_cfg1_graph.err || exit 1
find $OUT -name "*agent_info.csv" -delete
python3 - <<EOF
import json
for k in ("bench_cfg1_profiled","bench_cfg1_graph_profiled"):
    d=json.load(open("$OUT/%s.json"%k)); r=d["roofline"]; print(k, r["kernel_us"], r["frac"], d["config"]["launch"])
EOF
