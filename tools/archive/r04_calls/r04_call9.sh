#!/bin/bash
set -e
mkdir -p gpurun_out/r04i
python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py -m gpu -x -q -k "lanes_per_board or expand or random_boards or distributed or gathers or full_size" > gpurun_out/r04i/pytest.log 2>&1 || { tail -40 gpurun_out/r04i/pytest.log; exit 1; }
tail -3 gpurun_out/r04i/pytest.log
python tools/bpw_sweep.py > gpurun_out/r04i/bpw_sweep.log 2>&1
cat gpurun_out/r04i/bpw_sweep.log
python tools/lanes_sweep.py > gpurun_out/r04i/lanes_sweep.log 2>&1
cat gpurun_out/r04i/lanes_sweep.log
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-pipelined --no-other-configs --no-sibling > gpurun_out/r04i/bench_learner.json 2> gpurun_out/r04i/bench_learner.err
python -c "
import json; b=json.loads(open('gpurun_out/r04i/bench_learner.json').read().strip().splitlines()[-1]); print(b['cfg3_learner_side']); print({k:(v['us'] if isinstance(v,dict) else v) for k,v in b['entry_points'].items()})"
