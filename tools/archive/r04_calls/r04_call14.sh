#!/bin/bash
set -e
mkdir -p gpurun_out/r04p
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dealt_tiles or large_batches or full_size or fuzzed" > gpurun_out/r04p/pytest.log 2>&1 || { tail -40 gpurun_out/r04p/pytest.log; exit 1; }
tail -3 gpurun_out/r04p/pytest.log
python tools/deal_ab.py > gpurun_out/r04p/deal_ab.log 2>&1
cat gpurun_out/r04p/deal_ab.log
python tools/learner_side_sweep.py > gpurun_out/r04p/learner_sweep.log 2>&1
cat gpurun_out/r04p/learner_sweep.log
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 --no-other-configs --no-sibling --no-learner-side --no-cpu-baseline > gpurun_out/r04p/bench_steps20.json 2> gpurun_out/r04p/bench.err
python -c "
import json; b=json.loads(open('gpurun_out/r04p/bench_steps20.json').read().strip().splitlines()[-1]); print('steps20: value %.4g ms %.5f kernel_us %.2f frac %.4f' % (b['value'], b['ms_per_step'], b['roofline']['kernel_us'], b['roofline']['frac']))"
