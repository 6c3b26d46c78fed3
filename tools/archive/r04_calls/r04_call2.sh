#!/bin/bash
# round 4, GPU call 2: k_deal / streamed MT19937 / valid4 parity, piece policy by shape on contiguous memory, k_deal speed
set -e
mkdir -p gpurun_out/r04b
python - <<'PY' > gpurun_out/r04b/box.log 2>&1
import os
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("cpu.max", e)
PY
cat gpurun_out/r04b/box.log
python -m pytest tests/test_gpu_parity.py tests/test_mt19937_levels.py -m gpu -x -q --durations=12 -k "dealt_tiles or mt19937 or streamed or hip_generator or main_entry or large_batches or launch_hint or placement" > gpurun_out/r04b/pytest_new.log 2>&1 || { tail -40 gpurun_out/r04b/pytest_new.log; exit 1; }
tail -18 gpurun_out/r04b/pytest_new.log
python tools/scramble_timing.py > gpurun_out/r04b/scramble.log 2>&1
cat gpurun_out/r04b/scramble.log
python tools/deal_ab.py > gpurun_out/r04b/deal_ab.log 2>&1
cat gpurun_out/r04b/deal_ab.log
python tools/piece_by_shape.py > gpurun_out/r04b/piece_by_shape.log 2>&1
tail -40 gpurun_out/r04b/piece_by_shape.log
