set -e
mkdir -p gpurun_out
timeout -k 10 900 python tools/ooc_fuzz.py --onehot-only > gpurun_out/r04_ooc_fuzz_onehot.log 2>&1 || { tail -5 gpurun_out/r04_ooc_fuzz_onehot.log; exit 1; }
grep -c " ok " gpurun_out/r04_ooc_fuzz_onehot.log; grep -c MISMATCH gpurun_out/r04_ooc_fuzz_onehot.log || true
timeout -k 10 300 python tools/soak.py cfg2 200 > gpurun_out/r04_soak_cfg2.log 2>&1 || { tail -5 gpurun_out/r04_soak_cfg2.log; exit 1; }
tail -1 gpurun_out/r04_soak_cfg2.log
