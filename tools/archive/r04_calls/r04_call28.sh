set -e
mkdir -p gpurun_out
timeout -k 10 400 python tools/two_stream_order_probe.py base pf pffence > gpurun_out/r04_two_stream_order_probe.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_two_stream_order_probe.log
