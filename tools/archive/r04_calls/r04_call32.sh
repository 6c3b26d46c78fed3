set -e
mkdir -p gpurun_out
L=gpurun_out/r04_hybrid_stores.log
: > $L
echo "== cfg2" >> $L
timeout -k 10 300 python tools/variant_bench.py run --config cfg2 --rounds 6 --steps 100 >> $L 2>&1
echo "== cfg4" >> $L
timeout -k 10 300 python tools/variant_bench.py run --config cfg4 --rounds 6 --steps 100 >> $L 2>&1
for shape in 8,4,8,650000 6,3,4,1048576 4,2,2,4194304 12,8,16,400000; do
  echo "== $shape" >> $L
  timeout -k 10 300 python tools/variant_bench.py run --config cfg1 --shape $shape --rounds 6 --steps 60 >> $L 2>&1
done
grep -v "amdgpu.ids\|rounds x" $L
