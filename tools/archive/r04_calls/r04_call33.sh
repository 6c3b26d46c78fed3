set -e
mkdir -p gpurun_out
L=gpurun_out/r04_hybrid_stores2.log
: > $L
for shape in 6,3,4,1048576 6,3,4,2097152 6,6,4,1048576 8,4,8,650000 5,2,3,2097152 7,5,6,850000; do
  echo "== $shape" >> $L
  timeout -k 10 300 python tools/variant_bench.py run --config cfg1 --shape $shape --rounds 6 --steps 60 >> $L 2>&1
done
grep -v "amdgpu.ids\|rounds x" $L
