set -e
mkdir -p gpurun_out
L=gpurun_out/r04_lines_nt_loads.log
: > $L
echo "== cfg4" >> $L
timeout -k 10 300 python tools/variant_bench.py run --config cfg4 --rounds 6 --steps 100 >> $L 2>&1
for shape in 15,32,24,518400 14,20,20,300000 12,8,16,400000 9,4,9,600000 16,16,24,228000 24,30,60,100000 32,32,100,60000 20,10,40,250000; do
  echo "== $shape" >> $L
  timeout -k 10 300 python tools/variant_bench.py run --config cfg1 --shape $shape --rounds 5 --steps 40 >> $L 2>&1
done
grep -v "amdgpu.ids\|rounds x" $L
