set -e
mkdir -p gpurun_out
L=gpurun_out/r04_planes_first_$1.log
: > $L
for shape in 5,2,3,1048576 5,1,3,1048576 5,3,3,1048576 5,2,3,2097152 4,2,2,2097152 4,1,2,2097152 6,2,4,1048576 8,2,8,524288; do
  echo "== $shape with one-hot planes and reward, 16 candidates" >> $L
  timeout -k 10 300 python tools/variant_bench.py run --config cfg2 --shape $shape --onehot --rounds 6 --steps 60 --obs-candidates 16 >> $L 2>&1
done
grep -v "amdgpu.ids\|rounds x" $L
