set -e
mkdir -p gpurun_out
L=gpurun_out/r04_two_stream_variants3.log
: > $L
for shape in 5,2,3,1048576 5,4,3,1048576 5,8,3,524288 4,2,2,2097152 4,4,2,1048576 6,3,4,524288 6,6,4,524288 6,2,4,1048576 7,5,6,262144 7,2,6,524288 8,4,8,262144 8,8,8,131072 8,2,8,524288 3,1,0,4194304 3,2,0,4194304; do
  echo "== $shape with one-hot planes and reward, 16 candidates" >> $L
  timeout -k 10 300 python tools/variant_bench.py run --config cfg2 --shape $shape --onehot --rounds 6 --steps 60 --obs-candidates 16 --only base,pf >> $L 2>&1
done
grep -v "amdgpu.ids\|rounds x" $L
