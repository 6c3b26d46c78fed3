set -e
mkdir -p gpurun_out
L=gpurun_out/r04_two_stream_variants2.log
: > $L
echo "== cfg2, observation buffer chosen among 16 candidates" >> $L
timeout -k 10 300 python tools/variant_bench.py run --config cfg2 --rounds 8 --steps 100 --obs-candidates 16 >> $L 2>&1
echo "== cfg2, first allocation" >> $L
timeout -k 10 300 python tools/variant_bench.py run --config cfg2 --rounds 8 --steps 100 >> $L 2>&1
for shape in 4,2,2,2097152 6,3,4,524288 8,4,8,262144 5,6,3,1048576 3,1,0,4194304; do
  echo "== $shape with one-hot planes and reward, 16 candidates" >> $L
  timeout -k 10 300 python tools/variant_bench.py run --config cfg2 --shape $shape --onehot --rounds 6 --steps 60 --obs-candidates 16 >> $L 2>&1
done
grep -v "amdgpu.ids" $L
