set -e
mkdir -p gpurun_out
L=gpurun_out/r04_two_stream_variants.log
: > $L
for pick in slowest fastest; do
  echo "== cfg2, $pick of 8 allocations" >> $L
  timeout -k 10 300 python tools/variant_bench.py run --config cfg2 --rounds 6 --steps 100 --placement 7 --pick $pick >> $L 2>&1
done
for bpw in 16 64; do
  echo "== cfg2, slowest of 8 allocations, boards per wave $bpw" >> $L
  timeout -k 10 300 python tools/variant_bench.py run --config cfg2 --rounds 6 --steps 100 --placement 7 --pick slowest --tuning 8=$bpw --only base,pf,fence >> $L 2>&1
done
cat $L
