#!/bin/bash
# A/B on ONE box: the working-tree library against 3pre_amd/lib/libpre3_head.so (built from HEAD by tools/build_head_lib.sh)
B="python bench.py --no-cpu-baseline --no-extra-legs --steps 200 --warmup 10"
run() { eval "$1 timeout -k 10 120 $B $2" 2>&1 | grep -o '"value": [0-9.]*' | head -1; }
for rep in 1 2 3; do
echo "f32 new : $(run '' '')"
echo "f32 head: $(run 'PRE3_LIB=$PWD/3pre_amd/lib/libpre3_head.so' '')"
done
echo "f64 new : $(run '' '--landmarks 200 --dtype f64')"
echo "f64 head: $(run 'PRE3_LIB=$PWD/3pre_amd/lib/libpre3_head.so' '--landmarks 200 --dtype f64')"
