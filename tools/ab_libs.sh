#!/bin/bash
# same-box A/B of library variants: tools/ab_libs.sh <tag1> <tag2> ...   (tag "" = working-tree libpre3.so; others 3pre_amd/lib/libpre3_<tag>.so)
B="python bench.py --no-cpu-baseline --no-extra-legs --steps 200 --warmup 10 ${BENCH_ARGS}"
run() { if [ "$1" = "tree" ]; then L=""; else L="PRE3_LIB=$PWD/3pre_amd/lib/libpre3_$1.so"; fi; eval "$L timeout -k 10 120 $B" 2>&1 | grep -o '"value": [0-9.]*' | head -1; }
for rep in 1 2 3; do for t in "$@"; do echo "$t: $(run $t)"; done; done
