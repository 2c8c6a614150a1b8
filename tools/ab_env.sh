#!/bin/bash
# same-box A/B of an environment switch: tools/ab_env.sh VAR val1 val2 ...   (working-tree library, bench.py headline only, 3 rounds)
V=$1; shift
B="python bench.py --no-cpu-baseline --no-extra-legs --steps 200 --warmup 10 ${BENCH_ARGS}"
for rep in 1 2 3; do for x in "$@"; do echo "$V=$x: $(env $V=$x timeout -k 10 120 $B 2>&1 | grep -o '"value": [0-9.]*' | head -1)"; done; done
