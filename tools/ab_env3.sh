#!/bin/bash
# same-box A/B of environment settings on the headline: tools/ab_env3.sh "VAR=1" "VAR=0 OTHER=0" ...   (200-step value and the driver's 20-step window)
for rep in 1 2 3; do for kv in "$@"; do echo "$kv: $(env $kv python bench.py --no-cpu-baseline --no-extra-legs --steps 200 --warmup 10 2>&1 | grep -o '"value": [0-9.]*' | head -1) window $(env $kv python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 2>&1 | grep -o '"value": [0-9.]*' | head -1)"; done; done
