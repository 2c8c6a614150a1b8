#!/bin/bash
# same-box A/B of an environment switch on the headline and the two legacy legs: tools/ab_env2.sh VAR val1 val2 ...
V=$1; shift
for rep in 1 2 3; do for x in "$@"; do
  echo "$V=$x: $(env $V=$x timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs --no-check --steps ${STEPS:-200} --warmup ${WARM:-10} 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('headline', round(d['value'],1), 'hi rows', round(d['config']['mean_hi_rows'],1), '| thr05', round(d['thr05']['value'],1), '| no_hi', round(d['no_hi']['value'],1))
")"
done; done
