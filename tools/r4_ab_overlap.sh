#!/bin/bash
# same-box A/B: down-date consumers inside k_cholp (PRE3_K9_OVERLAP=1, default) against the K9 launch behind it (0); then a kernel trace of each
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for x in 1 0; do
  for args in "--steps 20 --warmup 5" "--steps 200 --warmup 10"; do
    echo "PRE3_K9_OVERLAP=$x [$args]: $(env PRE3_K9_OVERLAP=$x timeout -k 10 120 python bench.py --no-cpu-baseline --no-extra-legs --no-check $args 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], 'no_hi', d.get('no_hi', {}).get('value'))
")"
  done
done; done
cd /tmp && export TMPDIR=/tmp
for x in 1 0; do
  PRE3_K9_OVERLAP=$x rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_ov${x}_trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs --no-check --legacy-steps 0 --steps 40 --warmup 5 > $R/gpurun_out/r4_ov${x}_trace.log 2>&1
done
cd $R
for x in 1 0; do
  f=$(find gpurun_out/r4_ov${x}_trace -name "*kernel_stats.csv" | head -1); echo "== overlap=$x"; head -14 $f | cut -c1-160
done
