#!/bin/bash
# k_select_gather's duration with parts of it switched off (PRE3_SG_DBG: 1 no S entries, 2 no W rows, 3 neither: the selection prefix alone)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for d in ${DBGS:-0 1 2 3}; do
  PRE3_SG_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sg_dbg_$d -o t -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs --no-check --legacy-steps 0 --steps 30 --warmup 5 > $R/gpurun_out/sg_dbg_$d.log 2>&1
  echo "dbg $d: $(grep -E 'k_select_gather|k_ransac_score' $R/gpurun_out/sg_dbg_$d/t_kernel_stats.csv | cut -d, -f1-4 | tr '\n' ' ')"
done
