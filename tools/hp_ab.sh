#!/bin/bash
# per-kernel averages of the front of the step under an environment switch, one box: tools/hp_ab.sh VAR val1 val2 ...  (rocprofv3 --kernel-trace --stats of bench.py)
V=$1; shift
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --no-cpu-baseline --no-extra-legs --no-check --legacy-steps 0 --steps 40 --warmup 5"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export $V=$v
  rm -rf $R/gpurun_out/hp_ab_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hp_ab_$v -o p -- $B > /dev/null 2>&1
  python3 - "$V=$v" $(find $R/gpurun_out/hp_ab_$v -name "p_kernel_stats.csv" | head -1) <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if any(k in r["Name"] for k in ("k_ell_HP_build", "k_ransac_score", "k_predict", "k_innov", "k_select_gather", "k_jnorm", "k_hi_fused", "k_downdate_b3", "k_cholp")):
        print("%-16s %-44s calls %3s  avg %7.2f us  min %7.2f  max %7.2f" % (sys.argv[1], r["Name"].replace("void pre3::", "")[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
P
done
