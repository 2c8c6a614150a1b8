#!/bin/bash
# kernel trace of the headline workload (40 steps) -> gpurun_out/r4_tr_$1 ; prints the per-kernel stats head and three one-step timelines
R=$GRAFT_REPO_ROOT; tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_tr_$tag -o t -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs --no-check --legacy-steps 0 --steps 40 --warmup 5 > $R/gpurun_out/r4_tr_$tag.log 2>&1
cd $R
f=$(find gpurun_out/r4_tr_$tag -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-150
t=$(find gpurun_out/r4_tr_$tag -name "*kernel_trace.csv" | head -1)
for k in 3 7 12; do python3 tools/step_timeline.py $t $k; done
