#!/bin/bash
# kernel trace of the frame leg (map management + prediction + IC search + RANSAC + updates at N=500, K2=600) -> per-kernel stats and one frame's timeline
R=$GRAFT_REPO_ROOT; tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/r4_frame_$tag -o t -- python3 $R/tools/frame_trace.py ${FRAMES:-16} > $R/gpurun_out/r4_frame_$tag.log 2>&1
cd $R
f=$(find gpurun_out/r4_frame_$tag -name "*kernel_stats.csv" | head -1); head -30 $f | cut -d, -f1-4 | cut -c1-110
tail -2 gpurun_out/r4_frame_$tag.log | cut -c1-600
