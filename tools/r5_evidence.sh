#!/bin/bash
# Round 5: the evidence round 4's verdict found missing (item 7): kernel traces of the `frame` and `fp64_n200` legs, and FETCH / WRITE / SQ
# passes of the stand-alone K9 (k_downdate_b3 at r = 554 through pre3_bench_downdate).  Counters in passes of their own (--pmc + --kernel-trace only).
R=$GRAFT_REPO_ROOT; tag=${1:-r5}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_frame -o t -- python3 $R/tools/frame_trace.py 24 > $R/gpurun_out/${tag}_frame.log 2>&1 || echo "frame trace failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_fp64 -o t -- python3 $R/tools/fp64_trace.py > $R/gpurun_out/${tag}_fp64.log 2>&1 || echo "fp64 trace failed"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  t=$(echo $set | tr ' ' '_' | cut -c1-40)
  K9_ROWS=554 timeout -k 10 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_k9s/$t -o p -- python3 $R/tools/k9ab.py > $R/gpurun_out/${tag}_pmc_k9s_$t.log 2>&1 || echo "failed: $set"
done
cd $R
find gpurun_out/${tag}_* -name "*.csv" | head -40
