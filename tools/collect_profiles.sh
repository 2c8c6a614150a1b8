#!/bin/bash
# On the GPU box: the rocprofv3 passes behind profiles/<tag>_* (kernel trace + stats of bench.py, the two HBM-byte PMC passes of the
# same command -- separate passes, counters only with --kernel-trace --, the matcher's kernel trace, and the SQ counters of the
# factorisation launches).  Then: python tools/make_profiles.py <tag>
tag=${1:-r4}
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --no-cpu-baseline --no-extra-legs --no-check --legacy-steps 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -o $tag -- $B --steps 40 --warmup 5 > $R/gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_fetch -o f -- $B --steps 10 --warmup 2 > $R/gpurun_out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_write -o w -- $B --steps 10 --warmup 2 > $R/gpurun_out/${tag}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_match -o m -- python3 $R/tools/match_ab.py child > $R/gpurun_out/${tag}_match.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_matchf -o mf -- python3 $R/tools/match_float.py > $R/gpurun_out/${tag}_matchf.log 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  t=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_sq/$t -o p -- $B --steps 10 --warmup 2 > $R/gpurun_out/${tag}_pmc_sq_$t.log 2>&1 || echo "failed: $set"
done
cd $R
find gpurun_out/${tag}_* -name "*.csv" | head -30
