#!/bin/bash
# Round 5, on the GPU box: every rocprofv3 pass behind profiles/r5_* (counters in passes of their own: --pmc with --kernel-trace only).
#   headline (bench.py): kernel trace + stats, FETCH_SIZE / WRITE_SIZE, three SQ sets
#   frame leg, fp64_n200 leg: kernel trace + stats
#   stand-alone K9 (k_downdate_b3 at r = 554 through pre3_bench_downdate): FETCH / WRITE / SQ
#   matchers: kernel trace + stats
#   the in-launch tail (PRE3_TAIL=1): kernel trace of the headline + the device-side timeline of one launch (probe build)
# then: python tools/make_profiles_r5.py
tag=${1:-r5}
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --no-cpu-baseline --no-extra-legs --no-check --legacy-steps 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -o $tag -- $B --steps 40 --warmup 5 > $R/gpurun_out/${tag}_trace.log 2>&1 || echo "failed: trace"
PRE3_TAIL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace_tail -o $tag -- $B --steps 40 --warmup 5 > $R/gpurun_out/${tag}_trace_tail.log 2>&1 || echo "failed: tail trace"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  t=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc/$t -o p -- $B --steps 10 --warmup 2 > $R/gpurun_out/${tag}_pmc_$t.log 2>&1 || echo "failed: $set"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_frame -o t -- python3 $R/tools/frame_trace.py 24 > $R/gpurun_out/${tag}_frame.log 2>&1 || echo "failed: frame"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_fp64 -o t -- python3 $R/tools/fp64_trace.py > $R/gpurun_out/${tag}_fp64.log 2>&1 || echo "failed: fp64"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  t=$(echo $set | tr ' ' '_' | cut -c1-40)
  K9_ROWS=554 timeout -k 10 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_k9s/$t -o p -- python3 $R/tools/k9ab.py > $R/gpurun_out/${tag}_pmc_k9s_$t.log 2>&1 || echo "failed: k9 $set"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_match -o m -- python3 $R/tools/match_ab.py child > $R/gpurun_out/${tag}_match.log 2>&1 || echo "failed: match"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_matchf -o mf -- python3 $R/tools/match_float.py > $R/gpurun_out/${tag}_matchf.log 2>&1 || echo "failed: matchf"
cd $R
PRE3_LIB=3pre_amd/lib/libpre3_probe.so timeout -k 10 200 python3 tools/probe_tail.py 30 > gpurun_out/${tag}_probe_tail.txt 2>&1 || echo "failed: probe_tail"
$B --steps 20 --warmup 5 > gpurun_out/${tag}_bench_headline.json 2> gpurun_out/${tag}_bench_headline.err || echo "failed: bench"
find gpurun_out/${tag}_* -name "*.csv" | wc -l
