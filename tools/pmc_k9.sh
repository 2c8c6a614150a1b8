#!/bin/bash
# SQ counters of the K9 launch (k_downdate_b3; PRE3_K9_B3=0 + the filter below for the f32 MFMA form) (separate passes; tools/k9ab.py runs bench_downdate at r=320/640)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_k9/$tag -o p --output-format csv -- python3 $R/tools/k9ab.py > $R/gpurun_out/pmc_k9/$tag.log 2>&1 || echo "failed: $set"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_k9/*/p_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_downdate_b3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
    for k,v in acc.items():
        big=[x for x in v if x[1] > 0.8*max(y[1] for y in v)]      # r=640 launches
        print("%-32s r=640 launches %3d  mean %.4g   (kernel %.1f us)" % (k, len(big), sum(x[0] for x in big)/len(big), sum(x[1] for x in big)/len(big)/1e3))
PY
