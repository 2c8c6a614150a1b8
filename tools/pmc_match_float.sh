#!/bin/bash
# SQ counters of the float-class matcher's ranking phases (k_rank_tiled<0>, <1>) at 4096 x 4096 x 128: what bounds them?  (separate passes)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  t=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mf/$t -o p -- python3 $R/tools/match_float.py > /dev/null 2>&1 || echo "failed: $set"
done
cd $R
python3 - <<'P'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_mf/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_rank_tiled" in k or "k_rank_tail" in k:
            name = "tiled<0>" if "tiled<0>" in k else "tiled<1>" if "tiled<1>" in k else "tail<" + ("double" if "double" in k else "float") + ">"
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(acc):
    print(name, {c: round(sum(v) / len(v)) for c, v in sorted(acc[name].items())})
P
