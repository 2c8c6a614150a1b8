#!/bin/bash
# SQ / SQC counters of the factorisation launches (k_chol_step) inside bench.py steps: instruction fetch and wait states
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_chol
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 160 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_chol/$tag -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/pmc_chol/$tag.log 2>&1 || echo "failed: $set"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_chol/*/p_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for key in ("k_chol_step", "k_downdate_1t", "k_ransac_score"):
            if key in r["Kernel_Name"]:
                acc[(key, r["Counter_Name"])].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
    for k,v in sorted(acc.items()):
        print("%-16s %-30s launches %4d  mean %.5g   (kernel %.1f us)" % (k[0], k[1], len(v), sum(x[0] for x in v)/len(v), sum(x[1] for x in v)/len(v)/1e3))
PY
