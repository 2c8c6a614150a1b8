#!/bin/bash
# HBM-side bytes of the fused k_cholp launches (FETCH_SIZE / WRITE_SIZE, separate passes) on the headline workload -> prints the LI launches' means
R=$GRAFT_REPO_ROOT; tag=${1:-x}
B="python3 $R/bench.py --no-cpu-baseline --no-extra-legs --no-check --legacy-steps 0 --steps 10 --warmup 2"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcc_${tag}_f -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcc_${tag}_w -o w -- $B > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob
for nm, d in (("FETCH_SIZE", "f"), ("WRITE_SIZE", "w")):
    fn = glob.glob("gpurun_out/pmcc_${tag}_%s/**/*counter_collection.csv" % d, recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fn)) if "k_cholp" in r["Kernel_Name"] and r["Counter_Name"] == nm]
    big = [x for x in v if x > 0.6 * max(v)]
    print(nm, "LI launches %d  mean %.1f MB" % (len(big), sum(big) / len(big) / 1024.0))
PY
