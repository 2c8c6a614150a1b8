#!/bin/bash
# same-session A/B of the HBM-side bytes of the LI launches (k_cholp): working-tree library against 3pre_amd/lib/libpre3_<tag>.so  (FETCH_SIZE / WRITE_SIZE passes)
#   tools/pmc_ab_write.sh r5     (on the GPU box; summaries -> gpurun_out/pmc_ab_<tag>.txt)
tag=${1:-r5}
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --no-cpu-baseline --no-extra-legs --no-check --legacy-steps 0 --steps 10 --warmup 2"
cd /tmp && export TMPDIR=/tmp
for lib in tree $tag; do
  for set in FETCH_SIZE WRITE_SIZE; do
    if [ $lib = tree ]; then unset PRE3_LIB; else export PRE3_LIB=$R/3pre_amd/lib/libpre3_$lib.so; fi
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ab/$lib/$set -o p -- $B > $R/gpurun_out/pmc_ab_${lib}_$set.log 2>&1 || echo "failed: $lib $set"
  done
done
cd $R
python3 - <<'P' | tee gpurun_out/pmc_ab_$tag.txt
import csv, glob, collections
for lib in sorted(glob.glob("gpurun_out/pmc_ab/*")):
    for cdir in sorted(glob.glob(lib + "/*")):
        cc = [r for r in csv.DictReader(open(glob.glob(cdir + "/**/p_counter_collection.csv", recursive=True)[0])) if "k_cholp" in r["Kernel_Name"]]
        kt = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(glob.glob(cdir + "/**/p_kernel_trace.csv", recursive=True)[0])) if "k_cholp" in r["Kernel_Name"]}
        v = [(float(r["Counter_Value"]), kt.get(r["Dispatch_Id"], 0.0)) for r in cc]
        dmax = max(d for _, d in v)
        sel = [x for x in v if x[1] > 0.5 * dmax]
        print("%-28s %-11s LI launches %2d  mean %.1f KB-units (x1024 B = %.1f MB)  mean duration %.1f us" % (lib.split("/")[-1], cdir.split("/")[-1], len(sel), sum(x[0] for x in sel) / len(sel), sum(x[0] for x in sel) / len(sel) * 1024 / 1e6, sum(x[1] for x in sel) / len(sel)))
P
