#!/bin/bash
# (KERN="k_cholp k_downdate_b3" lists the kernels to report; per kernel the launches longer than half its longest)
# same-session WRITE_SIZE / FETCH_SIZE of the LI launches under an environment switch: tools/pmc_env_write.sh VAR val1 val2 ...
V=$1; shift
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --no-cpu-baseline --no-extra-legs --no-check --legacy-steps 0 --steps 10 --warmup 2"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_env
for x in "$@"; do for set in FETCH_SIZE WRITE_SIZE; do
  export $V=$x
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_env/$V=$x/$set -o p -- $B > /dev/null 2>&1 || echo "failed: $x $set"
done; done
cd $R
python3 - <<'P'
import csv, glob, os
for lib in sorted(glob.glob("gpurun_out/pmc_env/*")):
    for cdir in sorted(glob.glob(lib + "/*")):
      for kern in os.environ.get("KERN", "k_cholp").split():
        cc = [r for r in csv.DictReader(open(glob.glob(cdir + "/**/p_counter_collection.csv", recursive=True)[0])) if kern in r["Kernel_Name"]]
        kt = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(glob.glob(cdir + "/**/p_kernel_trace.csv", recursive=True)[0])) if kern in r["Kernel_Name"]}
        v = [(float(r["Counter_Value"]), kt.get(r["Dispatch_Id"], 0.0)) for r in cc]
        if not v: continue
        dmax = max(d for _, d in v)
        sel = [x for x in v if x[1] > 0.5 * dmax]
        print("%-24s %-11s %-16s long launches %2d  %.1f MB  mean duration %.1f us" % (lib.split("/")[-1], cdir.split("/")[-1], kern, len(sel), sum(x[0] for x in sel) / len(sel) * 1024 / 1e6, sum(x[1] for x in sel) / len(sel)))
P
