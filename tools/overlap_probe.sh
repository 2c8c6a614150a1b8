#!/bin/bash
# k_chol_step durations with a down-date neighbour of rank $1 on the same GPU (tools/overlap_probe.py)
for cfg in "0" "64" "-1 8 195 0" "-1 100 195 0" "-1 100 195 4" "-1 100 1024 0"; do
r=$(echo $cfg | tr " " "_")
mkdir -p gpurun_out/ov_$r
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ov_$r -o ov -- python3 $GRAFT_REPO_ROOT/tools/overlap_probe.py $cfg 2>&1 | grep neighbour)
f=$(find gpurun_out/ov_$r -name "ov_kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections
by=collections.defaultdict(list); k9=[]
for r in csv.DictReader(open("$f")):
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if "k_chol_step" in r["Kernel_Name"]: by[0].append(d)
    if "k_downdate_b3" in r["Kernel_Name"] or "nb_stream" in r["Kernel_Name"]: k9.append(d)
out=[]; tot=0
for g in sorted(by, reverse=True):
    v=sorted(by[g])
    if len(v)>20: out.append("%.1f" % v[len(v)//2]); tot+=v[len(v)//2]
print("  k_chol_step medians:", " ".join(out), " sum %.1f" % tot, "| neighbour + K9 launches %d, median %.1f us" % (len(k9), sorted(k9)[len(k9)//2]))
PY
done
