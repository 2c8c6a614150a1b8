#!/bin/bash
# k_cholp (persistent factorisation) durations with a down-date neighbour of rank $cfg running back to back in a second context on the
# same GPU (tools/overlap_probe.py): does MFMA / HBM work on the CUs the factorisation leaves idle slow its dependent chain?
for cfg in "0" "64" "320" "640"; do
r=$(echo $cfg | tr " " "_")
mkdir -p gpurun_out/ovc_$r
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ovc_$r -o ov -- python3 $GRAFT_REPO_ROOT/tools/overlap_probe.py $cfg 2>&1 | grep neighbour)
f=$(find gpurun_out/ovc_$r -name "ov_kernel_trace.csv" | head -1)
python3 - <<PY
import csv
cp=[]; k9=[]; ivs=[]
rows=list(csv.DictReader(open("$f")))
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"]); d=(e-s)/1e3
    if "k_cholp" in r["Kernel_Name"] and d>40: cp.append((s,e,d))
    if "k_downdate_b3" in r["Kernel_Name"]: k9.append((s,e,d))
# fraction of each k_cholp launch that overlapped some k_downdate_b3 launch
import bisect
k9.sort(); starts=[k[0] for k in k9]
ov=[]
for s,e,d in cp:
    i=max(0,bisect.bisect_left(starts,s)-2); t=0
    while i<len(k9) and k9[i][0]<e:
        t+=max(0,min(e,k9[i][1])-max(s,k9[i][0])); i+=1
    ov.append(t/(e-s))
v=sorted(d for _,_,d in cp)
print("  cfg $cfg: k_cholp launches %d  median %.1f us  p10 %.1f  p90 %.1f | mean overlap with K9 launches %.2f | K9 launches %d median %.1f us" % (len(v), v[len(v)//2], v[len(v)//10], v[9*len(v)//10], sum(ov)/max(1,len(ov)), len(k9), sorted(k[2] for k in k9)[len(k9)//2]))
PY
done
