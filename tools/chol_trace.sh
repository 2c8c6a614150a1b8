#!/bin/bash
# per-launch durations of the factorisation (k_chol_step) and a one-step timeline from a rocprofv3 kernel trace of bench.py (working-tree library)
# usage (on the GPU box): bash tools/chol_trace.sh <tag>
tag=${1:-t}
mkdir -p gpurun_out/tl_$tag
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_$tag -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 ${BENCH_ARGS} > /dev/null 2>&1)
f=$(find gpurun_out/tl_$tag -name "${tag}_kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections
by=collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    if "k_chol_step" in r["Kernel_Name"]: by[int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
tot=0; out=[]
for g in sorted(by, reverse=True):
    v=sorted(by[g])
    if len(v)>20: out.append("%.1f" % v[len(v)//2]); tot+=v[len(v)//2]
print("k_chol_step launches (median us):", " ".join(out), " sum %.1f" % tot)
PY
python3 tools/step_timeline.py $f 3
