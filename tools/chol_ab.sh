#!/bin/bash
# per-launch durations of the factorisation (k_chol_step) from a rocprofv3 kernel trace of bench.py: working tree against 3pre_amd/lib/libpre3_head.so
for v in "" _head; do
  mkdir -p gpurun_out/tl$v
  (cd /tmp && export TMPDIR=/tmp && PRE3_LIB=$GRAFT_REPO_ROOT/3pre_amd/lib/libpre3$v.so rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl$v -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-legs --steps 60 --warmup 5 ${BENCH_ARGS} > /dev/null 2>&1)
  python3 - <<PY
import csv, collections
by=collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/tl$v/t_kernel_trace.csv")):
    if "k_chol_step" in r["Kernel_Name"]: by[int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
tot=0; out=[]
for g in sorted(by, reverse=True):
    v=sorted(by[g])
    if len(v)>20: out.append("%.1f" % v[len(v)//2]); tot+=v[len(v)//2]
print("lib$v launches (median us):", " ".join(out), " sum %.1f" % tot)
PY
done
