#!/bin/bash
# per-count duration of k_hi_fused (and the general path's launches) from a rocprofv3 kernel trace; usage: tools/hi_fused_trace.sh <tag> [env...]
set -e
tag=$1; shift
out=gpurun_out/hf_$tag
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/time_hi_fused.py $HF_COUNTS > $out.txt 2>&1
python3 - $out <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0].replace('pre3::','').replace('void ','') for r in rows]
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
# steps: from k_predict to the next k_predict
idx = [i for i, n in enumerate(names) if n.startswith('k_predict')]
steps = []
for a, b in zip(idx, idx[1:] + [len(rows)]):
    d = collections.OrderedDict()
    for i in range(a, b):
        d[names[i]] = d.get(names[i], 0) + dur[i]
    wall = (int(rows[b-1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3
    steps.append((d, wall))
import statistics
for g in range(0, len(steps), 30):
    grp = steps[g:g+30][5:]
    if not grp: continue
    keys = list(grp[-1][0].keys())
    print('group %d: step wall median %.1f us; ' % (g // 30, statistics.median(w for _, w in grp)) + ', '.join('%s %.1f' % (k[:28], statistics.median(d.get(k, 0) for d, _ in grp)) for k in keys if not k.startswith('k_cholp') or True))
P
