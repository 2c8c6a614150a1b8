#!/bin/bash
# The N = 2000 step under rocprofv3 --kernel-trace: per-kernel time per step (sum over the steps of the timed region / steps).
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_n2000 -o t -- python3 tools/n2000_step.py > gpurun_out/tl_n2000.log 2>&1
python3 - <<'P'
import csv, glob, collections
f = glob.glob('gpurun_out/tl_n2000/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_predict' in r['Kernel_Name']]
i0, i1 = idx[-3], idx[-2]                       # one whole step late in the run
tot = collections.OrderedDict(); cnt = collections.Counter(); gap = 0.0
prev = rows[i0 - 1]
for r in rows[i0:i1]:
    nm = r['Kernel_Name'].split('(')[0].replace('pre3::', '').replace('void ', '')[:44]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot[nm] = tot.get(nm, 0.0) + d; cnt[nm] += 1
    gap += max(0.0, (int(r['Start_Timestamp']) - int(prev['End_Timestamp'])) / 1e3); prev = r
wall = (int(rows[i1]['Start_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3
for nm, d in tot.items():
    print('%-46s n %4d  total %9.1f us  avg %8.2f us' % (nm, cnt[nm], d, d / cnt[nm]))
print('step wall %.1f us, kernel busy %.1f us, gaps %.1f us' % (wall, sum(tot.values()), gap))
P
rm -rf gpurun_out/tl_n2000
