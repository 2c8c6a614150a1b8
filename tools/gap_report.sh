#!/bin/bash
# kernel-to-kernel gaps of any command under rocprofv3: tools/gap_report.sh <tag> <python script and args...>
export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap_$tag -o t -- python3 "$@" > gpurun_out/gap_$tag.log 2>&1
python3 - gpurun_out/gap_$tag <<'P'
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
gaps = collections.defaultdict(list); busy = 0
for a, b in zip(rows, rows[1:]):
    n = b['Kernel_Name'].split('(')[0].replace('pre3::','').replace('void ','')[:30]
    p = a['Kernel_Name'].split('(')[0].replace('pre3::','').replace('void ','')[:26]
    gaps[(p, n)].append((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in gaps.values())
print('kernels %d, span %.1f ms, sum of gaps %.1f ms' % (len(rows), (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6, tot / 1e3))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print('%-28s -> %-32s n=%4d median %7.2f  total %9.1f us' % (k[0], k[1], len(v), statistics.median(v), sum(v)))
P
rm -rf gpurun_out/gap_$tag
