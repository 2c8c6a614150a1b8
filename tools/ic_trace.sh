#!/bin/bash
# kernels of the frame leg's IC search under rocprofv3; usage: tools/ic_trace.sh <tag> [VAR=val ...]
tag=$1; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
out=gpurun_out/ic_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 tools/frame_trace.py 24 > $out.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - $f <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0].replace('pre3::', '').replace('void ', '')
    if any(k in n for k in ('k_ic', 'k_rank', 'k_match', 'k_project_inn', 'k_scan', 'k_map', 'k_inbox', 'fill')):
        print('%-40s calls %4s avg %.1f us' % (n[:40], r['Calls'], float(r['AverageNs']) / 1e3))
P
tail -1 $out.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('frames/s', round(d['frames_per_s'], 1), d['stage_us_synchronised'])"
rm -rf $out
