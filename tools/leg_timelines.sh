#!/bin/bash
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_fp64 -o t -- python3 tools/fp64_trace.py > gpurun_out/tl_fp64.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_frame -o t -- python3 tools/frame_trace.py 24 > gpurun_out/tl_frame.log 2>&1
python3 - <<'P'
import csv, glob
for leg, first in (("tl_fp64", "k_predict"), ("tl_frame", "k_scan_pull")):
    f = glob.glob('gpurun_out/%s/**/*kernel_trace.csv' % leg, recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]
    i0 = idx[len(idx) // 2]
    prev = rows[i0 - 1]
    n = 0
    for r in rows[i0:i0 + 26]:
        nm = r['Kernel_Name'].split('(')[0].replace('pre3::', '').replace('void ', '')[:34]
        print('%-36s gap %6.2f dur %6.2f' % (nm, (int(r['Start_Timestamp']) - int(prev['End_Timestamp'])) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
        prev = r
    print()
P
rm -rf gpurun_out/tl_fp64 gpurun_out/tl_frame
