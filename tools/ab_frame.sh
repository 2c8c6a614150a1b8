#!/bin/bash
# same-box A/B of the frame leg: working-tree libpre3.so against 3pre_amd/lib/libpre3_head.so (tools/build_head_lib.sh), alternating
for rep in 1 2 3 4; do
  for t in head tree; do
    if [ "$t" = "tree" ]; then L=""; else L="PRE3_LIB=$PWD/3pre_amd/lib/libpre3_head.so"; fi
    echo "$t: $(eval "$L timeout -k 10 120 python3 tools/frame_trace.py 120" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['frames_per_s'],1), {k: round(v) for k, v in d['stage_us_synchronised'].items()})")"
  done
done
