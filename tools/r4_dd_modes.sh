#!/bin/bash
for m in 1 2 0 3; do
  echo "== PRE3_DD_MODE=$m (bit 0: sc1 LDS-DMA, bit 1: one agent acquire per panel)"
  PRE3_DD_MODE=$m PRE3_K9_OVERLAP=1 timeout -k 10 200 python tools/probe_cholp.py 500 3 > gpurun_out/r4_probe_m$m.txt 2>&1; grep -A14 "down-date consumers" gpurun_out/r4_probe_m$m.txt | tail -5; grep "J= 9" gpurun_out/r4_probe_m$m.txt | head -1 | cut -c1-60
  PRE3_DD_MODE=$m timeout -k 10 300 python -m pytest tests/test_gpu_cholp.py -q -x -k consumers 2>&1 | tail -1
done
