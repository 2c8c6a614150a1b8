export HF_COUNTS="8 20 32 48 64"
PRE3_LIB=$PWD/3pre_amd/lib/libpre3_head.so bash tools/hi_fused_trace.sh head 2>&1 | grep -o "group [0-9]*:.*k_hi_fused [0-9.]*" | sed 's/step wall median//' | awk '{print "head", $0}' | cut -c1-60,150-
bash tools/hi_fused_trace.sh tree 2>&1 | grep -o "group [0-9]*:.*" > gpurun_out/hf_tree_groups.txt; python3 - <<'P'
import re
for l in open('gpurun_out/hf_tree_groups.txt'):
    m=re.search(r'group (\d+).*median ([0-9.]+) us.*k_hi_fused ([0-9.]+)', l)
    if m: print('tree group', m.group(1), 'wall', m.group(2), 'k_hi_fused', m.group(3))
P
timeout -k 10 600 python -m pytest tests/test_gpu_hi_fused.py tests/test_gpu_variants.py -x -q 2>&1 | tail -2
