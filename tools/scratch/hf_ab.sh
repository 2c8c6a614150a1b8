export HF_COUNTS="8 20 32 48"
for l in head tree head tree; do
if [ $l = head ]; then export PRE3_LIB=$PWD/3pre_amd/lib/libpre3_head.so; else unset PRE3_LIB; fi
bash tools/hi_fused_trace.sh $l 2>&1 | grep -o "group [0-9]*:.*" > gpurun_out/hf_groups.txt; python3 - $l <<'P'
import re,sys
out=[]
for l in open('gpurun_out/hf_groups.txt'):
    m=re.search(r'group (\d+).*k_hi_fused ([0-9.]+)', l)
    if m: out.append(m.group(2))
print(sys.argv[1], ' '.join(out))
P
done
