for rep in 1 2; do
for l in tree head; do
if [ $l = head ]; then export PRE3_LIB=$PWD/3pre_amd/lib/libpre3_head.so; else unset PRE3_LIB; fi
python bench.py --no-cpu-baseline --no-extra-legs --steps 200 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$l', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'frac', round(r['frac'],4), {k:(round(v,2) if isinstance(v,float) else v) for k,v in r.items() if k in ('avg_launch_us','launches','achieved')})"
done; done
