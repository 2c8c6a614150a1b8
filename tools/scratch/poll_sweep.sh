python tools/probe_cholp.py 500 6 2>/dev/null | sed -n 2,24p | cut -c1-200
timeout -k 10 600 python -m pytest tests/test_gpu_cholp.py tests/test_gpu_fullsize.py tests/test_gpu_variants.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
echo "tree: $(python bench.py --no-cpu-baseline --no-extra-legs --steps 200 --warmup 10 2>&1 | grep -o '"value": [0-9.]*' | head -1)"
echo "head: $(PRE3_LIB=$PWD/3pre_amd/lib/libpre3_head.so python bench.py --no-cpu-baseline --no-extra-legs --steps 200 --warmup 10 2>&1 | grep -o '"value": [0-9.]*' | head -1)"
done
