for cfg in "0 0" "600 6"; do set -- $cfg
  echo "== poll $1 from $2"
  PRE3_CHOLP_POLL=$1 PRE3_CHOLP_POLL_FROM=$2 python tools/probe_cholp.py 500 6 2>/dev/null | sed -n 6,9p | cut -c1-200
done
for rep in 1 2 3; do
for cfg in "0 0" "600 6"; do set -- $cfg
  echo "poll $1 from $2 bench: $(PRE3_CHOLP_POLL=$1 PRE3_CHOLP_POLL_FROM=$2 python bench.py --no-cpu-baseline --no-extra-legs --steps 200 --warmup 10 2>&1 | grep -o '"value": [0-9.]*' | head -1)"
done
echo "head: $(PRE3_LIB=$PWD/3pre_amd/lib/libpre3_head.so python bench.py --no-cpu-baseline --no-extra-legs --steps 200 --warmup 10 2>&1 | grep -o '"value": [0-9.]*' | head -1)"
done
