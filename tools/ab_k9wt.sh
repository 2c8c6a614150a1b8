#!/bin/bash
# same-box A/B of the write-through stores of P in k_downdate_b3 (PRE3_K9_WT) on the legs that launch it stand-alone
for w in 0 1 0 1; do
  PRE3_K9_WT=$w python3 bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null > /tmp/b_$w.json
  python3 - $w <<'PY'
import json, sys
w = sys.argv[1]
d = json.loads(open('/tmp/b_%s.json' % w).read().strip().splitlines()[-1])
print("wt", w, round(d["value"], 1), "n2000", round(d["n2000_step"]["value"], 2), json.dumps(d["n2000_step"].get("k9", {}))[:160], "k9sa", json.dumps(d["roofline"].get("k9_standalone", {}))[:120])
PY
done
