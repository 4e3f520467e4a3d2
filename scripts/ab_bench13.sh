#!/bin/bash
# lean work list: chain order vs stable two-class partition by path length inside 1024-chain tiles
OUT=$1; : > "$OUT"
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --no-rmse --steps 64 --warmup 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'standalone': d['roofline'].get('standalone', {}).get('avg_launch_ms'), 'accept_rate': d['accept_rate']}))" | tee -a "$OUT"
}
run LMC_SORT_PLAIN=0
run LMC_SORT_PLAIN=3
run LMC_SORT_PLAIN=0 LMC_OVERLAP=0
run LMC_SORT_PLAIN=3 LMC_OVERLAP=0
