#!/bin/bash
# second A/B round: technique sort inside 256-chain tiles
OUT=$1; : > "$OUT"
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --steps 64 --warmup 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'roofline_frac': d['roofline']['frac'], 'accept_rate': d['accept_rate']}))" | tee -a "$OUT"
}
run LMC_SORT_PLAIN=0 LMC_LEAN_BLOCK=64
run LMC_SORT_PLAIN=2 LMC_LEAN_BLOCK=256
run LMC_SORT_PLAIN=2 LMC_LEAN_BLOCK=64
run LMC_SORT_PLAIN=2 LMC_LEAN_BLOCK=128
run LMC_SORT_PLAIN=1 LMC_LEAN_BLOCK=256
