#!/bin/bash
# cache-query pre-filter A/B
OUT=$1; : > "$OUT"
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --steps 64 --warmup 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'roofline_frac': d['roofline']['frac'], 'accept_rate': d['accept_rate']}))" | tee -a "$OUT"
}
run LMC_OCC_FILTER=0
run LMC_GRID_DIMS=3
run LMC_GRID_DIMS=4
run LMC_GRID_DIMS=4 LMC_OVERLAP=0
