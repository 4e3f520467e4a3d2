#!/bin/bash
# where does the lean kernel's time go?  measurement aids (results are NOT valid renders): no film splats / no cache queries
OUT=$1; : > "$OUT"
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --steps 64 --warmup 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'large_ms': d['step_ms']['large_and_generic'], 'accept_rate': d['accept_rate']}))" | tee -a "$OUT"
}
run LMC_OVERLAP=0
run LMC_OVERLAP=0 LMC_EXP_NOSPLAT=1
run LMC_OVERLAP=0 LMC_EXP_NOQUERY=1
run LMC_OVERLAP=0 LMC_EXP_NOSPLAT=1 LMC_EXP_NOQUERY=1
