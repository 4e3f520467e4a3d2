#!/bin/bash
# same-address statistics atomics of the lean kernel (one set per 64-thread block): a serialisation point?
OUT=$1; : > "$OUT"
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --no-rmse --steps 64 --warmup 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'roofline_frac': d['roofline']['frac']}))" | tee -a "$OUT"
}
run LMC_OVERLAP=0 LMC_EXP_NOSTATS=0
run LMC_OVERLAP=0 LMC_EXP_NOSTATS=1
run LMC_OVERLAP=0 LMC_EXP_NOSTATS=1 LMC_EXP_NOSPLAT=1
run LMC_OVERLAP=0 LMC_LEAN_BLOCK=256
