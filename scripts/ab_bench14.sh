#!/bin/bash
# block sizes of the lean and the large-step launch, re-checked on the round's final kernels
OUT=$1; : > "$OUT"
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --no-rmse --steps 64 --warmup 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'large_ms': d['step_ms']['large_and_generic']}))" | tee -a "$OUT"
}
run LMC_LEAN_BLOCK=64 LMC_LARGE_BLOCK=64
run LMC_LEAN_BLOCK=128 LMC_LARGE_BLOCK=64
run LMC_LEAN_BLOCK=64 LMC_LARGE_BLOCK=128
run LMC_LEAN_BLOCK=64 LMC_LARGE_BLOCK=256
