#!/bin/bash
# queue priority of the side launches (large steps, cache-filling steps) relative to the lean launch
OUT=$1; : > "$OUT"
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --no-rmse --steps 64 --warmup 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'large_ms': d['step_ms']['large_and_generic'], 'frac': d['roofline']['frac']}))" | tee -a "$OUT"
}
run LMC_STREAM_PRIO=1
run LMC_STREAM_PRIO=0
run LMC_STREAM_PRIO=-1
run LMC_STREAM_PRIO=-1 LMC_LARGE_BLOCK=256
run LMC_OVERLAP=0
