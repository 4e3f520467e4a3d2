#!/bin/bash
# stream-priority / cache-filling-kernel A/B on the driver's window and on the steady state
OUT=$1; : > "$OUT"
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --no-rmse $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$* $ARGS', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'roofline_frac': d['roofline']['frac'], 'accept_rate': d['accept_rate']}))" | tee -a "$OUT"
}
ARGS="--steps 20 --warmup 5"
run LMC_STREAM_PRIO=0
run LMC_STREAM_PRIO=1
run LMC_STREAM_PRIO=1 LMC_LEAN_GRAD=0
ARGS="--steps 64 --warmup 40"
run LMC_STREAM_PRIO=0
run LMC_STREAM_PRIO=1
