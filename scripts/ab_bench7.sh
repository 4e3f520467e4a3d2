#!/bin/bash
# occupancy sweep of the lean kernel through its dynamic LDS size: latency-bound or throughput-bound?
OUT=$1; : > "$OUT"
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --no-rmse --steps 64 --warmup 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'roofline_frac': d['roofline']['frac'], 'accept_rate': d['accept_rate']}))" | tee -a "$OUT"
}
run LMC_OVERLAP=0 LMC_EXP_LDS_EXTRA=0
run LMC_OVERLAP=0 LMC_EXP_LDS_EXTRA=6144
run LMC_OVERLAP=0 LMC_EXP_LDS_EXTRA=12288
run LMC_OVERLAP=0 LMC_EXP_LDS_EXTRA=26000
run LMC_OVERLAP=0 LMC_EXP_LDS_EXTRA=66000
