#!/bin/bash
# A/B of two builds on the driver's window (fresh start: cache fill inside the window)
OUT=$1; : > "$OUT"
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --no-rmse --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'large_and_generic_ms': d['step_ms']['large_and_generic']}))" | tee -a "$OUT"
}
run LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_build/w3/liblmc_hip.so
run LMC_X=new
run LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_build/w3/liblmc_hip.so LMC_OVERLAP=0
run LMC_X=new LMC_OVERLAP=0
