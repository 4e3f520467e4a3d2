#!/bin/bash
# A/B of two builds of the library (LMC_LIB): previous build in csrc/_build/w3 vs the tree's
OUT=$1; : > "$OUT"
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --no-rmse --steps 64 --warmup 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'large_ms': d['step_ms']['large_and_generic'], 'standalone': d['roofline'].get('standalone', {}).get('avg_launch_ms')}))" | tee -a "$OUT"
}
run LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_build/w3/liblmc_hip.so LMC_OVERLAP=0
run LMC_X=new LMC_OVERLAP=0
run LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_build/w3/liblmc_hip.so
run LMC_X=new
