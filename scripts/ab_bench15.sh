#!/bin/bash
# lean kernel compiled with other GCN scheduler strategies (-mllvm -amdgpu-sched-strategy=...), A/B through LMC_LIB
OUT=$1; : > "$OUT"
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --no-rmse --steps 64 --warmup 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'standalone': d['roofline'].get('standalone', {}).get('avg_launch_ms'), 'accept_rate': d['accept_rate']}))" | tee -a "$OUT"
}
run LMC_X=default
run LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_build/s_max-memory-clause/liblmc_hip.so
run LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_build/s_max-ilp/liblmc_hip.so
run LMC_X=default
