#!/bin/bash
# lean kernel: 2 vs 3 waves per SIMD (register allocation target), same LDS layout (56 words per thread)
OUT=$1; : > "$OUT"
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --no-rmse --steps 64 --warmup 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'roofline_frac': d['roofline']['frac'], 'accept_rate': d['accept_rate']}))" | tee -a "$OUT"
}
run LMC_X=waves2
run LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_build/w3/liblmc_hip.so
run LMC_X=waves2 LMC_OVERLAP=0
run LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_build/w3/liblmc_hip.so LMC_OVERLAP=0
