#!/bin/bash
# A/B of launch-side choices of the chain step on one GPU box: one JSON line per variant.
# usage: scripts/ab_bench.sh OUT.jsonl [steps warmup]   (default: steady state, --steps 64 --warmup 40)
OUT=$1; : > "$OUT"
STEPS=${2:-64}; WARM=${3:-40}
run() {
  echo "== $*" >&2
  env "$@" timeout 240 python bench.py --no-cpu-baseline --steps $STEPS --warmup $WARM 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'variant': '$*', 'steps': $STEPS, 'warmup': $WARM, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'large_ms': d['step_ms']['large_and_generic'], 'roofline_frac': d['roofline']['frac'], 'accept_rate': d['accept_rate']}))" | tee -a "$OUT"
}
run LMC_BVH=sah LMC_SORT_PLAIN=0 LMC_LEAN_BLOCK=64 LMC_OVERLAP=0
run LMC_BVH=sah LMC_SORT_PLAIN=0 LMC_LEAN_BLOCK=64 LMC_OVERLAP=1
run LMC_BVH=sah LMC_SORT_PLAIN=1 LMC_LEAN_BLOCK=64 LMC_OVERLAP=1
run LMC_BVH=sah LMC_SORT_PLAIN=0 LMC_LEAN_BLOCK=128 LMC_OVERLAP=1
run LMC_BVH=sah LMC_SORT_PLAIN=0 LMC_LEAN_BLOCK=256 LMC_OVERLAP=1
run LMC_BVH=lbvh LMC_SORT_PLAIN=0 LMC_LEAN_BLOCK=64 LMC_OVERLAP=1
