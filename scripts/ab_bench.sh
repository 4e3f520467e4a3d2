#!/bin/bash
# A/B of build- or launch-side choices of the chain step on one GPU box: one JSON line per variant, appended to OUT.jsonl.
# A variant is a quoted list of environment assignments (LMC_* switches of host/context.cpp; LMC_LIB=<path> selects another
# build of liblmc_hip.so); "-" runs the tree's defaults.  The first variant is repeated at the end (box drift).
# usage: scripts/ab_bench.sh OUT.jsonl [-s STEPS] [-w WARMUP] [-b "extra bench.py args"] -- "VAR=1 VAR2=x" "VAR=2" ...
#   e.g. scripts/ab_bench.sh gpurun_out/ab.jsonl -- - "LMC_SORT_PLAIN=1" "LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_build/alt/liblmc_hip.so"
OUT=$1; shift
export LMC_BENCH_ALLOW_EXP=1  # variants may carry work-skipping LMC_EXP_* switches: every line names its variant
STEPS=64; WARM=40; EXTRA=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  case "$1" in
    -s) STEPS=$2; shift 2;;
    -w) WARM=$2; shift 2;;
    -b) EXTRA=$2; shift 2;;
    *) echo "unknown option $1" >&2; exit 2;;
  esac
done
shift
run() {
  local v="$1"
  [ "$v" = "-" ] && v="LMC_X=default"
  echo "== $v" >&2
  env $v timeout 300 python bench.py --no-cpu-baseline --no-rmse --no-configs --steps $STEPS --warmup $WARM $EXTRA 2>/dev/null | tail -1 | VARIANT="$v" python -c "
import json,os,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'variant': os.environ['VARIANT'], 'steps': d['steps'], 'warmup': d['warmup'], 'value': d['value'], 'value_from_step_counter': d.get('value_from_step_counter'), 'ms_per_step': d['ms_per_step'],
  'k_step_small_ms': d['step_ms']['k_step_small'], 'large_and_generic_ms': d['step_ms']['large_and_generic'], 'frac': r['frac'],
  'standalone_ms': r.get('standalone', {}).get('avg_launch_ms'), 'standalone_frac': r.get('standalone', {}).get('frac'), 'accept_rate': d['accept_rate']}))" | tee -a "$OUT"
}
first="$1"
for v in "$@"; do run "$v"; done
[ $# -gt 1 ] && run "$first"
