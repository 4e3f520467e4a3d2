#!/bin/bash
# A/B over ALL bench workloads (headline + the `configs` array of bench.py): one JSON line per variant with every workload's rate and
# kernel split.  Variants as in scripts/ab_bench.sh.   usage: scripts/ab_configs.sh OUT.jsonl -- "-" "LMC_LIB=..." ...
OUT=$1; shift; [ "$1" = "--" ] && shift
run() {
  local v="$1"; [ "$v" = "-" ] && v="LMC_X=default"
  echo "== $v" >&2
  env $v timeout 600 python bench.py --no-cpu-baseline --no-rmse --steps 48 --warmup 40 2>/dev/null | tail -1 | VARIANT="$v" python -c "
import json,os,sys
d=json.loads(sys.stdin.read())
row={'variant': os.environ['VARIANT'], 'headline': {'value': d['value'], 'ms_per_step': d['ms_per_step'], 'k_step_small_ms': d['step_ms']['k_step_small'], 'large_and_generic_ms': d['step_ms']['large_and_generic'], 'standalone_ms': d['roofline'].get('standalone', {}).get('avg_launch_ms')}}
row['configs']=[{'workload': c['workload'][:40], 'value': c.get('value'), 'ms_per_step': c.get('ms_per_step'), 'kernel_ms': c.get('kernel_ms_per_step'), 'accept_rate': c.get('accept_rate'), 'failed': c.get('failed')} for c in d.get('configs', [])]
print(json.dumps(row))" | tee -a "$OUT"
}
first="$1"
for v in "$@"; do run "$v"; done
[ $# -gt 1 ] && run "$first"
