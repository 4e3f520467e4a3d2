#!/bin/bash
# HBM-traffic A/B of the step kernels: for every variant (a list of environment assignments, "-" = the tree's defaults; LMC_LIB=<path> selects
# another build) two rocprofv3 --pmc passes over a short steady-state bench run (FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; never combined
# with trace domains other than --kernel-trace), condensed per kernel by scripts/pmc_summary.py.  One JSON line per variant: the lean kernel's
# bytes read (FETCH_SIZE KB x 1024 x 2: the gfx950 correction of MI355X_MICROARCH.md confirmed by lmc_stream_probe) / written per launch and per chain-step.
# usage (GPU box): scripts/pmc_ab.sh OUT.jsonl [-g "extra counter group"] -- "VAR=1" "-" ...
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export LMC_BENCH_ALLOW_EXP=1  # variants may carry work-skipping LMC_EXP_* switches: every line names its variant
GROUPS_=("FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum")
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  case "$1" in
    -g) GROUPS_+=("$2"); shift 2;;
    *) echo "unknown option $1" >&2; exit 2;;
  esac
done
shift
TMP=$(mktemp -d /tmp/pmc_ab.XXXXXX)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  vv="$v"; [ "$v" = "-" ] && vv="LMC_X=default"
  rm -rf "$TMP/run"; mkdir -p "$TMP/run"
  i=0
  for grp in "${GROUPS_[@]}"; do
    i=$((i+1))
    ( cd "$REPO" && env $vv timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$TMP/run/pass$i" -- python bench.py --no-cpu-baseline --no-rmse --no-configs --steps 32 --warmup 40 > "$TMP/run/pass$i.log" 2>&1 )
  done
  python "$REPO/scripts/pmc_summary.py" "$TMP/run" > "$TMP/run/summary.json"
  VARIANT="$vv" python - "$TMP/run/summary.json" "$TMP/run/pass1.log" <<'PY' | tee -a "$OUT"
import json, os, sys
d = json.load(open(sys.argv[1]))
steps = None
try:
    line = [l for l in open(sys.argv[2]) if l.startswith('{"metric"')][-1]
    b = json.loads(line)
    steps = b["roofline"]["chain_steps_per_launch"] if b["roofline"]["kernel"].startswith("k_step_small") else None
    rate = b["value"]
    extra = {k: b.get(k) for k in ("accept_rate", "cache_queries_per_step", "cache_hits_per_query")}
except Exception:
    rate = None
out = {"variant": os.environ["VARIANT"], "value_under_pmc": rate, "run": extra if rate else None, "lean_chain_steps_per_launch": steps, "kernels": {}}
for k, v in d.items():
    if not ("k_step" in k or "k_h2" in k or "k_mala" in k or "k_reloc" in k):
        continue
    rd = v.get("FETCH_SIZE", 0) * 1024 * 2
    wr = v.get("WRITE_SIZE", 0) * 1024
    e = {"launches": v.get("launches"), "read_MB": rd / 1e6, "written_MB": wr / 1e6, "tcc_hit_rate": (v.get("TCC_HIT_sum", 0) / max(v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0), 1))}
    for n, x in v.items():
        if n not in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "launches"):
            e[n] = x
    if "k_step_small<" in k and steps:
        e["read_B_per_chain_step"] = rd / steps
        e["written_B_per_chain_step"] = wr / steps
    out["kernels"][k] = e
print(json.dumps(out))
PY
done
rm -rf "$TMP"
