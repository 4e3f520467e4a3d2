#!/bin/bash
# closest-hit kernel alone (k_trace on 2^20 incoherent rays, scripts/trace_tcc_probe.py) under rocprofv3 --kernel-trace --stats: rays per second
# of the traversal by build.  usage (GPU box): scripts/trace_rate.sh OUT.jsonl [scene.xml] -- "-" "LMC_LIB=<path>" ...
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
SCENE=$REPO/scenes/torus/lmc.xml
if [ "$1" != "--" ]; then SCENE=$(realpath "$1"); shift; fi
shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  vv="$v"; [ "$v" = "-" ] && vv="LMC_X=default"
  D=$(mktemp -d /tmp/trace_rate.XXXXXX)
  ( cd "$REPO" && env $vv rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python scripts/trace_tcc_probe.py "$SCENE" 20 6 > "$D/log" 2>&1 )
  VARIANT="$vv" SCENE="$SCENE" python - "$D" <<'PY' | tee -a "$OUT"
import csv, glob, json, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Name"]:
            ns = float(r["AverageNs"])
            print(json.dumps({"variant": os.environ["VARIANT"], "scene": os.path.basename(os.path.dirname(os.environ["SCENE"])), "k_trace_avg_ms": ns / 1e6, "calls": int(r["Calls"]), "G_rays_per_s": (1 << 20) / ns}))
PY
  rm -rf "$D"
done
