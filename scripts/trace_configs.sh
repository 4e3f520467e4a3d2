#!/bin/bash
# Kernel timeline of bench workloads at steady state: per-step durations and the launches of one step.
# usage (GPU box): scripts/trace_configs.sh OUTDIR config [config ...]     (configs of scripts/run_one_config.py)
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
case "$OUT" in /*) ;; *) OUT="$REPO/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  ( cd "$REPO" && rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_$w" -- python scripts/run_one_config.py $w 6 > /dev/null 2>&1 )
  echo "== $w"
  python "$REPO/scripts/step_durations.py" "$OUT/tr_$w" 40 46
  python "$REPO/scripts/kernel_trace_summary.py" "$OUT/tr_$w" 6 44 | grep -A30 ^step
  rm -rf "$OUT/tr_$w"
done
