#!/bin/bash
# Per-dispatch durations of the step kernels over a fresh start (rocprofv3 --kernel-trace of scripts/step_trace.py).
# usage: scripts/kernel_trace_steps.sh <out.txt> [steps]   (GPU)
OUT=$(realpath -m "$1"); STEPS=${2:-26}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python "$ROOT/scripts/step_trace.py" "$STEPS" > /dev/null 2>/tmp/kt.err
python - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[1], "w") as o:
    t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
    for r in rows:
        n = r["Kernel_Name"]
        if "k_step" not in n and "k_build_lists" not in n: continue
        short = n.split("(")[0][:70]
        o.write("%10.3f ms  +%8.3f ms  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, short))
PY
