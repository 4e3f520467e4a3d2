#!/bin/bash
# the fill phase launch by launch on the current tree: per-step durations of the first 30 steps and the timelines of steps 5 and 22 (the two cache-ready transitions)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kfill && cd "$ROOT" && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kfill -- python scripts/run_one_config.py torus6 2 > /dev/null 2>&1
python scripts/step_durations.py /tmp/kfill 1 30
for st in 4 5 22; do python scripts/kernel_trace_summary.py /tmp/kfill 8 $st | grep -A60 ^step; done
