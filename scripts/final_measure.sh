#!/bin/bash
# The round's measurements of record (GPU box): bench lines, rocprofv3 kernel stats of the same command, PMC passes.
# usage: scripts/final_measure.sh <outdir under gpurun_out> ; then copy with scripts/collect_profiles.sh
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>"$OUT/bench_driver.err" | tail -1 > "$OUT/bench_steps20_warmup5.json"
timeout 400 python bench.py 2>"$OUT/bench_default.err" | tail -1 > "$OUT/bench_default.json"
# a complete run: every chain's 256 mutations from a fresh start (cache fill at the beginning, draining chains at the end)
timeout 400 python bench.py --steps 256 --warmup 0 --no-cpu-baseline --no-rmse 2>/dev/null | tail -1 > "$OUT/bench_full_run_256_steps.json"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstats && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python "$ROOT/bench.py" --no-cpu-baseline --no-rmse > "$OUT/rocprof_bench.log" 2>&1
  f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" )
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstats2 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats2 -- python "$ROOT/bench.py" --no-cpu-baseline --no-rmse --steps 20 --warmup 5 > "$OUT/rocprof_bench_driver.log" 2>&1
  f=$(find /tmp/kstats2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_steps20_warmup5.csv" )
timeout 1500 bash scripts/pmc_passes.sh "$OUT/pmc" --no-rmse > "$OUT/pmc.log" 2>&1
rm -rf "$OUT"/pmc/pass*/ "$OUT"/pmc/calib_*/   # keep the summaries, drop the raw csv trees
LMC_PROF=1 timeout 300 python scripts/lean_region_profile.py > "$OUT/lean_regions.json" 2>/dev/null
timeout 300 python scripts/step_trace.py 30 > "$OUT/step_trace.jsonl" 2>/dev/null
ls -la "$OUT"
