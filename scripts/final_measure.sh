#!/bin/bash
# The round's measurements of record (GPU box): the GPU test tier, bench lines, rocprofv3 kernel stats of the same commands, PMC passes of the
# headline kernel and of the H2MC launches, region profile, step traces.  usage: scripts/final_measure.sh <outdir under gpurun_out>
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.txt" 2>&1; tail -3 "$OUT/pytest_gpu.txt"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>"$OUT/bench_driver.err" | tail -1 > "$OUT/bench_steps20_warmup5.json"
timeout 400 python bench.py 2>"$OUT/bench_default.err" | tail -1 > "$OUT/bench_default.json"
# a complete run: every chain's 256 mutations from a fresh start (cache fill at the beginning, draining chains at the end)
timeout 400 python bench.py --steps 256 --warmup 0 --no-cpu-baseline --no-rmse --no-configs 2>/dev/null | tail -1 > "$OUT/bench_full_run_256_steps.json"
# --gpus N without N devices: refused (exit code 2, message on stderr, no JSON line); the in-process job itself with the bring-up switch
python bench.py --gpus 2 --chains 65536 --steps 8 --warmup 4 > "$OUT/bench_gpus2_refused.out" 2> "$OUT/bench_gpus2_refused.err"; echo "exit code $?" >> "$OUT/bench_gpus2_refused.err"
LMC_BENCH_OVERSUBSCRIBE=1 timeout 400 python bench.py --gpus 2 --in-process --chains 262144 --steps 20 --warmup 30 2>/dev/null | tail -1 > "$OUT/bench_inprocess_2_ranks_one_device.json"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstats && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python "$ROOT/bench.py" --no-cpu-baseline --no-rmse > "$OUT/rocprof_bench.log" 2>&1
  f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" )
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstats2 && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats2 -- python "$ROOT/bench.py" --no-cpu-baseline --no-rmse --steps 20 --warmup 5 > "$OUT/rocprof_bench_driver.log" 2>&1
  f=$(find /tmp/kstats2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_steps20_warmup5.csv" )
timeout 1500 bash scripts/pmc_passes.sh "$OUT/pmc" --no-rmse --no-configs > "$OUT/pmc.log" 2>&1
rm -rf "$OUT"/pmc/pass*/ "$OUT"/pmc/calib_*/   # keep the summaries, drop the raw csv trees
LMC_PROF=1 timeout 300 python scripts/lean_region_profile.py > "$OUT/lean_regions.json" 2>/dev/null
timeout 300 python scripts/step_trace.py 30 > "$OUT/step_trace.jsonl" 2>/dev/null
# the fill phase launch by launch: per-step durations of the first 30 steps of the headline workload and the timeline of step 12
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kfill && cd "$ROOT" && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kfill -- python scripts/run_one_config.py torus6 2 > /dev/null 2>&1
  python scripts/step_durations.py /tmp/kfill 1 30 > "$OUT/step_durations_fill_phase.txt" 2>&1
  python scripts/kernel_trace_summary.py /tmp/kfill 8 12 | grep -A30 ^step >> "$OUT/step_durations_fill_phase.txt" 2>&1 )
# H2MC (BASELINE.json configs[4]): rates, per-kernel counters and the step's timeline on both shipped scenes at 2^20 chains
timeout 600 python scripts/h2mc_rates.py both 20 16 40 > "$OUT/h2mc_rates.jsonl" 2>/dev/null
for sc in door torus; do
  scripts/session.sh "$(basename "$OUT")/h2mc_$sc" h2mc_pmc="$sc 20" > "$OUT/h2mc_pmc_$sc.log" 2>&1
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktr && cd "$ROOT" && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktr -- python scripts/h2mc_rates.py $sc 20 8 24 > /dev/null 2>&1
    python scripts/kernel_trace_summary.py /tmp/ktr 8 > "$OUT/h2mc_timeline_$sc.txt" 2>&1 )
done
ls -la "$OUT"
