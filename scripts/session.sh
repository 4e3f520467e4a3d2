#!/bin/bash
# ONE parameterised GPU session launcher (replaces the per-session one-off scripts of rounds 2 and 3).
# usage (on the GPU box, through gpurun):  scripts/session.sh <tag> <task> [<task> ...]
# Every task writes under gpurun_out/<tag>/.  A task is `name` or `name=argument string`:
#   tests[=pytest args]        pytest -m gpu (default: the whole GPU tier)             -> pytest.txt
#   bench[=bench.py args]      one bench line                                          -> bench.jsonl (appended)
#   ab=<file of variants>      scripts/ab_bench.sh with one variant per line           -> ab.jsonl
#   h2mc[=scene lg steps warm] scripts/h2mc_rates.py                                   -> h2mc.jsonl (appended)
#   h2mc_pmc[=scene lg]        rocprofv3 --pmc passes + kernel stats over h2mc_rates   -> pmc_<n>.json, kernel_stats_<n>.csv
#   pmc_cmd=<python command>   the same over any command (e.g. python scripts/run_one_config.py door)
#   stats=<python command>     rocprofv3 --kernel-trace --stats of any command         -> stats_<n>/
#   pmc=<bench.py args>        scripts/pmc_passes.sh                                   -> pmc/
#   final                      scripts/final_measure.sh                                -> final/
#   sh=<command>               anything else, logged                                   -> sh_<n>.log
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
n=0
for task in "$@"; do
  n=$((n+1))
  name=${task%%=*}; arg=""; [ "$task" != "$name" ] && arg=${task#*=}
  echo "=== [$n] $name $arg" | tee -a "$OUT/session.log"
  t0=$(date +%s)
  case "$name" in
    tests) timeout 2400 python -m pytest tests -m gpu -q -x ${arg} > "$OUT/pytest_$n.txt" 2>&1; tail -5 "$OUT/pytest_$n.txt";;
    bench) timeout 900 python bench.py ${arg} 2> "$OUT/bench_$n.err" | tail -1 | tee -a "$OUT/bench.jsonl";;
    ab) mapfile -t V < "$arg"; scripts/ab_bench.sh "$OUT/ab.jsonl" -- "${V[@]}" 2> "$OUT/ab_$n.err";;
    h2mc) timeout 900 python scripts/h2mc_rates.py ${arg} 2> "$OUT/h2mc_$n.err" | tee -a "$OUT/h2mc.jsonl";;
    h2mc_pmc|pmc_cmd)   # PMC passes + kernel stats over a command (h2mc_pmc=<scene lg>: scripts/h2mc_rates.py; pmc_cmd=<python command>); H2PMC=short: the two SQ groups only; LMC_LIB honoured
      if [ "$name" = h2mc_pmc ]; then CMD="python $REPO/scripts/h2mc_rates.py ${arg:-door 18} 6 24"; else CMD="${arg}"; fi
      mkdir -p "$OUT/pmc_tmp"
      GROUPS_=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
               "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM")
      [ "${H2PMC:-full}" = full ] && GROUPS_+=("FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
               "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_IFETCH SQC_ICACHE_MISSES")
      ( cd /tmp && export TMPDIR=/tmp
        i=0
        for grp in "${GROUPS_[@]}"; do
          i=$((i+1))
          ( cd "$REPO" && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc_tmp/pass$i" -- $CMD > "$OUT/pmc_tmp/pass$i.log" 2>&1 )
        done
        ( cd "$REPO" && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pmc_tmp/stats" -- $CMD > "$OUT/pmc_tmp/stats.log" 2>&1 ) )
      python scripts/pmc_summary.py "$OUT/pmc_tmp" > "$OUT/pmc_$n.json"
      find "$OUT/pmc_tmp/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_$n.csv" \;
      rm -rf "$OUT/pmc_tmp"
      python - "$OUT/pmc_$n.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "h2_" in k or "k_step" in k:
        print(k, {a: (round(b / 1e6, 2) if b > 1e4 else round(b, 2)) for a, b in sorted(v.items())})
PY
      ;;
    stats)
      ( cd /tmp && export TMPDIR=/tmp
        cd "$REPO" && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$n" -- ${arg} > "$OUT/stats_$n.log" 2>&1 )
      find "$OUT/stats_$n" -name "*kernel_stats.csv" -exec cp {} "$OUT/stats_${n}_kernel_stats.csv" \;
      find "$OUT/stats_$n" -name "*.csv" -size +2M -delete; find "$OUT/stats_$n" -name "*.db" -delete
      head -12 "$OUT/stats_${n}_kernel_stats.csv";;
    pmc) scripts/pmc_passes.sh "$OUT/pmc" ${arg} > "$OUT/pmc_$n.log" 2>&1; tail -40 "$OUT/pmc_$n.log"
         find "$OUT/pmc" -name "*.csv" -size +2M -delete; find "$OUT/pmc" -name "*.db" -delete;;
    final) scripts/final_measure.sh "$OUT/final" > "$OUT/final_$n.log" 2>&1; tail -30 "$OUT/final_$n.log";;
    sh) timeout 2400 bash -c "${arg}" > "$OUT/sh_$n.log" 2>&1; tail -30 "$OUT/sh_$n.log";;
    *) echo "unknown task $name" | tee -a "$OUT/session.log";;
  esac
  echo "    ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/session.log"
done
