#!/bin/bash
# rocprofv3 PMC passes over a short bench run (one counter group per run; never combined with trace domains other
# than --kernel-trace).  Usage (on the GPU box): scripts/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQ_INSTS_BRANCH" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -- python "$REPO/bench.py" --no-cpu-baseline --steps 32 --warmup 40 "$@" > "$OUT/pass$i.log" 2>&1
done
python "$REPO/scripts/pmc_summary.py" "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
# counter calibration on a known byte count (1 GiB read + 1 GiB written per launch, dword-per-lane pattern)
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/calib_$grp" -- python -c "
import importlib,sys
sys.path.insert(0,'$REPO')
p=importlib.import_module('langevin-mcmc_amd')
assert p.lib().lmc_stream_probe(1<<28, 4)==0" > "$OUT/calib_$grp.log" 2>&1
done
python "$REPO/scripts/pmc_summary.py" "$OUT" calib > "$OUT/calib.json"
cat "$OUT/calib.json"
