#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06_ax
for rep in 1 2; do
for v in "LMC_LARGE_REFILL=0" "LMC_LARGE_REFILL=1" "LMC_LARGE_REFILL=1 LMC_LARGE_PER_LANE=2" "LMC_LARGE_REFILL=1 LMC_LARGE_PER_LANE=5 LMC_LARGE_RETIRE_AT=24"; do
  for cfg in torus12 door door_h2mc; do
    echo -n "{\"variant\": \"$v\", \"run\": " ; env $v timeout 300 python scripts/run_one_config.py $cfg 48 2>/dev/null | tail -1 | tr -d '\n'; echo "}"
  done
done
done | tee gpurun_out/r06_ax/configs.jsonl
