#!/bin/bash
# GPU session 5: what bounds the large-step launch on the door scene + the cfg-1 twin tests
OUT=gpurun_out/r03_e; mkdir -p $OUT
ROOT=$(pwd)
timeout 600 python -m pytest "tests/test_gpu_parity.py::test_cfg1_twin_four_chains_thousand_mutations" tests/test_gpu_parity.py::test_group_of_ranks_equals_one_rank_through_the_cache_phase -q 2>&1 | grep -E "^E  |^FAILED|passed|failed|Error" | head -30 > $OUT/tests.txt
cd /tmp && export TMPDIR=/tmp
for cfgname in door torus12; do
  rm -rf /tmp/ks_$cfgname
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$cfgname -- python $ROOT/scripts/run_one_config.py $cfgname 24 > $ROOT/$OUT/${cfgname}_stats.log 2>&1
  f=$(find /tmp/ks_$cfgname -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $ROOT/$OUT/${cfgname}_kernel_stats.csv
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT" \
             "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$cfgname/pass$i -- python $ROOT/scripts/run_one_config.py $cfgname 24 > $ROOT/$OUT/pmc_${cfgname}_pass$i.log 2>&1
  done
  python $ROOT/scripts/pmc_summary.py $ROOT/$OUT/pmc_$cfgname > $ROOT/$OUT/pmc_${cfgname}_summary.json
  rm -rf $ROOT/$OUT/pmc_$cfgname
done
