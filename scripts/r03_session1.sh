#!/bin/bash
# GPU session 1 of round 3: full GPU test tier with the new parity bars, bench lines, and the two cheap experiments the verdict asked
# for (instruction counts of sorted vs unsorted lean launches; the scene's own L2 hit rate).
OUT=gpurun_out/r03_a; mkdir -p $OUT
ROOT=$(pwd)
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/gputests.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/bench_driver.err | tail -1 > $OUT/bench_steps20_warmup5.json
timeout 300 python bench.py --no-cpu-baseline --no-rmse 2>$OUT/bench_default.err | tail -1 > $OUT/bench_default.json
cd /tmp && export TMPDIR=/tmp
for sortv in 0 1 2; do
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
             "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    LMC_SORT_PLAIN=$sortv timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/$OUT/sort$sortv/pass$i -- python $ROOT/bench.py --no-cpu-baseline --no-rmse --steps 24 --warmup 40 > $ROOT/$OUT/sort${sortv}_pass$i.log 2>&1
  done
  python $ROOT/scripts/pmc_summary.py $ROOT/$OUT/sort$sortv > $ROOT/$OUT/sort${sortv}_pmc_summary.json
  rm -rf $ROOT/$OUT/sort$sortv
done
for scene in torus veachdoor; do
  i=0
  for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/$OUT/tcc_$scene/pass$i -- python $ROOT/scripts/trace_tcc_probe.py $ROOT/scenes/$scene/lmc.xml 20 6 > $ROOT/$OUT/tcc_${scene}_pass$i.log 2>&1
  done
  python $ROOT/scripts/pmc_summary.py $ROOT/$OUT/tcc_$scene > $ROOT/$OUT/tcc_${scene}_pmc_summary.json
  rm -rf $ROOT/$OUT/tcc_$scene
done
cd $ROOT
ls -la $OUT
