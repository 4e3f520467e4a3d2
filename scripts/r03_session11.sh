#!/bin/bash
OUT=gpurun_out/r03_k; mkdir -p $OUT
bash scripts/ab_bench.sh $OUT/ab_sort_block.jsonl -- - "LMC_LEAN_BLOCK=256" "LMC_SORT_PLAIN=2 LMC_LEAN_BLOCK=256" "LMC_SORT_PLAIN=1 LMC_LEAN_BLOCK=256" "LMC_SORT_PLAIN=2 LMC_LEAN_BLOCK=128"
