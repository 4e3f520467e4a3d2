#!/bin/bash
OUT=gpurun_out/r03_t; mkdir -p $OUT
bash scripts/ab_bench.sh $OUT/ab_stream_prio_driver.jsonl -s 20 -w 5 -- - "LMC_STREAM_PRIO=0" "LMC_STREAM_PRIO=-1" "LMC_LARGE_BLOCK=128" "LMC_SORT_GENERIC=0"
