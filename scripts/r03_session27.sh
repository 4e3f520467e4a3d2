#!/bin/bash
OUT=gpurun_out/r03_ad; mkdir -p $OUT
bash scripts/ab_bench.sh $OUT/ab_griddims.jsonl -- - "LMC_GRID_DIMS=3"
