#!/bin/bash
# A/B: lean launch specialised for scenes without light sub-paths (environment map only): LMC_LEAN_LIGHTLESS=0 = the general instantiation
OUT=gpurun_out/r03_aa; mkdir -p $OUT
timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -6 > $OUT/pytest_gpu.txt
bash scripts/ab_bench.sh $OUT/ab_lightless_steady.jsonl -- - "LMC_LEAN_LIGHTLESS=0"
bash scripts/ab_bench.sh $OUT/ab_lightless_driver.jsonl -s 20 -w 5 -- - "LMC_LEAN_LIGHTLESS=0"
