#!/bin/bash
OUT=gpurun_out/r03_y; mkdir -p $OUT
B=$PWD/langevin-mcmc_amd/csrc/_build/before_iso
timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -6 > $OUT/pytest_gpu3.txt
bash scripts/ab_bench.sh $OUT/ab_dirty_steady.jsonl -- - "LMC_LIB=$B/liblmc_hip_pss.so"
bash scripts/ab_bench.sh $OUT/ab_dirty_driver.jsonl -s 20 -w 5 -- - "LMC_LIB=$B/liblmc_hip_pss.so" "LMC_LIB=$B/liblmc_hip.so"
