#!/bin/bash
OUT=gpurun_out/r03_p; mkdir -p $OUT
B=$PWD/langevin-mcmc_amd/csrc/_build
bash scripts/ab_bench.sh $OUT/ab_gradblk_all_driver2.jsonl -s 20 -w 5 -- - "LMC_LIB=$B/gradblk1/liblmc_hip.so" "LMC_LIB=$B/gradblk2/liblmc_hip.so" "LMC_LIB=$B/gradblk3/liblmc_hip.so" "LMC_LIB=$B/gradblk4/liblmc_hip.so"
bash scripts/ab_bench.sh $OUT/ab_gradblk_all_full.jsonl -s 256 -w 0 -- - "LMC_LIB=$B/gradblk1/liblmc_hip.so" "LMC_LIB=$B/gradblk2/liblmc_hip.so" "LMC_LIB=$B/gradblk4/liblmc_hip.so"
