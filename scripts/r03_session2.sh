#!/bin/bash
# GPU session 2 of round 3: the whole GPU test tier, the LDS-staged BVH top A/B, the large-step occupancy A/B.
OUT=gpurun_out/r03_b; mkdir -p $OUT
B=$PWD/langevin-mcmc_amd/csrc/_build
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/gputests.txt
timeout 1500 bash scripts/ab_bench.sh $OUT/ab_bvh_lds_top.jsonl -- - "LMC_LIB=$B/ldstop21/liblmc_hip.so" "LMC_LEAN_BLOCK=256" "LMC_LEAN_BLOCK=256 LMC_LIB=$B/ldstop85/liblmc_hip.so" "LMC_LEAN_BLOCK=256 LMC_LIB=$B/ldstop21/liblmc_hip.so" 2> $OUT/ab_bvh_lds_top.err
timeout 1500 bash scripts/ab_configs.sh $OUT/ab_large_step_waves.jsonl -- - "LMC_LIB=$B/large2/liblmc_hip.so" "LMC_LIB=$B/large3/liblmc_hip.so" 2> $OUT/ab_large_step_waves.err
ls -la $OUT
