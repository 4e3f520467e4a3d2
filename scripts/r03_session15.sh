#!/bin/bash
# A/B: gradient of the cache-filling launches in blocks of B components (LMC_GRAD_BLOCK) -- driver window, steady state, timeline
OUT=gpurun_out/r03_p; mkdir -p $OUT
V="LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_build/$1/liblmc_hip.so"
bash scripts/ab_bench.sh $OUT/ab_$1_driver.jsonl -s 20 -w 5 -- - "$V"
bash scripts/ab_bench.sh $OUT/ab_$1_steady.jsonl -- - "$V"
python scripts/step_timeline.py 26 > $OUT/timeline_default.jsonl 2>&1
env $V python scripts/step_timeline.py 26 > $OUT/timeline_$1.jsonl 2>&1
