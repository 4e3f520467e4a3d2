#!/bin/bash
# A/B: warm-up launches of the step kernels at the end of MLTInit (LMC_NO_WARM_LAUNCH=1 = before)
OUT=gpurun_out/r03_x; mkdir -p $OUT
bash scripts/ab_bench.sh $OUT/ab_warm_full.jsonl -s 256 -w 0 -- - "LMC_NO_WARM_LAUNCH=1"
python scripts/step_timeline.py 4 > $OUT/timeline_warm.jsonl 2>&1
LMC_NO_WARM_LAUNCH=1 python scripts/step_timeline.py 4 > $OUT/timeline_nowarm.jsonl 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "chain_loop_parity or ragged or group_of_ranks" 2>&1 | tail -3 > $OUT/pytest.txt
