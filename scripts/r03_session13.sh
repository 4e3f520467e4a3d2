#!/bin/bash
OUT=gpurun_out/r03_n; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "group_of_ranks" 2>&1 | grep -E "^E  |^FAILED|passed|failed|Error" | head -40 > $OUT/cache.txt
