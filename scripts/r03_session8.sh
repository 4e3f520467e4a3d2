#!/bin/bash
OUT=gpurun_out/r03_h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_door.py -q -k "light_coordinate" 2>&1 | grep -E "^E  |^FAILED|passed|failed|Error" | head -30 > $OUT/lightcoord.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |^FAILED|passed|failed|Error" | head -40 > $OUT/gputests.txt
