#!/bin/bash
OUT=gpurun_out/r03_g; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -s -k "plugin" 2>&1 | grep -E "^E  |^FAILED|passed|failed|Error|plugin gradient" | head -30 > $OUT/plugin.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |^FAILED|passed|failed|Error" | head -40 > $OUT/gputests.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rmse 2>$OUT/bench.err | tail -1 > $OUT/bench_20_5.json
