#!/bin/bash
OUT=gpurun_out/r03_f; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |^FAILED|passed|failed|Error" | head -60 > $OUT/gputests.txt
timeout 400 python bench.py --no-cpu-baseline --no-rmse 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rmse --no-configs 2>>$OUT/bench.err | tail -1 > $OUT/bench_20_5.json
