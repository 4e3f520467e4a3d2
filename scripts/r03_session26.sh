#!/bin/bash
OUT=gpurun_out/r03_aa; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "without_light_subpaths" 2>&1 | grep -E "^E  |passed|failed" | head -20 > $OUT/pytest_fallback.txt
