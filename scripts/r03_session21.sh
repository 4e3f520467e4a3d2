#!/bin/bash
OUT=gpurun_out/r03_w; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "large_step_cache" 2>&1 | tail -6 > $OUT/pytest.txt
