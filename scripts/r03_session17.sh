#!/bin/bash
OUT=gpurun_out/r03_q; mkdir -p $OUT
timeout 3000 python -m pytest tests/ -q -m gpu 2>&1 | tail -15 > $OUT/pytest_gpu.txt
