#!/bin/bash
OUT=gpurun_out/r03_u; mkdir -p $OUT
python scripts/debug/h2mc_pair.py > $OUT/pair_lds.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_h2mc.py -q 2>&1 | tail -5 > $OUT/pytest_h2mc.txt
timeout 600 python scripts/h2mc_rates.py > $OUT/h2mc_rates_lds.txt 2>&1
