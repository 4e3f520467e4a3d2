#!/bin/bash
OUT=gpurun_out/r03_z; mkdir -p $OUT
B=$PWD/langevin-mcmc_amd/csrc/_build/before_q
timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -6 > $OUT/pytest_gpu.txt
bash scripts/ab_bench.sh $OUT/ab_queue_steady.jsonl -- - "LMC_LIB=$B/liblmc_hip.so"
