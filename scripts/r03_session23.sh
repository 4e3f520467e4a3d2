#!/bin/bash
# A/B: the isotropic Gaussian of a state is a flag, not 3 dim + 1 stored words (F_GAUSS_ISO); before = LMC_LIB=_build/before_iso
OUT=gpurun_out/r03_y; mkdir -p $OUT
V="LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_build/before_iso/liblmc_hip.so"
timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -6 > $OUT/pytest_gpu.txt
bash scripts/ab_bench.sh $OUT/ab_iso_steady.jsonl -- - "$V"
bash scripts/ab_bench.sh $OUT/ab_iso_driver.jsonl -s 20 -w 5 -- - "$V"
