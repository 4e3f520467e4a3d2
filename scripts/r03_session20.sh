#!/bin/bash
# A/B: rows x columns per Hessian pass of the H2MC step (DualS<R, Dual<W>>), upper-triangle blocks
OUT=gpurun_out/r03_v; mkdir -p $OUT
B=$PWD/langevin-mcmc_amd/csrc/_build
for v in r2w2 r1w2 r2w3 r3w2 r4w2; do LMC_LIB=$B/$v/liblmc_hip.so timeout 300 python scripts/h2mc_rates.py > $OUT/b_$v.txt 2>&1; done
