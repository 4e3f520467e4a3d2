#!/bin/bash
# A/B: when the H2MC step's large-step launch starts relative to the pipeline's head (LMC_H2_LARGE_AFTER: 0 at once, 1 behind k_h2_begin, 2 behind the first Hessian launches)
for rep in 1 2; do
  for v in 0 1 2; do
    echo "== LMC_H2_LARGE_AFTER=$v"
    LMC_H2_LARGE_AFTER=$v python scripts/h2mc_rates.py both 20 24 8
  done
done
