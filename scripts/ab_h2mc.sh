#!/bin/bash
# A/B of builds on the H2MC workloads: scripts/ab_h2mc.sh OUT.jsonl "<h2mc_rates args>" <variant dir names under csrc/_ab, or 'tree'> ...
OUT=$1; ARGS=$2; shift 2
for v in "$@" "$1"; do
  if [ "$v" = tree ]; then env -u LMC_LIB python scripts/h2mc_rates.py $ARGS | tee -a "$OUT"
  else LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_ab/$v/liblmc_hip.so python scripts/h2mc_rates.py $ARGS | tee -a "$OUT"; fi
done
