#!/bin/bash
# Region profile of the lean kernel (LMC_PROF=1) for the tree's build and variant builds: scripts/ab_regions.sh OUT.jsonl [full] -- tree <variant> ...
OUT=$1; shift
MODE=""; [ "$1" = full ] && { MODE=full; shift; }
[ "$1" = "--" ] && shift
for v in "$@"; do
  if [ "$v" = tree ]; then R=$(env -u LMC_LIB LMC_PROF=1 python scripts/lean_region_profile.py $MODE | tr -d '\n')
  else R=$(LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_ab/$v/liblmc_hip.so LMC_PROF=1 python scripts/lean_region_profile.py $MODE | tr -d '\n'); fi
  echo "{\"variant\": \"$v\", \"mode\": \"${MODE:-diffuse}\", \"profile\": $R}" | tee -a "$OUT"
done
