#!/bin/bash
# A/B build of liblmc_hip.so that differs from the tree's build in ONE translation unit's flags: copies the objects of the tree's build
# and recompiles only <tu>.  usage: scripts/build_variant.sh <name> <tu> "<flags>"  ->  langevin-mcmc_amd/csrc/_ab/<name>/liblmc_hip.so
# (selected at run time with LMC_LIB=<path>; _ab/ travels to the GPU box, unlike _build/)
set -e
NAME=$1; TU=$2; FLAGS=$3
D=langevin-mcmc_amd/csrc/_ab/$NAME
mkdir -p $D
cp -u langevin-mcmc_amd/csrc/_build/*.o langevin-mcmc_amd/csrc/_build/*.d $D/
rm -f $D/$TU.o
make -s -f langevin-mcmc_amd/csrc/Makefile OBJ=$D OUT=$D/liblmc_hip.so CLI=$D/dpt_amd EXTRA_$TU="$FLAGS" $D/liblmc_hip.so
rm -f $D/*.o $D/*.d   # the snapshot that travels to the GPU box is capped at 512 MiB: only the library stays
ls -la $D/liblmc_hip.so
