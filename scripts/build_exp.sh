#!/bin/bash
# The build of liblmc_hip.so that carries the work-skipping measurement switches (LMC_EXP_NOSPLAT / NOQUERY / QUERY_STOP / NOGRAD / NOSTATS / NOHESS /
# NOEIGEN / NOHESSLAUNCH, device/dstep_params.h): langevin-mcmc_amd/csrc/_ab/exp/liblmc_hip.so, selected at run time with LMC_LIB=<path>.
# The shipped library has none of them compiled in and refuses to run while one is set.  scripts/pmc_ab.sh / ab_bench.sh variants that use a switch
# name this library:  "LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_ab/exp/liblmc_hip.so LMC_EXP_NOQUERY=1"
set -e
D=langevin-mcmc_amd/csrc/_ab/exp
mkdir -p $D
make -s -f langevin-mcmc_amd/csrc/Makefile -j8 OBJ=$D OUT=$D/liblmc_hip.so CLI=$D/dpt_amd EXTRA="-DLMC_EXP_SWITCHES" $D/liblmc_hip.so
rm -f $D/*.o $D/*.d
ls -la $D/liblmc_hip.so
