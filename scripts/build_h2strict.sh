#!/bin/bash
# liblmc_hip.so with the H2MC step's Hessian program and eigen-solve in STRICT arithmetic (no fused multiply-add, correctly rounded division and
# square root, the deterministic float transcendentals of dtrans.h instead of the hardware's approximate ones): the round-3 arithmetic, kept as a
# variant so that a GPU test can show that what the shipped (fast) build's H2MC chains diverge by from the oracle is arithmetic only
# (tests/test_gpu_h2mc.py::test_h2mc_chain_parity_on_the_strict_build).  Only h2hess.o and h2gauss.o differ from the tree's build.
# -> langevin-mcmc_amd/csrc/_ab/h2strict/liblmc_hip.so (selected with LMC_LIB; travels to the GPU box with the snapshot)
set -e
D=langevin-mcmc_amd/csrc/_ab/h2strict
mkdir -p $D
cp -u langevin-mcmc_amd/csrc/_build/*.o langevin-mcmc_amd/csrc/_build/*.d $D/
# stale if the sources of the two units (or anything they include) are newer than the variant's objects: make's own dependency files decide
rm -f $D/h2hess.o $D/h2gauss.o
make -s -f langevin-mcmc_amd/csrc/Makefile OBJ=$D OUT=$D/liblmc_hip.so CLI=$D/dpt_amd H2HESS_FLAGS= H2GAUSS_FLAGS= EXTRA_h2hess=-DLMC_H2HESS_EXACT_MATH $D/liblmc_hip.so
rm -f $D/*.o $D/*.d
ls -la $D/liblmc_hip.so
