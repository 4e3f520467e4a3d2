// TEST HELPER: the deterministic float exp / log / pow of the product (langevin-mcmc_amd/csrc/device/dtrans.h) compiled by the host
// compiler with the product's arithmetic contract (-ffp-contract=off), for the accuracy test and as the CPU side of the GPU
// bit-equality test.
#include "../../langevin-mcmc_amd/csrc/device/dtrans.h"
extern "C" void lmc_test_trans_host(int n, int mode, const float *x, const float *y, float *o) {
    for (int i = 0; i < n; i++) o[i] = mode == 0 ? lmcd::lexpf(x[i]) : mode == 1 ? lmcd::llogf(x[i]) : lmcd::lpowf(x[i], y[i]);
}
