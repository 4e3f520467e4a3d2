// TEST HELPER (not part of the product library): instantiates the product's generic path program
// (langevin-mcmc_amd/csrc/device/pathfunc.h) with a plain host compiler so that `-m "not gpu"` tests can
// check its arithmetic against the reference's generated programs without a GPU.  The shipped C-ABI never
// calls this; it always launches the HIP kernels.
#include "../../langevin-mcmc_amd/csrc/device/pathfunc.h"

extern "C" void lmc_test_pathfunc_host(int c, int l, const float *primary, const float *scene, const float *vert, float *logLum, float *grad) {
    lmcd::ContigIn in{vert};
    if (logLum) *logLum = lmcd::PathFuncValue(c, l, primary, scene, in);
    if (grad) {
        float ll;
        lmcd::PathFuncGrad(c, l, primary, scene, in, &ll, grad);
    }
}
