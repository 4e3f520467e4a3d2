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

extern "C" void lmc_test_pathfunc_hess_host(int c, int l, const float *primary, const float *scene, const float *vert, float *logLum, float *grad, float *hess) {
    lmcd::ContigIn in{vert};
    lmcd::PathFuncHess(c, l, primary, scene, in, logLum, grad, hess);
}

// The reference's plugin symbol names over the same host instantiation: lets the CPU oracle (test infrastructure) take
// its gradients from the product's path program instead of the reference's generated code.
#define LMC_HOST_PLUGIN(C, Lg)                                                                                                                      \
    extern "C" void evaluate_path_bidir_mala_##C##_##Lg##_static(const float *, const float *primary, const float *scene, const float *vp, float *ll) { \
        lmc_test_pathfunc_host(C, Lg, primary, scene, vp, ll, nullptr);                                                                             \
    }                                                                                                                                               \
    extern "C" void evaluate_path_bidir_mala_##C##_##Lg##_static_derv(const float *, const float *primary, const float *scene, const float *vp, float *g) { \
        lmc_test_pathfunc_host(C, Lg, primary, scene, vp, nullptr, g);                                                                              \
    }                                                                                                                                               \
    extern "C" void evaluate_path_bidir_##C##_##Lg##_static_derv(const float *, const float *primary, const float *scene, const float *vp, float *g, float *h) { \
        float ll;                                                                                                                                   \
        lmc_test_pathfunc_hess_host(C, Lg, primary, scene, vp, &ll, g, h);                                                                          \
    }
LMC_HOST_PLUGIN(1, 2) LMC_HOST_PLUGIN(1, 3) LMC_HOST_PLUGIN(1, 4) LMC_HOST_PLUGIN(1, 5) LMC_HOST_PLUGIN(1, 6) LMC_HOST_PLUGIN(1, 7) LMC_HOST_PLUGIN(1, 8)
LMC_HOST_PLUGIN(2, 1) LMC_HOST_PLUGIN(2, 2) LMC_HOST_PLUGIN(2, 3) LMC_HOST_PLUGIN(2, 4) LMC_HOST_PLUGIN(2, 5) LMC_HOST_PLUGIN(2, 6) LMC_HOST_PLUGIN(2, 7)
LMC_HOST_PLUGIN(3, 0) LMC_HOST_PLUGIN(3, 1) LMC_HOST_PLUGIN(3, 2) LMC_HOST_PLUGIN(3, 3) LMC_HOST_PLUGIN(3, 4) LMC_HOST_PLUGIN(3, 5) LMC_HOST_PLUGIN(3, 6)
LMC_HOST_PLUGIN(4, 0) LMC_HOST_PLUGIN(4, 1) LMC_HOST_PLUGIN(4, 2) LMC_HOST_PLUGIN(4, 3) LMC_HOST_PLUGIN(4, 4) LMC_HOST_PLUGIN(4, 5)
LMC_HOST_PLUGIN(5, 0) LMC_HOST_PLUGIN(5, 1) LMC_HOST_PLUGIN(5, 2) LMC_HOST_PLUGIN(5, 3) LMC_HOST_PLUGIN(5, 4)
LMC_HOST_PLUGIN(6, 0) LMC_HOST_PLUGIN(6, 1) LMC_HOST_PLUGIN(6, 2) LMC_HOST_PLUGIN(6, 3)
LMC_HOST_PLUGIN(7, 0) LMC_HOST_PLUGIN(7, 1) LMC_HOST_PLUGIN(7, 2)
LMC_HOST_PLUGIN(8, 0) LMC_HOST_PLUGIN(8, 1)
LMC_HOST_PLUGIN(9, 0)
