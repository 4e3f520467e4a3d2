// Out-of-line device copies of the second-order path program (pathfunc.h PathFuncHessPass) for the two per-lane forms of the H2MC
// library symbols: lmc_hess_batch (one lane per item) and the single-call plugin kernel (one lane per pass), step_small_h2mc.hip.
// The H2MC step itself evaluates the program wave-cooperatively (h2hess.hip) and no longer uses these.
#pragma once
#include "dh2mc.h"
#include "dstep.h"

namespace lmcd {

#ifdef __HIPCC__
#ifndef LMC_PF_ATTR
#define LMC_PF_ATTR
#endif
template <class In>
__device__ __noinline__ LMC_PF_ATTR void PathFuncHessPassDevice(int c, int l, const float *primary, const float *scene, const In &vp, int i, int c0, float *logLum,
                                                                float *grad, float *hess, bool firstOfRow) {
    PathFuncHessPass(c, l, primary, scene, vp, i, c0, logLum, grad, hess, firstOfRow);
}
// all passes: ONE copy of the second-order program per kernel image (the single-call plugin kernel runs the passes side by side, one
// per lane, through the same copy)
template <class In>
__device__ __noinline__ LMC_PF_ATTR void PathFuncHessDevice(int c, int l, const float *primary, const float *scene, const In &vp, float *logLum, float *grad, float *hess) {
    const int dim = 2 * (c + l - 1 > 2 ? c + l - 1 : 2);
    for (int i = 0; i < dim; i++)
        for (int c0 = 0; c0 < dim; c0 += HC) PathFuncHessPassDevice(c, l, primary, scene, vp, i, c0, logLum, grad, hess, c0 == 0);
}
#endif

}  // namespace lmcd
