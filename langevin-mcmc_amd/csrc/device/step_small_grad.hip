// k_step<false, true, true, true>: see step_kernel.h.  The generic small-step launch runs during cache warm-up only, so
// it is compiled once, with the glossy BSDF code in (the material dispatch is a run-time test on DMaterial::type; a
// Lambertian-only scene just never takes the other branches).
#include "step_kernel.h"

using namespace lmcd;

void LaunchStepSmallGrad(const DScene &S, const DCache *cache, const ChainArrays &A, const Film &film, const StepParams &P, const int *list, const int *listCount,
                         const NextLists &next, float *gradBuf, int gradStride, bool /*glossy*/, int gridBlocks, hipStream_t s) {
    hipLaunchKernelGGL((k_step<false, true, true, true>), dim3(gridBlocks), dim3(256), 0, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
}
