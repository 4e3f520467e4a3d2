// k_step<false, true, false>: see step_kernel.h
#include "step_kernel.h"

using namespace lmcd;

void LaunchStepSmallPlain(const DScene &S, const DCache *cache, const ChainArrays &A, const Film &film, const StepParams &P, const int *list, const int *listCount,
                     const NextLists &next, float *gradBuf, int gradStride, int gridBlocks, hipStream_t s) {
    hipLaunchKernelGGL((k_step<false, true, false>), dim3(gridBlocks), dim3(256), 0, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
}
