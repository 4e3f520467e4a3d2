// k_step<true, false, false, ., ., MUX = true>: the multiplexed large step (`largestepmultiplexed`, mutation_large.h:45-58,87-102), a TU of
// its own so that the default large step keeps its registers and the two compile in parallel
#ifndef LMC_NO_RNG_JUMP_LDS
#define LMC_RNG_JUMP_LDS  // drng.h: the PCG jump constants of this launch live in LDS
#endif
#include "step_kernel.h"

using namespace lmcd;

void LaunchStepLargeMux(const DScene &S, const DCache *cache, const ChainArrays &A, const Film &film, const StepParams &P, const int *list, const int *listCount,
                     const NextLists &next, float *gradBuf, int gradStride, bool glossy, int gridBlocks, int bvhStackNeed, int blockThreads, hipStream_t s) {
    RequireJumpLdsBlock(blockThreads);
    if (bvhStackNeed <= BVH_LDS_STACK) {  // traversal stack in LDS; gridBlocks was sized for 256-thread blocks
        const int blocks = gridBlocks * (256 / blockThreads);
        const size_t ldsBytes = (size_t)blockThreads * ((bvhStackNeed + 7) / 8 * 8) * sizeof(int);  // the scene's own stack need, not the cap
        if (glossy) hipLaunchKernelGGL((k_step<true, false, false, true, true, 1>), dim3(blocks), dim3(blockThreads), ldsBytes, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
        else
            hipLaunchKernelGGL((k_step<true, false, false, false, true, 1>), dim3(blocks), dim3(blockThreads), ldsBytes, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
        return;
    }
    if (glossy) hipLaunchKernelGGL((k_step<true, false, false, true, false, 1>), dim3(gridBlocks), dim3(256), 0, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
    else
        hipLaunchKernelGGL((k_step<true, false, false, false, false, 1>), dim3(gridBlocks), dim3(256), 0, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
}
