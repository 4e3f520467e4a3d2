// k_step<true, false, false>: see step_kernel.h
#ifndef LMC_NO_RNG_JUMP_LDS
#define LMC_RNG_JUMP_LDS  // drng.h: the PCG jump constants of this launch live in LDS
#endif
#include "step_kernel.h"

using namespace lmcd;

void LaunchStepLarge(const DScene &S, const DCache *cache, const ChainArrays &A, const Film &film, const StepParams &P, const int *list, const int *listCount,
                     const NextLists &next, float *gradBuf, int gradStride, bool glossy, int gridBlocks, int bvhStackNeed, int blockThreads, hipStream_t s) {
    RequireJumpLdsBlock(blockThreads);
    if (bvhStackNeed <= BVH_LDS_STACK) {  // traversal stack in LDS; gridBlocks was sized for 256-thread blocks
        const int blocks = gridBlocks * (256 / blockThreads);
        const size_t ldsBytes = (size_t)blockThreads * ((bvhStackNeed + 7) / 8 * 8) * sizeof(int);  // the scene's own stack need, not the cap
        const bool quant = S.qnodes != nullptr;  // the scene's choice of node format (host/context.cpp UploadScene, dscene.h LdsStackT::kQuant)
        if (glossy && quant) hipLaunchKernelGGL((k_step<true, false, false, true, true, 0, true>), dim3(blocks), dim3(blockThreads), ldsBytes, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
        else if (glossy) hipLaunchKernelGGL((k_step<true, false, false, true, true>), dim3(blocks), dim3(blockThreads), ldsBytes, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
        else if (quant) hipLaunchKernelGGL((k_step<true, false, false, false, true, 0, true>), dim3(blocks), dim3(blockThreads), ldsBytes, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
        else
            hipLaunchKernelGGL((k_step<true, false, false, false, true>), dim3(blocks), dim3(blockThreads), ldsBytes, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
        return;
    }
    if (glossy) hipLaunchKernelGGL((k_step<true, false, false, true>), dim3(gridBlocks), dim3(256), 0, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
    else
        hipLaunchKernelGGL((k_step<true, false, false, false>), dim3(gridBlocks), dim3(256), 0, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
}
