// k_step<true, false, false>: see step_kernel.h
#ifndef LMC_NO_RNG_JUMP_LDS
#define LMC_RNG_JUMP_LDS  // drng.h: the PCG jump constants of this launch live in LDS
#endif
#include "step_kernel.h"
#include "dlarge.h"

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

// ---- the same launch with re-filled lanes (dlarge.h).  One wave per block; of the launch's blocks, as many take part as give every lane `perLane`
// chains of the work list on average (a wave that starts with 64 chains and never re-fills is the plain launch).
namespace lmcd {
template <bool GLOSSY, bool QUANT>
__global__ void __launch_bounds__(64, GLOSSY ? LMC_STEP_WAVES_GLOSSY_LARGE : LMC_STEP_WAVES) k_step_large_refill(DScene S, const DCache *cache, ChainArrays A, Film film, StepParams P, const int *list,
                                                                                                              const int *listCount, int *cursor, int retireAt, int perLane) {
    const int total = *listCount;
    const int waves = min((int)gridDim.x, (total + 64 * perLane - 1) / (64 * perLane));
    if ((int)blockIdx.x >= waves) return;
    LMC_RNG_JUMP_INIT();
    LMC_MAT_LDS_INIT(S);
    extern __shared__ int ldsStack[];
    StepStats st;
    LdsStackT<GLOSSY, QUANT> stk{ldsStack + threadIdx.x, (int)blockDim.x, 0};
    LargeStepsRefilled(S, *cache, A, film, P, list, total, cursor, retireAt, st, stk);
    __shared__ int sStats[9];
    BlockReduceStats(st, A.counters, A.weightSum, sStats);
}
}  // namespace lmcd

bool LaunchStepLargeRefill(const DScene &S, const DCache *cache, const ChainArrays &A, const Film &film, const StepParams &P, const int *list, const int *listCount, int *cursor,
                           int retireAt, int perLane, bool glossy, int maxWaves, int bvhStackNeed, hipStream_t s) {
    if (bvhStackNeed > BVH_LDS_STACK) return false;  // a tree too deep for the LDS stack: the plain launch (LaunchStepLarge)
    RequireJumpLdsBlock(64);
    const size_t ldsBytes = (size_t)64 * ((bvhStackNeed + 7) / 8 * 8) * sizeof(int);
    const bool quant = S.qnodes != nullptr;
    if (glossy && quant) hipLaunchKernelGGL((k_step_large_refill<true, true>), dim3(maxWaves), dim3(64), ldsBytes, s, S, cache, A, film, P, list, listCount, cursor, retireAt, perLane);
    else if (glossy) hipLaunchKernelGGL((k_step_large_refill<true, false>), dim3(maxWaves), dim3(64), ldsBytes, s, S, cache, A, film, P, list, listCount, cursor, retireAt, perLane);
    else if (quant) hipLaunchKernelGGL((k_step_large_refill<false, true>), dim3(maxWaves), dim3(64), ldsBytes, s, S, cache, A, film, P, list, listCount, cursor, retireAt, perLane);
    else
        hipLaunchKernelGGL((k_step_large_refill<false, false>), dim3(maxWaves), dim3(64), ldsBytes, s, S, cache, A, film, P, list, listCount, cursor, retireAt, perLane);
    return true;
}
