// The large-step launch with the vertex connections of a wave shared out over its lanes (dlargecoop.h): LargeStep::Mutate (mutation_large.h:31-128) on
// GeneratePathBidir, one-wave blocks.  Everything behind the generation -- technique pick, acceptance, splats, the state's write-out, cache push -- is
// StepChain's (dstep.h, PREGEN).  Selected by host/context.cpp for the default large step (not the multiplexed / cached forms) on trees that fit the LDS stack.
#ifndef LMC_NO_RNG_JUMP_LDS
#define LMC_RNG_JUMP_LDS  // drng.h: the PCG jump constants of this launch live in LDS
#endif
#include "step_kernel.h"
#include "dlargecoop.h"

using namespace lmcd;

template <bool GLOSSY, bool QUANT>
__global__ void __launch_bounds__(64, GLOSSY ? LMC_STEP_WAVES_GLOSSY_LARGE : LMC_STEP_WAVES) k_step_large_coop(DScene S, const DCache *cache, ChainArrays A, Film film, StepParams P, const int *list,
                                                                                                           const int *listCount, NextLists next, CoopScratch X, int stackWords) {
    if ((int)(blockIdx.x * 64) >= *listCount) return;
    LMC_RNG_JUMP_INIT();
    LMC_MAT_LDS_INIT(S);
    extern __shared__ int ldsStack[];  // [stackWords][64] traversal stack, then the task list
    unsigned short *taskList = reinterpret_cast<unsigned short *>(ldsStack + (size_t)stackWords * 64);
    StepStats st;
    const int total = *listCount, lane = threadIdx.x;
    for (int j0 = blockIdx.x * 64; j0 < total; j0 += gridDim.x * 64) {  // wave-uniform: lanes past the end of the list still work off connection tasks
        const int j = j0 + lane;
        const bool active = j < total;
        const int i = list[active ? j : j0];
        Rng rng = LoadChainRng(A, P.chainBegin, S.opt.seedOffset, i);
        LdsStackT<GLOSSY, QUANT> stk{ldsStack + lane, 64, 0};
        DPath prop;
        ContribSink sink{A.contribList, (size_t)A.N, (size_t)i, 0};
        GeneratePathBidirCoop(S, max(S.opt.minDepth, 3), S.opt.maxDepth, prop, sink, rng, stk, active, i, X, taskList);
        if (active) {
            GradWork gw{nullptr, 0, 0};
            StepChain<true, false, false, 0, true>(S, *cache, A, film, P, i, KIND_LARGE, rng, gw, st, stk, &prop, sink.count);
            QueueNext(S, *cache, A, P, i, rng);
            StoreChainRng(A, i, rng);
        }
    }
    __shared__ int sStats[9];
    BlockReduceStats(st, A.counters, A.weightSum, sStats);
}

size_t LargeCoopScratchFloats(int N) { return (size_t)N * ((size_t)MAXD * COOP_STATE_WORDS + COOP_CAM_WORDS + (size_t)MAXD * COOP_RES_WORDS); }

void LaunchStepLargeCoop(const DScene &S, const DCache *cache, const ChainArrays &A, const Film &film, const StepParams &P, const int *list, const int *listCount, const NextLists &next,
                         float *scratch, bool glossy, int gridBlocks, int bvhStackNeed, hipStream_t s) {
    RequireJumpLdsBlock(64);
    const size_t N = (size_t)A.N;
    CoopScratch X{scratch, scratch + N * MAXD * COOP_STATE_WORDS, scratch + N * (MAXD * COOP_STATE_WORDS + COOP_CAM_WORDS), N};
    const int stackWords = (bvhStackNeed + 7) / 8 * 8;
    const size_t ldsBytes = (size_t)64 * stackWords * sizeof(int) + COOP_MAX_TASKS * sizeof(unsigned short);
    const int blocks = gridBlocks * 4;  // gridBlocks was sized for 256-thread blocks
    const bool quant = S.qnodes != nullptr;
    if (glossy && quant) hipLaunchKernelGGL((k_step_large_coop<true, true>), dim3(blocks), dim3(64), ldsBytes, s, S, cache, A, film, P, list, listCount, next, X, stackWords);
    else if (glossy) hipLaunchKernelGGL((k_step_large_coop<true, false>), dim3(blocks), dim3(64), ldsBytes, s, S, cache, A, film, P, list, listCount, next, X, stackWords);
    else if (quant) hipLaunchKernelGGL((k_step_large_coop<false, true>), dim3(blocks), dim3(64), ldsBytes, s, S, cache, A, film, P, list, listCount, next, X, stackWords);
    else
        hipLaunchKernelGGL((k_step_large_coop<false, false>), dim3(blocks), dim3(64), ldsBytes, s, S, cache, A, film, P, list, listCount, next, X, stackWords);
}
