// The cache-filling launch: small steps of the chains whose dimension's cache is not ready yet, i.e. whose Gaussians come
// from the gradient of the path program (mutation_mala.h:94-130).  Same streamed body as the hot launch (dsmall.h) with the
// gradient branch compiled in; the program itself runs in a non-inlined function over the path record in HBM.  It replaces
// k_step<false, true, true, true> (step_small_grad.hip) on this list -- that kernel keeps the whole path in private memory
// and took 3.6 ms per wave even with the gradient program skipped (profiles/r02_h_*) -- which remains the fallback for BVHs
// deeper than the LDS traversal stack and for cache trees deeper than the LDS search frames.
#define LMC_LEAN_GRAD
#ifndef LMC_NO_RNG_JUMP_LDS
#define LMC_RNG_JUMP_LDS  // drng.h: the PCG jump constants of this launch live in LDS
#endif
#include "dsmall.h"
#include "step_kernel.h"

using namespace lmcd;

template <bool GLOSSY>
#ifndef LMC_LEANGRAD_WAVES
#define LMC_LEANGRAD_WAVES 2  // registers for two waves per SIMD: a wave of this launch then fits beside a resident wave of the hot launch
#endif
__global__ void __launch_bounds__(256, LMC_LEANGRAD_WAVES) k_step_small_grad(DScene S, const DCache *cache, ChainArrays A, Film film, StepParams P, const int *list,
                                                      const int *listCount, NextLists next, float *gradBuf, int gradStride, int stackWords) {
    if ((int)(blockIdx.x * blockDim.x) >= *listCount) return;  // a block past the end of the work list: nothing to set up, nothing to do
    LMC_RNG_JUMP_INIT();
    LMC_MAT_LDS_INIT(S);
    extern __shared__ float lds[];
    StepStats st;
    const int total = *listCount;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const LdsView L{lds + threadIdx.x, (int)blockDim.x, stackWords};
    for (int j = tid; j < total; j += gridDim.x * blockDim.x) {
        const int i = list[j];
        Rng rng = LoadChainRng(A, P.chainBegin, S.opt.seedOffset, i);
        LdsStackT<GLOSSY> stk{reinterpret_cast<int *>(L.base), L.stride, 0};
        NoProf prof;
        SmallStepLean<true>(S, *cache, A, film, P, i, rng, L, stk, st, prof, gradBuf, (size_t)gradStride, (size_t)tid);
        QueueNext(S, *cache, A, P, i, rng);
        StoreChainRng(A, i, rng);
    }
    BlockReduceStats(st, A.counters, A.weightSum, reinterpret_cast<int *>(lds));
}

// gridBlocks * blockThreads must not exceed gradStride (one serialisation slot per thread)
void LaunchStepSmallLeanGrad(const DScene &S, const DCache *cache, const ChainArrays &A, const Film &film, const StepParams &P, const int *list, const int *listCount,
                             const NextLists &next, float *gradBuf, int gradStride, bool glossy, int gridBlocks, int blockThreads, int bvhStackNeed, hipStream_t s) {
    RequireJumpLdsBlock(blockThreads);
    const int stackWords = LeanStackWords(bvhStackNeed);
    const size_t ldsBytes = (size_t)blockThreads * LeanLdsWordsPerThread(stackWords) * sizeof(float);
    if (glossy) hipLaunchKernelGGL((k_step_small_grad<true>), dim3(gridBlocks), dim3(blockThreads), ldsBytes, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride, stackWords);
    else
        hipLaunchKernelGGL((k_step_small_grad<false>), dim3(gridBlocks), dim3(blockThreads), ldsBytes, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride, stackWords);
}
