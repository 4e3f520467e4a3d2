// The chain-step kernel template; instantiated three times (step_large.hip, step_small_grad.hip,
// step_small_plain.hip) so that large steps, gradient-evaluating small steps and plain small steps run as
// separate launches: no large/small divergence inside a wave, and the common case (plain small step) does not
// carry the registers and scratch of the other two.
#pragma once
#include "dstep.h"
#include "kernels.h"

namespace lmcd {

// `sh`: 9 words of LDS (8 counters + the weight sum); the lean kernel passes its dynamic LDS (14 KB per 64-thread block) instead of
// declaring a static array next to it
__device__ __forceinline__ void BlockReduceStats(const StepStats &st, unsigned long long *counters, double *weightSum, int *sh) {
    int *sInt = sh;
    float &sW = *reinterpret_cast<float *>(sh + 8);
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 8; k++) sInt[k] = 0;
        sW = 0.f;
    }
    __syncthreads();
    int v[8] = {st.steps, st.large, st.accepted, st.gradCalls, st.cacheQueries, st.cacheHits, st.resets, st.lean};
    float w = st.wsum;
    for (int off = 32; off > 0; off >>= 1) {  // wave reduction (64 lanes), then one LDS atomic per wave
        for (int k = 0; k < 8; k++) v[k] += __shfl_down(v[k], off);
        w += __shfl_down(w, off);
    }
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 8; k++)
            if (v[k]) atomicAdd(&sInt[k], v[k]);
        atomicAdd(&sW, w);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 8; k++)
            if (sInt[k]) atomicAdd(&counters[k], (unsigned long long)sInt[k]);
        if (sW != 0.f) atomicAdd(weightSum, (double)sW);
    }
}

// LDS_STACK: the BVH traversal stack in dynamic LDS ([entry][thread], BVH_LDS_STACK entries per thread) instead of private memory
// LMC_STEP_WAVES: waves per SIMD the register allocation of these launches aims at.  Unconstrained, the glossy large-step
// instantiation takes 256 VGPRs + 19 AGPRs = ONE wave per SIMD; with a budget of 256 (48 B more spills) two waves fit and the launch
// is 2.9 x faster on the full-material torus (9.65 -> 3.32 ms), 1.3 x on the door scene (profiles/r03_b_ab_large_step_waves.jsonl);
// three waves (168 VGPRs, 384 B of spills) gave nothing more in round 3.  Round 5, after the load pinning (dscene.h LMC_PIN), measured again
// (profiles/r05_ao_ab_waves_per_simd.jsonl): the GLOSSY large-step launch compiled for three waves: veach-door LMC +3 %, H2MC +2.3 % (a 168-register wave
// fits beside two of the lean launch's), headline unchanged -- the Lambertian instantiation stays at two.
#ifndef LMC_STEP_WAVES
#define LMC_STEP_WAVES 2
#endif
#ifndef LMC_STEP_WAVES_GLOSSY_LARGE
#define LMC_STEP_WAVES_GLOSSY_LARGE 3
#endif
template <bool WITH_LARGE, bool WITH_SMALL, bool WITH_GRAD, bool GLOSSY, bool LDS_STACK = false, int MUX = 0, bool QUANT = false>
__global__ void __launch_bounds__(256, (GLOSSY && WITH_LARGE && !WITH_SMALL && LDS_STACK) ? LMC_STEP_WAVES_GLOSSY_LARGE : LMC_STEP_WAVES) k_step(DScene S, const DCache *cache, ChainArrays A, Film film, StepParams P, const int *list, const int *listCount,
                                              NextLists next, float *gradBuf, int gradStride) {
    if ((int)(blockIdx.x * blockDim.x) >= *listCount) return;  // a block past the end of the work list: nothing to set up, nothing to do
    LMC_RNG_JUMP_INIT();
    LMC_MAT_LDS_INIT(S);
    extern __shared__ int ldsStack[];
    StepStats st;
    const int total = *listCount;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int j = tid; j < total; j += gridDim.x * blockDim.x) {
        const int i = list[j];
        Rng rng = LoadChainRng(A, P.chainBegin, S.opt.seedOffset, i);
        GradWork gw{gradBuf, (size_t)gradStride, (size_t)tid};
        const int kind = WITH_LARGE ? KIND_LARGE : KIND_SMALL;  // decided (and its uniform drawn) at the end of the previous step
        if constexpr (LDS_STACK) {
            LdsStackT<GLOSSY, QUANT> stk{ldsStack + threadIdx.x, (int)blockDim.x, 0};
            StepChain<WITH_LARGE, WITH_SMALL, WITH_GRAD, MUX>(S, *cache, A, film, P, i, kind, rng, gw, st, stk);
        } else {
            LocalStackT<GLOSSY> stk;
            StepChain<WITH_LARGE, WITH_SMALL, WITH_GRAD, MUX>(S, *cache, A, film, P, i, kind, rng, gw, st, stk);
        }
        QueueNext(S, *cache, A, P, i, rng);
        StoreChainRng(A, i, rng);
    }
    __shared__ int sStats[9];
    BlockReduceStats(st, A.counters, A.weightSum, sStats);
}

}  // namespace lmcd
