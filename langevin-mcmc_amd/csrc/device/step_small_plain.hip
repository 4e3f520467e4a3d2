// The hot launch: plain small steps (dsmall.h) of the chains on the smallPlain list.  Everything indexed at run time
// lives in LDS (dsmall.h: 56 words per thread on the torus) and the path is streamed through registers.  Private memory: the kernel reports
// 2.9-3.4 KB per lane, which is (hipcc -S, round 4) the frames of the out-of-line functions -- the kd-tree search that 0.015 % of the queries
// reach, the once-per-2^32-draws table advance of the RNG, the trigonometry -- plus, in the body itself, 79 scratch loads and 83 scratch stores:
// the caller-saved registers around those (cold) call sites and, in the cache-query path, the 12-word query vector handed to the search by
// address with the LDS addresses it is filled from.  The steady-state step executes the query-path ones only (two queries per step).
// USE_LDS_STACK = false is the fallback for scenes whose LBVH is deeper than the 32-entry LDS traversal stack.
#include <cstdlib>
#include <type_traits>

#ifndef LMC_NO_RNG_JUMP_LDS
#define LMC_RNG_JUMP_LDS  // drng.h: the PCG jump constants of this launch live in LDS
#endif
#include "dsmall.h"
#include "step_kernel.h"

using namespace lmcd;

template <bool USE_LDS_STACK, bool GLOSSY, bool PROF = false, bool LIGHTLESS = false, bool QUANT = false>
#ifndef LMC_LEAN_K
#define LMC_LEAN_K 1
#endif
#ifndef LMC_LEAN_WAVES
#define LMC_LEAN_WAVES 2  // waves per SIMD the register allocation aims at
#endif
#ifndef LMC_LEAN_WAVES_GLOSSY_LIGHTLESS
// ... of the glossy instantiation without light sub-paths (full-material scenes lit by their environment map: BASELINE configs[2]): three.  Measured
// (profiles/r05_ao_ab_waves_per_simd.jsonl): that workload +4 %; the glossy instantiation WITH light sub-paths (veach-door) loses 3 % at three waves and the
// Lambertian one 5 % (it has no registers to give), so they stay at two.
#define LMC_LEAN_WAVES_GLOSSY_LIGHTLESS 3
#endif
__global__ void __launch_bounds__(256, (GLOSSY && LIGHTLESS) ? LMC_LEAN_WAVES_GLOSSY_LIGHTLESS : LMC_LEAN_WAVES) k_step_small(DScene S, const DCache *cache, ChainArrays A, Film film, StepParams P, const int *list,
                                                    const int *listCount, NextLists next, int stackWords) {
    extern __shared__ float lds[];
    if ((int)(blockIdx.x * blockDim.x) >= *listCount) return;  // a block past the end of the work list: nothing to set up, nothing to do
    LMC_RNG_JUMP_INIT();
    LMC_MAT_LDS_INIT(S);
    StepStats st;
    const int total = *listCount;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const LdsView L{lds + threadIdx.x, (int)blockDim.x, stackWords};
    typename std::conditional<PROF, WaveProf, NoProf>::type prof;
    if constexpr (PROF) prof.Start();
    for (int j = tid; j < total; j += gridDim.x * blockDim.x) {
        const int i = list[j];
        Rng rng = LoadChainRng(A, P.chainBegin, S.opt.seedOffset, i);
#if LMC_LEAN_K > 1  // A/B build (DESIGN.md "K small steps of a chain per launch"): the chain stays in its lane for up to K consecutive plain small steps
#pragma unroll 1
        for (int rep = 0;; rep++) {
#endif
        if (USE_LDS_STACK) {
            LdsStackT<GLOSSY, QUANT> stk{reinterpret_cast<int *>(L.base), L.stride, 0};
            SmallStepLean<false, LIGHTLESS>(S, *cache, A, film, P, i, rng, L, stk, st, prof);
        } else {
            LocalStackT<GLOSSY, QUANT> stk;
            SmallStepLean<false, LIGHTLESS>(S, *cache, A, film, P, i, rng, L, stk, st, prof);
        }
        const unsigned char nk = QueueNext(S, *cache, A, P, i, rng);
#if LMC_LEAN_K > 1
            if (rep + 1 >= LMC_LEAN_K || (nk & 3) != NEXT_SMALL_PLAIN) break;
        }
#else
        (void)nk;
#endif
        StoreChainRng(A, i, rng);
        prof.Mark(PR_QUEUE);
    }
    if constexpr (PROF) {  // one lane per wave adds the wave's region totals (A.prof: PR_COUNT cycle sums + the number of waves)
        if ((threadIdx.x & 63) == 0) {
            for (int r = 0; r < PR_COUNT; r++) atomicAdd(&A.prof[r], prof.acc[r]);
            atomicAdd(&A.prof[PR_COUNT], 1ull);
        }
    }
    if (!LMC_EXP(P.expFlags, 8)) BlockReduceStats(st, A.counters, A.weightSum, reinterpret_cast<int *>(lds));
}

void LaunchStepSmallPlain(const DScene &S, const DCache *cache, const ChainArrays &A, const Film &film, const StepParams &P, const int *list, const int *listCount,
                          const NextLists &next, int bvhDepth, bool glossy, int gridBlocks, int blockThreads, bool profile, hipStream_t s) {
    RequireJumpLdsBlock(blockThreads);
    const int stackWords = LeanStackWords(bvhDepth);  // bvhDepth: the tree's stack need (host/accel.cpp)
    size_t ldsBytes = (size_t)blockThreads * LeanLdsWordsPerThread(stackWords) * sizeof(float);
    if (const char *e = getenv("LMC_EXP_LDS_EXTRA")) ldsBytes += (size_t)atoi(e);  // measurement aid: lowers the occupancy without touching the code
    const bool lds = bvhDepth <= BVH_LDS_STACK;
#define LMC_LAUNCH_SMALL(...) hipLaunchKernelGGL((k_step_small<__VA_ARGS__>), dim3(gridBlocks), dim3(blockThreads), ldsBytes, s, S, cache, A, film, P, list, listCount, next, stackWords)
    const bool quant = S.qnodes != nullptr && lds && !profile;  // the scene's choice (host/context.cpp UploadScene); the profiling and fallback instantiations stay on the exact nodes
    if (profile && lds && glossy) LMC_LAUNCH_SMALL(true, true, true);
    else if (profile && lds && !glossy) LMC_LAUNCH_SMALL(true, false, true);
    else if (lds && !glossy && S.opt.leanLightless) {
        if (quant) LMC_LAUNCH_SMALL(true, false, false, true, true);
        else
            LMC_LAUNCH_SMALL(true, false, false, true);
    } else if (lds && glossy && S.opt.leanLightless) {
        if (quant) LMC_LAUNCH_SMALL(true, true, false, true, true);
        else
            LMC_LAUNCH_SMALL(true, true, false, true);
    } else if (lds && !glossy) {
        if (quant) LMC_LAUNCH_SMALL(true, false, false, false, true);
        else
            LMC_LAUNCH_SMALL(true, false);
    } else if (lds && glossy) {
        if (quant) LMC_LAUNCH_SMALL(true, true, false, false, true);
        else
            LMC_LAUNCH_SMALL(true, true);
    } else if (!glossy)
        LMC_LAUNCH_SMALL(false, false);
    else
        LMC_LAUNCH_SMALL(false, true);
#undef LMC_LAUNCH_SMALL
}
