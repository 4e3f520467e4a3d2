// k_step<true, false, false, ., ., MUX = 2>: LargeStepCache (`samplecache` with mala, mutation_large_cache.h:22-141), a TU of its own like
// step_large_mux.hip
#ifndef LMC_NO_RNG_JUMP_LDS
#define LMC_RNG_JUMP_LDS  // drng.h: the PCG jump constants of this launch live in LDS
#endif
#include "step_kernel.h"

using namespace lmcd;

void LaunchStepLargeCache(const DScene &S, const DCache *cache, const ChainArrays &A, const Film &film, const StepParams &P, const int *list, const int *listCount,
                     const NextLists &next, float *gradBuf, int gradStride, bool glossy, int gridBlocks, int bvhStackNeed, int blockThreads, hipStream_t s) {
    RequireJumpLdsBlock(blockThreads);
    if (bvhStackNeed <= BVH_LDS_STACK) {  // traversal stack in LDS; gridBlocks was sized for 256-thread blocks
        const int blocks = gridBlocks * (256 / blockThreads);
        const size_t ldsBytes = (size_t)blockThreads * ((bvhStackNeed + 7) / 8 * 8) * sizeof(int);  // the scene's own stack need, not the cap
        if (glossy) hipLaunchKernelGGL((k_step<true, false, false, true, true, 2>), dim3(blocks), dim3(blockThreads), ldsBytes, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
        else
            hipLaunchKernelGGL((k_step<true, false, false, false, true, 2>), dim3(blocks), dim3(blockThreads), ldsBytes, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
        return;
    }
    if (glossy) hipLaunchKernelGGL((k_step<true, false, false, true, false, 2>), dim3(gridBlocks), dim3(256), 0, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
    else
        hipLaunchKernelGGL((k_step<true, false, false, false, false, 2>), dim3(gridBlocks), dim3(256), 0, s, S, cache, A, film, P, list, listCount, next, gradBuf, gradStride);
}

// parity probe of the two cache-side pieces of LargeStepCache on the cache as it stands: item i draws a row with u[i] (sampleCache)
// and evaluates the kernel density at query[i] for technique cl[2 i], cl[2 i + 1] (evalPdfCache)
__global__ void k_cache_probe(const DCache *cache, int dim, int n, const float *u, int *row, const float *query, const int *cl, float *pdf) {
    const DCacheDim &D = cache->d[dim];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        row[i] = CacheSampleRow(D, u[i]);
        float q[MAXPSS];
        for (int k = 0; k < dim; k++) q[k] = query[(size_t)i * dim + k];
        pdf[i] = EvalPdfCache(D, dim, q, cl[2 * i], cl[2 * i + 1]);
    }
}
void LaunchCacheProbe(const DCache *cache, int dim, int n, const float *u, int *row, const float *query, const int *cl, float *pdf, hipStream_t s) {
    hipLaunchKernelGGL(k_cache_probe, dim3((n + 63) / 64), dim3(64), 0, s, cache, dim, n, u, row, query, cl, pdf);
}
