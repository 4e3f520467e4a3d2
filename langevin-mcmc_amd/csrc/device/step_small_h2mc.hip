// The per-lane forms of the H2MC library symbols (evaluate_path_bidir_<c>_<l>_static_derv with `hess`): k_hess_batch, one lane per item, and
// k_plugin_hess, one call = one wave with a lane per pass; the FULL matrix, as the reference's programs deliver it.  (The H2MC step itself
// needs one triangle only and evaluates it wave-cooperatively: h2hess.hip.)
// Both call ONE out-of-line copy of the second-order path program (PathFuncHessDevice); its building blocks are outlined too
// (LMC_PF_OUTLINE, pathfunc.h): this translation unit compiles in about two minutes instead of twenty-five.
#define LMC_PF_OUTLINE
#include "dh2step.h"
#include "step_kernel.h"

using namespace lmcd;

// lmc_hess_batch: value, gradient and Hessian (row i of item j at hessSoA[(i * dim + k) * n + j])
__global__ void __launch_bounds__(64) k_hess_batch(int c, int l, int n, const float *primarySoA, const float *scene, const float *vertSoA, float *logLum,
                                                  float *gradSoA, float *hessSoA) {
    __shared__ float sScene[38];
    if (threadIdx.x < 38) sScene[threadIdx.x] = scene[threadIdx.x];
    __syncthreads();
    const int L = c + l - 1 > 2 ? c + l - 1 : 2;
    const int dim = 2 * L;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float primary[2 * 8 + 1];
        for (int k = 0; k < dim + 1; k++) primary[k] = primarySoA[(size_t)k * n + i];
        StridedIn vin{vertSoA + i, (size_t)n};
        float ll, g[16], h[256];
        PathFuncHessDevice(c, l, primary, sScene, vin, &ll, g, h);
        if (logLum) logLum[i] = ll;
        if (gradSoA)
            for (int k = 0; k < dim; k++) gradSoA[(size_t)k * n + i] = g[k];
        for (int k = 0; k < dim * dim; k++) hessSoA[(size_t)k * n + i] = h[k];
    }
}


// The H2MC plugin symbols, one evaluation per call (evaluate_path_bidir_<c>_<l>_static_derv; caller mutation_h2mc.h:74-79): like
// plugin.hip, one wave, inputs / outputs in host-mapped pinned memory; lane t runs ONE pass of the second-order program (row
// t / chunks, column chunk t % chunks): the dim x ceil(dim / 8) passes side by side.  out: [logLum | grad 16 | hess 256]
__global__ void __launch_bounds__(64) k_plugin_hess(int c, int l, const float *in, float *stage, float *out) {
    // the inputs are staged in device memory (the slot's own 655 floats; L1-resident for the one wave): the out-of-line program takes its
    // accessor by reference, and an accessor over a __shared__ array would have to be a compile-time constant (unsupported address-space cast)
    const int L = c + l - 1 > 2 ? c + l - 1 : 2, dim = 2 * L, V = 238 + 59 * (c + l - 3);
    for (int k = threadIdx.x; k < 17 + 38 + V; k += 64) stage[k] = in[k];
    __threadfence_block();
    __syncthreads();
    const int chunks = (dim + HC - 1) / HC;
    const int t = threadIdx.x;
    if (t < dim * chunks) {
        const StridedIn vin{stage + 55, 1};  // the accessor type of the step kernel: the same copy of the program
        PathFuncHessPassDevice(c, l, stage, stage + 17, vin, t / chunks, (t % chunks) * HC, out, out + 1, out + 17, (t % chunks) == 0);
    }
}
void LaunchPluginHess(int c, int l, const float *in, float *stage, float *out, hipStream_t s) { hipLaunchKernelGGL(k_plugin_hess, dim3(1), dim3(64), 0, s, c, l, in, stage, out); }

void LaunchHessBatch(int c, int l, int n, const float *primarySoA, const float *scene, const float *vertSoA, float *logLum, float *gradSoA, float *hessSoA,
                     hipStream_t s) {
    hipLaunchKernelGGL(k_hess_batch, dim3((n + 63) / 64 < 4096 ? (n + 63) / 64 : 4096), dim3(64), 0, s, c, l, n, primarySoA, scene, vertSoA, logLum, gradSoA, hessSoA);
}
