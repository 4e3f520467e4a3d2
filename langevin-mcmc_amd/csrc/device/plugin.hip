// The reference's plugin ABI, ONE evaluation per call (evaluate_path_bidir_mala_<c>_<l>_static{,_derv}; caller mutation_mala.h:97-110):
// a single wave.  The caller's arrays arrive through host-mapped pinned memory and the result leaves the same way (host/context.cpp
// PluginSlot: no hipMalloc, no copy call, one launch + one stream sync per call); the wave stages the inputs in LDS and lane k computes
// d logLum / d primary[k + 1] with a one-wide dual number -- the gradient's components side by side instead of one after the other.
#include "kernels.h"
#include "pathfunc.h"

using namespace lmcd;

// in: [primary 17 | scene 38 | vertParams V]; out: [logLum | grad 16]
__global__ void __launch_bounds__(64) k_plugin_grad(int c, int l, const float *in, float *out, int wantGrad) {
    __shared__ float s[17 + 38 + 600];
    const int L = c + l - 1 > 2 ? c + l - 1 : 2, dim = 2 * L, V = 238 + 59 * (c + l - 3);
    for (int k = threadIdx.x; k < 17 + 38 + V; k += 64) s[k] = in[k];
    __syncthreads();
    const float *primary = s, *scene = s + 17;
    const ContigIn vin{s + 55};
    if (!wantGrad) {
        if (threadIdx.x == 0) out[0] = PathFuncValue(c, l, primary, scene, vin);
        return;
    }
    const int k = threadIdx.x;
    if (k < dim) {
        Dual<1> p[2 * 8 + 1];
        for (int j = 0; j <= dim; j++) p[j] = MakeDual<1>(primary[j]);
        p[k + 1].d[0] = 1.0f;
        const Dual<1> r = PathProgram<Dual<1>, ContigIn>(c, l, p, scene, vin);
        out[1 + k] = r.d[0];
        if (k == 0) out[0] = r.v;
    }
}

void LaunchPluginGrad(int c, int l, const float *in, float *out, int wantGrad, hipStream_t s) { hipLaunchKernelGGL(k_plugin_grad, dim3(1), dim3(64), 0, s, c, l, in, out, wantGrad); }
