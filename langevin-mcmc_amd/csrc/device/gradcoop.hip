// k_mala_grad: d log f / d pss for the MALA proposal of the cache-fill phase (the reference's evaluate_path_bidir_mala_<c>_<l>_static_derv
// programs, generator /root/reference/src/path.cpp:3419-3662; caller mutation_mala.h:101-107), WAVE-COOPERATIVE: the lanes of a wave are
// the Dual<2> passes of a few states of ONE technique (c,l) -- dim 12: six lanes per state, ten states per wave -- instead of one lane
// walking the six passes of its own chain's state over a record in private memory (dgrad.h ComputeGradient, which remains the form of the
// fall-back kernels).  Forward-mode components never mix and this translation unit keeps the strict arithmetic of the step kernels, so a
// gradient is bit-identical to ComputeGradient's.  The state's serialised record is staged in LDS once per wave and read by broadcast.
#ifndef LMC_NO_PF_VEC2
#define LMC_PF_VEC2  // pathfunc.h: Dual<2> on two-wide vectors (packed FP32 instructions); per component the same operations in the same order, so the
                     // gradient stays bit-identical to ComputeGradient's (tests/test_gpu_parity.py::test_cache_fill_pipeline_equals_the_single_launch_form)
#endif
#include "dh2coop.h"
#include "pathfunc.h"
#include "kernels.h"

using namespace lmcd;

namespace {

struct LdsIn {  // the state's vertParams in LDS
    const float *p;
    __device__ __forceinline__ float operator[](int k) const { return p[k]; }
};
// a lane's seeding of the float primary samples: components [b0, b0 + 2) are the active ones
struct SeedGrad {
    const float *p;
    int b0;
    __device__ __forceinline__ Dual<2> operator()(int k) const {
        Dual<2> r = MakeDual<2>(p[k]);
        const int v = k - 1;  // primary[0] is the (inactive) time
        r.d[0] = v == b0 ? 1.0f : 0.0f, r.d[1] = v == b0 + 1 ? 1.0f : 0.0f;
        return r;
    }
};
template <int LCLASS>
__device__ __forceinline__ void GradPass(int c, int l, int b0, const float *base, const float *scene, float *o) {
    const LdsIn vp{base + H2_REC_VP};
    const SeedGrad prim{base, b0};
    const Dual<2> r = PathProgramP<Dual<2>, LdsIn, SeedGrad, LCLASS>(c, l, prim, scene, vp);
    if (b0 == 0) o[MG_OUT_LOGLUM] = r.v;
    o[b0] = r.d[0], o[b0 + 1] = r.d[1];
}

}  // namespace

// 17 KB of records per wave: 13 states of dim 6 (314 words + pad), 8 of dim 12 (491).  Not more: with the task table a block stays under
// 20 KB, so two of its waves fit a SIMD -- and one fits BESIDE a resident wave of the hot launch (256 registers, 14 KB); compiled for one wave
// per SIMD (406 registers) the launch waited for a SIMD to drain, i.e. for the end of the hot launch (profiles/r04_fill_l_*)
constexpr int MG_LDS_WORDS = 4352;

struct SceneBlock38G {  // the serialised scene block (scene.cpp:164-169) as a kernel argument: uniform, read with scalar loads
    float v[38];
};
__global__ void __launch_bounds__(64, 2) k_mala_grad(const float *__restrict__ rec, H2Bins bins, int N, SceneBlock38G sceneArg, float *__restrict__ gout) {
    const float *scene = sceneArg.v;
    __shared__ float lds[MG_LDS_WORDS];
    __shared__ int taskIncl[H2_NBINS];
    const int lane = threadIdx.x;
    auto recWordsOf = [](int t) {
        int c, l;
        H2TechOf(t, c, l);
        return (H2_REC_VP + 238 + 59 * (c + l - 3) + 3) & ~3;
    };
    auto ipwOf = [&](int t) { return min(64 / (H2TechDim(t) / 2), MG_LDS_WORDS / recWordsOf(t)); };
    const int total = H2BuildTaskTable(bins.count, taskIncl, ipwOf);
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        const int wr = total - 1 - w;  // longest programs first
        const int bin = H2BinOfTask(taskIncl, wr);
        const int t = bin / H2_NSIG;
        const int j = wr - (bin ? taskIncl[bin - 1] : 0);
        const int tIpw = ipwOf(t), tRecW = recWordsOf(t), tCnt = bins.count[bin];
        const int passes = H2TechDim(t) / 2;
        const int first = j * tIpw, n = min(tIpw, tCnt - first);
        int c, l;
        H2TechOf(t, c, l);
        const int *items = bins.items + bins.start[bin] + first;
        for (int s = 0; s < n; s++) {  // stage the records: whole lines
            const float *src = rec + (size_t)items[s] * H2_REC_WORDS;
            for (int k = lane; k < tRecW; k += 64) lds[s * tRecW + k] = src[k];
        }
        __syncthreads();
        const int slot = lane / passes;
        if (slot < n) {
            const int b0 = 2 * (lane - slot * passes);
            const float *base = lds + slot * tRecW;
            float *o = gout + (size_t)items[slot] * MG_OUT_WORDS;
            // three copies of the program by what the technique keeps alive (pathfunc.h PathProgramP): the same arithmetic, fewer live registers
            if (l <= 1) GradPass<0>(c, l, b0, base, scene, o);
            else if (c == 1)
                GradPass<1>(c, l, b0, base, scene, o);
            else
                GradPass<2>(c, l, b0, base, scene, o);
        }
        __syncthreads();
    }
}

void LaunchMalaGrad(const float *rec, const H2Bins &bins, int N, const float *scene38 /* host */, float *gout, int gridBlocks, hipStream_t s) {
    SceneBlock38G sc;
    for (int k = 0; k < 38; k++) sc.v[k] = scene38[k];
    hipLaunchKernelGGL(k_mala_grad, dim3(gridBlocks), dim3(64), 0, s, rec, bins, N, sc, gout);
}
