// k_h2_hess: gradient and Hessian of log f for the H2MC proposal (the reference's evaluate_path_bidir_<c>_<l>_static_derv programs,
// generator /root/reference/src/path.cpp:3419-3662, chad.cpp:333-544; caller mutation_h2mc.h:60-93), WAVE-COOPERATIVE:
// the lanes of a wave are the 2 x 2 blocks of the Hessian triangle of a few states of ONE technique (c,l) -- dim 12: 21 lanes per
// state, three states per wave -- instead of one lane walking all 21 passes of its own chain's state one after the other
// (round 3: one lane per chain, 4 M chain-steps/s).  All lanes of a state walk the same path: same branches, the state's serialised
// record is staged in LDS once per wave and read by broadcast; a lane evaluates its block in two passes of 6-float values (four of
// 4-float values for the techniques with both sub-paths), seeded on the fly: no indexable array of second-order values, hence no private
// memory for it.  A wave's states also share a material signature (the stage's bins, dh2coop.h), so they take the same BSDF branches.
#ifndef LMC_H2HESS_EXACT_MATH  // the strict build (scripts/build_h2strict.sh; tests/test_gpu_h2mc.py runs the chain-parity tests on it): neither
#define LMC_PF_CONTRACT  // the dual-number arithmetic of this translation unit may fuse a * b + c (pathfunc.h)
#define LMC_PF_FASTMATH  // ... and sin / cos / exp / log / pow are the hardware's approximate instructions
#endif
#ifndef LMC_NO_PF_VEC2
#define LMC_PF_VEC2  // the inner Dual<2> of the second-order type on two-wide vectors: packed FP32 instructions without the shuffles (pathfunc.h)
#endif
#include "dh2coop.h"
#include "pathfunc.h"
#include "kernels.h"

using namespace lmcd;

namespace {

// A lane owns the 2 x 2 block (rows bi0, bi0 + 1; columns bc0, bc0 + 1) of one state's Hessian triangle and evaluates it in
// (2 / R) x (2 / W) passes of the program in DualS<R, Dual<W>>: (1 + R)(1 + W) floats per value.  The shape is chosen per copy of the
// program (below): 1 x 2 where one sub-path's state is alive (6 floats per value: the fastest shape measured, profiles/r04_d_*, r04_s_*),
// 1 x 1 for the techniques that keep BOTH sub-paths' states alive across the camera loop -- with 6-float values that copy spills most
// and the launch waits for private memory half of the time on the area-lit scene (profiles/r04_final_h2mc_pmc_door.json).
#ifndef LMC_H2HESS_R
#define LMC_H2HESS_R 1
#endif
#ifndef LMC_H2HESS_W
#define LMC_H2HESS_W 2
#endif
#ifndef LMC_H2HESS_W_BOTH
#define LMC_H2HESS_W_BOTH 1  // columns per pass of the copy with both sub-paths
#endif

struct LdsIn {  // the state's vertParams in LDS
    const float *p;
    __device__ __forceinline__ float operator[](int k) const { return p[k]; }
};
// a lane's seeding of the float primary samples: rows [i0, i0 + R) on the outer level, columns [c0, c0 + W) on the inner one
template <int R, int W>
struct SeedPrim {
    typedef DualS<R, Dual<W>> T2;
    const float *p;
    int i0, c0;
    __device__ __forceinline__ T2 operator()(int k) const {
        T2 r = Lift<T2>::Of(p[k]);
        const int v = k - 1;  // primary[0] is the (inactive) time
        for (int q = 0; q < W; q++) r.v.d[q] = v == c0 + q ? 1.0f : 0.0f;
        for (int q = 0; q < R; q++) r.d[q].v = v == i0 + q ? 1.0f : 0.0f;
        return r;
    }
};

// one lane's block: LCLASS selects the copy of the program (pathfunc.h PathProgramP)
template <int R, int W, int LCLASS>
__device__ __forceinline__ void HessBlock(int c, int l, int dim, int bi0, int bc0, const float *base, const float *scene, float *o) {
    typedef DualS<R, Dual<W>> T2;
    const LdsIn vp{base + H2_REC_VP};
#pragma unroll 1
    for (int sub = 0; sub < (2 / R) * (2 / W); sub++) {
        const int i0 = bi0 + (sub / (2 / W)) * R, c0 = bc0 + (sub % (2 / W)) * W;
        if (i0 > c0 + W - 1) continue;  // a pass of a diagonal block that lies entirely below the diagonal: nobody reads it
        const SeedPrim<R, W> prim{base, i0, c0};
        const T2 res = PathProgramP<T2, LdsIn, SeedPrim<R, W>, LCLASS>(c, l, prim, scene, vp);
        if (i0 == 0 && c0 == 0) o[H2_OUT_LOGLUM] = res.v.v;
        for (int q = 0; q < R; q++) {
            if (bc0 == bi0) o[i0 + q] = res.d[q].v;  // the forward directional derivative: exact (the reference's `g`)
            for (int k = 0; k < W; k++) o[H2_OUT_HESS + (i0 + q) * dim + c0 + k] = res.d[q].d[k];
        }
    }
}

}  // namespace

// LDS per wave: the records of the states of one task, at most H2_HESS_LDS_WORDS floats (a task takes fewer states when they do not fit)
constexpr int H2_HESS_LDS_WORDS = 3328;  // 13 KB: 10 states of dim 6 (314 words + pad), 3 of dim 12 (491)

#ifndef LMC_H2HESS_WAVES
#define LMC_H2HESS_WAVES 1
#endif
struct SceneBlock38 {  // the serialised scene block (scene.cpp:164-169) as a kernel argument: uniform, read with scalar loads
    float v[38];
};
__global__ void __launch_bounds__(64, LMC_H2HESS_WAVES) k_h2_hess(const float *__restrict__ rec, H2Bins bins, int N, SceneBlock38 sceneArg, float *__restrict__ hout) {
    const float *scene = sceneArg.v;
    __shared__ float lds[H2_HESS_LDS_WORDS];
    const int lane = threadIdx.x;
    // tasks of a bin: ceil(count / states per wave); every wave derives the same table (dh2coop.h)
    __shared__ int taskIncl[H2_NBINS];
    auto recWordsOf = [](int t) {
        int c, l;
        H2TechOf(t, c, l);
        return (H2_REC_VP + 238 + 59 * (c + l - 3) + 3) & ~3;
    };
    auto ipwOf = [&](int t) { return min(64 / H2BlocksOfDim(H2TechDim(t)), H2_HESS_LDS_WORDS / recWordsOf(t)); };
    const int total = H2BuildTaskTable(bins.count, taskIncl, ipwOf);
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        const int wr = total - 1 - w;  // longest programs first
        const int bin = H2BinOfTask(taskIncl, wr);
        const int t = bin / H2_NSIG;
        const int j = wr - (bin ? taskIncl[bin - 1] : 0);
        const int tIpw = ipwOf(t), tRecW = recWordsOf(t), tCnt = bins.count[bin];
        const int dim = H2TechDim(t);
        const int tNb = H2BlocksOfDim(dim);
        const int first = j * tIpw, n = min(tIpw, tCnt - first);
        int c, l;
        H2TechOf(t, c, l);
        const int *items = bins.items + bins.start[bin] + first;
        for (int s = 0; s < n; s++) {  // stage the records: whole lines
            const float *src = rec + (size_t)items[s] * H2_REC_WORDS;
            for (int k = lane; k < tRecW; k += 64) lds[s * tRecW + k] = src[k];
        }
        __syncthreads();
        const int slot = lane / tNb;
        if (slot < n) {
            int b = lane - slot * tNb, r = 0;
            const int m = dim / 2;
            while (b >= m - r) b -= m - r, r++;
            const int bi0 = 2 * r, bc0 = 2 * (r + b);
            const float *base = lds + slot * tRecW;
            float *o = hout + (size_t)items[slot] * H2_OUT_WORDS;
            // three copies of the program by what the technique keeps alive: only the camera state (l <= 1), only the light state (c == 1),
            // both (the light state parked across the camera loop): the register allocation of the first two does not pay for the third
#ifdef LMC_H2HESS_ONECLASS
            HessBlock<LMC_H2HESS_R, LMC_H2HESS_W, -1>(c, l, dim, bi0, bc0, base, scene, o);
#else
            if (l <= 1) HessBlock<LMC_H2HESS_R, LMC_H2HESS_W, 0>(c, l, dim, bi0, bc0, base, scene, o);
            else if (c == 1)
                HessBlock<LMC_H2HESS_R, LMC_H2HESS_W, 1>(c, l, dim, bi0, bc0, base, scene, o);
            else
                HessBlock<LMC_H2HESS_R, LMC_H2HESS_W_BOTH, 2>(c, l, dim, bi0, bc0, base, scene, o);
#endif
        }
        __syncthreads();
    }
}

void LaunchH2Hess(const float *rec, const H2Bins &bins, int N, const float *scene38 /* host */, float *hout, int gridBlocks, hipStream_t s) {
    SceneBlock38 sc;
    for (int k = 0; k < 38; k++) sc.v[k] = scene38[k];
    hipLaunchKernelGGL(k_h2_hess, dim3(gridBlocks), dim3(64), 0, s, rec, bins, N, sc, hout);
}
