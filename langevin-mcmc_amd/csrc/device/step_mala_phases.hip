// The cache-filling small steps as a pipeline of launches: MALASmallStep::Mutate (/root/reference/src/mutation_mala.h:38-290) inside the
// chain loop body of mlt.cpp:91-170 -- the small-step branch of dstep.h StepChain, statement for statement -- cut at the two places where
// a state's gradient is needed (mutation_mala.h:94-130 current, :188-222 proposal), so that the path program runs wave-cooperatively in
// between (gradcoop.hip; layout and hand-off state: dh2coop.h MalaPipe):
//   k_mala_begin   the uniform-mixing draw, the step's normal draws z; a current state that needs its gradient is serialised -> stage 0
//   [k_mala_grad on stage 0]
//   k_mala_mid     the current Gaussian (InitGaussianFor with the delivered gradient), offset = covL z + mean, py; PerturbPathBidir; the
//                  proposal is stored in the chain's second path buffer and, if it needs its gradient, serialised -> stage 1
//   [k_mala_grad on stage 1]
//   k_mala_finish  the proposal's Gaussian, px, acceptance, splats, accept / reject (mlt.cpp:103-170), the next step's kind
// While a dimension's cache fills, a chain of that dimension evaluates one or two gradients per small step; as a single lane-per-chain launch
// beside the hot launch those few chains held SIMD slots for 2 ms per step.  The RNG order of a chain is the reference's:
// nothing between these draws consumes numbers.
#ifndef LMC_NO_RNG_JUMP_LDS
#define LMC_RNG_JUMP_LDS  // drng.h: the PCG jump constants of this launch live in LDS
#endif
#include "dpipe.h"

using namespace lmcd;

namespace {

// does InitGaussianFor evaluate the path program for this state?  (dstep.h: the branch `inRange && !ready && haveDerv`, then sp.ssScore > 1e-10)
__device__ __forceinline__ bool WantsGradient(const DCache &cache, const StepParams &P, int c, int l, float ssScore) {
    return NeedsGradient(cache, P, c, l) && ssScore > 1e-10f && !LMC_EXP(P.expFlags, 4);
}

}  // namespace

__global__ void __launch_bounds__(64) k_mala_begin(DScene S, const DCache *cachePtr, ChainArrays A, StepParams P, MalaPipe M, const int *list, const int *listCount) {
    if ((int)(blockIdx.x * blockDim.x) >= *listCount) return;  // a block past the end of the work list: nothing to set up, nothing to do
    LMC_RNG_JUMP_INIT();
    const DCache &cache = *cachePtr;
    const int total = *listCount, N = A.N;
    const float sigma = S.opt.perturbStdDev;
    // every lane takes part in the wave-level enqueue, also past the end of the list
    for (int j0 = blockIdx.x * blockDim.x; j0 < total; j0 += gridDim.x * blockDim.x) {
        const int j = j0 + threadIdx.x;
        bool want = false;
        int t = 0, i = 0;
        if (j < total) {
            i = list[j];
            Rng rng = LoadChainRng(A, P.chainBegin, S.opt.seedOffset, i);
            const int flags = A.flags[i];
            const int c = __float_as_int(A.curContrib[i]), l = __float_as_int(A.curContrib[(size_t)N + i]);
            const float curSs = A.curContrib[(size_t)8 * N + i];
            const int dim = PathDimension(c, l);
            const bool mala = S.opt.mala && !(rng.Uniform() < S.opt.uniformMixingProbability);  // mutation_mala.h:46-51
            // GenerateSample (gaussian.cpp:38-55) draws z ~ N(0, 1); the uniform-mixing step draws N(0, sigma) = z sigma + 0 (mutation_small.h:34-37)
            NormalDist nd(0.0f, 1.0f);
            for (int k = 0; k < dim; k++) {
                const float z = nd(rng);
                M.offset[(size_t)k * N + i] = mala ? z : z * sigma + 0.0f;
            }
            int bits = mala ? MS_MALA : 0;
            if (mala && !(flags & F_GAUSS) && WantsGradient(cache, P, c, l, curSs)) {
                DPath path;
                LoadPath(CurPathBuf(A, flags), N, i, path);
                H2Serialize(S, path, M.rec + (size_t)i * H2_REC_WORDS);
                want = true, t = H2BinIndex(H2TechIndex(c, l), H2MaterialSignature(S, path));
                bits |= MS_GRAD_CUR;
            }
            M.step[i] = bits;
            StoreChainRng(A, i, rng);
        }
        H2Enqueue(M.bins[0], N, want, t, j < total ? i : -1);
    }
}

#ifndef LMC_MALA_MID_WAVES
#define LMC_MALA_MID_WAVES 2  // waves per SIMD the register allocation aims at (A/B: profiles/r04_fill_w_*)
#endif
template <bool LDS_STACK, bool GLOSSY>
__global__ void __launch_bounds__(64, LMC_MALA_MID_WAVES) k_mala_mid(DScene S, const DCache *cachePtr, ChainArrays A, StepParams P, MalaPipe M, const int *list, const int *listCount) {
    if ((int)(blockIdx.x * blockDim.x) >= *listCount) return;  // a block past the end of the work list: nothing to set up, nothing to do
    LMC_RNG_JUMP_INIT();
    LMC_MAT_LDS_INIT(S);
    extern __shared__ int ldsStack[];
    const DCache &cache = *cachePtr;
    StepStats st;
    const int total = *listCount, N = A.N;
    for (int j0 = blockIdx.x * blockDim.x; j0 < total; j0 += gridDim.x * blockDim.x) {
        const int j = j0 + threadIdx.x;
        bool want = false;
        int t = 0, i = 0;
        if (j < total) {
            i = list[j];
            Rng rng = LoadChainRng(A, P.chainBegin, S.opt.seedOffset, i);
            int flags = A.flags[i];
            int bits = M.step[i];
            const bool mala = bits & MS_MALA;
            const Contrib cur = LoadContrib(A.curContrib, A.N, i);
            DPath prop;
            LoadPath(CurPathBuf(A, flags), N, i, prop);  // proposalState.path = currentState.path
            const int dim = PathDimension(prop.camDepth, prop.lgtDepth);
            float offset[MAXPSS];
            for (int k = 0; k < dim; k++) offset[k] = M.offset[(size_t)k * N + i];
            GradWork gw{nullptr, 0, 0};
            if (mala) {
                if (!(flags & F_BUFFERED)) {  // mutation_mala.h:59-81; the vectors are zero already (dchain.h ClearBuffered)
                    flags |= F_BUFFERED;
                    flags &= ~F_QUERIED;
                }
                flags |= F_VDIRTY;  // InitGaussianFor writes chain->pss (and the moment vectors) unconditionally
                Gauss cg;
                if (!(flags & F_GAUSS)) {
                    InitGaussianFor<false>(S, cache, A, P, i, prop, cur, false, flags, cg, gw, st, (bits & MS_GRAD_CUR) ? M.gout + (size_t)i * MG_OUT_WORDS : nullptr);
                    StoreGauss(A, i, dim, flags, cg);
                    flags = (flags | F_GAUSS) & ~F_GAUSS_ISO;
                } else {
                    LoadGauss(S, A, i, dim, flags, cg);
                }
                for (int k = 0; k < dim; k++) offset[k] = cg.covL[k] * offset[k] + cg.mean[k];  // GenerateSample, gaussian.cpp:38-55
                for (int k = 0; k < dim; k++) M.offset[(size_t)k * N + i] = offset[k];
                M.py[i] = GaussianLogPdf(dim, offset, false, cg);
            }
            Contrib pc;
            pc.camDepth = pc.lightDepth = 0;
            pc.lsScore = pc.ssScore = 0.f;
            pc.screenPos = V2{0.f, 0.f};
            pc.contrib = V3{0.f, 0.f, 0.f};
            bool ok;
            if constexpr (LDS_STACK) {
                LdsStackT<GLOSSY> stk{ldsStack + threadIdx.x, (int)blockDim.x, 0};
                ok = PerturbPathBidir(S, offset, prop, pc, rng, stk);
            } else {
                LocalStackT<GLOSSY> stk;
                ok = PerturbPathBidir(S, offset, prop, pc, rng, stk);
            }
            if (ok) {
                bits |= MS_OK;
                ToSubpath(pc.camDepth, pc.lightDepth, prop);  // a small step keeps (c, l): nothing changes
                StorePath(PropPathBuf(A, flags), N, i, prop);
                StoreContrib(M.propContrib, N, i, pc);
                if (mala && WantsGradient(cache, P, prop.camDepth, prop.lgtDepth, pc.ssScore)) {
                    H2Serialize(S, prop, M.rec + (size_t)i * H2_REC_WORDS);
                    want = true, t = H2BinIndex(H2TechIndex(prop.camDepth, prop.lgtDepth), H2MaterialSignature(S, prop));
                    bits |= MS_GRAD_PROP;
                }
            }
            A.flags[i] = flags;
            M.step[i] = bits;
            StoreChainRng(A, i, rng);
        }
        H2Enqueue(M.bins[1], N, want, t, j < total ? i : -1);
    }
    __shared__ int sStats[9];
    BlockReduceStats(st, A.counters, A.weightSum, sStats);
}

__global__ void __launch_bounds__(64) k_mala_finish(DScene S, const DCache *cachePtr, ChainArrays A, Film film, StepParams P, MalaPipe M, const int *list, const int *listCount) {
    if ((int)(blockIdx.x * blockDim.x) >= *listCount) return;  // a block past the end of the work list: nothing to set up, nothing to do
    LMC_RNG_JUMP_INIT();
    const DCache &cache = *cachePtr;
    StepStats st;
    const int total = *listCount;
    const size_t N = A.N;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int i = list[j];
        Rng rng = LoadChainRng(A, P.chainBegin, S.opt.seedOffset, i);
        int flags = A.flags[i];
        const int bits = M.step[i];
        const bool curValid = flags & F_VALID, mala = bits & MS_MALA;
        const Contrib cur = LoadContrib(A.curContrib, A.N, i);
        Contrib pc;
        pc.camDepth = pc.lightDepth = 0;
        pc.lsScore = pc.ssScore = 0.f;
        pc.screenPos = V2{0.f, 0.f};
        pc.contrib = V3{0.f, 0.f, 0.f};
        float a = 0.0f;
        Gauss pg;
        st.steps++;
        if (bits & MS_OK) {
            pc = LoadContrib(M.propContrib, A.N, i);
            if (mala) {
                DPath prop;
                LoadPath(PropPathBuf(A, flags), (int)N, i, prop);
                const int dim = PathDimension(prop.camDepth, prop.lgtDepth);
                GradWork gw{nullptr, 0, 0};
                InitGaussianFor<false>(S, cache, A, P, i, prop, pc, true, flags, pg, gw, st, (bits & MS_GRAD_PROP) ? M.gout + (size_t)i * MG_OUT_WORDS : nullptr);
                float offset[MAXPSS];
                for (int k = 0; k < dim; k++) offset[k] = M.offset[(size_t)k * N + i];
                const float py = M.py[i];
                const float px = GaussianLogPdf(dim, offset, true, pg);
                a = Clampf(lexpf(px - py) * pc.ssScore / cur.ssScore, 0.0f, 1.0f);
            } else {
                a = Clampf(pc.ssScore / cur.ssScore, 0.0f, 1.0f);
            }
        }
        // ---- splats, mlt.cpp:103-112
        if (curValid && a < 1.0f) {
            const int n = A.curSplatCount[i];
            for (int k = 0; k < n; k++) {
                const float *p = A.curSplat + ((size_t)k * SPLAT_WORDS) * N + i;
                Splat(film, V2{p[0], p[N]}, (1.0f - a) * V3{p[2 * N], p[3 * N], p[4 * N]});
            }
        }
        // mutation_small.h:48 `contrib * (normalization / lsScore)` vs mutation_mala.h:271 `contrib * normalization / lsScore` (different rounding, kept)
        const V3 smallSplat = mala ? (pc.contrib * P.normalization) / pc.lsScore : pc.contrib * (P.normalization / pc.lsScore);
        if (a > 0.0f) Splat(film, pc.screenPos, a * smallSplat);
        st.wsum += curValid ? 1.0f : (a > 0.0f ? a : 0.0f);
        // ---- accept / reject, mlt.cpp:113-170
        const int sampleIdx = A.sampleIdx[i];
        A.pushDim[i] = 0;
        if (a > 0.0f && rng.Uniform() <= a) {
            st.accepted++;
            flags ^= F_SEL;  // the proposal's path buffer (filled by k_mala_mid) is the current one now
            StoreContrib(A.curContrib, A.N, i, pc);
            A.adjacentReject[i] = 0;
            float *p = A.curSplat + i;
            p[0] = pc.screenPos.x, p[N] = pc.screenPos.y, p[2 * N] = smallSplat.x, p[3 * N] = smallSplat.y, p[4 * N] = smallSplat.z;
            A.curSplatCount[i] = 1;
            if (mala) {  // mlt.cpp:133-142: chain.v1/v2 = prop_new_v1/v2 (whole vectors, whichever branch filled them last)
                for (int k = 0; k < MAXPSS; k++) {
                    A.chV1[(size_t)k * N + i] = A.chPropNewV1[(size_t)k * N + i];
                    A.chV2[(size_t)k * N + i] = A.chPropNewV2[(size_t)k * N + i];
                }
                flags = (flags | F_BUFFERED | F_GAUSS) & ~F_GAUSS_ISO;
                StoreGauss(A, i, PathDimension(pc.camDepth, pc.lightDepth), flags, pg);
            } else {
                flags &= ~(F_GAUSS | F_GAUSS_ISO);  // proposalState.gaussianInitialized = false, mutation_small.h:39
            }
            flags |= F_VALID;
        } else {
            int rej = A.adjacentReject[i] + 1;  // REMOVE_OUTLIERS, mlt.cpp:147-169
            A.adjacentReject[i] = rej;
            const bool strongReject = cur.lsScore > OUTLIER_RATIO_THRESHOLD * P.normalization;
            if (OutlierReset(rej, strongReject, P.expFlags)) {
                ResetToInitState(A, P.chainBegin, P.numChains, OUTLIER_RATIO_THRESHOLD * P.normalization, i, sampleIdx, CurPathBuf(A, flags));
                A.curSplatCount[i] = 0;
                flags &= ~(F_VALID | F_GAUSS | F_GAUSS_ISO);
                ClearBuffered(A, i, flags);
                st.resets++;
            }
        }
        A.flags[i] = flags & ~F_VSYNC;  // these kernels do not track the v1 / v2 equality (dchain.h)
        A.sampleIdx[i] = sampleIdx + 1;
        QueueNext(S, cache, A, P, i, rng);
        StoreChainRng(A, i, rng);
    }
    // the stages' bin counts are zero again for the next step (both stages have been consumed: this launch is queued behind them).  No fill
    // launch in front of k_mala_begin: queued beside the hot launch, that small launch waited 0.25 ms for a slot (profiles/r04_fill_s_*)
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < H2_COUNT_WORDS; k += blockDim.x) M.bins[0].count[k] = 0, M.bins[1].count[k] = 0;
    __shared__ int sStats[9];
    BlockReduceStats(st, A.counters, A.weightSum, sStats);
}

void LaunchMalaBegin(const DScene &S, const DCache *cache, const ChainArrays &A, const StepParams &P, const MalaPipe &M, const int *list, const int *listCount, int gridBlocks,
                     hipStream_t s) {
    hipLaunchKernelGGL(k_mala_begin, dim3(gridBlocks), dim3(64), 0, s, S, cache, A, P, M, list, listCount);
}
void LaunchMalaMid(const DScene &S, const DCache *cache, const ChainArrays &A, const StepParams &P, const MalaPipe &M, const int *list, const int *listCount, int bvhStackNeed,
                   bool glossy, int gridBlocks, hipStream_t s) {
    if (bvhStackNeed <= BVH_LDS_STACK) {
        const size_t ldsBytes = (size_t)64 * ((bvhStackNeed + 7) / 8 * 8) * sizeof(int);
        if (glossy) hipLaunchKernelGGL((k_mala_mid<true, true>), dim3(gridBlocks), dim3(64), ldsBytes, s, S, cache, A, P, M, list, listCount);
        else
            hipLaunchKernelGGL((k_mala_mid<true, false>), dim3(gridBlocks), dim3(64), ldsBytes, s, S, cache, A, P, M, list, listCount);
    } else {
        if (glossy) hipLaunchKernelGGL((k_mala_mid<false, true>), dim3(gridBlocks), dim3(64), 0, s, S, cache, A, P, M, list, listCount);
        else
            hipLaunchKernelGGL((k_mala_mid<false, false>), dim3(gridBlocks), dim3(64), 0, s, S, cache, A, P, M, list, listCount);
    }
}
void LaunchMalaFinish(const DScene &S, const DCache *cache, const ChainArrays &A, const Film &film, const StepParams &P, const MalaPipe &M, const int *list, const int *listCount,
                      int gridBlocks, hipStream_t s) {
    hipLaunchKernelGGL(k_mala_finish, dim3(gridBlocks), dim3(64), 0, s, S, cache, A, film, P, M, list, listCount);
}
