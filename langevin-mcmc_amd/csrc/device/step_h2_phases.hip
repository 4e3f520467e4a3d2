// The lane-per-chain phases of an H2MC small step: H2MCSmallStep::Mutate (/root/reference/src/mutation_h2mc.h:38-128) inside the chain
// loop body of mlt.cpp:91-170, cut at the two places where a state's proposal Gaussian is needed, so that the second-order path
// program and the eigen-solve run wave-cooperatively in between (h2hess.hip; layout and hand-off state: dh2coop.h):
//   k_h2_begin    the uniform-mixing draw, the step's normal draws z; a current state without a Gaussian is serialised -> stage 0
//   [k_h2_hess, k_h2_gauss on stage 0; k_h2_sample: offset = covL z + mean, py]
//   k_h2_perturb  PerturbPathBidir; the proposal is stored in the chain's second path buffer and serialised -> stage 1
//   [k_h2_hess, k_h2_gauss on stage 1: the proposal's Gaussian and px]
//   k_h2_finish   acceptance, splats, accept / reject (mlt.cpp:103-170), the next step's kind
// The RNG order of a chain is the reference's: nothing between these draws consumes numbers (the Gaussians draw nothing).
#ifndef LMC_NO_RNG_JUMP_LDS
#define LMC_RNG_JUMP_LDS  // drng.h: the PCG jump constants of this launch live in LDS
#endif
#include <cstdlib>
#include <cstring>

#include "dh2mc.h"
#include "dpipe.h"
#include "dwalk.h"

using namespace lmcd;

namespace {

__device__ __forceinline__ bool H2HaveDerv(const StepParams &P, int c, int l, int dim) {  // mutation_h2mc.h:63-66 (+ the library's range, path.cpp:4030-4037)
    return P.useGradient && GradAvailable(c, l) && c + l - 1 <= P.maxDervDepth && dim <= H2_MAXDIM;
}
// IsotropicGaussian(dim, sigma), gaussian.cpp:4-22: what a state without a derivative program gets (mutation_h2mc.h:62,92)
__device__ __forceinline__ float H2IsoLogDetNoDerv(int dim, float sigma) { return dim * fastlog(1.0f / (sigma * sigma)); }
// the isotropic outcome of ComputeGaussian (h2mc.cpp:84-92) as k_h2_gauss and the oracle spell it: n additions of log(1 / sigma^2)
__device__ __forceinline__ float H2IsoLogDetEarlyOut(int n, float sigma) {
    const float invSigmaSq = 1.0f / (sigma * sigma);
    float logDet = 0.f;
    for (int i = 0; i < n; i++) logDet += llogf(invSigmaSq);
    return logDet;
}
__device__ __forceinline__ float *H2GaussRec(const H2Arrays &H, int N, int i, bool second) { return H.gauss + (second ? (size_t)N * H2_GAUSS_AOS : 0) + (size_t)i * H2_GAUSS_AOS; }

}  // namespace

__global__ void __launch_bounds__(64) k_h2_begin(DScene S, ChainArrays A, StepParams P, H2Arrays H, const int *list, const int *listCount) {
    if ((int)(blockIdx.x * blockDim.x) >= *listCount) return;  // a block past the end of the work list: nothing to set up, nothing to do
    LMC_RNG_JUMP_INIT();
    StepStats st;
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
            int flags = A.flags[i];
            const int c = __float_as_int(A.curContrib[i]), l = __float_as_int(A.curContrib[(size_t)N + i]);
            const float curSs = A.curContrib[(size_t)8 * N + i];
            const int dim = PathDimension(c, l);
            const bool h2 = !(rng.Uniform() < S.opt.uniformMixingProbability);  // mutation_h2mc.h:49-55
            const bool useDense = dim <= H2_MAXDIM;  // longer states have no derivative program: isotropic, nothing stored
            const bool sample = h2 && useDense;
            // GenerateSample (gaussian.cpp:38-55) draws z ~ N(0, 1); the other kinds of step draw N(0, sigma) = z sigma + 0 (mutation_small.h:34-37)
            NormalDist nd(0.0f, 1.0f);
            for (int k = 0; k < dim; k++) {
                const float z = nd(rng);
                H.offset[(size_t)k * N + i] = sample ? z : sigma * z + 0.0f;
            }
            if (sample && !(flags & F_GAUSS)) {  // initGaussian(currentState), mutation_h2mc.h:60-98
                float *G = H2GaussRec(H, N, i, (flags & F_GSEL) != 0);
                if (!H2HaveDerv(P, c, l, dim)) {
                    G[H2_GAUSS_LOGDET] = H2IsoLogDetNoDerv(dim, sigma), G[H2_GAUSS_LOGDET + 1] = (float)H2K_ISO_NODERV;
                } else {
                    if (curSs > 1e-15f) st.gradCalls++;
                    if (curSs > 1e-15f && !LMC_EXP(P.expFlags, 16)) {
                        // serialised straight from the SoA path buffer (dwalk.h): no DPath in private memory
                        const unsigned sig = H2SerializeStreamed(S, SoAPathView{CurPathBuf(A, flags), (size_t)N, i}, H.rec + (size_t)i * H2_REC_WORDS);
                        want = true, t = H2BinIndex(H2TechIndex(c, l), sig);
                    } else {  // zero gradient and Hessian: the early-out of ComputeGaussian
                        G[H2_GAUSS_LOGDET] = H2IsoLogDetEarlyOut(dim, sigma), G[H2_GAUSS_LOGDET + 1] = (float)H2K_ISO_EARLYOUT;
                    }
                }
                flags |= F_GAUSS;
                A.flags[i] = flags;
            }
            H.step[i] = (h2 ? H2S_H2 : 0) | (useDense ? H2S_DENSE : 0);
            H.kind[i] = sample ? 1 : 0;
            StoreChainRng(A, i, rng);
        }
        H2Enqueue(H.bins[0], N, want, t, j < total ? i : -1);
    }
    __shared__ int sStats[9];
    BlockReduceStats(st, A.counters, A.weightSum, sStats);
}

template <bool LDS_STACK>
__global__ void __launch_bounds__(64, 2) k_h2_perturb(DScene S, ChainArrays A, StepParams P, H2Arrays H, const int *list, const int *listCount) {
    if ((int)(blockIdx.x * blockDim.x) >= *listCount) return;  // a block past the end of the work list: nothing to set up, nothing to do
    LMC_RNG_JUMP_INIT();
    LMC_MAT_LDS_INIT(S);
    extern __shared__ int ldsStack[];
    StepStats st;
    const int total = *listCount, N = A.N;
    const float sigma = S.opt.perturbStdDev;
    for (int j0 = blockIdx.x * blockDim.x; j0 < total; j0 += gridDim.x * blockDim.x) {
        const int j = j0 + threadIdx.x;
        bool want = false;
        int t = 0, i = 0;
        if (j < total) {
            i = list[j];
            Rng rng = LoadChainRng(A, P.chainBegin, S.opt.seedOffset, i);
            const int flags = A.flags[i];
            int bits = H.step[i];
            DPath prop;
            LoadPath(CurPathBuf(A, flags), N, i, prop);  // proposalState.path = currentState.path
            const int dim = PathDimension(prop.camDepth, prop.lgtDepth);
            float offset[MAXPSS];
            for (int k = 0; k < dim; k++) offset[k] = H.offset[(size_t)k * N + i];
            Contrib pc;
            pc.camDepth = pc.lightDepth = 0;
            pc.lsScore = pc.ssScore = 0.f;
            pc.screenPos = V2{0.f, 0.f};
            pc.contrib = V3{0.f, 0.f, 0.f};
            bool ok;
            if constexpr (LDS_STACK) {
                LdsStackT<true> stk{ldsStack + threadIdx.x, (int)blockDim.x, 0};
                ok = PerturbPathBidir(S, offset, prop, pc, rng, stk);
            } else {
                LocalStackT<true> stk;
                ok = PerturbPathBidir(S, offset, prop, pc, rng, stk);
            }
            if (ok) {
                bits |= H2S_OK;
                ToSubpath(pc.camDepth, pc.lightDepth, prop);
                StorePath(PropPathBuf(A, flags), N, i, prop);
                StoreContrib(H.propContrib, N, i, pc);
                if ((bits & H2S_H2) && (bits & H2S_DENSE)) {  // initGaussian(proposalState), mutation_h2mc.h:100-102
                    float *G = H2GaussRec(H, N, i, (flags & F_GSEL) == 0);
                    float logDet = 0.f;
                    bool iso = true;
                    if (!H2HaveDerv(P, prop.camDepth, prop.lgtDepth, dim)) {
                        logDet = H2IsoLogDetNoDerv(dim, sigma);
                        G[H2_GAUSS_LOGDET] = logDet, G[H2_GAUSS_LOGDET + 1] = (float)H2K_ISO_NODERV;
                    } else {
                        if (pc.ssScore > 1e-15f) st.gradCalls++;
                        if (pc.ssScore > 1e-15f && !LMC_EXP(P.expFlags, 16)) {
                            H2Serialize(S, prop, H.rec + (size_t)i * H2_REC_WORDS);
                            want = true, t = H2BinIndex(H2TechIndex(prop.camDepth, prop.lgtDepth), H2MaterialSignature(S, prop)), iso = false;
                        } else {
                            logDet = H2IsoLogDetEarlyOut(dim, sigma);
                            G[H2_GAUSS_LOGDET] = logDet, G[H2_GAUSS_LOGDET + 1] = (float)H2K_ISO_EARLYOUT;
                        }
                    }
                    if (iso) {  // px = GaussianLogPdf(-offset, isotropic), the dense form of gaussian.cpp:24-36 with a diagonal matrix
                        const float inv = 1.0f / (sigma * sigma);
                        float q = 0.f;
                        for (int k = 0; k < dim; k++) {
                            const float d = -offset[k] - 0.0f;
                            q += d * (inv * d);
                        }
                        float logPdf = dim * (-0.9189385332046727f);
                        logPdf += 0.5f * logDet;
                        logPdf -= 0.5f * q;
                        H.px[i] = logPdf;
                    }
                }
            }
            H.step[i] = bits;
            StoreChainRng(A, i, rng);
        }
        H2Enqueue(H.bins[1], N, want, t, j < total ? i : -1);
    }
    __shared__ int sStats[9];
    BlockReduceStats(st, A.counters, A.weightSum, sStats);
}

// The same phase with the path streamed (dwalk.h): vertex by vertex from the chain's current path buffer into its other buffer, the record for
// stage 1 serialised from that buffer afterwards; offsets and traversal stack in LDS ([word][lane]); no DPath in private memory.  The form for
// every render whose tree fits the LDS stack and that does not use light-coordinate sampling (LMC_H2_PERTURB=generic selects the kernel above).
__global__ void __launch_bounds__(64, 2) k_h2_perturb_streamed(DScene S, ChainArrays A, StepParams P, H2Arrays H, const int *list, const int *listCount, int stackWords) {
    if ((int)(blockIdx.x * blockDim.x) >= *listCount) return;  // a block past the end of the work list: nothing to set up, nothing to do
    LMC_RNG_JUMP_INIT();
    LMC_MAT_LDS_INIT(S);
    extern __shared__ int ldsStack[];
    StepStats st;
    const int total = *listCount, N = A.N;
    const float sigma = S.opt.perturbStdDev;
    float *const ldsF = reinterpret_cast<float *>(ldsStack) + threadIdx.x;
    for (int j0 = blockIdx.x * blockDim.x; j0 < total; j0 += gridDim.x * blockDim.x) {
        const int j = j0 + threadIdx.x;
        bool want = false;
        int t = 0, i = 0;
        if (j < total) {
            i = list[j];
            Rng rng = LoadChainRng(A, P.chainBegin, S.opt.seedOffset, i);
            const int flags = A.flags[i];
            int bits = H.step[i];
            const int c = __float_as_int(A.curContrib[i]), l = __float_as_int(A.curContrib[(size_t)N + i]);
            const int dim = PathDimension(c, l);
            const float *cur = CurPathBuf(A, flags);
            float *prop = PropPathBuf(A, flags);
            for (int k = 0; k < dim; k++) ldsF[(stackWords + k) * 64] = H.offset[(size_t)k * N + i];
            Contrib pc;
            pc.camDepth = pc.lightDepth = 0;
            pc.lsScore = pc.ssScore = 0.f;
            pc.screenPos = V2{0.f, 0.f};
            pc.contrib = V3{0.f, 0.f, 0.f};
            LdsStackT<true> stk{ldsStack + threadIdx.x, 64, 0};
            const bool ok = PerturbPathStreamed(S, cur, prop, (size_t)N, i, c, l, LdsOffsets{ldsF, 64, stackWords}, rng, stk, pc);
            if (ok) {
                bits |= H2S_OK;
                StoreContrib(H.propContrib, N, i, pc);
                if ((bits & H2S_H2) && (bits & H2S_DENSE)) {  // initGaussian(proposalState), mutation_h2mc.h:100-102
                    float *G = H2GaussRec(H, N, i, (flags & F_GSEL) == 0);
                    float logDet = 0.f;
                    bool iso = true;
                    if (!H2HaveDerv(P, c, l, dim)) {
                        logDet = H2IsoLogDetNoDerv(dim, sigma);
                        G[H2_GAUSS_LOGDET] = logDet, G[H2_GAUSS_LOGDET + 1] = (float)H2K_ISO_NODERV;
                    } else {
                        if (pc.ssScore > 1e-15f) st.gradCalls++;
                        if (pc.ssScore > 1e-15f && !LMC_EXP(P.expFlags, 16)) {
                            const unsigned sig = H2SerializeStreamed(S, SoAPathView{prop, (size_t)N, i}, H.rec + (size_t)i * H2_REC_WORDS);
                            want = true, t = H2BinIndex(H2TechIndex(c, l), sig), iso = false;
                        } else {
                            logDet = H2IsoLogDetEarlyOut(dim, sigma);
                            G[H2_GAUSS_LOGDET] = logDet, G[H2_GAUSS_LOGDET + 1] = (float)H2K_ISO_EARLYOUT;
                        }
                    }
                    if (iso) {  // px = GaussianLogPdf(-offset, isotropic), the dense form of gaussian.cpp:24-36 with a diagonal matrix
                        const float inv = 1.0f / (sigma * sigma);
                        float q = 0.f;
                        for (int k = 0; k < dim; k++) {
                            const float d = -ldsF[(stackWords + k) * 64] - 0.0f;
                            q += d * (inv * d);
                        }
                        float logPdf = dim * (-0.9189385332046727f);
                        logPdf += 0.5f * logDet;
                        logPdf -= 0.5f * q;
                        H.px[i] = logPdf;
                    }
                }
            }
            H.step[i] = bits;
            StoreChainRng(A, i, rng);
        }
        H2Enqueue(H.bins[1], N, want, t, j < total ? i : -1);
    }
    __shared__ int sStats[9];
    BlockReduceStats(st, A.counters, A.weightSum, sStats);
}

__global__ void __launch_bounds__(64) k_h2_finish(DScene S, const DCache *cache, ChainArrays A, Film film, StepParams P, H2Arrays H, const int *list, const int *listCount) {
    if ((int)(blockIdx.x * blockDim.x) >= *listCount) return;  // a block past the end of the work list: nothing to set up, nothing to do
    LMC_RNG_JUMP_INIT();
    StepStats st;
    const int total = *listCount;
    const size_t N = A.N;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int i = list[j];
        Rng rng = LoadChainRng(A, P.chainBegin, S.opt.seedOffset, i);
        int flags = A.flags[i];
        const int bits = H.step[i];
        const bool curValid = flags & F_VALID, h2 = bits & H2S_H2, useDense = bits & H2S_DENSE;
        const Contrib cur = LoadContrib(A.curContrib, A.N, i);
        Contrib pc;
        pc.camDepth = pc.lightDepth = 0;
        pc.lsScore = pc.ssScore = 0.f;
        pc.screenPos = V2{0.f, 0.f};
        pc.contrib = V3{0.f, 0.f, 0.f};
        float a = 0.0f;
        st.steps++;
        if (bits & H2S_OK) {
            pc = LoadContrib(H.propContrib, A.N, i);
            if (h2) {
                // states of more than 16 dimensions: both Gaussians are the same isotropic one, px == py (mutation_h2mc.h:62,104-105)
                const float px = useDense ? H.px[i] : 0.0f, py = useDense ? H.py[i] : 0.0f;
                a = Clampf(lexpf(px - py) * pc.ssScore / cur.ssScore, 0.0f, 1.0f);
            } else {
                a = Clampf(pc.ssScore / cur.ssScore, 0.0f, 1.0f);
            }
        }
        // ---- splats, mlt.cpp:103-112 (both small-step flavours: contrib * (normalization / lsScore), mutation_h2mc.h:119-121)
        if (curValid && a < 1.0f) {
            const int n = A.curSplatCount[i];
            for (int k = 0; k < n; k++) {
                const float *p = A.curSplat + ((size_t)k * SPLAT_WORDS) * N + i;
                Splat(film, V2{p[0], p[N]}, (1.0f - a) * V3{p[2 * N], p[3 * N], p[4 * N]});
            }
        }
        const V3 smallSplat = pc.contrib * (P.normalization / pc.lsScore);
        if (a > 0.0f) Splat(film, pc.screenPos, a * smallSplat);
        st.wsum += curValid ? 1.0f : (a > 0.0f ? a : 0.0f);
        // ---- accept / reject, mlt.cpp:113-170
        const int sampleIdx = A.sampleIdx[i];
        A.pushDim[i] = 0;
        if (a > 0.0f && rng.Uniform() <= a) {
            st.accepted++;
            flags ^= F_SEL;  // the proposal's path buffer (filled by k_h2_perturb) is the current one now
            StoreContrib(A.curContrib, A.N, i, pc);
            A.adjacentReject[i] = 0;
            float *p = A.curSplat + i;
            p[0] = pc.screenPos.x, p[N] = pc.screenPos.y, p[2 * N] = smallSplat.x, p[3 * N] = smallSplat.y, p[4 * N] = smallSplat.z;
            A.curSplatCount[i] = 1;
            if (h2) {  // std::swap(currentState, proposalState): the proposal's Gaussian is the current one now
                if (useDense) flags ^= F_GSEL;
                flags |= F_GAUSS;
            } else {
                flags &= ~F_GAUSS;  // mutation_small.h:39
            }
            flags |= F_VALID;
        } else {
            int rej = A.adjacentReject[i] + 1;  // REMOVE_OUTLIERS, mlt.cpp:147-169
            A.adjacentReject[i] = rej;
            const bool strongReject = cur.lsScore > OUTLIER_RATIO_THRESHOLD * P.normalization;
            if (OutlierReset(rej, strongReject, P.expFlags)) {
                ResetToInitState(A, P.chainBegin, P.numChains, OUTLIER_RATIO_THRESHOLD * P.normalization, i, sampleIdx, CurPathBuf(A, flags));
                A.curSplatCount[i] = 0;
                flags &= ~(F_VALID | F_GAUSS | F_BUFFERED);
                st.resets++;
            }
        }
        A.flags[i] = flags & ~F_VSYNC;
        A.sampleIdx[i] = sampleIdx + 1;
        QueueNext(S, *cache, A, P, i, rng);
        StoreChainRng(A, i, rng);
    }
    __shared__ int sStats[9];
    BlockReduceStats(st, A.counters, A.weightSum, sStats);
}

void LaunchH2Begin(const DScene &S, const ChainArrays &A, const StepParams &P, const H2Arrays &H, const int *list, const int *listCount, int gridBlocks, hipStream_t s) {
    hipLaunchKernelGGL(k_h2_begin, dim3(gridBlocks), dim3(64), 0, s, S, A, P, H, list, listCount);
}
void LaunchH2Perturb(const DScene &S, const ChainArrays &A, const StepParams &P, const H2Arrays &H, const int *list, const int *listCount, int bvhStackNeed, int gridBlocks,
                     hipStream_t s) {
    const bool genericForm = getenv("LMC_H2_PERTURB") && !strcmp(getenv("LMC_H2_PERTURB"), "generic");
    if (bvhStackNeed <= BVH_LDS_STACK && !S.opt.useLightCoord && !genericForm) {
        const int stackWords = LeanStackWords(bvhStackNeed);
        const size_t ldsBytes = (size_t)64 * LeanLdsWordsPerThread(stackWords) * sizeof(int);
        hipLaunchKernelGGL(k_h2_perturb_streamed, dim3(gridBlocks), dim3(64), ldsBytes, s, S, A, P, H, list, listCount, stackWords);
    } else if (bvhStackNeed <= BVH_LDS_STACK) {
        const size_t ldsBytes = (size_t)64 * ((bvhStackNeed + 7) / 8 * 8) * sizeof(int);
        hipLaunchKernelGGL(k_h2_perturb<true>, dim3(gridBlocks), dim3(64), ldsBytes, s, S, A, P, H, list, listCount);
    } else {
        hipLaunchKernelGGL(k_h2_perturb<false>, dim3(gridBlocks), dim3(64), 0, s, S, A, P, H, list, listCount);
    }
}
void LaunchH2Finish(const DScene &S, const DCache *cache, const ChainArrays &A, const Film &film, const StepParams &P, const H2Arrays &H, const int *list, const int *listCount,
                    int gridBlocks, hipStream_t s) {
    hipLaunchKernelGGL(k_h2_finish, dim3(gridBlocks), dim3(64), 0, s, S, cache, A, film, P, H, list, listCount);
}
