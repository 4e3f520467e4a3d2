// H2MC small step on the device: H2MCSmallStep::Mutate (/root/reference/src/mutation_h2mc.h:38-128) inside the chain loop
// body of mlt.cpp:91-170.  One thread = one chain; all small steps of an H2MC render run here (large steps are the same as
// for LMC and keep their own launch; the global cache and the Adam-style moments do not exist on this path).
// Per state: gradient and Hessian of log f through the path program (pathfunc.h: PathFuncHess, one nested-dual pass per
// Hessian row), symmetric eigen-solve + Gaussian (dh2mc.h), dense sample x = covL z + mean, dense log pdf.
// The chain's current Gaussian (mean, covL, invCov, logDet: up to 16 + 2 * 256 + 1 words) lives in HBM (A.h2Gauss, SoA).
#pragma once
#include "dh2mc.h"
#include "dstep.h"

namespace lmcd {

constexpr int H2_GAUSS_WORDS = H2_MAXDIM + 2 * H2_MAXDIM * H2_MAXDIM + 1;  // mean | covL | invCov | logDet

// The dense Gaussian of a state, in HBM: two buffers of H2_GAUSS_WORDS words per chain (A.h2Gauss, SoA), the current state's
// and the proposal's, selected by F_GSEL like the path buffers by F_SEL (acceptance flips the bit).  Only the mean and the
// log-determinant are ever held in private memory: with both matrices of both Gaussians on the stack the kernel needed more
// than 16 KB of scratch per lane, which faults on gfx950 (scripts/debug).
struct H2Slot {
    float *base;  // word 0 of this chain in the selected buffer
    size_t stride;
    LMC_D float &Mean(int k) const { return base[(size_t)k * stride]; }
    LMC_D MatRef CovL() const { return MatRef{base + (size_t)H2_MAXDIM * stride, stride}; }
    LMC_D MatRef InvCov() const { return MatRef{base + (size_t)(H2_MAXDIM + H2_MAXDIM * H2_MAXDIM) * stride, stride}; }
    LMC_D float &LogDet() const { return base[(size_t)(H2_GAUSS_WORDS - 1) * stride]; }
};
LMC_D H2Slot H2Buf(const ChainArrays &A, int i, bool second) { return H2Slot{A.h2Gauss + (second ? (size_t)H2_GAUSS_WORDS * A.N : 0) + i, (size_t)A.N}; }

#ifdef __HIPCC__
template <class In>
#ifndef LMC_PF_ATTR
#define LMC_PF_ATTR
#endif
__device__ __noinline__ LMC_PF_ATTR void PathFuncHessPassDevice(int c, int l, const float *primary, const float *scene, const In &vp, int i, int c0, float *logLum,
                                                                float *grad, float *hess, bool firstOfRow) {
    PathFuncHessPass(c, l, primary, scene, vp, i, c0, logLum, grad, hess, firstOfRow);
}
// all passes: ONE copy of the second-order program per kernel image (the single-call plugin kernel runs the passes side by side, one
// per lane, through the same copy)
template <class In>
__device__ __noinline__ LMC_PF_ATTR void PathFuncHessDevice(int c, int l, const float *primary, const float *scene, const In &vp, float *logLum, float *grad, float *hess) {
    const int dim = 2 * (c + l - 1 > 2 ? c + l - 1 : 2);
    for (int i = 0; i < dim; i++)
        for (int c0 = 0; c0 < dim; c0 += HC) PathFuncHessPassDevice(c, l, primary, scene, vp, i, c0, logLum, grad, hess, c0 == 0);
}
// What the H2MC step needs of it: h2mc.cpp:78 hands the rows to Eigen as a column-major matrix and SelfAdjointEigenSolver reads its
// lower triangle = the UPPER triangle of the rows as delivered (dh2mc.h mirrors exactly that), so row i is only evaluated from its
// diagonal on, in blocks of R rows x W columns that cover the upper triangle (dim 12, 2 x 2: 21 passes of 9 floats per value
// against round 2's 24 of 18).  The entries below the diagonal are read by two tests only, the all-finite test
// (mutation_h2mc.h:80-84) and the Frobenius norm of the early-out (h2mc.cpp:84-92, `hnorm < 0.5 / sigma^2`): they are filled with
// the mirror image of the upper triangle.  The reference's matrix is asymmetric where chad's adjoint overwrite is active
// (DESIGN.md §2), so that norm is the norm of the symmetrised matrix here: the early-out can differ for a state whose norm sits
// within the asymmetry of the threshold.  Leaving them ZERO instead halves the norm and costs 6 % of the chains their agreement
// with the oracle within 30 steps (profiles/r03_r_h2mc_upper_triangle.txt).
// Block shape of a pass, DualS<R, Dual<W>> = (1 + R)(1 + W) floats per value: 2 x 2 measured best (21 passes of 9 floats for
// dim 12; profiles/r03_v_ab_h2mc_hessian_blocks.txt: 1x4 74.1 ms per step, 2x4 68.4, 2x3 68.0, 3x2 69.0, 4x2 68.4, 3x3 71.0, 1x2 80.1,
// 2x2 65.3; 4x4 needs more than the ~16 KB of private memory per lane at which gfx950 faults)
#ifndef LMC_HESS_ROW_CHUNK
#define LMC_HESS_ROW_CHUNK 2  // W: columns per pass
#endif
#ifndef LMC_HESS_ROW_BLOCK
#define LMC_HESS_ROW_BLOCK 2  // R: rows per pass
#endif
template <class In>
__device__ __noinline__ LMC_PF_ATTR void PathFuncHessRowPassDevice(int c, int l, const float *primary, const float *scene, const In &vp, int i0, int c0, float *logLum,
                                                                   float *grad, float *hess) {
    PathFuncHessRowPass<LMC_HESS_ROW_BLOCK, LMC_HESS_ROW_CHUNK>(c, l, primary, scene, vp, i0, c0, logLum, grad, hess);
}
template <class In>
__device__ __noinline__ LMC_PF_ATTR void PathFuncHessUpperDevice(int c, int l, const float *primary, const float *scene, const In &vp, float *logLum, float *grad, float *hess) {
    const int dim = 2 * (c + l - 1 > 2 ? c + l - 1 : 2);
    for (int i0 = 0; i0 < dim; i0 += LMC_HESS_ROW_BLOCK) {
        for (int c0 = i0; c0 < dim; c0 += LMC_HESS_ROW_CHUNK) PathFuncHessRowPassDevice(c, l, primary, scene, vp, i0, c0, logLum, grad, hess);
        for (int i = i0; i < dim && i < i0 + LMC_HESS_ROW_BLOCK; i++)  // after the block's passes: they also wrote the block's own below-diagonal entries
            for (int k = 0; k < i; k++) hess[i * dim + k] = hess[k * dim + i];
    }
}
#endif

constexpr int H2_LDS_DIM = 12;  // H2MC differentiates states of up to 16 dimensions; up to 12 (every state of the shipped scenes' depth 8 .. 6) solve in LDS
#ifdef __HIPCC__
// out of line: its eigen-solve work space then shares stack with the (already returned) path program instead of adding to it
__device__ __noinline__ LMC_PF_ATTR void ComputeGaussianH2MCDevice(const H2MCParam &param, int n, float sc, const float *grad, float *hess, float *mean, MatRef covL,
                                                       MatRef invCov, float &logDet) {
    float work[H2_MAXDIM * H2_MAXDIM + 4 * H2_MAXDIM];
    // the matrix the Jacobi rotations work on lives in LDS, [entry][thread] (H2_LDS_DIM^2 words per thread, allocated by the launch):
    // a rotation reads and writes 4 n of its entries with run-time indices, which in private memory is a dependent round trip to
    // HBM-backed scratch each -- the eigen-solve was 25 of the step's 80 ms (profiles/r03_u_h2mc_ablation.txt)
    extern __shared__ float h2Lds[];
    if (n <= H2_LDS_DIM) {
        MatRef A{h2Lds + threadIdx.x, blockDim.x};
        for (int k = 0; k < n * n; k++) A[k] = hess[k];
        ComputeGaussianH2MCT(param, n, sc, grad, A, mean, covL, invCov, logDet, work);
        return;
    }
    ComputeGaussianH2MC(param, n, sc, grad, hess, mean, covL, invCov, logDet, work);
}
#endif

// initGaussian lambda of H2MCSmallStep::Mutate (mutation_h2mc.h:60-93): fills `slot` (HBM) and returns mean / logDet
LMC_D void InitGaussianH2MC(const DScene &S, const StepParams &P, const H2MCParam &param, const DPath &path, const Contrib &sp, const H2Slot &slot, float *mean,
                            float &logDet, GradWork &gw, StepStats &st) {
    const int dim = PathDimension(path.camDepth, path.lgtDepth);
    const bool haveDerv = P.useGradient && GradAvailable(path.camDepth, path.lgtDepth) && path.camDepth + path.lgtDepth - 1 <= P.maxDervDepth && dim <= H2_MAXDIM;
    MatRef covL = slot.CovL(), invCov = slot.InvCov();
    if (!haveDerv) {  // IsotropicGaussian(dim, sigma), gaussian.cpp:4-22
        const float sigma = param.sigma;
        for (int i = 0; i < dim; i++) {
            mean[i] = 0.f;
            for (int j = 0; j < dim; j++) covL[i * dim + j] = (i == j) ? sigma : 0.f, invCov[i * dim + j] = (i == j) ? 1.0f / (sigma * sigma) : 0.f;
        }
        logDet = dim * fastlog(1.0f / (sigma * sigma));
    } else {
        float vGrad[H2_MAXDIM], vHess[H2_MAXDIM * H2_MAXDIM];
        for (int k = 0; k < dim; k++) vGrad[k] = 0.f;
        for (int k = 0; k < dim * dim; k++) vHess[k] = 0.f;
        if (sp.ssScore > 1e-15f) {
            float primary[2 * MAXD + 1];
            StridedOut o{gw.buf + gw.slot, gw.stride, 0};
            SerializePath(S, path, primary, o);
            StridedIn vin{gw.buf + gw.slot, gw.stride};
            float logLum;
            if (!(P.expFlags & 16)) PathFuncHessUpperDevice(path.camDepth, path.lgtDepth, primary, S.sceneParams, vin, &logLum, vGrad, vHess);
            st.gradCalls++;
            bool finite = true;
            for (int k = 0; k < dim; k++) finite = finite && isfinite(vGrad[k]);
            for (int k = 0; k < dim * dim; k++) finite = finite && isfinite(vHess[k]);
            if (!finite) {
                for (int k = 0; k < dim; k++) vGrad[k] = 0.f;
                for (int k = 0; k < dim * dim; k++) vHess[k] = 0.f;
            }
        }
        ComputeGaussianH2MCDevice(param, dim, (P.expFlags & 32) ? 0.f : sp.ssScore, vGrad, vHess, mean, covL, invCov, logDet);  // sc = 0: isotropic early-out, no eigen-solve
    }
    for (int k = 0; k < dim; k++) slot.Mean(k) = mean[k];
    slot.LogDet() = logDet;
}

template <class Stk>
LMC_D void StepChainH2MC(const DScene &S, const ChainArrays &A, const Film &film, const StepParams &P, int i, Rng &rng, GradWork &gw, StepStats &st, Stk &stk) {
    const size_t N = A.N;
    int flags = A.flags[i];
    const bool curValid = flags & F_VALID;
    const Contrib cur = LoadContrib(A.curContrib, A.N, i);
    DPath prop;
    Contrib pc;
    pc.camDepth = pc.lightDepth = 0;
    pc.lsScore = pc.ssScore = 0.f;
    pc.screenPos = V2{0.f, 0.f};
    pc.contrib = V3{0.f, 0.f, 0.f};
    float a = 1.0f;
    st.steps++;
    LoadPath(CurPathBuf(A, flags), A.N, i, prop);  // proposalState.path = currentState.path
    const int dim = PathDimension(prop.camDepth, prop.lgtDepth);
    float offset[MAXPSS];
    const bool h2 = !(rng.Uniform() < S.opt.uniformMixingProbability);  // mutation_h2mc.h:49-55
    const H2MCParam param = MakeH2MCParam(S.opt.perturbStdDev);
    float mean[H2_MAXDIM], logDet = 0.f, py = 0.f;  // of the Gaussian in use: the current state's until py is known, then the proposal's
    const bool gsel = (flags & F_GSEL) != 0;
    const bool useDense = dim <= H2_MAXDIM;  // longer states have no derivative program: isotropic, nothing stored
    if (!h2) {  // SmallStep::Mutate, mutation_small.h:16-56
        NormalDist nd(0.0f, S.opt.perturbStdDev);
        for (int k = 0; k < dim; k++) offset[k] = nd(rng);
    } else {
        const H2Slot cs = H2Buf(A, i, gsel);
        if (useDense) {
            if (!(flags & F_GAUSS)) {
                InitGaussianH2MC(S, P, param, prop, cur, cs, mean, logDet, gw, st);
                flags |= F_GAUSS;
            } else {
                for (int k = 0; k < dim; k++) mean[k] = cs.Mean(k);
                logDet = cs.LogDet();
            }
        }
        NormalDist nd(0.0f, 1.0f);  // GenerateSample, gaussian.cpp:38-55
        float z[MAXPSS];
        for (int k = 0; k < dim; k++) z[k] = nd(rng);
        if (useDense) {
            DenseGaussianMap(dim, z, mean, cs.CovL(), offset);
            py = DenseGaussianLogPdf(dim, offset, false, mean, cs.InvCov(), logDet);  // GaussianLogPdf(offset, currentState.gaussian): draws nothing, so it can be taken now
        } else
            for (int k = 0; k < dim; k++) offset[k] = param.sigma * z[k] + 0.0f;
    }
    if (PerturbPathBidir(S, offset, prop, pc, rng, stk)) {
        if (h2) {
            float px;
            if (useDense) {
                const H2Slot ps = H2Buf(A, i, !gsel);
                InitGaussianH2MC(S, P, param, prop, pc, ps, mean, logDet, gw, st);
                px = DenseGaussianLogPdf(dim, offset, true, mean, ps.InvCov(), logDet);
            } else {  // both isotropic with the same sigma: the dense form with a diagonal matrix, written out
                const float inv = 1.0f / (param.sigma * param.sigma), logDet = dim * fastlog(inv);
                float q = 0.f;
                for (int k = 0; k < dim; k++) q += offset[k] * (inv * offset[k]);
                py = dim * (-0.9189385332046727f);
                py += 0.5f * logDet;
                py -= 0.5f * q;
                px = py;
            }
            a = Clampf(expf(px - py) * pc.ssScore / cur.ssScore, 0.0f, 1.0f);
        } else {
            a = Clampf(pc.ssScore / cur.ssScore, 0.0f, 1.0f);
        }
    } else {
        a = 0.0f;
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
        ToSubpath(pc.camDepth, pc.lightDepth, prop);
        StorePath(PropPathBuf(A, flags), A.N, i, prop);
        flags ^= F_SEL;
        StoreContrib(A.curContrib, A.N, i, pc);
        A.adjacentReject[i] = 0;
        float *p = A.curSplat + i;
        p[0] = pc.screenPos.x, p[N] = pc.screenPos.y, p[2 * N] = smallSplat.x, p[3 * N] = smallSplat.y, p[4 * N] = smallSplat.z;
        A.curSplatCount[i] = 1;
        if (h2) {  // std::swap(currentState, proposalState): the proposal's Gaussian is the current one now
            if (useDense) flags ^= F_GSEL;  // the proposal's buffer is the current one now
            flags |= F_GAUSS;
        } else {
            flags &= ~F_GAUSS;  // mutation_small.h:39
        }
        flags |= F_VALID;
    } else {
        int rej = A.adjacentReject[i] + 1;  // REMOVE_OUTLIERS, mlt.cpp:147-169
        A.adjacentReject[i] = rej;
        const bool strongReject = cur.lsScore > OUTLIER_RATIO_THRESHOLD * P.normalization;
        if (rej > OUTLIER_WEAK_REJECT_CNT || (strongReject && rej > OUTLIER_STRONG_REJECT_CNT)) {
            ResetToInitState(A, P.chainBegin, P.numChains, OUTLIER_RATIO_THRESHOLD * P.normalization, i, sampleIdx, CurPathBuf(A, flags));
            A.curSplatCount[i] = 0;
            flags &= ~(F_VALID | F_GAUSS | F_BUFFERED);
            st.resets++;
        }
    }
    A.flags[i] = flags & ~F_VSYNC;
    A.sampleIdx[i] = sampleIdx + 1;
}

}  // namespace lmcd
