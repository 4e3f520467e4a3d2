// The hot kernel body: one plain small step (isotropic SmallStep or MALASmallStep with the global cache ready or
// not applicable) of one chain, written so that NOTHING lives in scratch memory and as little as possible in registers:
//   * the path is streamed vertex by vertex from the chain's current SoA buffer in HBM into registers and the
//     perturbed vertex is streamed out to the chain's other buffer (double buffering, one select bit per chain;
//     acceptance flips the bit instead of copying) -- coalesced reads/writes of exactly the words the
//     reference's PerturbPathBidir touches (path.cpp:1953-2160);
//   * the Gaussians are streamed too: one dimension at a time from / to the chain's two Gaussian buffers (same select-bit
//     scheme), so that neither the current nor the proposal Gaussian ever exists as a register array;
//   * ONE closest-hit traversal site (light and camera sub-path share a loop) and ONE any-hit site (the connection
//     strategies record their shadow ray, DeferOcclusion in dpath.h): the kernel's code is a fraction of the fully inlined
//     form, which matters on a 64 KB instruction cache shared by two CUs;
//   * everything indexed at run time (BVH stack, kd-tree search frames / per-dimension distances, the proposal offsets,
//     the new primary-sample vector) lives in LDS, laid out [word][thread].
// Same arithmetic, same RNG order as the generic StepChain (dstep.h), which remains the implementation for
// large steps and for gradient-evaluating small steps and which the parity tests cross-check against this one.
//
// LMC_LEAN_GRAD (step_small_leangrad.hip) compiles the same body with the gradient branch of InitGaussianFor in
// (mutation_mala.h:94-130): the launch that serves the chains whose cache is still filling.  The path program runs in a
// non-inlined function on the path record the step has just streamed to HBM, so its private memory does not touch the body.
#pragma once
#include "dstep.h"
#ifdef LMC_LEAN_GRAD
#include "dgrad.h"
#endif

namespace lmcd {

// Dimensions that can carry a non-isotropic Gaussian / a cache query (mutation_mala.h:94-96): states of dimension above
// PSS_MAX_LENGTH always get IsotropicGaussian(malaStdDev), which needs no per-dimension storage at all.
constexpr int MD = PSS_MAX_LENGTH;

// LDS words per thread, S = the launch's BVH stack entries (the scene's stack need rounded up to 8, at least 32, at most
// BVH_LDS_STACK: 32 for the torus = 56 words = 14 KB per wave, 11 waves per CU; 40 for the veach-door scene):
//   [0, S)          the BVH stack; between traversals the staged Gaussian / moment vectors / clipped gradient (2 MD <= 32 words)
//   [S, S + 12)     the proposal offsets of a state with dim <= MD
//   [S + 12, S + 24) its new primary-sample vector
//   [S, S + 24)     the offsets of a state with dim > MD (up to 2 * MAXD): such a state is never looked up, it has no Q
// The kd-tree search frames used to live here too (80 words, 8 waves per CU).  Since the existence test (dchain.h) the search
// runs for 0.015 % of the queries: it now keeps its frames in private memory, out of line (KdRadiusSearchRare).
constexpr int LDS_MIN_STACK_WORDS = 32;
static_assert(2 * MD <= LDS_MIN_STACK_WORDS, "the staging words live in the (idle) stack region");
LMC_HD int LeanStackWords(int bvhStackNeed) {
    const int w = (bvhStackNeed + 7) / 8 * 8;
    return w < LDS_MIN_STACK_WORDS ? LDS_MIN_STACK_WORDS : w;
}
LMC_HD int LeanLdsWordsPerThread(int stackWords) { return stackWords + MAXPSS; }

struct LdsView {
    float *base;  // &lds[threadIdx.x]
    int stride;   // blockDim.x
    int off;      // first word behind the BVH stack (= the launch's stack entries)
    LMC_D float &U(int w) const { return base[w * stride]; }                  // any word
    LMC_D float &Q(int k) const { return base[(off + MD + k) * stride]; }      // new pss
};

// word offsets inside the SoA path record (DPath layout)
enum : int {
    PW_TIME = 0, PW_SCREEN0, PW_SCREEN1, PW_LGTPOS0, PW_LGTPOS1, PW_LGTDIR0, PW_LGTDIR1, PW_LGTLIGHT, PW_LGTPRIM, PW_ENVPRIM, PW_CAMDEPTH,
    PW_LGTDEPTH, PW_CAMCOUNT, PW_LGTCOUNT, PW_LENS0, PW_LENS1
};
// LMC_NT_STATE (A/B build): the streamed path words are loaded / stored non-temporally -- every word of a chain's path record is touched once
// per step and the resident population (2-3 GB) is a hundred times the L2, so a line of it that stays in L2 only evicts scene data
#ifdef LMC_NT_STATE
LMC_D float LdS(const float *p) { return __builtin_nontemporal_load(p); }
LMC_D void StS(float *p, float v) { __builtin_nontemporal_store(v, p); }
#else
LMC_D float LdS(const float *p) { return *p; }
LMC_D void StS(float *p, float v) { *p = v; }
#endif
LMC_D int VertWord(bool lgt, int d, int field) { return DPATH_HEAD_WORDS + ((lgt ? MAXD : 0) + d) * DVERTEX_WORDS + field; }

LMC_D DVertex LoadVertex(const float *buf, size_t N, int i, bool lgt, int d) {
    const float *p = buf + (size_t)VertWord(lgt, d, 0) * N + i;
    DVertex v;
    v.tri = __float_as_int(LdS(p));
    v.st0 = LdS(p + N), v.st1 = LdS(p + 2 * N), v.rnd0 = LdS(p + 3 * N), v.rnd1 = LdS(p + 4 * N), v.bsdfDiscrete = LdS(p + 5 * N), v.useAbs = LdS(p + 6 * N), v.rrWeight = LdS(p + 7 * N);
    v.dirLight = __float_as_int(LdS(p + 8 * N)), v.dirPrim = __float_as_int(LdS(p + 9 * N));
    v.dirRnd0 = LdS(p + 10 * N), v.dirRnd1 = LdS(p + 11 * N);
    return v;
}
LMC_D void StoreVertex(float *buf, size_t N, int i, bool lgt, int d, const DVertex &v) {
    float *p = buf + (size_t)VertWord(lgt, d, 0) * N + i;
    StS(p, __int_as_float(v.tri));
    StS(p + N, v.st0), StS(p + 2 * N, v.st1), StS(p + 3 * N, v.rnd0), StS(p + 4 * N, v.rnd1), StS(p + 5 * N, v.bsdfDiscrete), StS(p + 6 * N, v.useAbs), StS(p + 7 * N, v.rrWeight);
    StS(p + 8 * N, __int_as_float(v.dirLight)), StS(p + 9 * N, __int_as_float(v.dirPrim));
    StS(p + 10 * N, v.dirRnd0), StS(p + 11 * N, v.dirRnd1);
}

// The proposal offsets are consumed in PerturbPathBidir's order through a cursor
struct OffsetCursor {
    const LdsView &L;
    int w;  // next word
    LMC_D float Pop() { return L.U(w++); }
};
// the new primary-sample vector of a state with dim <= MD (cache query / reuse test / chain->pss); longer states are never
// looked up and their offsets occupy the words Q would use
struct PssSink {
    const LdsView &L;
    bool keep;
    int n = 0;
    LMC_D void Push(float v) {
        if (keep && n < MD) L.Q(n) = v;
        n++;
    }
};

// The nanoflann-ordered radius search (dchain.h KdRadiusSearch) for the rare query that has a point within its radius.
// Out of line: its frame stack lives in private memory that the common path never touches.
// The query point is read from the caller's LDS words here, not handed over as a private array: an array whose address escapes into a call lives in
// scratch memory, and the caller -- the hot path, in which 99.98 % of the queries end at the existence test -- paid twelve scratch stores (and the
// reloads of the spilled LDS addresses they were filled from) per query for it (hipcc -S of the round-4 kernel).
__device__ __noinline__ int KdRadiusSearchRare(const DCacheDim &C, int dim, const float *ldsQ, int ldsStride, float radiusSq, int knn, int *idx, float *dist) {
    float q[MD];
#pragma unroll
    for (int k = 0; k < MD; k++) q[k] = k < dim ? ldsQ[k * ldsStride] : 0.f;
    return KdRadiusSearch(C, dim, q, radiusSq, knn, idx, dist);
}

// Where the moment vectors (v1, v2) behind a state's Gaussian come from (mutation_mala.h:131-164 and :224-257, cache /
// isotropic branches; chains that would evaluate a gradient run the generic kernel instead).
struct VSource {
    bool wrotePss;  // chain->pss was written (the cache of this dimension is still filling)
    int mode;  // 0: IsotropicGaussian(malaStdDev); 1: chain->v1 / v2 re-used; 2: inverse-distance blend of nMatches cache entries
    int nMatches;
    int idx[5];
    float w[5];
    double sum_w;
};
enum : int { VS_ISOTROPIC = 0, VS_REUSE = 1, VS_BLEND = 2, VS_GRAD = 3 };  // VS_GRAD: clipped gradient in LDS words [0, dim), nMatches = `first`

// What the gradient branch needs to know about the state whose Gaussian is being initialised
struct GradState {
    const float *pathBuf;  // its path record (current or proposal buffer of the chain)
    int c, l;
    float ssScore;
    bool isProposal;
    float *workBuf;  // serialisation buffer (dgrad.h GradWork)
    size_t workStride, workSlot;
};
#ifdef LMC_LEAN_GRAD
__device__ __noinline__ void LeanStateGradient(const DScene &S, const float *pathBuf, int N, int i, float *workBuf, size_t workStride, size_t workSlot, float *grad) {
    DPath path;
    LoadPath(pathBuf, N, i, path);
    GradWork gw{workBuf, workStride, workSlot};
    Contrib unused;
    unused.camDepth = path.camDepth, unused.lightDepth = path.lgtDepth;
    ComputeGradient(S, path, unused, grad, gw);
}
#endif

// First half of InitGaussianFor for a state whose pss is in L.Q: bookkeeping writes, re-use test, cache query.
template <bool WITH_GRAD>
LMC_D void PrepareGaussianLean(const DScene &S, const DCache &cache, const ChainArrays &A, const StepParams &P, int i, int dim, float lsScore, int flags,
                               const LdsView &L, VSource &vs, StepStats &st, const GradState &gs, bool skipQuery = false) {
    const size_t N = A.N;
    vs.mode = VS_ISOTROPIC;
    vs.nMatches = 0;
    vs.sum_w = 0;
    vs.wrotePss = false;
    A.pathWeight[i] = lsScore;
    if (dim > MD) return;  // PSS_MAX_LENGTH: no cache, chain->pss is never read for such a state
    // GetPathPss(path, chain->pss).  chain->pss has one reader, the cache push of an accepted large step (mlt.cpp:120-127), which only
    // happens while the cache of this dimension is still filling: once it is ready (and it stays ready) the copy is a dead store
    if (dim < PSS_MIN_LENGTH || !cache.d[dim].ready) {
#pragma unroll 1
        for (int k = 0; k < dim; k++) A.chPss[(size_t)k * N + i] = L.Q(k);
        vs.wrotePss = true;
    }
    if (dim < PSS_MIN_LENGTH) return;
#ifdef LMC_LEAN_GRAD
    if (WITH_GRAD && !cache.d[dim].ready && P.useGradient && GradAvailable(gs.c, gs.l) && gs.c + gs.l - 1 <= P.maxDervDepth) {  // mutation_mala.h:94-130
        float g[MD];
#pragma unroll
        for (int k = 0; k < MD; k++) g[k] = 0.f;
        if (gs.ssScore > 1e-10f) {
            if (!LMC_EXP(P.expFlags, 4)) LeanStateGradient(S, gs.pathBuf, (int)N, i, gs.workBuf, gs.workStride, gs.workSlot, g);
            st.gradCalls++;
            bool finite = true;
            for (int k = 0; k < dim; k++) finite = finite && isfinite(g[k]);
            if (!finite)
                for (int k = 0; k < dim; k++) g[k] = 0.f;
        }
        float norm = 0.f;
        const float drift = S.opt.malaGN;
        for (int k = 0; k < dim; k++) norm += g[k] * g[k];
        norm = sqrtf(norm);
        const float *newV2 = gs.isProposal ? A.chPropNewV2 : A.chCurrNewV2;
        bool first = true;
        for (int k = 0; k < dim; k++) {
            L.U(k) = g[k] * (drift / fmaxf(drift, norm));
            if (first && newV2[(size_t)k * N + i] > 1e-10f) first = false;
        }
        vs.mode = VS_GRAD;
        vs.nMatches = first ? 1 : 0;
        return;
    }
#endif
    if (!cache.d[dim].ready) return;
    if (skipQuery) return;
    if (flags & F_QUERIED) {
        float dist_sqr = 0.f;
        for (int k0 = 0; k0 < dim; k0 += 4) {  // chain->last_pss in branch-free groups of four; the sum keeps its order
            float lp[4];
#pragma unroll
            for (int j = 0; j < 4; j++) lp[j] = A.chLastPss[(size_t)min(k0 + j, dim - 1) * N + i];
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (k0 + j < dim) {
                    const float diff = L.Q(k0 + j) - lp[j];
                    dist_sqr += diff * diff;
                }
        }
        if (dist_sqr < dim * (PSS_REUSE_DIST * PSS_REUSE_DIST)) {
            vs.mode = VS_REUSE;
            return;
        }
    }
    st.cacheQueries++;
    if (LMC_EXP(P.expFlags, 256)) return;  // LMC_EXP_QUERY_STOP=1 (measurement): the query ends before its cell is computed
    const DCacheDim &C = cache.d[dim];
    const float radiusSq = dim * (PSS_QUERY_DIST * PSS_QUERY_DIST);
    int knnStop = 5;
    if (C.gridWords) {  // exact existence test (dchain.h): no candidate within the radius => query() finds nothing
        int cell = 0;
        for (int k = 0; k < C.gridM; k++) cell = cell * C.gridG + CacheGridCell(L.Q(C.gridCoord[k]), C.gridG);
        const uint2 word = C.gridWords[cell >> 5];
        const unsigned bit = 1u << (cell & 31);
        if (!(word.x & bit) || LMC_EXP(P.expFlags, 512)) return;  // the common case: the point is read no further (LMC_EXP_QUERY_STOP=2: every cell counts as empty)
        const int r = (int)word.y + __popc(word.x & (bit - 1u));
        const int s0 = C.gridCellStart[r], s1 = C.gridCellStart[r + 1];
        float q[MD];
#pragma unroll
        for (int k = 0; k < MD; k++) q[k] = k < dim ? L.Q(k) : 0.f;
        bool any = false;
        int nMatch = 0, oneIdx = 0;  // the candidates within the radius: how many, and (if one) which
        float oneD = 0.f;
#ifndef LMC_QUERY_BATCH
#define LMC_QUERY_BATCH 4
#endif
        // LMC_QUERY_BATCH candidates per round: their row indices in flight together, then their rows together (indices clamped, the surplus slots repeat the
        // last candidate).  A third to a half of the queries land in a non-empty cell, cells near the states' clusters list tens of candidates, and one candidate
        // at a time was two dependent fetches each -- the wave runs the loop of its longest lane: 11 % of the lean kernel (profiles/r06_bj_*: LMC_EXP_QUERY_STOP=2)
        for (int j = s0; j < s1; j += LMC_QUERY_BATCH) {
            int id[LMC_QUERY_BATCH];
#pragma unroll
            for (int b = 0; b < LMC_QUERY_BATCH; b++) id[b] = C.gridIdx[min(j + b, s1 - 1)];
            float2 pr[LMC_QUERY_BATCH][MD / 2];
#pragma unroll
            for (int b = 0; b < LMC_QUERY_BATCH; b++) {
                const float2 *row = reinterpret_cast<const float2 *>(C.pts + (size_t)id[b] * dim);
#pragma unroll
                for (int k = 0; k < MD / 2; ++k) pr[b][k] = row[min(k, dim / 2 - 1)];
            }
#pragma unroll
            for (int b = 0; b < LMC_QUERY_BATCH; b++) {
                float d = 0.f;  // same arithmetic, same order as the leaf scan of the search
#pragma unroll
                for (int k = 0; k < MD / 2; ++k)
                    if (2 * k < dim) {
                        const float2 p = pr[b][k];
                        const float diff0 = q[2 * k] - p.x;
                        d += diff0 * diff0;
                        const float diff1 = q[2 * k + 1] - p.y;
                        d += diff1 * diff1;
                    }
                any = any || d < radiusSq;
                if (j + b < s1 && d < radiusSq) nMatch++, oneIdx = id[b], oneD = d;
            }
        }
        if (!any) return;
#ifndef LMC_QUERY_ALWAYS_SEARCH
        // The cell's candidates are a superset of the rows within the radius and each was measured with the search's own arithmetic: if exactly ONE lies within it,
        // the search (unsorted, traversal order, dchain.h KdRadiusSearch) can only return that row with that distance -- and need not run.  It is a serial walk of
        // one lane through a tree whose planes prune little in 6-12 dimensions: ~0.1 ms, and although only ~20 queries of a launch get this far, one of their waves
        // is nearly always among the launch's last, so the search was the TAIL of the lean launch: 9 % of it, 7 % of the step (profiles/r06_bo_*, r06_bq_*;
        // LMC_QUERY_ALWAYS_SEARCH: A/B build).  Two or more rows within the radius: their order is the tree's, the search runs.
        if (nMatch == 1) {
            st.cacheHits++;
            vs.mode = VS_BLEND;
            vs.nMatches = 1;
            vs.idx[0] = oneIdx;
            vs.w[0] = inverse(oneD * oneD + 1e-6f);
            vs.sum_w += vs.w[0];
            return;
        }
        knnStop = min(nMatch, 5);
#endif
    }
    float dist[5];
    int idx[5];  // not vs.idx: an array handed to the search by address lives in private memory, and with it would the whole of vs
    // (the search stops at its knn-th match, nanoflann.hpp:256-262 as modified by the reference: 5 -- or, when the existence test has COUNTED the rows within the radius,
    // that count: the matches lie in the leaves the walk reaches first, everything behind its last match is backtracking that finds nothing)
    const int n = KdRadiusSearchRare(C, dim, &L.Q(0), L.stride, radiusSq, knnStop, idx, dist);
    if (n > 0) {  // global_cache.h:106-123
        st.cacheHits++;
        vs.mode = VS_BLEND;
        vs.nMatches = n;
#pragma unroll
        for (int m = 0; m < 5; m++)
            if (m < n) {
                vs.idx[m] = idx[m];
                vs.w[m] = inverse(dist[m] * dist[m] + 1e-6f);
                vs.sum_w += vs.w[m];
            }
    }
}

// chain->v1 / v2 of a re-using state into LDS words [0, 2 MD) (the BVH stack is idle whenever a Gaussian is built): all loads
// in flight together instead of one HBM round trip per dimension in GaussianDim's loop
LMC_D void StageReuseVectors(const ChainArrays &A, int i, int dim, const LdsView &L) {
    const size_t N = A.N;
    for (int k0 = 0; k0 < dim; k0 += 4) {  // branch-free groups of four (see the stored-Gaussian staging in SmallStepLean)
        float a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const size_t kk = (size_t)min(k0 + j, dim - 1);
            a[j] = A.chV1[kk * N + i], b[j] = A.chV2[kk * N + i];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) L.U(k0 + j) = a[j], L.U(MD + k0 + j) = b[j];
    }
}

// Second half, one dimension at a time: (v1[k], v2[k]) -> M -> ComputeGaussian (mala.cpp:7-52) or the isotropic values.
// A blend also performs query()'s writes of chain->v1 / v2 and last_pss (global_cache.h:107-123, mutation_mala.h:147-151).
struct GaussK {
    float mean, covL, invCov;
};
LMC_D GaussK GaussianDim(const DScene &S, const DCacheDim &C, const ChainArrays &A, int i, int dim, int k, const VSource &vs, float ssScore, const LdsView &L,
                         float &logDet, bool isProposal) {
    const size_t N = A.N;
    const float ss = S.opt.malaStepsize, shk = S.opt.malaStdDev;
    GaussK g;
    if (vs.mode == VS_ISOTROPIC) {  // gaussian.cpp:4-22
        g.mean = 0.0f, g.covL = shk, g.invCov = 1.0f / (shk * shk);
        return g;
    }
    float v1, v2;
    if (vs.mode == VS_REUSE) {
        v1 = L.U(k), v2 = L.U(MD + k);  // staged by StageReuseVectors
    } else if (vs.mode == VS_GRAD) {  // first / second moment update of the clipped gradient, mutation_mala.h:108-128 and :199-221
        const float gk = L.U(k), ov1 = A.chV1[(size_t)k * N + i], ov2 = A.chV2[(size_t)k * N + i];
        const bool first = vs.nMatches != 0;
        v1 = first ? gk : 0.9f * ov1 + 0.1f * gk;
        v2 = first ? gk * gk : 0.999f * ov2 + 0.001f * gk * gk;
        (isProposal ? A.chPropNewV2 : A.chCurrNewV2)[(size_t)k * N + i] = v2;
        if (isProposal) A.chPropNewV1[(size_t)k * N + i] = v1;
    } else {
        v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int m = 0; m < 5; m++)
            if (m < vs.nMatches) {
                v1 += C.v1[(size_t)vs.idx[m] * dim + k] * vs.w[m];
                v2 += C.v2[(size_t)vs.idx[m] * dim + k] * vs.w[m];
            }
        v1 = (float)((double)v1 / vs.sum_w);
        v2 = (float)((double)v2 / vs.sum_w);
        A.chV1[(size_t)k * N + i] = v1, A.chV2[(size_t)k * N + i] = v2;
        A.chLastPss[(size_t)k * N + i] = L.Q(k);  // last_pss = pss
    }
    const float shrk = inverse(shk * shk);
    if (ssScore <= 1e-10f) {
        g.mean = 0.0f, g.invCov = shrk, g.covL = shk;
        return g;
    }
    const float M = Clampf(1.0f / (1e-3f + sqrtf(v2)), PCD_MIN, PCD_MAX);
    const float cov_t = ss * ss * (M + 1.0f);
    const float invcov = inverse(cov_t) + shrk;
    const float cov = inverse(invcov);
    g.invCov = invcov;
    g.covL = sqrtf(cov);
    g.mean = Clampf(v1, MTM_MIN, MTM_MAX) * cov / 2;
    logDet += fastlog(invcov);
    return g;
}
// logDet of the branches that do not accumulate it dimension by dimension
LMC_D bool LogDetIsClosedForm(const VSource &vs, float ssScore) { return vs.mode == VS_ISOTROPIC || ssScore <= 1e-10f; }
LMC_D float ClosedFormLogDet(const DScene &S, const VSource &vs, int dim) {
    const float shk = S.opt.malaStdDev;
    return vs.mode == VS_ISOTROPIC ? dim * fastlog(1.0f / (shk * shk)) : dim * fastlog(inverse(shk * shk));
}

// Region timer of the profiling instantiation (LMC_PROF=1 selects it at run time; the production kernel is compiled with
// NoProf): the shader clock is read at marks placed at wave-convergent and divergent points of the step; the cycles since the
// previous mark -- as the WAVE experienced them -- are charged to the region the mark names.  Scalar registers only.
enum : int { PR_PROLOGUE = 0, PR_GAUSS_CUR, PR_OFFSETS, PR_VERTEX_LOAD, PR_TRAVERSE, PR_SHADE, PR_LOOP_EXIT, PR_SHADOW, PR_GAUSS_PROP, PR_SPLAT, PR_ACCEPT, PR_QUEUE, PR_ISO, PR_RESET, PR_STAGE, PR_COUNT };
static_assert(PR_COUNT < 16, "lmc_prof_read hands out 16 words: the regions + the wave count (include/lmc_abi.h LMC_PROF_REGIONS)");
struct NoProf {
    LMC_D void Mark(int) {}
};
struct WaveProf {
    unsigned long long last, acc[PR_COUNT];
    LMC_D void Start() {
        for (int r = 0; r < PR_COUNT; r++) acc[r] = 0;
        last = __builtin_readcyclecounter();
    }
    LMC_D void Mark(int r) {
        const unsigned long long now = __builtin_readcyclecounter();
        acc[r] += now - last;
        last = now;
    }
};

// One plain small step of chain i.  Returns nothing; all state changes go to HBM.
// LIGHTLESS: the caller guarantees l <= 1 (DOptions::leanLightless): the light-sub-path half of the walk, ConnectVertex and the
// registers they hold are compiled out
template <bool WITH_GRAD, bool LIGHTLESS = false, class Stk, class Prof>
LMC_D void SmallStepLean(const DScene &S, const DCache &cache, const ChainArrays &A, const Film &film, const StepParams &P, int i, Rng &rng,
                         const LdsView &L, Stk &stk, StepStats &st, Prof &prof, float *workBuf = nullptr, size_t workStride = 0, size_t workSlot = 0) {
    const size_t N = A.N;
    int flags = A.flags[i];
    const int sel = (flags & F_SEL) ? 1 : 0;
    const float *cur = sel ? A.pathBuf1 : A.curPath;
    float *prop = sel ? A.curPath : A.pathBuf1;
    const bool curValid = flags & F_VALID;  // always true for a small step
    const int c = __float_as_int(A.curContrib[i]), l = __float_as_int(A.curContrib[N + i]);
    const float curLs = A.curContrib[7 * N + i], curSs = A.curContrib[8 * N + i];
    if constexpr (LIGHTLESS) __builtin_assume(l <= 1);
    const int dim = PathDimension(c, l);
    const int camCount = max(c - 1, 0), lgtCount = max(l - 1, 0);
    const bool shortState = dim <= MD;  // may carry a stored Gaussian / be looked up in the cache
    const int offBase = L.off;
    const DCacheDim &C = cache.d[shortState ? dim : 0];
    st.steps++;
    st.lean++;

    DVertex nextV = LoadVertex(cur, N, i, l > 1, 0);  // first vertex of the walk below, requested before the Gaussian work
    // ... and so are the head words the walk perturbs first: time, screen position (each was a round trip of its own to the chain's state in HBM,
    // at the point of use; dscene.h LMC_PIN)
    float headTime = LdS(&cur[(size_t)PW_TIME * N + i]), headScreen0 = LdS(&cur[(size_t)PW_SCREEN0 * N + i]), headScreen1 = LdS(&cur[(size_t)PW_SCREEN1 * N + i]);
    LMC_PIN3(headTime, headScreen0, headScreen1);
    prof.Mark(PR_PROLOGUE);
    // ---- proposal offsets
    const bool mala = S.opt.mala && !(rng.Uniform() < S.opt.uniformMixingProbability);  // mutation_mala.h:46-51
    float py = 0.f;
    if (!mala) {  // SmallStep::Mutate, mutation_small.h:29-37
        NormalDist nd(0.0f, S.opt.perturbStdDev);
#pragma unroll 1
        for (int k = 0; k < dim; k++) L.U(offBase + k) = nd(rng);
#ifndef LMC_PROF_FINE
        prof.Mark(PR_ISO);
#endif
    } else {
        if (!(flags & F_BUFFERED)) {  // mutation_mala.h:59-81; the vectors are zero already (dchain.h ClearBuffered)
            flags |= F_BUFFERED | F_VSYNC;  // all four vectors are zero
            flags &= ~F_QUERIED;
        }
#ifndef LMC_PROF_FINE
        prof.Mark(PR_RESET);
#endif
        // currentState.gaussian: stored (F_GAUSS) or initialised now from the cache / isotropic (mutation_mala.h:83-166);
        // GenerateSample (gaussian.cpp:38-55) and GaussianLogPdf(offset, currentState.gaussian) (gaussian.cpp:24-36) are
        // fused into the same pass over the dimensions (the affine map draws nothing)
        float *G = CurGaussBuf(A, flags);
        const bool stored = (flags & F_GAUSS) && shortState && !(flags & F_GAUSS_ISO);  // F_GAUSS_ISO: isotropic, nothing was stored (vs stays VS_ISOTROPIC below)
        VSource vs;
        vs.mode = VS_ISOTROPIC;
        float logDet = 0.f;
        bool keepCur = false;  // the Gaussian initialised below has to be written to the chain's buffer
        if (!(flags & F_GAUSS)) {
            // GetPathPss(currentState.path) into LDS, path.cpp:2588-2632
            PssSink qs{L, shortState};
            if (l > 1) {
                qs.Push(LdS(&cur[(size_t)PW_LGTPOS0 * N + i])), qs.Push(LdS(&cur[(size_t)PW_LGTPOS1 * N + i]));
                qs.Push(LdS(&cur[(size_t)PW_LGTDIR0 * N + i])), qs.Push(LdS(&cur[(size_t)PW_LGTDIR1 * N + i]));
                for (int d = 0; d < lgtCount - 1; d++) qs.Push(LdS(&cur[(size_t)VertWord(true, d, 3) * N + i])), qs.Push(LdS(&cur[(size_t)VertWord(true, d, 4) * N + i]));
            }
            if (c > 1) {
                qs.Push(LdS(&cur[(size_t)PW_SCREEN0 * N + i])), qs.Push(LdS(&cur[(size_t)PW_SCREEN1 * N + i]));
                for (int d = 0; d < camCount - 1; d++) qs.Push(LdS(&cur[(size_t)VertWord(false, d, 3) * N + i])), qs.Push(LdS(&cur[(size_t)VertWord(false, d, 4) * N + i]));
                if (l == 1) qs.Push(LdS(&cur[(size_t)VertWord(false, camCount - 1, 10) * N + i])), qs.Push(LdS(&cur[(size_t)VertWord(false, camCount - 1, 11) * N + i]));
            }
            const GradState gs{cur, c, l, curSs, false, workBuf, workStride, workSlot};
            PrepareGaussianLean<WITH_GRAD>(S, cache, A, P, i, dim, curLs, flags, L, vs, st, gs, LMC_EXP(P.expFlags, 2));
            if (LMC_EXP(P.expFlags, 4096) && (vs.mode == VS_REUSE || vs.mode == VS_BLEND)) vs.mode = VS_ISOTROPIC;  // LMC_EXP_NOREUSE (measurement): a found neighbour is not used
            if (vs.mode == VS_BLEND) flags = (flags | F_QUERIED) & ~F_VSYNC;  // the blend rewrote chain->v1 / v2
            if (vs.wrotePss || vs.mode == VS_BLEND || vs.mode == VS_GRAD) flags |= F_VDIRTY;
            flags |= F_GAUSS;
            keepCur = shortState && vs.mode != VS_ISOTROPIC;
            flags = keepCur ? (flags & ~F_GAUSS_ISO) : (flags | F_GAUSS_ISO);
            prof.Mark(PR_GAUSS_CUR);
        }
        NormalDist nd(0.0f, 1.0f);
        float q = 0.f;
        float storedLogDet = 0.f;
        if (stored) {
            // The stored Gaussian is streamed from HBM: every load is a full memory round trip, so all of them are put in
            // flight together (independent iterations, staged through LDS words that are free here: Q and the BVH stack)
            // instead of one round trip per dimension inside the loop below, whose RNG calls the loads cannot be moved across.
            storedLogDet = G[(size_t)(3 * MAXPSS) * N + i];
            // branch-free groups of four dimensions (indices clamped, surplus slots written with duplicates): with a test
            // per element the compiler emits "load, wait, write" once per word
            for (int k0 = 0; k0 < dim; k0 += 4) {
                float m[4], c[4], v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const size_t kk = (size_t)min(k0 + j, dim - 1);
                    m[j] = G[kk * N + i], c[j] = G[(MAXPSS + kk) * N + i], v[j] = G[(2 * MAXPSS + kk) * N + i];
                }
#pragma unroll
                for (int j = 0; j < 4; j++) L.Q(k0 + j) = m[j], L.U(k0 + j) = c[j], L.U(MD + k0 + j) = v[j];
            }
        } else if (vs.mode == VS_REUSE) {
            StageReuseVectors(A, i, dim, L);
        }
        prof.Mark(PR_STAGE);
#pragma unroll 1
        for (int k = 0; k < dim; k++) {
            GaussK g;
            if (stored) {
                g.mean = L.Q(k), g.covL = L.U(k), g.invCov = L.U(MD + k);
            } else {
                g = GaussianDim(S, C, A, i, dim, k, vs, curSs, L, logDet, false);
                if (keepCur) G[(size_t)k * N + i] = g.mean, G[(size_t)(MAXPSS + k) * N + i] = g.covL, G[(size_t)(2 * MAXPSS + k) * N + i] = g.invCov;
            }
            const float o = g.covL * nd(rng) + g.mean;
            L.U(offBase + k) = o;
            const float d = o - g.mean;
            q += d * (g.invCov * d);
        }
        if (stored) {
            logDet = storedLogDet;
        } else {
            if (LogDetIsClosedForm(vs, curSs)) logDet = ClosedFormLogDet(S, vs, dim);
            if (keepCur) G[(size_t)(3 * MAXPSS) * N + i] = logDet;
        }
        py = dim * (-0.9189385332046727f);
        py += 0.5f * logDet;
        py -= 0.5f * q;
    }

    prof.Mark(PR_OFFSETS);
    // ---- PerturbPathBidir, path.cpp:1953-2160, streamed.  Light and camera sub-path share one loop so that the closest-hit
    // traversal (and the hit reconstruction behind it) is instantiated once; the connection strategies defer their shadow ray
    Contrib pc;
    pc.camDepth = pc.lightDepth = 0;
    pc.lsScore = pc.ssScore = 0.f;
    pc.screenPos = V2{0.f, 0.f};
    pc.contrib = V3{0.f, 0.f, 0.f};
    bool ok = false;
    DeferOcclusion occ;
    {
        OffsetCursor off{L, offBase};
        PssSink qs{L, shortState};
        NormalDist normDist(0.0f, S.opt.discreteStdDev);
        const float time = Modulo1(headTime + normDist(rng));
        StS(&prop[(size_t)PW_TIME * N + i], time);
        StS(&prop[(size_t)PW_CAMDEPTH * N + i], __int_as_float(c)), StS(&prop[(size_t)PW_LGTDEPTH * N + i], __int_as_float(l));
        StS(&prop[(size_t)PW_CAMCOUNT * N + i], __int_as_float(camCount)), StS(&prop[(size_t)PW_LGTCOUNT * N + i], __int_as_float(lgtCount));
        int envPrim = (l == 0) ? __float_as_int(LdS(&cur[(size_t)PW_ENVPRIM * N + i])) : -1;  // ToSubpath: -1 unless lgtDepth == 0
        BPS lps, cps;
        DVertex lastLgt;
        lastLgt.tri = -1;
        V3 org, dir;
        float tnear = c_IsectEpsilon, tfar = INFINITY;
        V2 screenPos{0.f, 0.f};
        int lgtLight = -1;
        bool lightPhase = false;
        auto BeginCamera = [&]() {  // EmitFromCamera with the perturbed screen position, path.cpp:2032-2038
            const float screen0 = Modulo1(headScreen0 + off.Pop());
            const float screen1 = Modulo1(headScreen1 + off.Pop());
            StS(&prop[(size_t)PW_SCREEN0 * N + i], screen0), StS(&prop[(size_t)PW_SCREEN1 * N + i], screen1);
            qs.Push(screen0), qs.Push(screen1);
            screenPos = V2{screen0, screen1};
            EmitFromCamera(S, screenPos, org, dir, cps);
            tnear = PrimaryMinT(S, screenPos, tfar);
            lightPhase = false;
        };
        if (l > 1) {
            lightPhase = true;
            lgtLight = __float_as_int(LdS(&cur[(size_t)PW_LGTLIGHT * N + i]));
            const float lightPickProb = PickLightProb(S, lgtLight);
            DPath hd;  // only the emitter fields are used by EmitFromLight
            hd.lgtPos0 = Modulo1(LdS(&cur[(size_t)PW_LGTPOS0 * N + i]) + off.Pop());
            hd.lgtPos1 = Modulo1(LdS(&cur[(size_t)PW_LGTPOS1 * N + i]) + off.Pop());
            hd.lgtDir0 = Modulo1(LdS(&cur[(size_t)PW_LGTDIR0 * N + i]) + off.Pop());
            hd.lgtDir1 = Modulo1(LdS(&cur[(size_t)PW_LGTDIR1 * N + i]) + off.Pop());
            hd.lgtLight = lgtLight;
            hd.lgtPrim = __float_as_int(LdS(&cur[(size_t)PW_LGTPRIM * N + i]));
            qs.Push(hd.lgtPos0), qs.Push(hd.lgtPos1), qs.Push(hd.lgtDir0), qs.Push(hd.lgtDir1);
            EmitFromLight(S, lightPickProb, hd, org, dir, lps);
            StS(&prop[(size_t)PW_LGTPOS0 * N + i], hd.lgtPos0), StS(&prop[(size_t)PW_LGTPOS1 * N + i], hd.lgtPos1);
            StS(&prop[(size_t)PW_LGTDIR0 * N + i], hd.lgtDir0), StS(&prop[(size_t)PW_LGTDIR1 * N + i], hd.lgtDir1);
            StS(&prop[(size_t)PW_LGTLIGHT * N + i], __int_as_float(lgtLight)), StS(&prop[(size_t)PW_LGTPRIM * N + i], __int_as_float(hd.lgtPrim));
        } else {
            BeginCamera();
        }
        int depth = 0;  // vertex index inside the current sub-path
#ifdef LMC_NO_VERTEX_PREFETCH
        bool firstSegment = true;
#endif
        // every iteration = one path segment; `break` = the step's contribution is decided (ok) or the path died
        while (lightPhase || depth < camCount) {
            prof.Mark(PR_SHADE);  // the previous segment's vertex work (the first time: the sub-path head)
#ifdef LMC_NO_VERTEX_PREFETCH  // A/B build: every vertex record but the first is fetched where it is used (twelve registers fewer across the traversal)
            DVertex sv = firstSegment ? nextV : LoadVertex(cur, N, i, lightPhase, depth);
            firstSegment = false;
#else
            DVertex sv = nextV;
            {   // the record of the vertex the next iteration perturbs (if the path goes on) is requested now: the path is
                // streamed from HBM, and its round trip then runs behind this segment's traversal instead of in front of the next
                bool nl = lightPhase;
                int nd = depth + 1;
                if (lightPhase && depth == lgtCount - 1) nl = false, nd = 0;
                if (nl ? nd < lgtCount : nd < camCount) nextV = LoadVertex(cur, N, i, nl, nd);
            }
#endif
            SurfHit hit;
            hit.tri = -1;
            hit.st = V2{0.f, 0.f};
            Isect isect;
            isect.position = isect.shadingNormal = isect.geomNormal = V3{0.f, 0.f, 0.f};
#ifndef LMC_PROF_FINE
            prof.Mark(PR_VERTEX_LOAD);
#endif
            const bool hitSurface = IntersectSurface(S, org, dir, tnear, tfar, hit, isect, stk, sv.tri, LMC_EXP(P.expFlags, 1024));  // sv.tri: the current state's triangle, tried first (dscene.h)
#if LMC_MAT_CARRY
            // the vertex's material by the index the hit record carried (dshade.h SurfHit::material), not through S.tris[tri].material again.  (Requesting the
            // record HERE, so that it travels while ConvertMIS and the draws run, was measured too: -1.4 % -- eight more registers live across that code,
            // profiles/r05_bg_*)
            auto HitMaterial = [&]() -> DMaterial { return LoadMaterialIdx<Stk::kGlossy>(S, hit.material); };
#endif
            prof.Mark(PR_TRAVERSE);
            if (lightPhase) {
                if (!hitSurface) break;
                lps.isect = isect;
                sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
                lps.wi = -dir;
                sv.bsdfDiscrete = Modulo1(sv.bsdfDiscrete + normDist(rng));
                ConvertMIS(S, depth, lgtLight, org, dir, lps);
                if (depth == lgtCount - 1 && c == 1) {
                    ok = ConnectToCamera(S, depth, lps, sv, pc, stk, occ);
                    if (!LMC_EXP(P.expFlags, 2048)) StoreVertex(prop, N, i, true, depth, sv);
                    break;
                }
                if (depth == lgtCount - 1) {
                    if (!LMC_EXP(P.expFlags, 2048)) StoreVertex(prop, N, i, true, depth, sv);
                    lastLgt = sv;
                    BeginCamera();
                    depth = 0;
                    continue;
                }
                sv.rnd0 = Modulo1(sv.rnd0 + off.Pop());
                sv.rnd1 = Modulo1(sv.rnd1 + off.Pop());
                qs.Push(sv.rnd0), qs.Push(sv.rnd1);
                V3 bsdfContrib;
#if LMC_MAT_CARRY
                if (!BSDFSamplingM<true, true, Stk::kGlossy>(S, HitMaterial(), lps, sv, lps, dir, bsdfContrib)) break;
#else
                if (!BSDFSampling<true, true, Stk::kGlossy>(S, lps, sv, lps, dir, bsdfContrib)) break;
#endif
                if (!LMC_EXP(P.expFlags, 2048)) StoreVertex(prop, N, i, true, depth, sv);
                lps.throughput = lps.throughput * sv.rrWeight;
                org = lps.isect.position;
                depth++;
                continue;
            }
            if (hitSurface) cps.isect = isect;
            sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
            cps.wi = -dir;
            if (hitSurface) ConvertMIS(S, depth, -1, org, dir, cps);
            if (depth == camCount - 1 && l == 0) {
#ifdef LMC_PROF_FINE  // A/B build: isotropic_offsets = the emitter-hit terminal, vertex_load = the direct-lighting terminal, buffered_reset = PrepareGaussianLean(proposal)
                prof.Mark(PR_SHADE);
#endif
                const int light = HitLightOf(S, hitSurface, hit);
                if (light >= 0) ok = HandleHitLight(S, depth, light, hitSurface, dir, screenPos, cps, envPrim, pc);
                if (!LMC_EXP(P.expFlags, 2048)) StoreVertex(prop, N, i, false, depth, sv);
#ifdef LMC_PROF_FINE
                prof.Mark(PR_ISO);
#endif
                break;
            }
            if (!hitSurface) break;
            sv.bsdfDiscrete = Modulo1(sv.bsdfDiscrete + normDist(rng));
            if (depth == camCount - 1) {
#ifdef LMC_PROF_FINE
                prof.Mark(PR_SHADE);
#endif
                if (l == 1) {
                    const float directLightPickProb = PickLightProb(S, sv.dirLight);
                    sv.dirRnd0 = Modulo1(sv.dirRnd0 + off.Pop());
                    sv.dirRnd1 = Modulo1(sv.dirRnd1 + off.Pop());
                    qs.Push(sv.dirRnd0), qs.Push(sv.dirRnd1);
#if LMC_MAT_CARRY
                    ok = DirectLightingM(S, HitMaterial(), depth, cps, screenPos, directLightPickProb, sv, pc, stk, occ);
#else
                    ok = DirectLighting(S, depth, cps, screenPos, directLightPickProb, sv, pc, stk, occ);
#endif
                } else {
                    ok = ConnectVertex(S, depth, lgtCount - 1, lps, lastLgt, cps, sv, screenPos, pc, stk, occ);
                }
                if (!LMC_EXP(P.expFlags, 2048)) StoreVertex(prop, N, i, false, depth, sv);
#ifdef LMC_PROF_FINE
                prof.Mark(PR_VERTEX_LOAD);
#endif
                break;
            }
            sv.rnd0 = Modulo1(sv.rnd0 + off.Pop());
            sv.rnd1 = Modulo1(sv.rnd1 + off.Pop());
            qs.Push(sv.rnd0), qs.Push(sv.rnd1);
            V3 bsdfContrib;
#if LMC_MAT_CARRY
            if (!BSDFSamplingM<false, true, Stk::kGlossy>(S, HitMaterial(), cps, sv, cps, dir, bsdfContrib)) break;
#else
            if (!BSDFSampling<false, true, Stk::kGlossy>(S, cps, sv, cps, dir, bsdfContrib)) break;
#endif
            if (!LMC_EXP(P.expFlags, 2048)) StoreVertex(prop, N, i, false, depth, sv);
            cps.throughput = cps.throughput * sv.rrWeight;
            org = cps.isect.position;
            tnear = c_IsectEpsilon;
            tfar = INFINITY;
            depth++;
        }
        StS(&prop[(size_t)PW_ENVPRIM * N + i], __int_as_float(envPrim));
    }
    prof.Mark(PR_LOOP_EXIT);  // the last segment's vertex work: connection strategy / emitter hit
    // the one shadow ray of the step (scene.cpp:128-149), cast after its strategy has been evaluated
    if (ok && occ.pending && !LMC_EXP(P.expFlags, 1024)) ok = !Occluded(S, occ.org, occ.dir, occ.dist, stk);
    prof.Mark(PR_SHADOW);

    // ---- proposal Gaussian + acceptance probability (mutation_mala.h:174-267)
    float a = 0.0f;
    bool keepProp = false;  // the proposal's Gaussian was written to the chain's other buffer (it is not the isotropic one)
    if (ok) {
        if (mala) {
            float *G = PropGaussBuf(A, flags);
            VSource vs;
            const GradState gs{prop, c, l, pc.ssScore, true, workBuf, workStride, workSlot};
            PrepareGaussianLean<WITH_GRAD>(S, cache, A, P, i, dim, pc.lsScore, flags, L, vs, st, gs, LMC_EXP(P.expFlags, 2));
            if (LMC_EXP(P.expFlags, 4096) && (vs.mode == VS_REUSE || vs.mode == VS_BLEND)) vs.mode = VS_ISOTROPIC;
#ifdef LMC_PROF_FINE
            prof.Mark(PR_RESET);
#endif
            if (vs.mode == VS_BLEND) flags = (flags | F_QUERIED) & ~F_VSYNC;
            if (vs.mode == VS_GRAD) flags &= ~F_VSYNC;  // the moment update rewrote prop_new_v1 / v2
            if (vs.wrotePss || vs.mode == VS_BLEND || vs.mode == VS_GRAD) flags |= F_VDIRTY;
            if (vs.mode == VS_REUSE) StageReuseVectors(A, i, dim, L);
            keepProp = shortState && vs.mode != VS_ISOTROPIC;
            float logDet = 0.f, q = 0.f;  // GaussianLogPdf(-offset, proposalState.gaussian)
#pragma unroll 1
            for (int k = 0; k < dim; k++) {
                const GaussK g = GaussianDim(S, C, A, i, dim, k, vs, pc.ssScore, L, logDet, true);
                if (keepProp) G[(size_t)k * N + i] = g.mean, G[(size_t)(MAXPSS + k) * N + i] = g.covL, G[(size_t)(2 * MAXPSS + k) * N + i] = g.invCov;
                const float d = -L.U(offBase + k) - g.mean;
                q += d * (g.invCov * d);
            }
            if (LogDetIsClosedForm(vs, pc.ssScore)) logDet = ClosedFormLogDet(S, vs, dim);
            if (keepProp) G[(size_t)(3 * MAXPSS) * N + i] = logDet;
            float px = dim * (-0.9189385332046727f);
            px += 0.5f * logDet;
            px -= 0.5f * q;
            a = Clampf(lexpf(px - py) * pc.ssScore / curSs, 0.0f, 1.0f);
        } else {
            a = Clampf(pc.ssScore / curSs, 0.0f, 1.0f);
        }
    }

    prof.Mark(PR_GAUSS_PROP);
    // ---- splats, mlt.cpp:103-112
    const bool doSplat = !LMC_EXP(P.expFlags, 1);
    if (curValid && a < 1.0f && doSplat) {
        const int n = A.curSplatCount[i];
        for (int k = 0; k < n; k++) {
            const float *p = A.curSplat + ((size_t)k * SPLAT_WORDS) * N + i;
            float s0 = p[0], s1 = p[N], s2 = p[2 * N], s3 = p[3 * N], s4 = p[4 * N];
            LMC_PIN5(s0, s1, s2, s3, s4);  // one round trip for the pending splat's five words
            Splat(film, V2{s0, s1}, (1.0f - a) * V3{s2, s3, s4});
        }
    }
    const V3 smallSplat = mala ? (pc.contrib * P.normalization) / pc.lsScore : pc.contrib * (P.normalization / pc.lsScore);
    if (a > 0.0f && doSplat) Splat(film, pc.screenPos, a * smallSplat);
    st.wsum += curValid ? 1.0f : (a > 0.0f ? a : 0.0f);

    prof.Mark(PR_SPLAT);
    // ---- accept / reject, mlt.cpp:113-170
    const int sampleIdx = A.sampleIdx[i];
    A.pushDim[i] = 0;
    if (a > 0.0f && rng.Uniform() <= a) {
        st.accepted++;
        flags ^= F_SEL;  // the proposal buffer becomes the current path (ToSubpath: counts / depths already written)
        StoreContrib(A.curContrib, A.N, i, pc);
        A.adjacentReject[i] = 0;
        float *p = A.curSplat + i;
        p[0] = pc.screenPos.x, p[N] = pc.screenPos.y, p[2 * N] = smallSplat.x, p[3 * N] = smallSplat.y, p[4 * N] = smallSplat.z;
        A.curSplatCount[i] = 1;
        if (mala) {  // mlt.cpp:133-142: chain.v1 / v2 = prop_new_v1 / v2 (whole vectors) -- unless they are known to be equal
            if (!(flags & F_VSYNC)) {
#pragma unroll 4
                for (int k = 0; k < MAXPSS; k++) {
                    A.chV1[(size_t)k * N + i] = A.chPropNewV1[(size_t)k * N + i];
                    A.chV2[(size_t)k * N + i] = A.chPropNewV2[(size_t)k * N + i];
                }
            }
            flags |= F_BUFFERED | F_GAUSS | F_VSYNC;
            if (keepProp) {
                flags ^= F_GSEL;  // the proposal's Gaussian (streamed into the other buffer above) becomes the current one
                flags &= ~F_GAUSS_ISO;
            } else {
                flags |= F_GAUSS_ISO;  // isotropic: nothing was written, nothing will be read
            }
        } else {
            flags &= ~(F_GAUSS | F_GAUSS_ISO);
        }
        flags |= F_VALID;
    } else {
        int rej = A.adjacentReject[i] + 1;  // REMOVE_OUTLIERS, mlt.cpp:147-169
        A.adjacentReject[i] = rej;
        const bool strongReject = curLs > OUTLIER_RATIO_THRESHOLD * P.normalization;
        if (OutlierReset(rej, strongReject, P.expFlags)) {
            ResetToInitState(A, P.chainBegin, P.numChains, OUTLIER_RATIO_THRESHOLD * P.normalization, i, sampleIdx, sel ? A.pathBuf1 : A.curPath);
            A.curSplatCount[i] = 0;
            flags &= ~(F_VALID | F_GAUSS | F_GAUSS_ISO);
            ClearBuffered(A, i, flags);
            st.resets++;
        }
    }
    A.flags[i] = flags;
    A.sampleIdx[i] = sampleIdx + 1;
    prof.Mark(PR_ACCEPT);
}

}  // namespace lmcd
