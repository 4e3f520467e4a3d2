// Per-chain Markov state in HBM (SoA: word w of chain i lives at base[w * N + i], so a wave touching the
// same field of 64 consecutive chains reads one or two cache lines) and the device side of
// /root/reference/src/{mlt.cpp:91-170, mutation_large.h:31-128, mutation_small.h:16-56,
// mutation_mala.h:35-278, mala.cpp:7-52, gaussian.cpp:4-55, global_cache.h:96-124}.
#pragma once
#include "dpath.h"

namespace lmcd {

constexpr int GAUSS_WORDS = 3 * MAXPSS + 1;  // mean, covL_d, invCov_d, logDet
constexpr int CONTRIB_WORDS = 9;
constexpr int CACHE_ROW_EXTRA = DPATH_WORDS + CONTRIB_WORDS;  // a `samplecache` cache row beyond pss | v1 | v2 | weight: its path, its contribution
constexpr int SPLAT_WORDS = 5;

// F_SEL: which of the two path buffers holds the chain's current path (the other receives the proposal; acceptance
// flips the bit instead of copying the path)
// F_GSEL: the same for the two Gaussian buffers (the lean kernel streams the proposal's Gaussian into the other buffer)
// F_VSYNC: chain->v1 / v2 are known to equal prop_new_v1 / prop_new_v2 word for word, so the copy that follows an accepted
// MALA step (mlt.cpp:133-142) has nothing to move.  Maintained by the lean kernel; every other kernel just clears it.
enum : int { F_VALID = 1, F_GAUSS = 2, F_BUFFERED = 4, F_QUERIED = 8, F_LAST_MALA = 16, F_SEL = 32, F_GSEL = 64, F_VSYNC = 128, F_GAUSS_ISO = 256, F_VDIRTY = 512 };
// F_VDIRTY: one of the chain's seven MALA vectors (v1, v2, curr_new_v2, prop_new_v1 / v2, pss, last_pss) may be non-zero.  Once the
// caches are built a chain writes none of them (no gradient moments, a cache query that finds nothing, chain->pss only feeds the
// cache push), so ClearBuffered has nothing to zero.
// F_GAUSS_ISO (only meaningful together with F_GAUSS): the current state's Gaussian is IsotropicGaussian(malaStdDev) -- the outcome of
// 99.98 % of the initialisations once the caches are built (a cache query that finds nothing, mutation_mala.h:155-161) -- and is NOT
// stored: every reader re-creates the same constants instead of streaming 3 dim + 1 words in and out per step.
enum : int { KIND_SMALL = 0, KIND_LARGE = 1 };
enum : unsigned char { NEXT_DONE = 0, NEXT_LARGE = 1, NEXT_SMALL_GENERIC = 2, NEXT_SMALL_PLAIN = 3 };

// global_cache.h:8-14, mala.h:9-13, mutation.h:5-8
constexpr int PSS_MIN_LENGTH = 2, PSS_MAX_LENGTH = 12, PSS_MAX_SIZE = 3000;
constexpr float PSS_QUERY_DIST = 0.01f, PSS_REUSE_DIST = 0.10f;
constexpr float PCD_MIN = 0.01f, PCD_MAX = 100.f, MTM_MIN = -5.0f, MTM_MAX = 5.0f, LS_RATIO = 0.1f;
constexpr int OUTLIER_WEAK_REJECT_CNT = 10000, OUTLIER_STRONG_REJECT_CNT = 1000;
constexpr float OUTLIER_RATIO_THRESHOLD = 30.0f;
// REMOVE_OUTLIERS, mlt.cpp:147-169: the chain is reset after this many adjacent rejections.  expFlags bit 128 (LMC_EXP_OUTLIER_TEST=1, tests only)
// lowers the two counts to 6 / 2 so that a short run resets thousands of chains -- the path the counts of the reference make all but
// unreachable in a test (tests/test_gpu_relocate.py: the reset walks CHAIN ids, which relocated chains carry in A.chainId)
LMC_HD bool OutlierReset(int rej, bool strongReject, int expFlags) {
    const int weak = (expFlags & 128) ? 6 : OUTLIER_WEAK_REJECT_CNT, strong = (expFlags & 128) ? 2 : OUTLIER_STRONG_REJECT_CNT;
    return rej > weak || (strongReject && rej > strong);
}

LMC_HD int CacheGridG(int dim) {  // cells per axis: the largest G with 1 / G >= radius = sqrt(dim) * PSS_QUERY_DIST
    int G = (int)(1.0f / (sqrtf((float)dim) * PSS_QUERY_DIST));
    return G < 1 ? 1 : (G > 64 ? 64 : G);
}
LMC_HD int CacheGridCell(float x, int G) {
    int c = (int)(x * (float)G);
    return c < 0 ? 0 : (c > G - 1 ? G - 1 : c);
}
constexpr int KD_MAX_NODES = 6144;  // 3000 points: a binary tree has fewer than 2 * 3000 nodes whatever the split (the host checks)
constexpr int KD_STACK = 160;  // deepest kd-tree the in-kernel search accepts (the host refuses deeper ones)

struct KdNode {  // nanoflann Node flattened (host/kdtree.cpp builds it exactly like nanoflann's divideTree)
    int child1, child2;  // -1,-1 => leaf
    int left, right;     // leaf: [left,right) into vind
    int divfeat;
    float divlow, divhigh;
    int pad;
};
struct DCacheDim {
    int ready;
    int deep;  // tree deeper than the LDS search of the lean small-step kernel accepts: chains of this dim use the generic kernel
    const KdNode *nodes;
    const int *vind;
    const float *pts, *v1, *v2;  // PSS_MAX_SIZE x dim, row-major
    // Exact existence test in front of the radius query (lean kernel): a uniform grid over gridM coordinates (gridCoord: the ones in which the
    // cache rows collide least, chosen when the dim becomes ready -- host/accel.h ChooseGridCoords) with cells no smaller than the query radius
    // (G = floor(1 / sqrt(dim) / 0.01)).  Every cell lists the rows of its 3^gridM neighbourhood, so a row within the radius of a query is
    // among the candidates of the query's own cell, and testing them with the search's own distance arithmetic decides "any match?" exactly.
    // Only then (0.015 % of the queries on the torus) is the nanoflann-ordered search needed.  The search itself cost 40 % of the lean kernel
    // (profiles/r02_f_ablation_splat_query.jsonl): in 6-12 dimensions its split planes prune little.
    // Layout (round 5): COMPACT, so that what a query touches stays in L2 instead of 22 MB of offsets + 30 MB of copied rows that every
    // query missed (profiles/r05_a_pmc_read_attribution_*: 377 B of HBM-side reads per chain-step, a sixth of the kernel's):
    //   gridWords[cell / 32] = {occupancy bits of 32 cells, number of non-empty cells before them}   (0.15 - 0.64 MB per dim)
    //   gridCellStart[r], [r + 1] = the candidate range of the r-th non-empty cell                    (<= 1 MB)
    //   gridIdx[j] = row index of candidate j (16 bit), the row itself read from `pts` (72 - 144 KB)  (<= 0.5 MB)
    // An empty cell -- 52 - 67 % of the queries of the headline workload -- ends at one 8-byte load.
    const uint2 *gridWords;
    const int *gridCellStart;
    const unsigned short *gridIdx;
    int gridG, gridM;
    int gridCoord[4];
    float rootLow[MAXPSS], rootHigh[MAXPSS];
    // `samplecache` (LargeStepCache, global_cache.h:21-23,42-47,57-58,126-164): every row's path and contribution (CACHE_ROW_EXTRA
    // words), its weight, PiecewiseConstant1D over the weights (cdf of PSS_MAX_SIZE + 1 entries), their double sum in row order,
    // the Gaussian kernel's constants.  nullptr / 0 unless the option is set.
    const float *extra, *weight, *distCdf;
    double scoreSum;
    float invSigmaSq, factor;
};
struct DCache {
    DCacheDim d[PSS_MAX_LENGTH + 1];
};
// Fill side of the cache: dims 6, 8, 10, 12 (MLT states have path length >= 3, mlt.h:76) <-> slot (dim - 6) / 2
constexpr int CACHE_SLOTS = 4;
struct CachePushTargets {
    float *pss[CACHE_SLOTS], *v1[CACHE_SLOTS], *v2[CACHE_SLOTS], *weight[CACHE_SLOTS];  // PSS_MAX_SIZE rows each (nullptr: dim not in use)
    float *extra[CACHE_SLOTS];  // `samplecache`: PSS_MAX_SIZE x CACHE_ROW_EXTRA (the row's DPath, then its Contrib); nullptr otherwise
    int *count;                                                                           // [CACHE_SLOTS] rows filled
};

// the stage of the multi-rank push (kernels.hip k_push_apply): one buffer of floats per rank, [16 header words: rows per slot]
// followed, per slot, by pss | v1 | v2 (PSS_MAX_SIZE x dim each) | weight (PSS_MAX_SIZE); offsets in floats
struct PushStageLayout {
    int pss[CACHE_SLOTS], v1[CACHE_SLOTS], v2[CACHE_SLOTS], weight[CACHE_SLOTS], extra[CACHE_SLOTS];
    int totalFloats;
};
LMC_HD PushStageLayout MakePushStageLayout(bool withPaths) {  // withPaths (`samplecache`): every slot also carries PSS_MAX_SIZE x CACHE_ROW_EXTRA words
    PushStageLayout l;
    int off = 16;
    for (int sl = 0; sl < CACHE_SLOTS; sl++) {
        const int dim = 6 + 2 * sl;
        l.pss[sl] = off, l.v1[sl] = off + PSS_MAX_SIZE * dim, l.v2[sl] = off + 2 * PSS_MAX_SIZE * dim, l.weight[sl] = off + 3 * PSS_MAX_SIZE * dim;
        off += PSS_MAX_SIZE * (3 * dim + 1);
        l.extra[sl] = off;
        if (withPaths) off += PSS_MAX_SIZE * CACHE_ROW_EXTRA;
    }
    l.totalFloats = off;
    return l;
}

struct ChainArrays {
    int N;
    uint64_t *rngState;
    uint32_t *rngTab;  // N x 64 (AoS): a chain's extension table once its stream has ticked (drng.h); until then synthesised from the seed, never read
    unsigned char *rngTicked;  // N: the chain's stream has ticked, its table lives in rngTab
    float *curPath;    // DPATH_WORDS x N: path buffer 0
    float *pathBuf1;   // DPATH_WORDS x N: path buffer 1 (see F_SEL)
    float *curContrib; // CONTRIB_WORDS x N
    float *scoreSum;   // N
    int *flags;        // N
    float *gaussian;   // GAUSS_WORDS x N: Gaussian buffer 0
    float *gaussian1;  // GAUSS_WORDS x N: Gaussian buffer 1 (see F_GSEL)
    float *curSplat;   // MAXCONTRIB*SPLAT_WORDS x N
    int *curSplatCount;
    float *chV1, *chV2, *chCurrNewV2, *chPropNewV1, *chPropNewV2, *chPss, *chLastPss;  // MAXPSS x N each
    float *pathWeight, *lastScoreSum, *lastScore;
    float *chPath, *chContrib;  // `samplecache` only (else nullptr): chain.path / chain.spContrib of mutation_mala.h:90-91,185-186, DPATH_WORDS / CONTRIB_WORDS x N
    int *adjacentReject, *sampleIdx, *numSamples;
    float *contribList;  // MAXCONTRIB*CONTRIB_WORDS x N (GeneratePathBidir scratch)
    int *slotOf;         // N, or nullptr = the identity: the inverse of chainId (the cache pushes are packed in chain order, kernels.hip k_push_*)
    int *chainId;        // N, or nullptr = the identity: the (rank-local) id of the chain that lives in a slot, once chains are relocated (relocate.hip)
    unsigned char *stepKind;  // N, or nullptr: the launch (NEXT_*) that runs the chain's CURRENT step, recorded by k_build_lists for the relocation
    unsigned char *nextKind;  // N: which launch runs the chain's next step (NEXT_*), turned into id-ordered work lists by k_build_lists
    int *pushDim;        // N: dim of a pending global-cache push (0 = none)
    float *pushData;     // (3*MAXPSS+1) x N: pss, v1, v2, weight snapshot for the push
    // init states (outlier reset, mlt.cpp:147-169): full states of THIS rank's chains (N), and for every chain of the job (the
    // reset walks global chain ids) the two things the loop reads of a state that is then marked invalid: lsScore and technique
    float *initPath, *initContrib, *initScoreSum;
    const float *initLsAll;          // numChainsTotal
    const unsigned char *initCLAll;  // numChainsTotal, c * 16 + l
    // per-launch counters: [0] steps, [1] large, [2] accepted, [3] gradCalls, [4] cacheQueries, [5] cacheHits, [6] resets
    unsigned long long *counters;
    unsigned long long *prof;  // region cycle sums of the profiling instantiation of the lean kernel (dsmall.h WaveProf), 16 words
    double *weightSum;
};

// A chain's RNG for the duration of a kernel: the LCG word from HBM, the extension table synthesised from the chain's seed
// RNG(chainId + seedOffset) (mlt.cpp:61-62; i is the SLOT, the chain that lives in it: A.chainId) unless the stream has ticked.
LMC_D Rng LoadChainRng(const ChainArrays &A, int chainBegin, int seedOffset, int i) {
    Rng rng;
    rng.state = A.rngState[i];
    rng.tab = A.rngTab + (size_t)i * 64;
    rng.ticks = 0;
    if (!A.rngTicked[i]) rng.SetSynth((uint64_t)(chainBegin + (A.chainId ? A.chainId[i] : i) + seedOffset));
    return rng;
}
LMC_D void StoreChainRng(const ChainArrays &A, int i, const Rng &rng) {
    A.rngState[i] = rng.state;
    if (rng.ticks) A.rngTicked[i] = 1;  // Rng::Tick has materialised the table into the chain's slot
}

// Technique key of a state (c,l), 6 bits: path length first, then the light-subpath length.  Work lists (dstep.h QueueNext, kernels.hip)
// and the relocation of chains (relocate.hip) group by it.
LMC_HD unsigned char TechniqueKey(int c, int l) {
    const int L = c + l - 1 > 3 ? c + l - 1 : 3;
    const int k = (L - 3) * 6 + (l < 5 ? l : 5);
    return (unsigned char)(k < 63 ? k : 63);
}
LMC_D void LoadWords(const float *base, int N, int i, float *dst, int n) {
    for (int w = 0; w < n; w++) dst[w] = base[(size_t)w * N + i];
}
LMC_D void StoreWords(float *base, int N, int i, const float *src, int n) {
    for (int w = 0; w < n; w++) base[(size_t)w * N + i] = src[w];
}
LMC_D void LoadPath(const float *base, int N, int i, DPath &p) {
    float *w = reinterpret_cast<float *>(&p);
    // only the used vertices are moved: head + camCount / lgtCount vertices
    for (int k = 0; k < DPATH_HEAD_WORDS; k++) w[k] = base[(size_t)k * N + i];
    for (int v = 0; v < p.camCount; v++)
        for (int k = 0; k < DVERTEX_WORDS; k++) {
            int o = DPATH_HEAD_WORDS + v * DVERTEX_WORDS + k;
            w[o] = base[(size_t)o * N + i];
        }
    for (int v = 0; v < p.lgtCount; v++)
        for (int k = 0; k < DVERTEX_WORDS; k++) {
            int o = DPATH_HEAD_WORDS + (MAXD + v) * DVERTEX_WORDS + k;
            w[o] = base[(size_t)o * N + i];
        }
}
LMC_D void StorePath(float *base, int N, int i, const DPath &p) {
    const float *w = reinterpret_cast<const float *>(&p);
    for (int k = 0; k < DPATH_HEAD_WORDS; k++) base[(size_t)k * N + i] = w[k];
    for (int v = 0; v < p.camCount; v++)
        for (int k = 0; k < DVERTEX_WORDS; k++) {
            int o = DPATH_HEAD_WORDS + v * DVERTEX_WORDS + k;
            base[(size_t)o * N + i] = w[o];
        }
    for (int v = 0; v < p.lgtCount; v++)
        for (int k = 0; k < DVERTEX_WORDS; k++) {
            int o = DPATH_HEAD_WORDS + (MAXD + v) * DVERTEX_WORDS + k;
            base[(size_t)o * N + i] = w[o];
        }
}
// chain.buffered = false (mlt.cpp:121-132 after an accepted large step, :147-169 outlier reset).  The reference zeroes the
// chain's MALA vectors at the NEXT MALA step (mutation_mala.h:59-81); nothing reads them in between (the cache push that may
// precede this call is the last reader), so they are zeroed here instead, where the stores of a wave are dense: inside the hot
// small-step kernel the same zeroing ran as 168 sparsely populated store instructions in almost every wave-step and made up a
// third of the kernel's stores.  Invariant: F_BUFFERED clear => the seven vectors are zero; F_VDIRTY clear => they are zero as well
// (nothing was written since the last zeroing), and the 168 stores are skipped.
LMC_D void ClearBuffered(const ChainArrays &A, int i, int &flags) {
    if ((flags & F_BUFFERED) && (flags & F_VDIRTY)) {
        const size_t N = A.N;
#pragma unroll 4
        for (int k = 0; k < MAXPSS; k++) {
            const size_t o = (size_t)k * N + i;
            A.chV1[o] = A.chV2[o] = A.chCurrNewV2[o] = A.chPropNewV1[o] = A.chPropNewV2[o] = A.chPss[o] = A.chLastPss[o] = 0.f;
        }
    }
    flags &= ~(F_BUFFERED | F_VDIRTY);
}
// REMOVE_OUTLIERS, mlt.cpp:151-158: currentState = initStates[_chainId] for the first id (walk: (_chainId + sampleIdx + cnt++) %
// numChains, over the chains of the WHOLE job) whose lsScore is below the outlier threshold.  The state is marked invalid by the
// caller, and of an invalid state the chain loop reads exactly one thing, spContrib.lsScore (strongReject, mlt.cpp:148; a large step
// ignores the current state when it is invalid, mutation_large.h:87-116) -- so for a chain that lives on another rank the technique
// and the score of the job-wide tables are copied and the path words stay as they are.
LMC_D void ResetToInitState(const ChainArrays &A, int chainBegin, int numChainsTotal, float threshold, int i, int sampleIdx, float *curPathBuf) {
    const size_t N = A.N;
    int chainId = chainBegin + (A.chainId ? A.chainId[i] : i), cnt = 0;  // i is the slot; the walk starts at the chain's own id
    for (;;) {
        if (A.initLsAll[chainId] < threshold) break;
        chainId = (int)(((long long)chainId + sampleIdx + cnt++) % numChainsTotal);
    }
    const int li = chainId - chainBegin;
    if (li >= 0 && li < A.N) {
#pragma unroll 1
        for (int w = 0; w < DPATH_WORDS; w++) curPathBuf[(size_t)w * N + i] = A.initPath[(size_t)w * N + li];
#pragma unroll 1
        for (int w = 0; w < CONTRIB_WORDS; w++) A.curContrib[(size_t)w * N + i] = A.initContrib[(size_t)w * N + li];
        A.scoreSum[i] = A.initScoreSum[li];
    } else {
        const int cl = A.initCLAll[chainId];
        A.curContrib[i] = __int_as_float(cl >> 4), A.curContrib[N + i] = __int_as_float(cl & 15);
#pragma unroll 1
        for (int w = 2; w < CONTRIB_WORDS; w++) A.curContrib[(size_t)w * N + i] = 0.f;
        A.curContrib[7 * N + i] = A.initLsAll[chainId];
        A.scoreSum[i] = 0.f;
    }
}
LMC_D float *CurPathBuf(const ChainArrays &A, int flags) { return (flags & F_SEL) ? A.pathBuf1 : A.curPath; }
LMC_D float *PropPathBuf(const ChainArrays &A, int flags) { return (flags & F_SEL) ? A.curPath : A.pathBuf1; }
LMC_D float *CurGaussBuf(const ChainArrays &A, int flags) { return (flags & F_GSEL) ? A.gaussian1 : A.gaussian; }
LMC_D float *PropGaussBuf(const ChainArrays &A, int flags) { return (flags & F_GSEL) ? A.gaussian : A.gaussian1; }
LMC_D Contrib LoadContrib(const float *base, int N, int i) {
    Contrib c;
    c.camDepth = __float_as_int(base[0 * (size_t)N + i]);
    c.lightDepth = __float_as_int(base[1 * (size_t)N + i]);
    c.screenPos = V2{base[2 * (size_t)N + i], base[3 * (size_t)N + i]};
    c.contrib = V3{base[4 * (size_t)N + i], base[5 * (size_t)N + i], base[6 * (size_t)N + i]};
    c.lsScore = base[7 * (size_t)N + i];
    c.ssScore = base[8 * (size_t)N + i];
    return c;
}
LMC_D void StoreContrib(float *base, int N, int i, const Contrib &c) {
    base[0 * (size_t)N + i] = __int_as_float(c.camDepth);
    base[1 * (size_t)N + i] = __int_as_float(c.lightDepth);
    base[2 * (size_t)N + i] = c.screenPos.x;
    base[3 * (size_t)N + i] = c.screenPos.y;
    base[4 * (size_t)N + i] = c.contrib.x;
    base[5 * (size_t)N + i] = c.contrib.y;
    base[6 * (size_t)N + i] = c.contrib.z;
    base[7 * (size_t)N + i] = c.lsScore;
    base[8 * (size_t)N + i] = c.ssScore;
}

struct Film {
    float *rgb;  // W*H*3
    int W, H;
};
// image.h:66-77: nearest pixel, drop non-finite, float atomics (hardware global_atomic_add_f32)
LMC_D void Splat(const Film &film, V2 screenPos, V3 contrib) {
    int ix = Clampi((int)(screenPos.x * film.W), 0, film.W - 1);
    int iy = Clampi((int)(screenPos.y * film.H), 0, film.H - 1);
    if (AllFinite(contrib)) {
        float *px = film.rgb + ((size_t)iy * film.W + ix) * 3;
        unsafeAtomicAdd(px + 0, contrib.x);
        unsafeAtomicAdd(px + 1, contrib.y);
        unsafeAtomicAdd(px + 2, contrib.z);
    }
}

// ---------------------------------------------------------------------------------------------- Gaussian
struct Gauss {
    float mean[MAXPSS], covL[MAXPSS], invCov[MAXPSS];
    float logDet;
};
LMC_D void IsotropicGaussian(int dim, float sigma, Gauss &g) {  // gaussian.cpp:4-22
    for (int i = 0; i < dim; i++) g.mean[i] = 0.0f, g.covL[i] = sigma, g.invCov[i] = 1.0f / (sigma * sigma);
    g.logDet = dim * fastlog(1.0f / (sigma * sigma));
}
LMC_D float GaussianLogPdf(int dim, const float *offset, bool negate, const Gauss &g) {  // gaussian.cpp:24-36, summed left to right
    float logPdf = dim * (-0.9189385332046727f);
    logPdf += 0.5f * g.logDet;
    float q = 0.f;
    for (int i = 0; i < dim; i++) {
        float d = (negate ? -offset[i] : offset[i]) - g.mean[i];
        q += d * (g.invCov[i] * d);
    }
    logPdf -= 0.5f * q;
    return logPdf;
}
// mala.cpp:7-52
LMC_D void ComputeGaussianMALA(int dim, const float *v1, float ss, float shk, const float *M, float sc, Gauss &g) {
    g.logDet = 0.0f;
    const float shrk = inverse(shk * shk);
    if (sc <= 1e-10f) {
        for (int i = 0; i < dim; i++) g.mean[i] = 0.0f, g.invCov[i] = shrk, g.covL[i] = shk;
        g.logDet = dim * fastlog(inverse(shk * shk));
    } else {
        for (int i = 0; i < dim; i++) {
            float cov_t = ss * ss * (M[i] + 1.0f);
            float invcov = inverse(cov_t) + shrk;
            float cov = inverse(invcov);
            g.invCov[i] = invcov;
            g.covL[i] = sqrtf(cov);
            g.mean[i] = Clampf(v1[i], MTM_MIN, MTM_MAX) * cov / 2;
            g.logDet += fastlog(invcov);
        }
    }
}

// ---------------------------------------------------------------------------------------------- cache query
// nanoflann searchLevel (nanoflann.hpp:1359-1422) unrolled onto an explicit stack; the reference-modified
// RadiusResultSet stops the whole search after `knn` matches (nanoflann.hpp:256-262).  Matches are NOT
// sorted (SearchParams::sorted defaults to false in the reference's copy, nanoflann.hpp:567).
LMC_D int KdRadiusSearch(const DCacheDim &C, int dim, const float *q, float radiusSq, int knn, int *idx, float *dist) {
    float dists[MAXPSS];
    float distsq = 0.f;
    for (int i = 0; i < dim; i++) {
        dists[i] = 0.f;
        if (q[i] < C.rootLow[i]) {
            dists[i] = (q[i] - C.rootLow[i]) * (q[i] - C.rootLow[i]);
            distsq += dists[i];
        }
        if (q[i] > C.rootHigh[i]) {
            dists[i] = (q[i] - C.rootHigh[i]) * (q[i] - C.rootHigh[i]);
            distsq += dists[i];
        }
    }
    // frame = one activation of searchLevel; divfeat / cut_dist / the two children are recomputed from the node
    struct Frame {
        int node, phase;
        float mindistsq, dst;
    };
    Frame st[KD_STACK];
    int sp = 0;
    int count = 0;
    st[sp++] = Frame{0, 0, distsq, 0.f};
    while (sp > 0) {
        Frame &f = st[sp - 1];
        const KdNode nd = C.nodes[f.node];
        if (nd.child1 < 0 && nd.child2 < 0) {
            for (int i = nd.left; i < nd.right; ++i) {
                const int index = C.vind[i];
                float d = 0.f;
                for (int k = 0; k < dim; ++k) {
                    const float diff = q[k] - C.pts[(size_t)index * dim + k];
                    d += diff * diff;
                }
                if (d < radiusSq) {
                    idx[count] = index;
                    dist[count] = d;
                    count++;
                    if (count >= knn) return count;
                }
            }
            sp--;
            continue;
        }
        const int id = nd.divfeat;
        const float val = q[id];
        const float diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
        int bestChild, otherChild;
        float cut_dist;
        if ((diff1 + diff2) < 0) {
            bestChild = nd.child1, otherChild = nd.child2;
            cut_dist = (val - nd.divhigh) * (val - nd.divhigh);
        } else {
            bestChild = nd.child2, otherChild = nd.child1;
            cut_dist = (val - nd.divlow) * (val - nd.divlow);
        }
        if (f.phase == 0) {
            f.phase = 1;
            const float m = f.mindistsq;
            if (sp >= KD_STACK) return -1;  // deeper than the host-checked bound: cannot happen (host/context.cpp refuses such trees)
            st[sp++] = Frame{bestChild, 0, m, 0.f};
            continue;
        }
        if (f.phase == 1) {
            const float dst = dists[id];
            const float mindistsq = f.mindistsq + cut_dist - dst;
            f.dst = dst;
            dists[id] = cut_dist;
            f.phase = 2;
            if (mindistsq * 1.0f <= radiusSq) {  // epsError = 1 + eps, eps = 0
                if (sp >= KD_STACK) return -1;
                st[sp++] = Frame{otherChild, 0, mindistsq, 0.f};
                continue;
            }
        }
        dists[id] = f.dst;  // phase 2: restore and return to the caller
        sp--;
    }
    return count;
}

// global_cache.h:96-124
LMC_D bool CacheQuery(const DCacheDim &C, int dim, const float *pss, float *v1, float *v2) {
    if (!C.ready) return false;
    const float radius = dim * (PSS_QUERY_DIST * PSS_QUERY_DIST);
    int idx[5];
    float dist[5];
    const int nMatches = KdRadiusSearch(C, dim, pss, radius, 5, idx, dist);
    if (nMatches <= 0) return false;  // 0 matches, or -1: tree deeper than the stack (refused on the host)
    double sum_w = 0;
    for (int i = 0; i < dim; i++) v1[i] = 0.f, v2[i] = 0.f;
    for (int k = 0; k < nMatches; k++) {
        int index = idx[k];
        float d = dist[k];
        float w = inverse(d * d + 1e-6f);
        for (int i = 0; i < dim; i++) {
            v1[i] += C.v1[(size_t)index * dim + i] * w;
            v2[i] += C.v2[(size_t)index * dim + i] * w;
        }
        sum_w += w;
    }
    for (int i = 0; i < dim; i++) {
        v1[i] = (float)((double)v1[i] / sum_w);
        v2[i] = (float)((double)v2[i] / sum_w);
    }
    return true;
}

}  // namespace lmcd
