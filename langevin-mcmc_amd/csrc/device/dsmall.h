// The hot kernel body: one plain small step (isotropic SmallStep or MALASmallStep with the global cache ready or
// not applicable) of one chain, written so that NOTHING lives in scratch memory:
//   * the path is streamed vertex by vertex from the chain's current SoA buffer in HBM into registers and the
//     perturbed vertex is streamed out to the chain's other buffer (double buffering, one select bit per chain;
//     acceptance flips the bit instead of copying) -- coalesced reads/writes of exactly the words the
//     reference's PerturbPathBidir touches (path.cpp:1953-2160);
//   * the proposal offsets sit in 16 registers consumed through a shifting queue (static indexing only);
//   * everything indexed at run time (BVH stack, kd-tree search frames / per-dimension distances, the new
//     primary-sample vector) lives in LDS, laid out [word][thread].
// Same arithmetic, same RNG order as the generic StepChain (dstep.h), which remains the implementation for
// large steps and for gradient-evaluating small steps and which the parity tests cross-check against this one.
#pragma once
#include "dstep.h"

namespace lmcd {

// LDS words per thread: a union of the BVH stack (32) and the kd search state (16 per-dimension distances + KD_LDS_DEPTH
// two-word frames), followed by the new pss vector (16): 80 words = 320 B per thread, 80 KB per 256-thread block, two
// blocks (8 waves) per CU.
constexpr int LDS_UNION_WORDS = 16 + 2 * KD_LDS_DEPTH;  // 64 >= BVH_LDS_STACK
static_assert(LDS_UNION_WORDS >= BVH_LDS_STACK, "BVH stack must fit the union region");
constexpr int LDS_WORDS_PER_THREAD = LDS_UNION_WORDS + MAXPSS;

struct LdsView {
    float *base;  // &lds[threadIdx.x]
    int stride;   // blockDim.x
    LMC_D float &U(int w) const { return base[w * stride]; }                          // union region
    LMC_D float &Q(int k) const { return base[(LDS_UNION_WORDS + k) * stride]; }      // new pss
};

// word offsets inside the SoA path record (DPath layout)
enum : int {
    PW_TIME = 0, PW_SCREEN0, PW_SCREEN1, PW_LGTPOS0, PW_LGTPOS1, PW_LGTDIR0, PW_LGTDIR1, PW_LGTLIGHT, PW_LGTPRIM, PW_ENVPRIM, PW_CAMDEPTH,
    PW_LGTDEPTH, PW_CAMCOUNT, PW_LGTCOUNT, PW_LENS0, PW_LENS1
};
LMC_D int VertWord(bool lgt, int d, int field) { return DPATH_HEAD_WORDS + ((lgt ? MAXD : 0) + d) * DVERTEX_WORDS + field; }

LMC_D DVertex LoadVertex(const float *buf, size_t N, int i, bool lgt, int d) {
    const float *p = buf + (size_t)VertWord(lgt, d, 0) * N + i;
    DVertex v;
    v.tri = __float_as_int(p[0]);
    v.st0 = p[N], v.st1 = p[2 * N], v.rnd0 = p[3 * N], v.rnd1 = p[4 * N], v.bsdfDiscrete = p[5 * N], v.useAbs = p[6 * N], v.rrWeight = p[7 * N];
    v.dirLight = __float_as_int(p[8 * N]), v.dirPrim = __float_as_int(p[9 * N]);
    v.dirRnd0 = p[10 * N], v.dirRnd1 = p[11 * N];
    return v;
}
LMC_D void StoreVertex(float *buf, size_t N, int i, bool lgt, int d, const DVertex &v) {
    float *p = buf + (size_t)VertWord(lgt, d, 0) * N + i;
    p[0] = __int_as_float(v.tri);
    p[N] = v.st0, p[2 * N] = v.st1, p[3 * N] = v.rnd0, p[4 * N] = v.rnd1, p[5 * N] = v.bsdfDiscrete, p[6 * N] = v.useAbs, p[7 * N] = v.rrWeight;
    p[8 * N] = __int_as_float(v.dirLight), p[9 * N] = __int_as_float(v.dirPrim);
    p[10 * N] = v.dirRnd0, p[11 * N] = v.dirRnd1;
}

// The proposal offsets are consumed in PerturbPathBidir's order through a cursor over U[32,48): above the BVH stack,
// inside the region the kd search later reuses for frames (the offsets are pulled into registers before that search).
constexpr int LDS_OFFSET_WORD = BVH_LDS_STACK;
struct OffsetCursor {
    const LdsView &L;
    int k = 0;
    LMC_D float Pop() { return L.U(LDS_OFFSET_WORD + k++); }
};

// kd-tree radius search with all run-time indexed state in LDS: the traversal of KdRadiusSearch (dchain.h), i.e.
// nanoflann's searchLevel with the reference's stop-after-knn result set, reorganised for a wave:
//   * "while-while": every lane first walks its frame stack until its top frame is a leaf (or the search is over), then
//     all lanes scan their leaf's points together.  As one loop with a leaf branch, a wave spent most of its
//     vector-memory instructions scanning leaves with 2-3 active lanes (profiles/r01_e);
//   * the points are read from a copy stored in leaf order (C.ptsLeaf), two coordinates per load, the query from
//     registers.
// Per lane the sequence of visited nodes, tested points and matches is unchanged.
LMC_D int KdRadiusSearchLds(const DCacheDim &C, int dim, const LdsView &L, float radiusSq, int knn, int *idx, float *dist) {
    // union layout: [0,16) dists, then KD_LDS_DEPTH x (node | phase << 30, mindistsq until phase 2 / saved dists[id] afterwards)
    float q[MAXPSS];
    float distsq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXPSS; i++) {
        q[i] = 0.f;
        if (i < dim) {
            q[i] = L.Q(i);
            float d = 0.f;
            if (q[i] < C.rootLow[i]) d = (q[i] - C.rootLow[i]) * (q[i] - C.rootLow[i]);
            if (q[i] > C.rootHigh[i]) d = (q[i] - C.rootHigh[i]) * (q[i] - C.rootHigh[i]);
            L.U(i) = d;
            distsq += d;
        }
    }
    int sp = 0;
    int count = 0;
    auto FN = [&](int lvl) -> float & { return L.U(16 + 2 * lvl); };
    auto FM = [&](int lvl) -> float & { return L.U(16 + 2 * lvl + 1); };
    FN(0) = __int_as_float(0), FM(0) = distsq;  // node 0, phase 0 (phase in the top 2 bits)
    sp = 1;
    for (;;) {
        // ---- walk until the top frame is a leaf
        int leafLeft = 0, leafRight = 0;
        bool atLeaf = false;
        while (sp > 0) {
            const int lvl = sp - 1;
            const int packed = __float_as_int(FN(lvl));
            const int node = packed & 0x3fffffff, phase = (unsigned)packed >> 30;
            const KdNode nd = C.nodes[node];
            if (nd.child1 < 0 && nd.child2 < 0) {
                leafLeft = nd.left, leafRight = nd.right;
                atLeaf = true;
                break;
            }
            const int id = nd.divfeat;
            const float val = L.Q(id);
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
            if (phase == 0) {
                FN(lvl) = __int_as_float(node | (1 << 30));
                if (sp >= KD_LDS_DEPTH) return -1;  // the host routes deeper trees to the generic kernel
                FM(sp) = FM(lvl), FN(sp) = __int_as_float(bestChild);
                sp++;
                continue;
            }
            if (phase == 1) {
                const float dst = L.U(id);
                const float mindistsq = FM(lvl) + cut_dist - dst;
                FM(lvl) = dst;  // the frame's mindistsq is dead from here on: the word now keeps dists[id] for the restore
                L.U(id) = cut_dist;
                FN(lvl) = __int_as_float(node | (2 << 30));
                if (mindistsq * 1.0f <= radiusSq) {
                    if (sp >= KD_LDS_DEPTH) return -1;
                    FM(sp) = mindistsq, FN(sp) = __int_as_float(otherChild);
                    sp++;
                    continue;
                }
            }
            L.U(id) = FM(lvl);
            sp--;
        }
        if (!atLeaf) break;
        // ---- scan the leaf
        for (int i = leafLeft; i < leafRight; ++i) {
            const float2 *row = reinterpret_cast<const float2 *>(C.ptsLeaf + (size_t)i * dim);
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < MAXPSS / 2; ++k)
                if (2 * k < dim) {
                    const float2 p = row[k];
                    const float diff0 = q[2 * k] - p.x;
                    d += diff0 * diff0;
                    const float diff1 = q[2 * k + 1] - p.y;
                    d += diff1 * diff1;
                }
            if (d < radiusSq) {
                idx[count] = C.vind[i];
                dist[count] = d;
                count++;
                if (count >= knn) return count;
            }
        }
        sp--;
    }
    return count;
}

struct GaussR {  // Gaussian in registers: statically indexed arrays
    float mean[MAXPSS], covL[MAXPSS], invCov[MAXPSS];
    float logDet;
};

// cache / isotropic branch of InitGaussianFor (mutation_mala.h:131-164 and :224-257) for a state whose pss is in L.Q;
// never evaluates a gradient (chains that need one are dispatched to the gradient-capable launch).
LMC_D void InitGaussianLean(const DScene &S, const DCache &cache, const ChainArrays &A, int i, int dim, float lsScore, float ssScore, int &flags,
                            const LdsView &L, GaussR &g, StepStats &st) {
    const size_t N = A.N;
#pragma unroll
    for (int k = 0; k < MAXPSS; k++)
        if (k < dim) A.chPss[(size_t)k * N + i] = L.Q(k);
    A.pathWeight[i] = lsScore;
    const bool inRange = dim >= PSS_MIN_LENGTH && dim <= PSS_MAX_LENGTH;
    const bool ready = inRange && cache.d[dim].ready;
    const float ss = S.opt.malaStepsize, shk = S.opt.malaStdDev;
    bool fromV = false;
    float v1[MAXPSS], v2[MAXPSS];
    if (ready) {
        bool reuse = false;
        if (flags & F_QUERIED) {
            float dist_sqr = 0.f;
#pragma unroll
            for (int k = 0; k < MAXPSS; k++)
                if (k < dim) {
                    float diff = L.Q(k) - A.chLastPss[(size_t)k * N + i];
                    dist_sqr += diff * diff;
                }
            if (dist_sqr < dim * (PSS_REUSE_DIST * PSS_REUSE_DIST)) reuse = true;
        }
        if (reuse) {
#pragma unroll
            for (int k = 0; k < MAXPSS; k++)
                if (k < dim) v1[k] = A.chV1[(size_t)k * N + i], v2[k] = A.chV2[(size_t)k * N + i];
            fromV = true;
        } else {
            st.cacheQueries++;
            const DCacheDim &C = cache.d[dim];
            int idx[5];
            float dist[5];
            const int nMatches = KdRadiusSearchLds(C, dim, L, dim * (PSS_QUERY_DIST * PSS_QUERY_DIST), 5, idx, dist);
            if (nMatches > 0) {  // global_cache.h:106-123
                st.cacheHits++;
                double sum_w = 0;
#pragma unroll
                for (int k = 0; k < MAXPSS; k++) v1[k] = 0.f, v2[k] = 0.f;
#pragma unroll
                for (int m = 0; m < 5; m++)
                    if (m < nMatches) {
                        const int index = idx[m];
                        const float d = dist[m];
                        const float w = inverse(d * d + 1e-6f);
#pragma unroll
                        for (int k = 0; k < MAXPSS; k++)
                            if (k < dim) {
                                v1[k] += C.v1[(size_t)index * dim + k] * w;
                                v2[k] += C.v2[(size_t)index * dim + k] * w;
                            }
                        sum_w += w;
                    }
#pragma unroll
                for (int k = 0; k < MAXPSS; k++) {
                    if (k < dim) {
                        v1[k] = (float)((double)v1[k] / sum_w);
                        v2[k] = (float)((double)v2[k] / sum_w);
                    }
                    A.chV1[(size_t)k * N + i] = k < dim ? v1[k] : 0.f;
                    A.chV2[(size_t)k * N + i] = k < dim ? v2[k] : 0.f;
                    A.chLastPss[(size_t)k * N + i] = A.chPss[(size_t)k * N + i];  // last_pss = pss (whole vector)
                }
                flags |= F_QUERIED;
                fromV = true;
            }
        }
    }
    if (fromV) {  // M + ComputeGaussian, mala.cpp:7-52
        g.logDet = 0.0f;
        const float shrk = inverse(shk * shk);
        if (ssScore <= 1e-10f) {
#pragma unroll
            for (int k = 0; k < MAXPSS; k++) g.mean[k] = 0.0f, g.invCov[k] = shrk, g.covL[k] = shk;
            g.logDet = dim * fastlog(inverse(shk * shk));
        } else {
#pragma unroll
            for (int k = 0; k < MAXPSS; k++)
                if (k < dim) {
                    const float M = Clampf(1.0f / (1e-3f + sqrtf(v2[k])), PCD_MIN, PCD_MAX);
                    float cov_t = ss * ss * (M + 1.0f);
                    float invcov = inverse(cov_t) + shrk;
                    float cov = inverse(invcov);
                    g.invCov[k] = invcov;
                    g.covL[k] = sqrtf(cov);
                    g.mean[k] = Clampf(v1[k], MTM_MIN, MTM_MAX) * cov / 2;
                    g.logDet += fastlog(invcov);
                }
        }
    } else {  // IsotropicGaussian, gaussian.cpp:4-22
#pragma unroll
        for (int k = 0; k < MAXPSS; k++) g.mean[k] = 0.0f, g.covL[k] = shk, g.invCov[k] = 1.0f / (shk * shk);
        g.logDet = dim * fastlog(1.0f / (shk * shk));
    }
}

// One plain small step of chain i.  Returns nothing; all state changes go to HBM.
template <class Stk>
LMC_D void SmallStepLean(const DScene &S, const DCache &cache, const ChainArrays &A, const Film &film, const StepParams &P, int i, Rng &rng,
                         const LdsView &L, Stk &stk, StepStats &st) {
    const size_t N = A.N;
    int flags = A.flags[i];
    const int sel = (flags & F_SEL) ? 1 : 0;
    const float *cur = sel ? A.pathBuf1 : A.curPath;
    float *prop = sel ? A.curPath : A.pathBuf1;
    const bool curValid = flags & F_VALID;  // always true for a small step
    const int c = __float_as_int(A.curContrib[i]), l = __float_as_int(A.curContrib[N + i]);
    const float curLs = A.curContrib[7 * N + i], curSs = A.curContrib[8 * N + i];
    const int dim = PathDimension(c, l);
    const int camCount = max(c - 1, 0), lgtCount = max(l - 1, 0);
    st.steps++;
    st.lean++;

    // ---- proposal offsets
    const bool mala = S.opt.mala && !(rng.Uniform() < S.opt.uniformMixingProbability);  // mutation_mala.h:46-51
    OffsetCursor off{L};
    float py = 0.f;
    if (!mala) {  // SmallStep::Mutate, mutation_small.h:29-37
        NormalDist nd(0.0f, S.opt.perturbStdDev);
        for (int k = 0; k < dim; k++) L.U(LDS_OFFSET_WORD + k) = nd(rng);
    } else {
        if (!(flags & F_BUFFERED)) {  // mutation_mala.h:59-81
            for (int k = 0; k < MAXPSS; k++) {
                size_t o = (size_t)k * N + i;
                A.chV1[o] = A.chV2[o] = A.chCurrNewV2[o] = A.chPropNewV1[o] = A.chPropNewV2[o] = A.chPss[o] = A.chLastPss[o] = 0.f;
            }
            flags |= F_BUFFERED;
            flags &= ~F_QUERIED;
        }
        GaussR cg;
        if (!(flags & F_GAUSS)) {
            // GetPathPss(currentState.path) into LDS, path.cpp:2588-2632
            int k = 0;
            if (l > 1) {
                L.Q(k++) = cur[(size_t)PW_LGTPOS0 * N + i], L.Q(k++) = cur[(size_t)PW_LGTPOS1 * N + i];
                L.Q(k++) = cur[(size_t)PW_LGTDIR0 * N + i], L.Q(k++) = cur[(size_t)PW_LGTDIR1 * N + i];
                for (int d = 0; d < lgtCount - 1; d++)
                    L.Q(k++) = cur[(size_t)VertWord(true, d, 3) * N + i], L.Q(k++) = cur[(size_t)VertWord(true, d, 4) * N + i];
            }
            if (c > 1) {
                L.Q(k++) = cur[(size_t)PW_SCREEN0 * N + i], L.Q(k++) = cur[(size_t)PW_SCREEN1 * N + i];
                for (int d = 0; d < camCount - 1; d++)
                    L.Q(k++) = cur[(size_t)VertWord(false, d, 3) * N + i], L.Q(k++) = cur[(size_t)VertWord(false, d, 4) * N + i];
                if (l == 1) L.Q(k++) = cur[(size_t)VertWord(false, camCount - 1, 10) * N + i], L.Q(k++) = cur[(size_t)VertWord(false, camCount - 1, 11) * N + i];
            }
            InitGaussianLean(S, cache, A, i, dim, curLs, curSs, flags, L, cg, st);
#pragma unroll
            for (int k2 = 0; k2 < MAXPSS; k2++)
                if (k2 < dim) {
                    A.gaussian[(size_t)k2 * N + i] = cg.mean[k2];
                    A.gaussian[(size_t)(MAXPSS + k2) * N + i] = cg.covL[k2];
                    A.gaussian[(size_t)(2 * MAXPSS + k2) * N + i] = cg.invCov[k2];
                }
            A.gaussian[(size_t)(3 * MAXPSS) * N + i] = cg.logDet;
            flags |= F_GAUSS;
        } else {
#pragma unroll
            for (int k = 0; k < MAXPSS; k++)
                if (k < dim) {
                    cg.mean[k] = A.gaussian[(size_t)k * N + i];
                    cg.covL[k] = A.gaussian[(size_t)(MAXPSS + k) * N + i];
                    cg.invCov[k] = A.gaussian[(size_t)(2 * MAXPSS + k) * N + i];
                }
            cg.logDet = A.gaussian[(size_t)(3 * MAXPSS) * N + i];
        }
        NormalDist nd(0.0f, 1.0f);  // GenerateSample, gaussian.cpp:38-55 (the affine map draws nothing, so it is fused)
        float q = 0.f;              // GaussianLogPdf(offset, currentState.gaussian), gaussian.cpp:24-36
#pragma unroll
        for (int k = 0; k < MAXPSS; k++)
            if (k < dim) {
                const float o = cg.covL[k] * nd(rng) + cg.mean[k];
                L.U(LDS_OFFSET_WORD + k) = o;
                const float d = o - cg.mean[k];
                q += d * (cg.invCov[k] * d);
            }
        py = dim * (-0.9189385332046727f);
        py += 0.5f * cg.logDet;
        py -= 0.5f * q;
    }

    // ---- PerturbPathBidir, path.cpp:1953-2160, streamed
    Contrib pc;
    pc.camDepth = pc.lightDepth = 0;
    pc.lsScore = pc.ssScore = 0.f;
    bool ok = false;
    int qn = 0;  // number of new pss values written to L.Q
    {
        NormalDist normDist(0.0f, S.opt.discreteStdDev);
        const float time = Modulo1(cur[(size_t)PW_TIME * N + i] + normDist(rng));
        prop[(size_t)PW_TIME * N + i] = time;
        prop[(size_t)PW_CAMDEPTH * N + i] = __int_as_float(c), prop[(size_t)PW_LGTDEPTH * N + i] = __int_as_float(l);
        prop[(size_t)PW_CAMCOUNT * N + i] = __int_as_float(camCount), prop[(size_t)PW_LGTCOUNT * N + i] = __int_as_float(lgtCount);
        int envPrim = (l == 0) ? __float_as_int(cur[(size_t)PW_ENVPRIM * N + i]) : -1;  // ToSubpath: -1 unless lgtDepth == 0
        BPS lps;
        DVertex lastLgt;
        lastLgt.tri = -1;
        V3 org, dir;
        bool done = false;  // a terminal strategy has been evaluated (or the path died)
        int lgtLight = -1;
        if (l > 1) {
            lgtLight = __float_as_int(cur[(size_t)PW_LGTLIGHT * N + i]);
            const float lightPickProb = PickLightProb(S, lgtLight);
            DPath hd;  // only the emitter fields are used by EmitFromLight
            hd.lgtPos0 = Modulo1(cur[(size_t)PW_LGTPOS0 * N + i] + off.Pop());
            hd.lgtPos1 = Modulo1(cur[(size_t)PW_LGTPOS1 * N + i] + off.Pop());
            hd.lgtDir0 = Modulo1(cur[(size_t)PW_LGTDIR0 * N + i] + off.Pop());
            hd.lgtDir1 = Modulo1(cur[(size_t)PW_LGTDIR1 * N + i] + off.Pop());
            hd.lgtLight = lgtLight;
            hd.lgtPrim = __float_as_int(cur[(size_t)PW_LGTPRIM * N + i]);
            L.Q(qn++) = hd.lgtPos0, L.Q(qn++) = hd.lgtPos1, L.Q(qn++) = hd.lgtDir0, L.Q(qn++) = hd.lgtDir1;
            EmitFromLight(S, lightPickProb, hd, org, dir, lps);
            prop[(size_t)PW_LGTPOS0 * N + i] = hd.lgtPos0, prop[(size_t)PW_LGTPOS1 * N + i] = hd.lgtPos1;
            prop[(size_t)PW_LGTDIR0 * N + i] = hd.lgtDir0, prop[(size_t)PW_LGTDIR1 * N + i] = hd.lgtDir1;
            prop[(size_t)PW_LGTLIGHT * N + i] = __int_as_float(lgtLight), prop[(size_t)PW_LGTPRIM * N + i] = __int_as_float(hd.lgtPrim);
            for (int lgtDepth = 0; lgtDepth < lgtCount && !done; lgtDepth++) {
                DVertex sv = LoadVertex(cur, N, i, true, lgtDepth);
                SurfHit hit;
                if (!IntersectSurface(S, org, dir, c_IsectEpsilon, INFINITY, hit, lps.isect, stk)) {
                    done = true;
                    break;
                }
                sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
                lps.wi = -dir;
                sv.bsdfDiscrete = Modulo1(sv.bsdfDiscrete + normDist(rng));
                ConvertMIS(S, lgtDepth, lgtLight, org, dir, lps);
                if (lgtDepth == lgtCount - 1 && c == 1) {
                    ok = ConnectToCamera(S, lgtDepth, lps, sv, pc, stk);
                    StoreVertex(prop, N, i, true, lgtDepth, sv);
                    done = true;
                    break;
                }
                if (lgtDepth == lgtCount - 1) {
                    StoreVertex(prop, N, i, true, lgtDepth, sv);
                    lastLgt = sv;
                    break;
                }
                sv.rnd0 = Modulo1(sv.rnd0 + off.Pop());
                sv.rnd1 = Modulo1(sv.rnd1 + off.Pop());
                L.Q(qn++) = sv.rnd0, L.Q(qn++) = sv.rnd1;
                V3 bsdfContrib;
                if (!BSDFSampling<true, true, Stk::kGlossy>(S, lps, sv, lps, dir, bsdfContrib)) {
                    done = true;
                    break;
                }
                StoreVertex(prop, N, i, true, lgtDepth, sv);
                lps.throughput = lps.throughput * sv.rrWeight;
                org = lps.isect.position;
            }
        }
        if (!done) {
            const float screen0 = Modulo1(cur[(size_t)PW_SCREEN0 * N + i] + off.Pop());
            const float screen1 = Modulo1(cur[(size_t)PW_SCREEN1 * N + i] + off.Pop());
            prop[(size_t)PW_SCREEN0 * N + i] = screen0, prop[(size_t)PW_SCREEN1 * N + i] = screen1;
            L.Q(qn++) = screen0, L.Q(qn++) = screen1;
            const V2 screenPos{screen0, screen1};
            BPS cps;
            EmitFromCamera(S, screenPos, org, dir, cps);
            float tnear, tfar;
            tnear = PrimaryMinT(S, screenPos, tfar);
            for (int camDepth = 0; camDepth < camCount; camDepth++) {
                DVertex sv = LoadVertex(cur, N, i, false, camDepth);
                SurfHit hit;
                hit.tri = -1;
                hit.st = V2{0.f, 0.f};
                const bool hitSurface = IntersectSurface(S, org, dir, tnear, tfar, hit, cps.isect, stk);
                sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
                cps.wi = -dir;
                if (hitSurface) ConvertMIS(S, camDepth, -1, org, dir, cps);
                if (camDepth == camCount - 1 && l == 0) {
                    const int light = HitLightOf(S, hitSurface, hit.tri);
                    if (light >= 0) ok = HandleHitLight(S, camDepth, light, hitSurface, dir, screenPos, cps, envPrim, pc);
                    StoreVertex(prop, N, i, false, camDepth, sv);
                    break;
                }
                if (!hitSurface) break;
                sv.bsdfDiscrete = Modulo1(sv.bsdfDiscrete + normDist(rng));
                if (camDepth == camCount - 1) {
                    if (l == 1) {
                        const float directLightPickProb = PickLightProb(S, sv.dirLight);
                        sv.dirRnd0 = Modulo1(sv.dirRnd0 + off.Pop());
                        sv.dirRnd1 = Modulo1(sv.dirRnd1 + off.Pop());
                        L.Q(qn++) = sv.dirRnd0, L.Q(qn++) = sv.dirRnd1;
                        ok = DirectLighting(S, camDepth, cps, screenPos, directLightPickProb, sv, pc, stk);
                    } else {
                        ok = ConnectVertex(S, camDepth, lgtCount - 1, lps, lastLgt, cps, sv, screenPos, pc, stk);
                    }
                    StoreVertex(prop, N, i, false, camDepth, sv);
                    break;
                }
                sv.rnd0 = Modulo1(sv.rnd0 + off.Pop());
                sv.rnd1 = Modulo1(sv.rnd1 + off.Pop());
                L.Q(qn++) = sv.rnd0, L.Q(qn++) = sv.rnd1;
                V3 bsdfContrib;
                if (!BSDFSampling<false, true, Stk::kGlossy>(S, cps, sv, cps, dir, bsdfContrib)) break;
                StoreVertex(prop, N, i, false, camDepth, sv);
                cps.throughput = cps.throughput * sv.rrWeight;
                org = cps.isect.position;
                tnear = c_IsectEpsilon;
                tfar = INFINITY;
            }
        }
        prop[(size_t)PW_ENVPRIM * N + i] = __int_as_float(envPrim);
    }

    // ---- proposal Gaussian + acceptance probability
    float a = 0.0f;
    GaussR pg;
    if (ok) {
        if (mala) {
            float offKeep[MAXPSS];  // the kd search reuses their LDS words
#pragma unroll
            for (int k = 0; k < MAXPSS; k++) offKeep[k] = (k < dim) ? L.U(LDS_OFFSET_WORD + k) : 0.f;
            InitGaussianLean(S, cache, A, i, dim, pc.lsScore, pc.ssScore, flags, L, pg, st);
            float q = 0.f;  // GaussianLogPdf(-offset, proposalState.gaussian)
#pragma unroll
            for (int k = 0; k < MAXPSS; k++)
                if (k < dim) {
                    const float d = -offKeep[k] - pg.mean[k];
                    q += d * (pg.invCov[k] * d);
                }
            float px = dim * (-0.9189385332046727f);
            px += 0.5f * pg.logDet;
            px -= 0.5f * q;
            a = Clampf(expf(px - py) * pc.ssScore / curSs, 0.0f, 1.0f);
        } else {
            a = Clampf(pc.ssScore / curSs, 0.0f, 1.0f);
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
    const V3 smallSplat = mala ? (pc.contrib * P.normalization) / pc.lsScore : pc.contrib * (P.normalization / pc.lsScore);
    if (a > 0.0f) Splat(film, pc.screenPos, a * smallSplat);
    st.wsum += curValid ? 1.0f : (a > 0.0f ? a : 0.0f);

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
        if (mala) {  // mlt.cpp:133-142
            for (int k = 0; k < MAXPSS; k++) {
                A.chV1[(size_t)k * N + i] = A.chPropNewV1[(size_t)k * N + i];
                A.chV2[(size_t)k * N + i] = A.chPropNewV2[(size_t)k * N + i];
            }
            flags |= F_BUFFERED | F_GAUSS;
#pragma unroll
            for (int k = 0; k < MAXPSS; k++)
                if (k < dim) {
                    A.gaussian[(size_t)k * N + i] = pg.mean[k];
                    A.gaussian[(size_t)(MAXPSS + k) * N + i] = pg.covL[k];
                    A.gaussian[(size_t)(2 * MAXPSS + k) * N + i] = pg.invCov[k];
                }
            A.gaussian[(size_t)(3 * MAXPSS) * N + i] = pg.logDet;
        } else {
            flags &= ~F_GAUSS;
        }
        flags |= F_VALID;
    } else {
        int rej = A.adjacentReject[i] + 1;  // REMOVE_OUTLIERS, mlt.cpp:147-169
        A.adjacentReject[i] = rej;
        const bool strongReject = curLs > OUTLIER_RATIO_THRESHOLD * P.normalization;
        if (rej > OUTLIER_WEAK_REJECT_CNT || (strongReject && rej > OUTLIER_STRONG_REJECT_CNT)) {
            int chainId = i + P.chainBegin, cnt = 0;
            for (;;) {
                const float ls = A.initContrib[7 * (size_t)P.numChains + chainId];
                if (ls < OUTLIER_RATIO_THRESHOLD * P.normalization) break;
                chainId = (int)(((long long)chainId + sampleIdx + cnt++) % P.numChains);
            }
            float *curW = sel ? A.pathBuf1 : A.curPath;
            for (int w = 0; w < DPATH_WORDS; w++) curW[(size_t)w * N + i] = A.initPath[(size_t)w * P.numChains + chainId];
            for (int w = 0; w < CONTRIB_WORDS; w++) A.curContrib[(size_t)w * N + i] = A.initContrib[(size_t)w * P.numChains + chainId];
            A.scoreSum[i] = A.initScoreSum[chainId];
            A.curSplatCount[i] = 0;
            flags &= ~(F_VALID | F_GAUSS | F_BUFFERED);
            st.resets++;
        }
    }
    A.flags[i] = flags;
    A.sampleIdx[i] = sampleIdx + 1;
}

}  // namespace lmcd
