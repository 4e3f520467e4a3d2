// One Markov-chain step per thread: mlt.cpp:91-170 with LargeStep / SmallStep / MALASmallStep inlined.
// Scheduling contract: all chains advance in lock step; global-cache readiness and contents are
// frozen for the duration of a step and the step's pushes are applied afterwards in chain-id order.
#pragma once
#include "dchain.h"
#include "dgrad.h"
#include "dstep_params.h"

namespace lmcd {

struct StepStats {  // per-thread increments, block-reduced by the kernel
    int steps = 0, large = 0, accepted = 0, gradCalls = 0, cacheQueries = 0, cacheHits = 0, resets = 0, lean = 0;
    float wsum = 0.f;
};

// the chain's current Gaussian lives in the buffer F_GSEL selects (dchain.h)
LMC_D void LoadGauss(const DScene &S, const ChainArrays &A, int i, int dim, int flags, Gauss &g) {
    if (flags & F_GAUSS_ISO) {  // left by the lean kernel: the isotropic Gaussian is never stored (dchain.h)
        IsotropicGaussian(dim, S.opt.malaStdDev, g);
        return;
    }
    const size_t N = A.N;
    const float *G = CurGaussBuf(A, flags);
    for (int k = 0; k < dim; k++) {
        g.mean[k] = G[(size_t)k * N + i];
        g.covL[k] = G[(size_t)(MAXPSS + k) * N + i];
        g.invCov[k] = G[(size_t)(2 * MAXPSS + k) * N + i];
    }
    g.logDet = G[(size_t)(3 * MAXPSS) * N + i];
}
LMC_D void StoreGauss(const ChainArrays &A, int i, int dim, int flags, const Gauss &g) {
    const size_t N = A.N;
    float *G = CurGaussBuf(A, flags);
    for (int k = 0; k < dim; k++) {
        G[(size_t)k * N + i] = g.mean[k];
        G[(size_t)(MAXPSS + k) * N + i] = g.covL[k];
        G[(size_t)(2 * MAXPSS + k) * N + i] = g.invCov[k];
    }
    G[(size_t)(3 * MAXPSS) * N + i] = g.logDet;
}

// The twin blocks of MALASmallStep::Mutate (mutation_mala.h:83-166 current, :174-260 proposal).
// gradIn: the state's gradient where the step is run as a pipeline of launches (step_mala_phases.hip), else it is evaluated here.
// Persistent Chain vectors (mutation.h:28-43) live in HBM: v1, v2, curr_new_v2, prop_new_v1, prop_new_v2, pss,
// last_pss (g / curr_new_v1 / curr_new_g / prop_new_g / M are write-only or recomputed in the reference).
template <bool WITH_GRAD>
LMC_D void InitGaussianFor(const DScene &S, const DCache &cache, const ChainArrays &A, const StepParams &P, int i, const DPath &path,
                           const Contrib &sp, bool isProposal, int &flags, Gauss &g, GradWork &gw, StepStats &st, const float *gradIn = nullptr) {
    const size_t N = A.N;
    const int dim = PathDimension(path.camDepth, path.lgtDepth);
    float pss[MAXPSS];
    for (int k = 0; k < MAXPSS; k++) pss[k] = 0.f;
    GetPathPss(path, pss);
    for (int k = 0; k < dim; k++) A.chPss[(size_t)k * N + i] = pss[k];  // GetPathPss(path, chain->pss)
    A.pathWeight[i] = sp.lsScore;
    if (A.chPath) {  // chain->path / chain->spContrib, mutation_mala.h:90-91,185-186 (`samplecache`)
        StorePath(A.chPath, A.N, i, path);
        StoreContrib(A.chContrib, A.N, i, sp);
    }
    const bool inRange = dim >= PSS_MIN_LENGTH && dim <= PSS_MAX_LENGTH;
    const bool ready = inRange && cache.d[dim].ready;
    const bool haveDerv = P.useGradient && GradAvailable(path.camDepth, path.lgtDepth) && path.camDepth + path.lgtDepth - 1 <= P.maxDervDepth;
    const float ss = S.opt.malaStepsize, shk = S.opt.malaStdDev;
    float M[MAXPSS];
    if (inRange && !ready && haveDerv) {
        float vGrad[MAXPSS];
        for (int k = 0; k < dim; k++) vGrad[k] = 0.f;
        if (sp.ssScore > 1e-10f) {
            // chains are only dispatched to a launch without the gradient code once the cache of their dim is
            // ready (NeedsGradient below), so this branch is unreachable there; NaN -> zeroed keeps it defined
            if (gradIn) {  // evaluated by the launch in front of this one (step_mala_phases.hip, gradcoop.hip)
                for (int k = 0; k < dim; k++) vGrad[k] = gradIn[k];
            } else if (WITH_GRAD && !LMC_EXP(P.expFlags, 4)) ComputeGradient(S, path, sp, vGrad, gw);
            else
                for (int k = 0; k < dim; k++) vGrad[k] = NAN;
            st.gradCalls++;
            bool finite = true;
            for (int k = 0; k < dim; k++) finite = finite && isfinite(vGrad[k]);
            if (!finite)
                for (int k = 0; k < dim; k++) vGrad[k] = 0.f;
        }
        float norm = 0.f, drift = S.opt.malaGN;
        for (int k = 0; k < dim; k++) norm += vGrad[k] * vGrad[k];
        norm = sqrtf(norm);
        for (int k = 0; k < dim; k++) vGrad[k] *= drift / fmaxf(drift, norm);
        float *newV2 = isProposal ? A.chPropNewV2 : A.chCurrNewV2;
        bool first = true;
        for (int k = 0; k < dim; k++)
            if (newV2[(size_t)k * N + i] > 1e-10f) {
                first = false;
                break;
            }
        float nv1[MAXPSS];
        for (int k = 0; k < dim; k++) {
            float gk = vGrad[k];
            float v1 = A.chV1[(size_t)k * N + i], v2 = A.chV2[(size_t)k * N + i];
            nv1[k] = first ? gk : 0.9f * v1 + 0.1f * gk;
            float nv2 = first ? gk * gk : 0.999f * v2 + 0.001f * gk * gk;
            newV2[(size_t)k * N + i] = nv2;
            if (isProposal) A.chPropNewV1[(size_t)k * N + i] = nv1[k];
            M[k] = Clampf(1.0f / (1e-3f + sqrtf(nv2)), PCD_MIN, PCD_MAX);
        }
        ComputeGaussianMALA(dim, nv1, ss, shk, M, sp.ssScore, g);
    } else if (ready) {
        bool reuse = false;
        if (flags & F_QUERIED) {
            float dist_sqr = 0.f;
            for (int k = 0; k < dim; k++) {
                float diff = pss[k] - A.chLastPss[(size_t)k * N + i];
                dist_sqr += diff * diff;
            }
            if (dist_sqr < dim * (PSS_REUSE_DIST * PSS_REUSE_DIST)) reuse = true;
        }
        float v1[MAXPSS], v2[MAXPSS];
        bool fromV = false;
        if (reuse) {
            for (int k = 0; k < dim; k++) v1[k] = A.chV1[(size_t)k * N + i], v2[k] = A.chV2[(size_t)k * N + i];
            fromV = true;
        } else {
            st.cacheQueries++;
            if (CacheQuery(cache.d[dim], dim, pss, v1, v2)) {
                st.cacheHits++;
                // query() zero-fills the whole 2*maxDepth vectors before accumulating (global_cache.h:107-108)
                for (int k = 0; k < MAXPSS; k++) {
                    A.chV1[(size_t)k * N + i] = k < dim ? v1[k] : 0.f;
                    A.chV2[(size_t)k * N + i] = k < dim ? v2[k] : 0.f;
                }
                flags |= F_QUERIED;
                for (int k = 0; k < MAXPSS; k++) A.chLastPss[(size_t)k * N + i] = A.chPss[(size_t)k * N + i];  // last_pss = pss
                fromV = true;
            }
        }
        if (fromV) {
            for (int k = 0; k < dim; k++) M[k] = Clampf(1.0f / (1e-3f + sqrtf(v2[k])), PCD_MIN, PCD_MAX);
            ComputeGaussianMALA(dim, v1, ss, shk, M, sp.ssScore, g);
        } else {
            IsotropicGaussian(dim, shk, g);
        }
    } else {
        IsotropicGaussian(dim, shk, g);
    }
}

// mlt.cpp:96-97.  Drawn by the launch that precedes the step so that large and small steps can be
// dispatched as separate, divergence-free launches (the RNG order is unchanged: nothing draws in between).
LMC_D int DecideKind(const DScene &S, const ChainArrays &A, int i, Rng &rng) {
    if (!(A.flags[i] & F_VALID)) return KIND_LARGE;
    const int sampleIdx = A.sampleIdx[i];
    const float lsScale = ((float)sampleIdx > (float)A.numSamples[i] * LS_RATIO) ? S.opt.largeStepProbScale : 1.0f;
    return (rng.Uniform() < S.opt.largeStepProbability * lsScale) ? KIND_LARGE : KIND_SMALL;
}

// Will the next small step of a chain whose state has dimension `dim` evaluate a gradient?  (mutation_mala.h:94-96)
LMC_D bool NeedsGradient(const DCache &cache, const StepParams &P, int camDepth, int lgtDepth) {
    const int dim = PathDimension(camDepth, lgtDepth);
    return P.useGradient && dim >= PSS_MIN_LENGTH && dim <= PSS_MAX_LENGTH && !cache.d[dim].ready && GradAvailable(camDepth, lgtDepth) &&
           camDepth + lgtDepth - 1 <= P.maxDervDepth;
}

// ... or query a cache tree too deep for the lean kernel's LDS search (dsmall.h)?  Such chains run the generic kernel.
LMC_D bool NeedsGeneric(const DCache &cache, const StepParams &P, int camDepth, int lgtDepth) {
    const int dim = PathDimension(camDepth, lgtDepth);
    const bool deep = dim >= PSS_MIN_LENGTH && dim <= PSS_MAX_LENGTH && cache.d[dim].ready && cache.d[dim].deep;
    return deep || NeedsGradient(cache, P, camDepth, lgtDepth);
}

// A.nextKind byte: bits 0-1 = which launch runs the chain's next step (NEXT_*), bits 2-7 = sort key of plain small steps.
// k_build_lists groups the plain entries of every 1024-chain tile by this key so that the 64 chains of a wave retrace the
// same technique (c,l): same number of rays, same terminal strategy, same code path.  Path length first, then the
// light-subpath length.
// (TechniqueKey itself: dchain.h)
// End of a step: decide the next step's kind now (mlt.cpp:96-97; nothing else draws in between, so the RNG order is the
// reference's) and publish it for k_build_lists.
LMC_D unsigned char QueueNext(const DScene &S, const DCache &cache, const ChainArrays &A, const StepParams &P, int i, Rng &rng) {
    unsigned char nk = NEXT_DONE;
    if (A.sampleIdx[i] < A.numSamples[i]) {
        if (DecideKind(S, A, i, rng) == KIND_LARGE) {
            nk = NEXT_LARGE;
        } else {
            const int c = __float_as_int(A.curContrib[i]), l = __float_as_int(A.curContrib[(size_t)A.N + i]);
            if (S.opt.h2mc) nk = (unsigned char)(NEXT_SMALL_GENERIC | (TechniqueKey(c, l) << 2));  // every H2MC small step runs k_step_h2mc (the "generic" slot of the launch plan); key: list sort
            else if (S.opt.useLightCoord || S.opt.sampleCache || (S.opt.leanLightless && l > 1) || (S.opt.mala && NeedsGeneric(cache, P, c, l))) nk = (unsigned char)(NEXT_SMALL_GENERIC | (TechniqueKey(c, l) << 2));  // key: k_build_lists may still re-route it
            else nk = (unsigned char)(NEXT_SMALL_PLAIN | (TechniqueKey(c, l) << 2));
        }
    }
    A.nextKind[i] = nk;
    return nk;
}

// PiecewiseConstant1D::SampleDiscrete / Pmf (distribution.h:43-53) on StepParams' lengthDist
LMC_D int LengthSampleDiscrete(const StepParams &P, float u) {
    int pos = P.lengthCount + 1;  // std::upper_bound(cdf, cdf + count + 1, u): first element > u
    for (int k = P.lengthCount; k >= 0; k--)
        if (P.lengthCdf[k] > u) pos = k;
    return Clampi(pos - 1, 0, P.lengthCount - 1);
}
LMC_D float LengthPmf(const StepParams &P, int length) { return P.lengthFunc[length] / (P.lengthFuncInt * float(P.lengthCount)); }

// ---- LargeStepCache (`samplecache` with mala), mutation_large_cache.h:22-141 + global_cache.h:126-164
constexpr float CACHE_SIG = 0.15f, CACHE_PROB = 0.50f;  // global_cache.h:13-14

// global_cache_t::evalPdfCache: kernel density of the cache rows of technique (camDepth, lgtDepth) at pssQuery, toroidal distance
LMC_D float EvalPdfCache(const DCacheDim &D, int dim, const float *pssQuery, int camDepth, int lgtDepth) {
    float ret = 0.0f;
    for (int r = 0; r < PSS_MAX_SIZE; r++) {
        const float *x = D.extra + (size_t)r * CACHE_ROW_EXTRA + DPATH_WORDS;  // the row's Contrib: camDepth, lightDepth first
        if (__float_as_int(x[0]) != camDepth || __float_as_int(x[1]) != lgtDepth) continue;
        float sumDistSqr = 0.f;
        for (int j = 0; j < dim; j++) {
            const float c = D.pts[(size_t)r * dim + j], q = pssQuery[j];
            const float d1 = fabsf(q - c);
            const float d2 = 1.0f - d1;
            const float d = fminf(d1, d2);
            sumDistSqr += d * d;
        }
        const float expo = -0.5f * sumDistSqr * D.invSigmaSq;
        const float scale = float(double(D.factor * D.weight[r]) / D.scoreSum);
        ret += lexpf(expo) * scale;
    }
    return ret;
}

// global_cache_t::sampleCache (global_cache.h:126-137): data_distrib->SampleDiscrete(u) = clamp(upper_bound(cdf, u) - 1)
LMC_D int CacheSampleRow(const DCacheDim &D, float u) {
    int lo = 0, hi = PSS_MAX_SIZE + 1;  // std::upper_bound(cdf, cdf + count + 1, u)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (D.distCdf[mid] > u) hi = mid;
        else
            lo = mid + 1;
    }
    return Clampi(lo - 1, 0, PSS_MAX_SIZE - 1);
}

// proposes into `prop` / `pc` (and the sink, for the splats), sets the acceptance a
template <class Stk>
LMC_D void LargeStepCacheMutate(const DScene &S, const DCache &cache, const ChainArrays &A, const StepParams &P, int i, int flags, bool curValid, const Contrib &cur,
                                DPath &prop, Contrib &pc, ContribSink &sink, float &a, Rng &rng, Stk &stk) {
    const int proposalLength = LengthSampleDiscrete(P, rng.Uniform());
    const int proposalDim = proposalLength * 2;
    const int currentLength = cur.camDepth + cur.lightDepth - 1;
    const int currentDim = currentLength * 2;
    const bool proposalCacheAvailable = proposalDim >= PSS_MIN_LENGTH && proposalDim <= PSS_MAX_LENGTH && cache.d[proposalDim].ready;
    const bool currentCacheAvailable = currentDim >= PSS_MIN_LENGTH && currentDim <= PSS_MAX_LENGTH && cache.d[currentDim].ready;
    float propPss[MAXPSS];
    bool got = false;
    if (!proposalCacheAvailable || rng.Uniform() > CACHE_PROB) {  // uniform, as in multiplexed MLT
        const int lgtLength = Clampi(int(rng.Uniform() * float(proposalLength + 1)), 0, proposalLength);
        const int camLength = proposalLength - lgtLength + 1;
        GenerateSubpath(S, camLength, lgtLength, prop, sink, rng, stk);
        if (sink.count > 0) {
            got = true;
            pc = sink.Get(0);
            ToSubpath(pc.camDepth, pc.lightDepth, prop);
            GetPathPss(prop, propPss);
        }
    } else {  // a cached path, drawn by its weight, perturbed by N(0, CACHE_SIG) in every primary sample
        const DCacheDim &D = cache.d[proposalDim];
        const int idx = CacheSampleRow(D, rng.Uniform());
        const float *row = D.extra + (size_t)idx * CACHE_ROW_EXTRA;
        float *w = reinterpret_cast<float *>(&prop);
        for (int k = 0; k < DPATH_WORDS; k++) w[k] = row[k];
        ToSubpath(__float_as_int(row[DPATH_WORDS]), __float_as_int(row[DPATH_WORDS + 1]), prop);
        NormalDist nd(0.0f, CACHE_SIG);
        float offset[MAXPSS];
        for (int k = 0; k < MAXPSS; k++) offset[k] = 0.f;
        for (int k = 0; k < proposalDim; k++) {
            offset[k] = nd(rng);
            propPss[k] = Modulo1(D.pts[(size_t)idx * proposalDim + k] + offset[k]);
        }
        if (PerturbPathBidir(S, offset, prop, pc, rng, stk)) {
            got = true;
            sink.Push(pc);
        }
    }
    if (!got) {
        a = 0.0f;
        return;
    }
    a = 1.0f;
    if (curValid) {
        DPath curPath;
        LoadPath(CurPathBuf(A, flags), A.N, i, curPath);  // a subpath already: ToSubpath (mutation_large_cache.h:105) changes nothing
        float curPss[MAXPSS];
        GetPathPss(curPath, curPss);
        const float proposalJacobian = pc.ssScore / pc.lsScore, currentJacobian = cur.ssScore / cur.lsScore;
        const float proposalTechniquePickProb = inverse(float(proposalLength) + 1.0f), currentTechniquePickProb = inverse(float(currentLength) + 1.0f);
        const float proposalUniformPdf = 1.0f * proposalTechniquePickProb * proposalJacobian, currentUniformPdf = 1.0f * currentTechniquePickProb * currentJacobian;
        const float proposalCachePdf = proposalCacheAvailable ? EvalPdfCache(cache.d[proposalDim], proposalDim, propPss, pc.camDepth, pc.lightDepth) : 0.0f;
        const float currentCachePdf = currentCacheAvailable ? EvalPdfCache(cache.d[currentDim], currentDim, curPss, cur.camDepth, cur.lightDepth) : 0.0f;
        const float proposalPdf = !proposalCacheAvailable ? proposalUniformPdf : (1 - CACHE_PROB) * proposalUniformPdf + CACHE_PROB * proposalCachePdf;
        const float currentPdf = !currentCacheAvailable ? currentUniformPdf : (1 - CACHE_PROB) * currentUniformPdf + CACHE_PROB * currentCachePdf;
        a = Clampf(pc.ssScore * currentPdf * LengthPmf(P, currentLength) / (cur.ssScore * proposalPdf * LengthPmf(P, proposalLength)), 0.0f, 1.0f);
    }
}

template <bool WITH_LARGE, bool WITH_SMALL, bool WITH_GRAD, int MUX = 0, class Stk>
LMC_D void StepChain(const DScene &S, const DCache &cache, const ChainArrays &A, const Film &film, const StepParams &P, int i, int kind, Rng &rng,
                     GradWork &gw, StepStats &st, Stk &stk) {
    const size_t N = A.N;
    int flags = A.flags[i];
    const bool curValid = flags & F_VALID;
    const Contrib cur = LoadContrib(A.curContrib, A.N, i);
    DPath prop;
    Contrib pc;
    pc.camDepth = pc.lightDepth = 0;
    pc.lsScore = pc.ssScore = 0.f;
    float a = 1.0f;
    float propScoreSum = 0.f;
    bool lastMala = false;
    Gauss pg;
    ContribSink sink{A.contribList, N, (size_t)i, 0};
    st.steps++;

    if (WITH_LARGE && (kind == KIND_LARGE || !WITH_SMALL)) {  // LargeStep::Mutate, mutation_large.h:31-128; MUX = 1: largeStepMultiplexed, 2: LargeStepCache
        st.large++;
        if constexpr (MUX == 2) {
            LargeStepCacheMutate(S, cache, A, P, i, flags, curValid, cur, prop, pc, sink, a, rng, stk);
            propScoreSum = pc.lsScore;
        } else if constexpr (MUX == 1) {  // mutation_large.h:45-58: a length from lengthDist, a uniform split of it, one technique
            const int length = LengthSampleDiscrete(P, rng.Uniform());
            const int lgtLength = Clampi(int(rng.Uniform() * float(length + 1)), 0, length);
            const int camLength = length - lgtLength + 1;
            GenerateSubpath(S, camLength, lgtLength, prop, sink, rng, stk);
        } else {
            GeneratePathBidir(S, max(S.opt.minDepth, 3), S.opt.maxDepth, prop, sink, rng, stk);
        }
        if (MUX != 2 && sink.count > 0) {
            float scoreSum = 0.f;
            for (int k = 0; k < sink.count; k++) scoreSum += sink.LsScore(k);  // contribCdf.back()
            const float invSc = inverse(scoreSum);
            const float u = rng.Uniform();
            // std::upper_bound(cdf*invSc, u): first element > u; contribId = clamp(pos - 1, 0, n - 1)
            int pos = sink.count + 1;
            float cdf = 0.f;
            if (u < cdf * invSc) pos = 0;
            for (int k = 0; k < sink.count && pos > sink.count; k++) {
                cdf += sink.LsScore(k);
                if (u < cdf * invSc) pos = k + 1;
            }
            int contribId = Clampi(pos - 1, 0, sink.count - 1);
            pc = sink.Get(contribId);
            propScoreSum = scoreSum;
            if (curValid && MUX == 1) {  // mutation_large.h:87-102
                const int currentLength = cur.camDepth + cur.lightDepth - 1, proposalLength = pc.camDepth + pc.lightDepth - 1;
                const float invProposalTechniquesPmf = float(proposalLength) + 1.0f, invCurrentTechniquesPmf = float(currentLength) + 1.0f;
                a = Clampf((invProposalTechniquesPmf * pc.lsScore / LengthPmf(P, proposalLength)) / (invCurrentTechniquesPmf * cur.lsScore / LengthPmf(P, currentLength)),
                           0.0f, 1.0f);
            } else if (curValid) {
                const float probProposal = pc.lsScore / scoreSum;
                const float probLast = A.lastScore[i] / A.lastScoreSum[i];
                a = Clampf((pc.lsScore * probLast) / (cur.lsScore * probProposal), 0.0f, 1.0f);
            }
        } else if (MUX != 2) {
            a = 0.0f;
        }
    } else if (WITH_SMALL) {
        LoadPath(CurPathBuf(A, flags), A.N, i, prop);  // proposalState.path = currentState.path
        const int dim = PathDimension(prop.camDepth, prop.lgtDepth);
        float offset[MAXPSS];
        const bool mala = S.opt.mala && !(rng.Uniform() < S.opt.uniformMixingProbability);  // mutation_mala.h:46-51
        Gauss cg;
        if (!mala) {  // SmallStep::Mutate, mutation_small.h:16-56
            NormalDist nd(0.0f, S.opt.perturbStdDev);
            for (int k = 0; k < dim; k++) offset[k] = nd(rng);
        } else {
            lastMala = true;
            if (!(flags & F_BUFFERED)) {  // mutation_mala.h:59-81; the vectors are zero already (dchain.h ClearBuffered)
                flags |= F_BUFFERED;
                flags &= ~F_QUERIED;
            }
            flags |= F_VDIRTY;  // InitGaussianFor writes chain->pss (and the moment vectors) unconditionally
            if (!(flags & F_GAUSS)) {
                InitGaussianFor<WITH_GRAD>(S, cache, A, P, i, prop, cur, false, flags, cg, gw, st);
                StoreGauss(A, i, dim, flags, cg);
                flags = (flags | F_GAUSS) & ~F_GAUSS_ISO;
            } else {
                LoadGauss(S, A, i, dim, flags, cg);
            }
            NormalDist nd(0.0f, 1.0f);  // GenerateSample, gaussian.cpp:38-55
            for (int k = 0; k < dim; k++) offset[k] = nd(rng);
            for (int k = 0; k < dim; k++) offset[k] = cg.covL[k] * offset[k] + cg.mean[k];
        }
        if (PerturbPathBidir(S, offset, prop, pc, rng, stk)) {
            if (mala) {
                InitGaussianFor<WITH_GRAD>(S, cache, A, P, i, prop, pc, true, flags, pg, gw, st);
                float py = GaussianLogPdf(dim, offset, false, cg);
                float px = GaussianLogPdf(dim, offset, true, pg);
                a = Clampf(lexpf(px - py) * pc.ssScore / cur.ssScore, 0.0f, 1.0f);
            } else {
                a = Clampf(pc.ssScore / cur.ssScore, 0.0f, 1.0f);
            }
        } else {
            a = 0.0f;
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
    // small-step splat value: mutation_small.h:48 `contrib * (normalization / lsScore)` vs mutation_mala.h:271
    // `contrib * normalization / lsScore` (different rounding, kept)
    V3 smallSplat{0, 0, 0};
    if (WITH_SMALL && kind == KIND_SMALL) smallSplat = lastMala ? (pc.contrib * P.normalization) / pc.lsScore : pc.contrib * (P.normalization / pc.lsScore);
    const bool isLarge = WITH_LARGE && (kind == KIND_LARGE || !WITH_SMALL);
    if (a > 0.0f) {
        if (isLarge) {
            const float scale = P.normalization / propScoreSum;
            for (int k = 0; k < sink.count; k++) {
                Contrib c = sink.Get(k);
                Splat(film, c.screenPos, a * (c.contrib * scale));
            }
        } else {
            Splat(film, pc.screenPos, a * smallSplat);
        }
    }
    st.wsum += curValid ? 1.0f : (a > 0.0f ? a : 0.0f);

    // ---- accept / reject, mlt.cpp:113-170
    const int sampleIdx = A.sampleIdx[i];
    A.pushDim[i] = 0;
    if (a > 0.0f && rng.Uniform() <= a) {
        st.accepted++;
        const int oldDim = PathDimension(cur.camDepth, cur.lightDepth);  // GetDimension(proposalState.path) after the swap, mlt.cpp:121
        ToSubpath(pc.camDepth, pc.lightDepth, prop);
        StorePath(PropPathBuf(A, flags), A.N, i, prop);  // the proposal buffer becomes the current one
        flags ^= F_SEL;
        StoreContrib(A.curContrib, A.N, i, pc);
        A.adjacentReject[i] = 0;
        if (isLarge) {
            A.scoreSum[i] = propScoreSum;
            const float scale = P.normalization / propScoreSum;
            for (int k = 0; k < sink.count; k++) {
                Contrib c = sink.Get(k);
                float *p = A.curSplat + ((size_t)k * SPLAT_WORDS) * N + i;
                V3 v = c.contrib * scale;
                p[0] = c.screenPos.x, p[N] = c.screenPos.y, p[2 * N] = v.x, p[3 * N] = v.y, p[4 * N] = v.z;
            }
            A.curSplatCount[i] = sink.count;
            // the old current state was valid iff the chain had run a MALA step since (chain.buffered)
            if ((flags & F_BUFFERED) && A.pathWeight[i] > 1e-10f) {
                if (oldDim >= PSS_MIN_LENGTH && oldDim <= PSS_MAX_LENGTH && !cache.d[oldDim].ready) {
                    A.pushDim[i] = oldDim;
                    for (int k = 0; k < oldDim; k++) {
                        A.pushData[(size_t)k * N + i] = A.chPss[(size_t)k * N + i];
                        A.pushData[(size_t)(MAXPSS + k) * N + i] = A.chV1[(size_t)k * N + i];
                        A.pushData[(size_t)(2 * MAXPSS + k) * N + i] = A.chV2[(size_t)k * N + i];
                    }
                    A.pushData[(size_t)(3 * MAXPSS) * N + i] = A.pathWeight[i];
                }
            }
            A.lastScoreSum[i] = propScoreSum;
            A.lastScore[i] = pc.lsScore;
            flags &= ~(F_GAUSS | F_GAUSS_ISO);
            ClearBuffered(A, i, flags);
        } else {
            float *p = A.curSplat + i;
            p[0] = pc.screenPos.x, p[N] = pc.screenPos.y, p[2 * N] = smallSplat.x, p[3 * N] = smallSplat.y, p[4 * N] = smallSplat.z;
            A.curSplatCount[i] = 1;
            if (lastMala) {  // mlt.cpp:133-142: chain.v1/v2 = prop_new_v1/v2 (whole vectors, whichever branch filled them last)
                for (int k = 0; k < MAXPSS; k++) {
                    A.chV1[(size_t)k * N + i] = A.chPropNewV1[(size_t)k * N + i];
                    A.chV2[(size_t)k * N + i] = A.chPropNewV2[(size_t)k * N + i];
                }
                flags = (flags | F_BUFFERED | F_GAUSS) & ~F_GAUSS_ISO;
                StoreGauss(A, i, PathDimension(pc.camDepth, pc.lightDepth), flags, pg);
            } else {
                flags &= ~(F_GAUSS | F_GAUSS_ISO);  // proposalState.gaussianInitialized = false, mutation_small.h:39
            }
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
    A.flags[i] = flags & ~F_VSYNC;  // this kernel does not track the v1 / v2 equality (dchain.h)
    A.sampleIdx[i] = sampleIdx + 1;
}

}  // namespace lmcd
