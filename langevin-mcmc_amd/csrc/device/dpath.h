// Bidirectional path generation (large step / MLT init) and perturbation (small step) on the device:
// /root/reference/src/path.cpp:529-1449 (helpers + GeneratePathBidir), :1660-1673, :1953-2160
// (PerturbPathBidir).  One thread = one chain; RNG consumption order is the reference's
// (SURVEY.md Appendix A).  lensContrib / lensScore are not computed: nothing on the LMC path reads them.
#pragma once
#include "drng.h"
#include "dshade.h"

namespace lmcd {

constexpr int MAXD = 12;       // largest supported <dpt maxdepth> (shipped scenes use 8, BASELINE configs[2] 12; PSS dims <= 24)
constexpr int MAXPSS = 2 * MAXD;
// contributions one GeneratePathBidir call can emit: one per (c,l) with 3 <= c+l-1 <= maxDepth, i.e. sum of (L + 1)
constexpr int MAXCONTRIB = 85;
LMC_HD int MaxContribs(int maxDepth) {
    int n = 0;
    for (int L = 3; L <= maxDepth; L++) n += L + 1;
    return n > 0 ? n : 1;
}

struct DVertex {  // SurfaceVertex, path.h:24-32
    int tri;
    float st0, st1;
    float rnd0, rnd1;  // bsdfRndParam
    float bsdfDiscrete, useAbs, rrWeight;
    int dirLight, dirPrim;  // directLightInst
    float dirRnd0, dirRnd1;
};
constexpr int DVERTEX_WORDS = 12;

struct DPath {  // Path, path.h:38-56
    float time;
    float screen0, screen1;
    float lgtPos0, lgtPos1, lgtDir0, lgtDir1;
    int lgtLight, lgtPrim;
    int envPrim;  // envLightInst.lPrimID (valid when lgtDepth == 0 and the last ray escaped)
    int camDepth, lgtDepth;
    int camCount, lgtCount;
    float lensPos0, lensPos1;  // padding to 16 words (lensVertexPos is only read by the unused lens path)
    DVertex cam[MAXD], lgt[MAXD];
};
constexpr int DPATH_HEAD_WORDS = 16;
constexpr int DPATH_WORDS = DPATH_HEAD_WORDS + 2 * MAXD * DVERTEX_WORDS;
static_assert(sizeof(DPath) == DPATH_WORDS * 4, "DPath must be a plain word array");

struct Contrib {  // SubpathContrib, path.h:12-21 (lensScore, misWeight dropped)
    int camDepth, lightDepth;
    V2 screenPos;
    V3 contrib;
    float lsScore, ssScore;
};

// where GeneratePathBidir appends its contributions: SoA in HBM, stride = number of slots
struct ContribSink {
    float *base;    // 9 words per entry: [k*9 + w] * stride + slot
    size_t stride;  // number of slots
    size_t slot;
    int count;
    LMC_D void Push(const Contrib &c) {
        if (count >= MAXCONTRIB) return;
        float *p = base + (size_t)count * 9 * stride + slot;
        p[0 * stride] = __int_as_float(c.camDepth);
        p[1 * stride] = __int_as_float(c.lightDepth);
        p[2 * stride] = c.screenPos.x;
        p[3 * stride] = c.screenPos.y;
        p[4 * stride] = c.contrib.x;
        p[5 * stride] = c.contrib.y;
        p[6 * stride] = c.contrib.z;
        p[7 * stride] = c.lsScore;
        p[8 * stride] = c.ssScore;
        count++;
    }
    LMC_D Contrib Get(int k) const {
        const float *p = base + (size_t)k * 9 * stride + slot;
        Contrib c;
        c.camDepth = __float_as_int(p[0 * stride]);
        c.lightDepth = __float_as_int(p[1 * stride]);
        c.screenPos = V2{p[2 * stride], p[3 * stride]};
        c.contrib = V3{p[4 * stride], p[5 * stride], p[6 * stride]};
        c.lsScore = p[7 * stride];
        c.ssScore = p[8 * stride];
        return c;
    }
    LMC_D float LsScore(int k) const { return base[((size_t)k * 9 + 7) * stride + slot]; }
};

struct BPS {  // BidirPathState, path.cpp:529-540
    Isect isect;
    V3 wi;
    float accMISWPrev, accMISWThis;
    V3 throughput;
    float ssJacobian;
};

LMC_D float MIS(float pdf) { return square(pdf); }

template <bool adjoint>
LMC_D float ShadingNormalCorrection(V3 wi, const Isect &isect, V3 wo) {  // path.cpp:34-54
    const float cosWi = Dot(isect.shadingNormal, wi);
    const float cosWo = Dot(isect.shadingNormal, wo);
    float wiDotGeoN = Dot(isect.geomNormal, wi);
    float woDotGeoN = Dot(isect.geomNormal, wo);
    if (wiDotGeoN * cosWi <= 0.0f || woDotGeoN * cosWo <= 0.0f) return 0.0f;
    if (adjoint) return fabsf((woDotGeoN * cosWi) / (wiDotGeoN * cosWo));
    return 1.0f;
}

// Vector2(uniDist(rng), uniDist(rng)): gcc evaluates the arguments right to left (see oracle/path.cpp)
template <class R>
LMC_D V2 RndVec2(R &rng) {
    float first, second;
    rng.Uniform2(first, second);
    return V2{second, first};
}

LMC_D void EmitFromCamera(const DScene &S, V2 screenPos, V3 &org, V3 &dir, BPS &ps) {  // path.cpp:554-574
    V3 cOrg, camDir;
    SamplePrimary(S, V2{0.5f, 0.5f}, cOrg, camDir);
    SamplePrimary(S, screenPos, org, dir);
    const float cosAtCamera = Dot(camDir, dir);
    const float imagePointToCameraDist = S.cam.dist / cosAtCamera;
    const float imageToSolidAngleFactor = square(imagePointToCameraDist) / cosAtCamera;
    const float screenPixelCount = float(S.cam.width * S.cam.height);
    ps.throughput = V3{1, 1, 1};
    ps.accMISWPrev = MIS(screenPixelCount / imageToSolidAngleFactor);
    ps.accMISWThis = 0.0f;
    ps.ssJacobian = 1.0f;
}

LMC_D void EmitFromLight(const DScene &S, float lightPickProb, DPath &path, V3 &org, V3 &dir, BPS &ps) {  // path.cpp:588-618
    float cosLight, emissionPdf, directPdf;
    LightEmit(S, path.lgtLight, V2{path.lgtPos0, path.lgtPos1}, V2{path.lgtDir0, path.lgtDir1}, path.lgtPrim, org, dir, ps.throughput, cosLight,
              emissionPdf, directPdf);
    emissionPdf *= lightPickProb;
    directPdf *= lightPickProb;
    ps.throughput = ps.throughput * inverse(lightPickProb);
    ps.accMISWPrev = MIS(directPdf / emissionPdf);
    ps.accMISWThis = LightIsDelta(S, path.lgtLight) ? 0.0f : MIS(cosLight / emissionPdf);
    ps.ssJacobian = 1.0f;
}

// light < 0 means nullptr (camera subpath)
LMC_D void ConvertMIS(const DScene &S, int depth, int light, V3 rayOrg, V3 rayDir, BPS &ps) {  // path.cpp:620-631
    if (depth > 0 || light < 0 || LightIsFinite(S, light)) ps.accMISWPrev *= MIS(DistanceSquared(rayOrg, ps.isect.position));
    float invCosTheta = inverse(MIS(fabsf(Dot(rayDir, ps.isect.shadingNormal))));
    ps.accMISWPrev *= invCosTheta;
    ps.accMISWThis *= invCosTheta;
}

// How the three connection strategies below test visibility.  TraceOcclusion casts the shadow ray on the spot (the
// reference's order).  DeferOcclusion only records it and answers "visible": the caller (the lean small-step kernel,
// dsmall.h) then casts it from ONE place after the strategy has been evaluated and drops the contribution if the ray is
// blocked -- the strategies draw no random numbers and have no other side effects, so the result is the same, and the
// kernel carries one copy of the any-hit traversal instead of three.
struct TraceOcclusion {
    template <class Stk>
    LMC_D bool Test(const DScene &S, V3 org, V3 dir, float dist, Stk &stk) {
        return Occluded(S, org, dir, dist, stk);
    }
};
struct DeferOcclusion {
    V3 org, dir;
    float dist;
    bool pending = false;
    template <class Stk>
    LMC_D bool Test(const DScene &, V3 o, V3 d, float t, Stk &) {
        org = o, dir = d, dist = t, pending = true;
        return false;
    }
};

// path.cpp:633-745; returns true and fills `out` when a contribution is produced
template <class Stk, class Occ>
LMC_D bool ConnectToCamera(const DScene &S, int lgtDepth, const BPS &ps, const DVertex &lgtVertex, Contrib &out, Stk &stk, Occ &occ) {
    V3 camOrg, camDir;
    SamplePrimary(S, V2{0.5f, 0.5f}, camOrg, camDir);
    V3 dirToCamera = camOrg - ps.isect.position;
    if (-Dot(camDir, dirToCamera) <= 0.0f) return false;
    V2 screenPos;
    if (!ProjectPoint(S, ps.isect.position, screenPos)) return false;
    const float distSq = LengthSquared(dirToCamera);
    const float dist = sqrtf(distSq);
    dirToCamera = dirToCamera * inverse(dist);
    if (occ.Test(S, ps.isect.position, dirToCamera, dist, stk)) return false;
    const DMaterial m = LoadMaterial<Stk::kGlossy>(S, lgtVertex.tri);
    V2 st{lgtVertex.st0, lgtVertex.st1};
    V3 bsdfContrib;
    float cosToCamera, bsdfPdf, bsdfRevPdf;
    BsdfEvaluate<Stk::kGlossy>(S, m, true, ps.wi, ps.isect.shadingNormal, dirToCamera, st, bsdfContrib, cosToCamera, bsdfPdf, bsdfRevPdf);
    if (IsZero(bsdfContrib)) return false;
    const float factor = ShadingNormalCorrection<true>(ps.wi, ps.isect, dirToCamera);
    if (factor <= 0.0f) return false;
    bsdfContrib = bsdfContrib * factor;
    const float cosAtCamera = -Dot(camDir, dirToCamera);
    const float imagePointToCameraDist = S.cam.dist / cosAtCamera;
    const float imageToSolidAngleFactor = square(imagePointToCameraDist) / cosAtCamera;
    const float imageToSurfaceFactor = imageToSolidAngleFactor * fabsf(cosToCamera) / distSq;
    const float screenPixelCount = float(S.cam.width * S.cam.height);
    const float wLight = MIS(imageToSurfaceFactor / screenPixelCount) * (ps.accMISWPrev + ps.accMISWThis * MIS(bsdfRevPdf));
    const float misWeight = inverse(wLight + 1.0f);
    const float surfaceToImageFactor = cosToCamera / imageToSurfaceFactor;
    V3 contrib = misWeight * bsdfContrib / (screenPixelCount * surfaceToImageFactor);
    contrib = cmul(contrib, ps.throughput);
    const float score = Luminance(contrib);
    if (score > 0.0f) {
        out = Contrib{1, 2 + lgtDepth, screenPos, contrib, score, score * ps.ssJacobian};
        return true;
    }
    return false;
}

// path.cpp:747-900.  `in` and `out` may alias (the reference passes the same object in the camera loop);
// every field of `in` is read before the aliased field of `out` is written.
// lcJacobian (path.cpp:793-797,825; read by the light-coordinate branch of GeneratePathBidir only): handed out through a pointer so
// that BPS -- live across every traversal of the hot kernel -- does not carry it
// (the M form takes the vertex's material record: a caller that has just reconstructed the hit knows the material index -- SurfHit::material --
// and can have the record on its way before the arithmetic between the hit and the BSDF)
template <bool adjoint, bool perturb, bool GLOSSY>
LMC_D bool BSDFSamplingM(const DScene &S, const DMaterial &m, const BPS &in, DVertex &v, BPS &out, V3 &dir, V3 &bsdfContrib, float *lcJacobian = nullptr) {
    V2 st{v.st0, v.st1};
    float cosWo, bsdfPdf, bsdfRevPdf;
    v.useAbs = (BsdfRoughness<GLOSSY>(S, m, st, v.bsdfDiscrete) > S.opt.roughnessThreshold) ? 1.0f : 0.0f;
    const V3 wi = in.wi;
    const float inSsJac = in.ssJacobian, inAccThis = in.accMISWThis, inAccPrev = in.accMISWPrev;
    const V3 inThr = in.throughput;
    if (!perturb || v.useAbs == 0.0f) {
        if (!BsdfSample<GLOSSY>(S, m, adjoint, wi, in.isect.shadingNormal, st, V2{v.rnd0, v.rnd1}, v.bsdfDiscrete, dir, bsdfContrib, cosWo, bsdfPdf, bsdfRevPdf))
            return false;
        if (v.useAbs == 1.0f) {
            float jacobian;
            V2 sc = ToSphericalCoord(dir, jacobian);
            v.rnd0 = sc.x, v.rnd1 = sc.y;
            if (lcJacobian) *lcJacobian = inverse(jacobian);
            jacobian *= bsdfPdf;
            out.ssJacobian = inSsJac * jacobian;
        } else if (lcJacobian) {
            *lcJacobian = bsdfPdf;
        }
    } else {
        float jacobian;
        dir = SampleSphere(V2{v.rnd0, v.rnd1}, jacobian);
        if (lcJacobian) *lcJacobian = inverse(jacobian);
        BsdfEvaluate<GLOSSY>(S, m, adjoint, wi, in.isect.shadingNormal, dir, st, bsdfContrib, cosWo, bsdfPdf, bsdfRevPdf);
        if (IsZero(bsdfContrib) || bsdfPdf <= 0.0f) return false;
        bsdfContrib = bsdfContrib * inverse(bsdfPdf);
        jacobian *= bsdfPdf;
        out.ssJacobian = inSsJac * jacobian;
    }
    float factor = ShadingNormalCorrection<adjoint>(wi, in.isect, dir);
    if (factor <= 0.0f) return false;
    bsdfContrib = bsdfContrib * factor;
    out.accMISWThis = MIS(cosWo / bsdfPdf) * (inAccThis * MIS(bsdfRevPdf) + inAccPrev);
    out.accMISWPrev = MIS(inverse(bsdfPdf));
    out.throughput = cmul(inThr, bsdfContrib);
    return true;
}
template <bool adjoint, bool perturb, bool GLOSSY>
LMC_D bool BSDFSampling(const DScene &S, const BPS &in, DVertex &v, BPS &out, V3 &dir, V3 &bsdfContrib, float *lcJacobian = nullptr) {
    return BSDFSamplingM<adjoint, perturb, GLOSSY>(S, LoadMaterial<GLOSSY>(S, v.tri), in, v, out, dir, bsdfContrib, lcJacobian);
}

// path.cpp:902-967 (bidirMIS = true)
LMC_D bool HandleHitLight(const DScene &S, int camDepth, int light, bool hitSurface, V3 rayDir, V2 screenPos, const BPS &ps, int &envPrim,
                          Contrib &out) {
    int lPrimID = 0;
    V3 emission;
    float directPdf, emissionPdf;
    LightEmission(S, light, rayDir, ps.isect.shadingNormal, lPrimID, emission, directPdf, emissionPdf);
    if (emission.x + emission.y + emission.z > 0.0f) {
        V3 contrib = cmul(ps.throughput, emission);
        if (camDepth > 0) {
            float lightPickProb = PickLightProb(S, light);
            directPdf *= lightPickProb;
            emissionPdf *= lightPickProb;
            float wCamera = MIS(directPdf) * ps.accMISWPrev + MIS(emissionPdf) * ps.accMISWThis;
            float misWeight = inverse(1.0f + wCamera);
            contrib = contrib * misWeight;
        }
        float score = Luminance(contrib);
        if (score > 0.0f) {
            if (!hitSurface) envPrim = lPrimID;
            out = Contrib{2 + camDepth, 0, screenPos, contrib, score, score * ps.ssJacobian};
            return true;
        }
    }
    return false;
}

// path.cpp:969-1089 (doOcclusion = true, bidirMIS = true)
template <class Stk, class Occ>
LMC_D bool DirectLightingM(const DScene &S, const DMaterial &m, int camDepth, const BPS &ps, V2 screenPos, float lightPickProb, DVertex &camVertex, Contrib &out, Stk &stk, Occ &occ) {  // m: camVertex's material (BSDFSamplingM)
    const int light = camVertex.dirLight;
    V3 dirToLight, lightContrib;
    float dist, cosAtLight, directPdf, emissionPdf;
    if (!LightSampleDirect(S, light, ps.isect.position, V2{camVertex.dirRnd0, camVertex.dirRnd1}, camVertex.dirPrim, dirToLight, dist, lightContrib,
                           cosAtLight, directPdf, emissionPdf))
        return false;
    if (occ.Test(S, ps.isect.position, dirToLight, dist, stk)) return false;
    V3 bsdfContrib;
    float cosToLight, bsdfPdf, bsdfRevPdf;
    BsdfEvaluate<Stk::kGlossy>(S, m, false, ps.wi, ps.isect.shadingNormal, dirToLight, V2{camVertex.st0, camVertex.st1}, bsdfContrib, cosToLight, bsdfPdf, bsdfRevPdf);
    if (IsZero(bsdfContrib)) return false;
    const float factor = ShadingNormalCorrection<false>(ps.wi, ps.isect, dirToLight);
    if (factor <= 0.0f) return false;
    bsdfContrib = bsdfContrib * factor;
    V3 contrib = cmul(ps.throughput, bsdfContrib);
    contrib = cmul(contrib, lightContrib) * inverse(lightPickProb);
    float wLight = LightIsDelta(S, light) ? 0.0f : MIS(bsdfPdf / (lightPickProb * directPdf));
    float wCamera = MIS(emissionPdf * cosToLight / (directPdf * cosAtLight)) * (ps.accMISWPrev + ps.accMISWThis * MIS(bsdfRevPdf));
    float misWeight = inverse(wLight + 1.0f + wCamera);
    contrib = contrib * misWeight;
    const float score = Luminance(contrib);
    if (score > 0.0f) {
        out = Contrib{2 + camDepth, 1, screenPos, contrib, score, score * ps.ssJacobian};
        return true;
    }
    return false;
}
template <class Stk, class Occ>
LMC_D bool DirectLighting(const DScene &S, int camDepth, const BPS &ps, V2 screenPos, float lightPickProb, DVertex &camVertex, Contrib &out, Stk &stk, Occ &occ) {
    return DirectLightingM(S, LoadMaterial<Stk::kGlossy>(S, camVertex.tri), camDepth, ps, screenPos, lightPickProb, camVertex, out, stk, occ);
}

// the generators below shade a vertex right behind its hit reconstruction: the material by the index that hit carried (dshade.h LMC_MAT_CARRY)
#if LMC_MAT_CARRY
#define MAT_BSDF(adj, pert) BSDFSamplingM<adj, pert, Stk::kGlossy>
#define MAT_ARG LoadMaterialIdx<Stk::kGlossy>(S, hit.material),
#define MAT_DIRECT(S, d, ps, sp, pp, sv, c, stk, tr) DirectLightingM(S, LoadMaterialIdx<Stk::kGlossy>(S, hit.material), d, ps, sp, pp, sv, c, stk, tr)
#else
#define MAT_BSDF(adj, pert) BSDFSampling<adj, pert, Stk::kGlossy>
#define MAT_ARG
#define MAT_DIRECT DirectLighting
#endif
// path.cpp:1091-1235 (doOcclusion = true)
template <class Stk, class Occ>
LMC_D bool ConnectVertex(const DScene &S, int camDepth, int lgtDepth, const BPS &lps, const DVertex &lgtVertex, const BPS &cps,
                         const DVertex &camVertex, V2 screenPos, Contrib &out, Stk &stk, Occ &occ) {
    V3 dirToLight = lps.isect.position - cps.isect.position;
    const float distSq = LengthSquared(dirToLight);
    const float dist = sqrtf(distSq);
    dirToLight = dirToLight * inverse(dist);
    if (occ.Test(S, cps.isect.position, dirToLight, dist, stk)) return false;
    V3 camBsdfFactor;
    float cosCamera, camBsdfPdf, camBsdfRevPdf;
    BsdfEvaluate<Stk::kGlossy>(S, LoadMaterial<Stk::kGlossy>(S, camVertex.tri), false, cps.wi, cps.isect.shadingNormal, dirToLight, V2{camVertex.st0, camVertex.st1}, camBsdfFactor,
                 cosCamera, camBsdfPdf, camBsdfRevPdf);
    if (IsZero(camBsdfFactor)) return false;
    float camFactor = ShadingNormalCorrection<false>(cps.wi, cps.isect, dirToLight);
    if (camFactor <= 0.0f) return false;
    camBsdfFactor = camBsdfFactor * camFactor;
    V3 lgtBsdfFactor;
    float cosLight, lgtBsdfPdf, lgtBsdfRevPdf;
    BsdfEvaluate<Stk::kGlossy>(S, LoadMaterial<Stk::kGlossy>(S, lgtVertex.tri), true, lps.wi, lps.isect.shadingNormal, -dirToLight, V2{lgtVertex.st0, lgtVertex.st1}, lgtBsdfFactor,
                 cosLight, lgtBsdfPdf, lgtBsdfRevPdf);
    if (IsZero(lgtBsdfFactor)) return false;
    float lgtFactor = ShadingNormalCorrection<true>(lps.wi, lps.isect, -dirToLight);
    if (lgtFactor <= 0.0f) return false;
    lgtBsdfFactor = lgtBsdfFactor * lgtFactor;
    const float geometryTerm = inverse(distSq);
    const float camBsdfDirPdfA = camBsdfPdf * cosLight * geometryTerm;
    const float lgtBsdfDirPdfA = lgtBsdfPdf * cosCamera * geometryTerm;
    const float wLight = MIS(camBsdfDirPdfA) * (lps.accMISWPrev + lps.accMISWThis * MIS(lgtBsdfRevPdf));
    const float wCamera = MIS(lgtBsdfDirPdfA) * (cps.accMISWPrev + cps.accMISWThis * MIS(camBsdfRevPdf));
    const float misWeight = inverse(wLight + 1.0f + wCamera);
    const V3 throughput = cmul(lps.throughput, cps.throughput);
    V3 contrib = cmul(throughput, camBsdfFactor);
    contrib = cmul(contrib, lgtBsdfFactor) * geometryTerm;
    contrib = contrib * misWeight;
    const float ssJacobian = lps.ssJacobian * cps.ssJacobian;
    const float score = Luminance(contrib);
    if (score > 0.0f) {
        out = Contrib{2 + camDepth, 2 + lgtDepth, screenPos, contrib, score, score * ssJacobian};
        return true;
    }
    return false;
}

template <class R>
LMC_D bool RussianRoulette(int depth, V3 bsdfContrib, float &rrWeight, V3 &throughput, R &rng) {  // path.cpp:388-404
    float rrProb = 1.0f;
    if (depth >= 3) rrProb = fminf(MaxCoeff(bsdfContrib), 0.95f);
    if (rng.Uniform() > rrProb) return false;
    rrWeight = inverse(rrProb);
    throughput = throughput * rrWeight;
    return true;
}

LMC_D int HitLightOf(const DScene &S, bool hitSurface, int tri) {  // GetHitLight, path.cpp:105-120; -1 = none
    if (!hitSurface) return S.envLight;
    return S.tris[tri].areaLight;
}
LMC_D int HitLightOf(const DScene &S, bool hitSurface, const SurfHit &hit) {  // ... of a hit just reconstructed
#if LMC_MAT_CARRY
    return hitSurface ? hit.areaLight : S.envLight;
#else
    return HitLightOf(S, hitSurface, hit.tri);
#endif
}

// GeneratePathBidir, path.cpp:1237-1449 with screenPosi = (-1,-1)
template <class Stk>
LMC_D void GeneratePathBidir(const DScene &S, int minDepth, int maxDepth, DPath &path, ContribSink &sink, Rng &rng, Stk &stk) {
    TraceOcclusion trace;
    path.camCount = path.lgtCount = 0;
    path.envPrim = -1;
    path.time = rng.Uniform();
    BPS lightStates[MAXD];
    int numLightStates = 1;
    float lightPickProb = 1.0f;
    {  // EmitFromLightInit, path.cpp:576-586
        V2 p = RndVec2(rng), d = RndVec2(rng);
        path.lgtPos0 = p.x, path.lgtPos1 = p.y, path.lgtDir0 = d.x, path.lgtDir1 = d.y;
        path.lgtLight = PickLight(S, rng.Uniform(), lightPickProb);
        path.lgtPrim = LightSampleDiscrete(S, path.lgtLight, rng.Uniform());
    }
    V3 org, dir;
    EmitFromLight(S, lightPickProb, path, org, dir, lightStates[0]);
    for (int lgtDepth = 0;; lgtDepth++) {
        DVertex &sv = path.lgt[lgtDepth];
        SurfHit hit;
        bool hitSurface = IntersectSurface(S, org, dir, c_IsectEpsilon, INFINITY, hit, lightStates[lgtDepth].isect, stk);
        if (!hitSurface) {
            numLightStates--;
            break;
        }
        path.lgtCount = lgtDepth + 1;
        sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
        sv.bsdfDiscrete = rng.Uniform();
        lightStates[lgtDepth].wi = -dir;
        ConvertMIS(S, lgtDepth, path.lgtLight, org, dir, lightStates[lgtDepth]);
        if (lgtDepth + 2 >= minDepth) {
            Contrib c;
            if (ConnectToCamera(S, lgtDepth, lightStates[lgtDepth], sv, c, stk, trace)) sink.Push(c);
        }
        if (maxDepth != -1 && lgtDepth + 2 >= maxDepth) break;
        if (lgtDepth + 1 >= MAXD) break;  // storage bound (never reached for maxDepth <= MAXD)
        numLightStates++;
        V2 r = RndVec2(rng);
        sv.rnd0 = r.x, sv.rnd1 = r.y;
        V3 bsdfContrib;
        if (!BSDFSampling<true, false, Stk::kGlossy>(S, lightStates[lgtDepth], sv, lightStates[lgtDepth + 1], dir, bsdfContrib)) {
            numLightStates--;
            break;
        }
        // a fresh BidirPathState() is value-initialised: ssJacobian stays 0 unless BSDFSampling set it (non-absolute vertices)
        if (sv.useAbs == 0.0f) lightStates[lgtDepth + 1].ssJacobian = 0.0f;
        if (!RussianRoulette(lgtDepth, bsdfContrib, sv.rrWeight, lightStates[lgtDepth + 1].throughput, rng)) {
            numLightStates--;
            break;
        }
        org = lightStates[lgtDepth].isect.position;
    }

    BPS cps;
    {  // EmitFromCameraInit with screenPosi = (-1,-1): Vector2(u, u), right-to-left
        V2 s = RndVec2(rng);
        path.screen0 = s.x, path.screen1 = s.y;
    }
    V2 screenPos{path.screen0, path.screen1};
    EmitFromCamera(S, screenPos, org, dir, cps);
    float tnear, tfar;
    tnear = PrimaryMinT(S, screenPos, tfar);
    float lcJac = 0.0f;  // camPathState.lcJacobian of the last BSDF sampling
    for (int camDepth = 0;; camDepth++) {
        if (camDepth >= MAXD) break;
        DVertex &sv = path.cam[camDepth];
        path.camCount = camDepth + 1;
        SurfHit hit;
        hit.tri = -1;
        bool hitSurface = IntersectSurface(S, org, dir, tnear, tfar, hit, cps.isect, stk);
        sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
        cps.wi = -dir;
        if (hitSurface) ConvertMIS(S, camDepth, -1, org, dir, cps);
        if (camDepth + 1 >= minDepth) {
            int light = HitLightOf(S, hitSurface, hit);
            if (light >= 0) {
                if (S.opt.useLightCoord && camDepth > 1 && S.lights[light].type == LIGHT_AREA) {  // path.cpp:1339-1360
                    // area light: the BSDF sampling coordinates of the previous vertex become the light's direct sampling coordinates
                    DVertex &prev = path.cam[camDepth - 1];
                    const V2 sp = TriangleSampleParam(S, hit.tri, cps.isect.position);
                    prev.rnd0 = sp.x, prev.rnd1 = sp.y;
                    V3 dirToPrev = cps.isect.position - org;
                    const float distSq = LengthSquared(dirToPrev);
                    const float invDistSq = inverse(distSq);
                    const float invDist = sqrtf(invDistSq);
                    dirToPrev = dirToPrev * invDist;
                    cps.ssJacobian *= fabsf(Dot(dirToPrev, cps.isect.shadingNormal) * invDistSq) * (lcJac * S.meshes[S.tris[hit.tri].mesh].invTotalArea);
                }
                Contrib c;
                if (HandleHitLight(S, camDepth, light, hitSurface, dir, screenPos, cps, path.envPrim, c)) sink.Push(c);
                return;
            }
        }
        if (!hitSurface || (maxDepth != -1 && camDepth + 1 >= maxDepth)) break;
        sv.bsdfDiscrete = rng.Uniform();
        if (camDepth + 2 >= minDepth) {
            float directLightPickProb = 1.0f;
            sv.dirLight = PickLight(S, rng.Uniform(), directLightPickProb);  // DirectLightingInit, path.cpp:184-193
            V2 r = RndVec2(rng);
            sv.dirRnd0 = r.x, sv.dirRnd1 = r.y;
            sv.dirPrim = LightSampleDiscrete(S, sv.dirLight, rng.Uniform());
            Contrib c;
            if (MAT_DIRECT(S, camDepth, cps, screenPos, directLightPickProb, sv, c, stk, trace)) sink.Push(c);
        }
        int maxLgtDepth = maxDepth == -1 ? (numLightStates - 1) : min(maxDepth - camDepth - 3, numLightStates - 1);
        for (int lgtDepth = 0; lgtDepth <= maxLgtDepth; lgtDepth++) {
            if (camDepth + lgtDepth + 3 >= minDepth) {
                Contrib c;
                if (ConnectVertex(S, camDepth, lgtDepth, lightStates[lgtDepth], path.lgt[lgtDepth], cps, sv, screenPos, c, stk, trace)) sink.Push(c);
            }
        }
        V2 r = RndVec2(rng);
        sv.rnd0 = r.x, sv.rnd1 = r.y;
        V3 bsdfContrib;
        if (!MAT_BSDF(false, false)(S, MAT_ARG cps, sv, cps, dir, bsdfContrib, &lcJac)) break;
        if (!RussianRoulette(camDepth, bsdfContrib, sv.rrWeight, cps.throughput, rng)) break;
        org = cps.isect.position;
        tnear = c_IsectEpsilon;
        tfar = INFINITY;
    }
}

// GenerateSubpath, path.cpp:1451-1658 (screenPosi = (-1,-1), bidirMIS = true): ONE technique -- camLength camera vertices, lgtLength
// light vertices, the counts include the camera / the light itself -- and no Russian roulette (rrWeight = 1).  The generator of the
// multiplexed large step (mutation_large.h:45-57).  Differences from GeneratePathBidir that matter: the light subpath is only
// started for lgtLength > 1 (no random numbers drawn otherwise), ONE light state is advanced in place (its ssJacobian is carried,
// not reset, across non-absolute vertices), and the area-light re-parameterisation of the last bounce (path.cpp:1549-1571) is
// unconditional: this function never reads options->useLightCoordinateSampling.
template <class Stk>
LMC_D void GenerateSubpath(const DScene &S, int camLength, int lgtLength, DPath &path, ContribSink &sink, Rng &rng, Stk &stk) {
    TraceOcclusion trace;
    path.camCount = path.lgtCount = 0;
    path.envPrim = -1;
    path.time = rng.Uniform();
    BPS lps;
    V3 org, dir;
    if (lgtLength > 1) {
        float lightPickProb = 1.0f;
        {  // EmitFromLightInit, path.cpp:576-586
            V2 p = RndVec2(rng), d = RndVec2(rng);
            path.lgtPos0 = p.x, path.lgtPos1 = p.y, path.lgtDir0 = d.x, path.lgtDir1 = d.y;
            path.lgtLight = PickLight(S, rng.Uniform(), lightPickProb);
            path.lgtPrim = LightSampleDiscrete(S, path.lgtLight, rng.Uniform());
        }
        EmitFromLight(S, lightPickProb, path, org, dir, lps);
        for (int lgtDepth = 0;; lgtDepth++) {
            if (lgtDepth >= MAXD) return;  // storage bound (never reached for maxDepth <= MAXD)
            DVertex &sv = path.lgt[lgtDepth];
            SurfHit hit;
            if (!IntersectSurface(S, org, dir, c_IsectEpsilon, INFINITY, hit, lps.isect, stk)) return;
            path.lgtCount = lgtDepth + 1;
            sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
            sv.bsdfDiscrete = rng.Uniform();
            lps.wi = -dir;
            ConvertMIS(S, lgtDepth, path.lgtLight, org, dir, lps);
            if (lgtDepth + 2 == lgtLength) {
                if (camLength == 1) {
                    Contrib c;
                    if (ConnectToCamera(S, lgtDepth, lps, sv, c, stk, trace)) sink.Push(c);
                    return;
                }
                break;
            }
            V2 r = RndVec2(rng);
            sv.rnd0 = r.x, sv.rnd1 = r.y;
            V3 bsdfContrib;
            if (!MAT_BSDF(true, false)(S, MAT_ARG lps, sv, lps, dir, bsdfContrib)) return;
            sv.rrWeight = 1.0f;
            org = lps.isect.position;
        }
    }
    BPS cps;
    {  // EmitFromCameraInit with screenPosi = (-1,-1)
        V2 s = RndVec2(rng);
        path.screen0 = s.x, path.screen1 = s.y;
    }
    const V2 screenPos{path.screen0, path.screen1};
    EmitFromCamera(S, screenPos, org, dir, cps);
    float tnear, tfar;
    tnear = PrimaryMinT(S, screenPos, tfar);
    float lcJac = 0.0f;
    for (int camDepth = 0;; camDepth++) {
        if (camDepth >= MAXD) return;
        DVertex &sv = path.cam[camDepth];
        path.camCount = camDepth + 1;
        SurfHit hit;
        hit.tri = -1;
        const bool hitSurface = IntersectSurface(S, org, dir, tnear, tfar, hit, cps.isect, stk);
        sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
        cps.wi = -dir;
        if (hitSurface) ConvertMIS(S, camDepth, -1, org, dir, cps);
        if (camDepth + 2 >= camLength && lgtLength == 0) {
            const int light = HitLightOf(S, hitSurface, hit);
            if (light >= 0) {
                if (camDepth > 1 && S.lights[light].type == LIGHT_AREA) {
                    DVertex &prev = path.cam[camDepth - 1];
                    const V2 sp = TriangleSampleParam(S, hit.tri, cps.isect.position);
                    prev.rnd0 = sp.x, prev.rnd1 = sp.y;
                    V3 dirToPrev = cps.isect.position - org;
                    const float distSq = LengthSquared(dirToPrev);
                    const float invDistSq = inverse(distSq);
                    const float invDist = sqrtf(invDistSq);
                    dirToPrev = dirToPrev * invDist;
                    cps.ssJacobian *= fabsf(Dot(dirToPrev, cps.isect.shadingNormal) * invDistSq) * (lcJac * S.meshes[S.tris[hit.tri].mesh].invTotalArea);
                }
                Contrib c;
                if (HandleHitLight(S, camDepth, light, hitSurface, dir, screenPos, cps, path.envPrim, c)) sink.Push(c);
            }
            return;
        }
        if (!hitSurface) return;
        sv.bsdfDiscrete = rng.Uniform();
        if (camDepth + 2 == camLength) {
            Contrib c;
            if (lgtLength == 1) {
                float directLightPickProb = 1.0f;
                sv.dirLight = PickLight(S, rng.Uniform(), directLightPickProb);  // DirectLightingInit, path.cpp:184-193
                V2 r = RndVec2(rng);
                sv.dirRnd0 = r.x, sv.dirRnd1 = r.y;
                sv.dirPrim = LightSampleDiscrete(S, sv.dirLight, rng.Uniform());
                if (MAT_DIRECT(S, camDepth, cps, screenPos, directLightPickProb, sv, c, stk, trace)) sink.Push(c);
            } else {
                if (ConnectVertex(S, camDepth, lgtLength - 2, lps, path.lgt[lgtLength - 2], cps, sv, screenPos, c, stk, trace)) sink.Push(c);
            }
            return;
        }
        V2 r = RndVec2(rng);
        sv.rnd0 = r.x, sv.rnd1 = r.y;
        V3 bsdfContrib;
        if (!MAT_BSDF(false, false)(S, MAT_ARG cps, sv, cps, dir, bsdfContrib, &lcJac)) return;
        sv.rrWeight = 1.0f;
        org = cps.isect.position;
        tnear = c_IsectEpsilon;
        tfar = INFINITY;
    }
}

LMC_D void ToSubpath(int camDepth, int lgtDepth, DPath &path) {  // path.cpp:1660-1669
    path.camCount = max(camDepth - 1, 0);
    path.lgtCount = max(lgtDepth - 1, 0);
    if (lgtDepth != 0) path.envPrim = -1;
    path.camDepth = camDepth;
    path.lgtDepth = lgtDepth;
}

LMC_D int PathDimension(int camDepth, int lgtDepth) { return max(camDepth + lgtDepth - 1, 2) * 2; }  // path.h:108-115

// GetPathPss, path.cpp:2588-2632
LMC_D int GetPathPss(const DPath &path, float *pss) {
    int k = 0;
    if (path.lgtDepth > 1) {
        pss[k++] = path.lgtPos0, pss[k++] = path.lgtPos1, pss[k++] = path.lgtDir0, pss[k++] = path.lgtDir1;
        for (int d = 0; d < path.lgtCount; d++) {
            if (d == path.lgtCount - 1 && path.camDepth == 1) return k;
            if (d == path.lgtCount - 1) break;
            pss[k++] = path.lgt[d].rnd0, pss[k++] = path.lgt[d].rnd1;
        }
    }
    pss[k++] = path.screen0, pss[k++] = path.screen1;
    for (int d = 0; d < path.camCount; d++) {
        if (d == path.camCount - 1) {
            if (path.lgtDepth == 1) pss[k++] = path.cam[d].dirRnd0, pss[k++] = path.cam[d].dirRnd1;
            return k;
        }
        pss[k++] = path.cam[d].rnd0, pss[k++] = path.cam[d].rnd1;
    }
    return k;
}

// LightCoordinateSampling, path.cpp:1881-1951: the last bounce towards an area light, re-sampled in the light's own coordinates
// (the point the vertex's two primary samples select on the triangle the path's last vertex lies on)
template <bool GLOSSY, class Stk>
LMC_D bool LightCoordinateSampling(const DScene &S, const DVertex &cur, int nextTri, BPS &ps, V3 &dir, V3 &bsdfContrib, Stk &stk) {
    const DMaterial &m = MaterialOfTri(S, cur.tri);
    V3 nextPosition, nextNormal;
    float shapePdf;
    SampleTriangle(S, nextTri, V2{cur.rnd0, cur.rnd1}, nextPosition, nextNormal, shapePdf);
    dir = nextPosition - ps.isect.position;
    const float distToLightSq = LengthSquared(dir);
    const float distToLight = sqrtf(distToLightSq);
    dir = dir * inverse(distToLight);
    if (Occluded(S, ps.isect.position, dir, distToLight, stk)) return false;
    float cosWo, bsdfPdf, bsdfRevPdf;
    BsdfEvaluate<GLOSSY>(S, m, false, ps.wi, ps.isect.shadingNormal, dir, V2{cur.st0, cur.st1}, bsdfContrib, cosWo, bsdfPdf, bsdfRevPdf);
    if (IsZero(bsdfContrib)) return false;
    bsdfContrib = bsdfContrib * inverse(bsdfPdf);
    ps.throughput = cmul(ps.throughput, bsdfContrib);
    ps.ssJacobian *= fabsf(Dot(dir, nextNormal) * inverse(distToLightSq)) * bsdfPdf;
    ps.accMISWThis = MIS(cosWo / bsdfPdf) * (ps.accMISWThis * MIS(bsdfRevPdf) + ps.accMISWPrev);
    ps.accMISWPrev = MIS(inverse(bsdfPdf));
    return true;
}

// PerturbPathBidir, path.cpp:1953-2160.  Returns true and fills `out` when the perturbed path carries light.
template <class Stk>
LMC_D bool PerturbPathBidir(const DScene &S, const float *offset, DPath &path, Contrib &out, Rng &rng, Stk &stk) {
    NormalDist normDist(0.0f, S.opt.discreteStdDev);
    TraceOcclusion trace;
    int offsetId = 0;
    path.time = Modulo1(path.time + normDist(rng));
    BPS lps;
    V3 org, dir;
    if (path.lgtDepth > 1) {
        const float lightPickProb = PickLightProb(S, path.lgtLight);
        path.lgtPos0 = Modulo1(path.lgtPos0 + offset[offsetId++]);
        path.lgtPos1 = Modulo1(path.lgtPos1 + offset[offsetId++]);
        path.lgtDir0 = Modulo1(path.lgtDir0 + offset[offsetId++]);
        path.lgtDir1 = Modulo1(path.lgtDir1 + offset[offsetId++]);
        EmitFromLight(S, lightPickProb, path, org, dir, lps);
        for (int lgtDepth = 0; lgtDepth < path.lgtCount; lgtDepth++) {
            DVertex &sv = path.lgt[lgtDepth];
            SurfHit hit;
            if (!IntersectSurface(S, org, dir, c_IsectEpsilon, INFINITY, hit, lps.isect, stk, sv.tri)) return false;  // sv.tri: the state's triangle first (dscene.h)
            sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
            lps.wi = -dir;
            sv.bsdfDiscrete = Modulo1(sv.bsdfDiscrete + normDist(rng));
            ConvertMIS(S, lgtDepth, path.lgtLight, org, dir, lps);
            if (lgtDepth == path.lgtCount - 1 && path.camDepth == 1) return ConnectToCamera(S, lgtDepth, lps, sv, out, stk, trace);
            if (lgtDepth == path.lgtCount - 1) break;
            sv.rnd0 = Modulo1(sv.rnd0 + offset[offsetId++]);
            sv.rnd1 = Modulo1(sv.rnd1 + offset[offsetId++]);
            V3 bsdfContrib;
            if (!BSDFSampling<true, true, Stk::kGlossy>(S, lps, sv, lps, dir, bsdfContrib)) return false;
            lps.throughput = lps.throughput * sv.rrWeight;
            org = lps.isect.position;
        }
    }
    path.screen0 = Modulo1(path.screen0 + offset[offsetId++]);
    path.screen1 = Modulo1(path.screen1 + offset[offsetId++]);
    V2 screenPos{path.screen0, path.screen1};
    BPS cps;
    EmitFromCamera(S, screenPos, org, dir, cps);
    float tnear, tfar;
    tnear = PrimaryMinT(S, screenPos, tfar);
    for (int camDepth = 0; camDepth < path.camCount; camDepth++) {
        DVertex &sv = path.cam[camDepth];
        SurfHit hit;
        hit.tri = -1;
        bool hitSurface = IntersectSurface(S, org, dir, tnear, tfar, hit, cps.isect, stk, sv.tri);
        sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
        cps.wi = -dir;
        if (hitSurface) ConvertMIS(S, camDepth, -1, org, dir, cps);
        if (camDepth == path.camCount - 1 && path.lgtDepth == 0) {
            int light = HitLightOf(S, hitSurface, hit);
            if (light >= 0) return HandleHitLight(S, camDepth, light, hitSurface, dir, screenPos, cps, path.envPrim, out);
            return false;
        }
        if (!hitSurface) return false;
        sv.bsdfDiscrete = Modulo1(sv.bsdfDiscrete + normDist(rng));
        if (camDepth == path.camCount - 1) {
            if (path.lgtDepth == 1) {
                const float directLightPickProb = PickLightProb(S, sv.dirLight);
                sv.dirRnd0 = Modulo1(sv.dirRnd0 + offset[offsetId++]);
                sv.dirRnd1 = Modulo1(sv.dirRnd1 + offset[offsetId++]);
                return DirectLighting(S, camDepth, cps, screenPos, directLightPickProb, sv, out, stk, trace);
            }
            return ConnectVertex(S, camDepth, path.lgtCount - 1, lps, path.lgt[path.lgtCount - 1], cps, sv, screenPos, out, stk, trace);
        }
        sv.rnd0 = Modulo1(sv.rnd0 + offset[offsetId++]);
        sv.rnd1 = Modulo1(sv.rnd1 + offset[offsetId++]);
        bool useLightCoordinatesPerturb = false;  // path.cpp:2120-2128: the path's last vertex (as stored) lies on an area light
        if (S.opt.useLightCoord && camDepth == path.camCount - 2 && path.lgtDepth == 0) {
            const int lastTri = path.cam[path.camCount - 1].tri;
            if (lastTri >= 0 && S.tris[lastTri].areaLight >= 0) useLightCoordinatesPerturb = true;
        }
        V3 bsdfContrib;
        if (useLightCoordinatesPerturb) {
            if (!LightCoordinateSampling<Stk::kGlossy>(S, sv, path.cam[path.camCount - 1].tri, cps, dir, bsdfContrib, stk)) return false;
        } else if (!BSDFSampling<false, true, Stk::kGlossy>(S, cps, sv, cps, dir, bsdfContrib))
            return false;
        cps.throughput = cps.throughput * sv.rrWeight;
        org = cps.isect.position;
        tnear = c_IsectEpsilon;
        tfar = INFINITY;
    }
    return false;
}

}  // namespace lmcd
