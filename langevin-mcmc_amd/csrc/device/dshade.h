// Surface hit reconstruction, BSDFs, lights and the camera on the device.
// Scalar semantics of /root/reference/src/{trianglemesh.cpp:30-79,189-236,291-365, lambertian.cpp:15-93,
// envlight.cpp:120-248, arealight.cpp:28-104, pointlight.cpp:20-80, camera.cpp:38-84, scene.cpp:151-158,
// distribution.h:38-46}; every early-out of the reference is kept.
#pragma once
#include "dscene.h"

// LMC_MAT_CARRY (default 1): the walks that shade a vertex right behind its hit reconstruction (dsmall.h, dwalk.h, dpath.h) take the vertex's material --
// and GetHitLight's answer -- from the index the hit record carried (SurfHit::material / areaLight) instead of fetching S.tris[tri].material again: one
// dependent round trip less per vertex.  With LMC_TEX_INLINE (dscene.h): headline +1.3 %, full-material torus +0.9 %, the others unchanged; same
// results (profiles/r05_bg_ab_material_index_carried_texture_header_inline.jsonl).  0: the look-up through the triangle record (A/B).
#ifndef LMC_MAT_CARRY
#define LMC_MAT_CARRY 1
#endif

namespace lmcd {

struct Isect {
    V3 position, shadingNormal, geomNormal;
};
struct SurfHit {
    int tri;  // global triangle id
    V2 st;
    int material;  // the triangle's material index (TriData::material): the record is in registers when a hit is reconstructed, so the caller's
                   // BSDF need not fetch S.tris[tri].material again -- one dependent round trip less per vertex (LoadMaterialIdx below)
    int areaLight; // ... and TriData::areaLight, what GetHitLight asks of a surface hit (dpath.h HitLightOf)
};

// path.cpp:91-103 = scene.cpp:106-126 (BVH) + TriangleMesh::Intersect (recompute from primID)
template <class Stk>
LMC_D bool IntersectSurface(const DScene &S, V3 org, V3 dir, float tnear, float tfar, SurfHit &hit, Isect &isect, Stk &stk, int hint = -1, bool hintOnly = false) {
    float tB;
    int id = BvhIntersect(S, org, dir, tnear, tfar, tB, stk, hint, hintOnly);
    if (id < 0) return false;
    TriData T = S.tris[id];  // the whole record at once: by value, and pinned (dscene.h LMC_PIN) -- hipcc fetched the normals and the st / material words
                             // where they are first used, two more dependent round trips per hit
    LMC_PIN3(T.n0[0], T.n1[0], T.n2[0]);
    LMC_PIN3(T.n0[2], T.n1[2], T.n2[2]);
    LMC_PIN4(T.st[0], T.st[3], T.st[5], T.hasST);
    V3 p0{T.p0[0], T.p0[1], T.p0[2]}, e1{T.e1[0], T.e1[1], T.e1[2]}, e2{T.e2[0], T.e2[1], T.e2[2]};
    isect.geomNormal = Normalize(Cross(e1, e2));
    V3 s1 = Cross(dir, e2);
    float divisor = Dot(s1, e1);
    if (divisor == 0.0f) return false;
    float invDivisor = inverse(divisor);
    V3 s = org - p0;
    float u = Dot(s, s1) * invDivisor;
    V3 s2 = Cross(s, e1);
    float v = Dot(dir, s2) * invDivisor;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    float t = Dot(e2, s2) * invDivisor;
    float w = 1.0f - u - v;
    isect.position = org + t * dir;
    V3 n0{T.n0[0], T.n0[1], T.n0[2]}, n1{T.n1[0], T.n1[1], T.n1[2]}, n2{T.n2[0], T.n2[1], T.n2[2]};
    isect.shadingNormal = Normalize(w * n0 + u * n1 + v * n2);
    if (Dot(isect.geomNormal, isect.shadingNormal) < 0.0f) isect.geomNormal = -isect.geomNormal;
    hit.tri = id;
    hit.material = T.material, hit.areaLight = T.areaLight;
    if (T.hasST) {
        hit.st.x = (1.0f - u - v) * T.st[0] + u * T.st[2] + v * T.st[4];
        hit.st.y = (1.0f - u - v) * T.st[1] + u * T.st[3] + v * T.st[5];
    } else {
        hit.st = V2{u, v};
    }
    return true;
}

// PiecewiseConstant1D::SampleDiscrete (distribution.h:38-46): std::upper_bound over cdf[0..count]
LMC_D int SampleDiscrete1D(const float *func, const float *cdf, int count, float funcInt, float u, float *pdf) {
    int lo = 0, len = count + 1;
    while (len > 0) {
        int half = len >> 1;
        if (!(u < cdf[lo + half])) {
            lo += half + 1;
            len -= half + 1;
        } else
            len = half;
    }
    int offset = Clampi(lo - 1, 0, count - 1);
    if (pdf) *pdf = func[offset] / (funcInt * count);
    return offset;
}

LMC_D int PickLight(const DScene &S, float u, float &prob) { return SampleDiscrete1D(S.lightFunc, S.lightCdf, S.numLights, S.lightFuncInt, u, &prob); }
LMC_D float PickLightProb(const DScene &S, int light) { return S.lights[light].samplingWeight / S.lightWeightSum; }

// ---------------------------------------------------------------------------------------------- BSDF
LMC_D const DMaterial &MaterialOfTri(const DScene &S, int tri) { return S.materials[S.tris[tri].material]; }
// ... the record by value, all the words the instantiation reads in ONE round of loads (dscene.h LMC_PIN: through the reference above every field
// was fetched where it is first used, a dependent round trip each -- `twoSided` behind the cosine test, Ks / exponent / KsWeight one after the other)
#if defined(LMC_MAT_LDS) && defined(__HIP_DEVICE_COMPILE__)
// A/B build option, measured and NOT taken (profiles/r05_aj_ab_material_records_in_lds_rejected.jsonl: veach-door LMC -4.5 %, full-material torus -2 %, headline
// -1 %: 29 LDS reads per BSDF evaluation cost more than the two vector loads they replace).  A translation unit that defines LMC_MAT_LDS keeps the scene's first LMC_MAT_LDS_MAX material records (116 B each; the torus
// scene has 7, the veach-door scene 12) in LDS, filled once per block: a BSDF's parameters then are LDS reads, outside the in-order `vmcnt` queue
// of the vector-memory loads (drng.h LMC_RNG_JUMP_LDS has the reasoning).  A scene with more materials reads the rest from memory as before.
constexpr int LMC_MAT_LDS_MAX = 32;
__device__ __forceinline__ DMaterial *MaterialLds() {
    __shared__ DMaterial t[LMC_MAT_LDS_MAX];
    return t;
}
__device__ __forceinline__ void MaterialLdsInit(const DScene &S) {
    const int words = min(S.numMaterials, LMC_MAT_LDS_MAX) * (int)(sizeof(DMaterial) / 4);
    const int *src = reinterpret_cast<const int *>(S.materials);
    int *dst = reinterpret_cast<int *>(MaterialLds());
    for (int w = threadIdx.x; w < words; w += blockDim.x) dst[w] = src[w];
    __syncthreads();
}
#define LMC_MAT_LDS_INIT(S) lmcd::MaterialLdsInit(S)
#else
#define LMC_MAT_LDS_INIT(S) ((void)0)
#endif
template <bool GLOSSY>
LMC_D DMaterial LoadMaterialIdx(const DScene &S, int mi) {  // by material index (SurfHit::material of a hit just reconstructed)
#if defined(LMC_MAT_LDS) && defined(__HIP_DEVICE_COMPILE__)
    if (mi < LMC_MAT_LDS_MAX) return MaterialLds()[mi];
#endif
    DMaterial m = S.materials[mi];
    LMC_PIN4(m.type, m.twoSided, m.Kd.bitmap, m.Kd.value[0]);
    LMC_PIN4(m.Kd.value[1], m.Kd.value[2], m.Kd.sScale, m.Kd.tScale);
    if constexpr (GLOSSY) {
        LMC_PIN4(m.Ks.bitmap, m.Ks.value[0], m.Ks.value[1], m.Ks.value[2]);
        LMC_PIN4(m.Ks.sScale, m.Ks.tScale, m.Kt.bitmap, m.Kt.value[0]);
        LMC_PIN4(m.Kt.value[1], m.Kt.value[2], m.Kt.sScale, m.Kt.tScale);
        LMC_PIN4(m.expOrAlpha.bitmap, m.expOrAlpha.value[0], m.expOrAlpha.sScale, m.expOrAlpha.tScale);
        LMC_PIN3(m.eta, m.invEta, m.KsWeight);
    }
    return m;
}
template <bool GLOSSY>
LMC_D DMaterial LoadMaterial(const DScene &S, int tri) {
    return LoadMaterialIdx<GLOSSY>(S, S.tris[tri].material);
}

// Texture::Eval.  Bitmaps: periodic bilinear lookup standing in for OIIO's TextureSystem::texture() with zero filter
// width, then fastpow(max(v,0), gamma) (bitmaptexture.h:72-97); the same arithmetic as host/scene.cpp:EvalTexture.
LMC_D V3 EvalTex(const DScene &S, const DTexRef &t, V2 st) {
    if (t.bitmap < 0) return V3{t.value[0], t.value[1], t.value[2]};
    // every load below is pinned where it stands (dscene.h LMC_PIN): hipcc fetched the bitmap's fields and the twelve texel words one by one
    // where each is used -- a chain of up to fourteen dependent round trips per look-up (hipcc -S of the round-5 door kernels: 77 of the 88
    // loads of this function's inlined copies were waited for on their own)
#if LMC_TEX_INLINE
    const float *pix = S.texPool + t.bitmap;
    const int W = __float_as_int(t.value[0]), H = __float_as_int(t.value[1]);
    const float gamma = t.value[2];
#else
    const DBitmap *bp = S.bitmaps + t.bitmap;
    const float *pix = bp->pix;
    int W = bp->W, H = bp->H;
    float gamma = bp->gamma;
    LMC_PIN4(pix, W, H, gamma);
#endif
    const float fs = t.sScale * st.x * W - 0.5f, ft = t.tScale * st.y * H - 0.5f;
    const float x0f = floorf(fs), y0f = floorf(ft);
    const float dx = fs - x0f, dy = ft - y0f;
    // periodic wrap of floor(coordinate) and its right / upper neighbour.  The reference arithmetic is (long long) % n; inside +-2^30 the 32-bit
    // remainder is the same number (and a tenth of the instructions: a 64-bit remainder is a software loop on this hardware)
    auto wrap2 = [](float vf, int n, int &a, int &b) {
        if (fabsf(vf) < 1073741824.0f) {
            const int v = (int)vf;
            int r = v % n;
            r = r < 0 ? r + n : r;
            a = r, b = r + 1 == n ? 0 : r + 1;
        } else {
            const long long v = (long long)vf;
            long long r = v % n, r1 = (v + 1) % n;
            a = (int)(r < 0 ? r + n : r), b = (int)(r1 < 0 ? r1 + n : r1);
        }
    };
    int x0, x1, y0, y1;
    wrap2(x0f, W, x0, x1), wrap2(y0f, H, y0, y1);
    const float *p00 = pix + ((size_t)y0 * W + x0) * 3, *p10 = pix + ((size_t)y0 * W + x1) * 3;
    const float *p01 = pix + ((size_t)y1 * W + x0) * 3, *p11 = pix + ((size_t)y1 * W + x1) * 3;
    float a00[3] = {p00[0], p00[1], p00[2]}, a10[3] = {p10[0], p10[1], p10[2]}, a01[3] = {p01[0], p01[1], p01[2]}, a11[3] = {p11[0], p11[1], p11[2]};
    LMC_PIN3(a00[0], a00[1], a00[2]);
    LMC_PIN3(a10[0], a10[1], a10[2]);
    LMC_PIN3(a01[0], a01[1], a01[2]);
    LMC_PIN3(a11[0], a11[1], a11[2]);
    float o[3];
    for (int k = 0; k < 3; k++) {
        const float v = (1 - dx) * (1 - dy) * a00[k] + dx * (1 - dy) * a10[k] + (1 - dx) * dy * a01[k] + dx * dy * a11[k];
        o[k] = fastpow(fmaxf(v, 0.f), gamma);
    }
    return V3{o[0], o[1], o[2]};
}
LMC_D V3 EvalKd(const DScene &S, const DMaterial &m, V2 st) { return EvalTex(S, m.Kd, st); }

template <bool GLOSSY>
LMC_D float BsdfRoughness(const DScene &S, const DMaterial &m, V2 st, float) {
    if (GLOSSY && m.type == BSDF_ROUGHDIELECTRIC) return EvalTex(S, m.expOrAlpha, st).x;  // roughdielectric.h:61-63
    return 1.0f;                                                                // lambertian.h, phong.cpp:155-157
}

// ---- microfacet.h:6-70,165-185 (scalar versions)
LMC_D float BeckmennDistributionTerm(V3 localH, float alphaU, float alphaV) {
    const float cosTheta = localH.z, mu = localH.x, mv = localH.y;
    const float cosTheta2 = square(cosTheta);
    const float beckmannExponent = (square(mu) / square(alphaU) + square(mv) / square(alphaV)) / cosTheta2;
    return lexpf(-beckmannExponent) / (c_PI * alphaU * alphaV * square(cosTheta2));
}
LMC_D float BeckmennGeometryTerm1(float alpha, float cosTheta) {
    const float tanTheta = sqrtf(fabsf(1.0f - square(cosTheta))) / cosTheta;
    if (tanTheta <= 0.0f) return 1.0f;
    const float a = 1.0f / (alpha * tanTheta);
    if (a >= 1.6f) return 1.0f;
    const float aSqr = a * a;
    return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
}
LMC_D float BeckmennGeometryTerm(float alpha, float cosWi, float cosWo) { return BeckmennGeometryTerm1(alpha, cosWi) * BeckmennGeometryTerm1(alpha, cosWo); }
LMC_D float FresnelDielectricExt(float cosThetaI_, float &cosThetaT_, float eta, float invEta) {
    const float scale = (cosThetaI_ > 0) ? invEta : eta;
    const float cosThetaTSqr = 1.0f - (1.0f - square(cosThetaI_)) * square(scale);
    if (cosThetaTSqr <= 0.0f) {
        cosThetaT_ = 0.0f;
        return 1.0f;
    }
    const float cosThetaI = fabsf(cosThetaI_);
    const float cosThetaT = sqrtf(cosThetaTSqr);
    const float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    const float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (square(Rs) + square(Rp));
}
LMC_D V3 SampleMicronormal(V2 rndParam, float alpha, float &pdfW) {
    const float phiM = c_TWOPI * rndParam.y;
    const float sinPhiM = lsinf(phiM), cosPhiM = lcosf(phiM);
    const float alphaSqr = square(alpha);
    const float tanThetaMSqr = alphaSqr * (-llogf(fmaxf(1.0f - rndParam.x, 1e-6f)));
    const float cosThetaM = 1.0f / sqrtf(1.0f + tanThetaMSqr);
    const float cosThetaMSqr = square(cosThetaM);
    pdfW = (1.0f - rndParam.x) / (c_PI * alphaSqr * cosThetaM * cosThetaMSqr);
    const float sinThetaM = sqrtf(fmaxf(1.0f - cosThetaMSqr, 0.0f));  // ADEpsilon<Float>() == 0
    return V3{sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM};
}

// lambertian.cpp:15-43 (pdf / revPdf are left untouched on the early-out, as in the reference)
LMC_D void LambertianEvaluate(const DScene &S, const DMaterial &m, V3 wi, V3 normal, V3 wo, V2 st, V3 &contrib, float &cosWo, float &pdf, float &revPdf) {
    float cosWi = Dot(normal, wi);
    V3 normal_ = normal;
    if (m.twoSided && cosWi < 0.0f) {
        cosWi = -cosWi;
        normal_ = -normal_;
    }
    cosWo = Dot(normal_, wo);
    contrib = V3{0, 0, 0};
    if (cosWi < c_CosEpsilon || cosWo < c_CosEpsilon) return;
    float fwdScalar = cosWo * c_INVPI;
    float revScalar = cosWi * c_INVPI;
    contrib = fwdScalar * EvalKd(S, m, st);
    pdf = fwdScalar;
    revPdf = revScalar;
}
// lambertian.cpp:45-93
LMC_D bool LambertianSample(const DScene &S, const DMaterial &m, V3 wi, V3 normal, V2 st, V2 rnd, V3 &wo, V3 &contrib, float &cosWo, float &pdf, float &revPdf) {
    float cosWi = Dot(wi, normal);
    V3 normal_ = normal;
    if (fabsf(cosWi) < c_CosEpsilon) return false;
    if (cosWi < 0.0f) {
        if (m.twoSided) {
            cosWi = -cosWi;
            normal_ = -normal_;
        } else
            return false;
    }
    V3 b0, b1;
    CoordinateSystem(normal_, b0, b1);
    V3 ret = SampleCosHemisphere(rnd);
    wo = ret.x * b0 + ret.y * b1 + ret.z * normal_;
    cosWo = ret.z;
    pdf = ret.z * c_INVPI;
    if (cosWo < c_CosEpsilon) return false;
    revPdf = cosWi * c_INVPI;
    contrib = EvalKd(S, m, st);
    return true;
}

// phong.cpp:22-68
LMC_D void PhongEvaluate(const DScene &S, const DMaterial &m, V3 wi, V3 normal, V3 wo, V2 st, V3 &contrib, float &cosWo, float &pdf, float &revPdf) {
    contrib = V3{0, 0, 0};
    pdf = 0.0f;
    revPdf = 0.0f;
    float cosWi = Dot(normal, wi);
    V3 normal_ = normal;
    if (m.twoSided && cosWi < 0.0f) {
        cosWi = -cosWi;
        normal_ = -normal_;
    }
    cosWo = Dot(normal_, wo);
    if (cosWi <= c_CosEpsilon || cosWo <= c_CosEpsilon) return;
    const float KsWeight = m.KsWeight;
    if (KsWeight > 0.0f) {
        const float alpha = fmaxf(Dot(Reflect(wi, normal_), wo), 0.0f);
        const float expo = EvalTex(S, m.expOrAlpha, st).x;
        const float weight = lpowf(alpha, expo) * c_INVTWOPI;
        const float expoConst1 = (expo + 1.0f);
        const float expoConst2 = (expo + 2.0f);
        if (weight > 1e-10f) {
            contrib = EvalTex(S, m.Ks, st) * (expoConst2 * weight);
            pdf = KsWeight * expoConst1 * weight;
            revPdf = pdf;
        }
    }
    if (KsWeight < 1.0f) {
        pdf += (1.0f - KsWeight) * cosWo * c_INVPI;
        revPdf += (1.0f - KsWeight) * cosWi * c_INVPI;
        contrib = contrib + EvalTex(S, m.Kd, st) * c_INVPI;
    }
    contrib = contrib * cosWo;
    if (MaxCoeff(contrib) < 1e-10f) contrib = V3{0, 0, 0};
}
// phong.cpp:70-153 (the lobe choice re-uses rndParam[0]; revPdf accumulates onto the caller's value when KsWeight == 0)
LMC_D bool PhongSample(const DScene &S, const DMaterial &m, V3 wi, V3 normal, V2 st, V2 rndParam, V3 &wo, V3 &contrib, float &cosWo, float &pdf, float &revPdf) {
    float cosWi = Dot(wi, normal);
    if (fabsf(cosWi) < c_CosEpsilon) return false;
    V3 normal_ = normal;
    if (cosWi < 0.0f) {
        if (m.twoSided) {
            cosWi = -cosWi;
            normal_ = -normal_;
        } else
            return false;
    }
    const float KsWeight = m.KsWeight;
    const float expo = EvalTex(S, m.expOrAlpha, st).x;
    const V3 R = Reflect(wi, normal_);
    float g;
    V3 n;
    const float uDiscrete = rndParam.x;
    float rndParam0;
    if (uDiscrete > KsWeight) {
        g = 1.0f;
        n = normal_;
        rndParam0 = (uDiscrete - KsWeight) / (1.0f - KsWeight + 1e-10f);
    } else {
        g = expo;
        n = R;
        rndParam0 = uDiscrete / (KsWeight + 1e-10f);
    }
    const float power = 1.0f / (g + 1.0f);
    const float cosAlpha = lpowf(rndParam.y, power);
    const float sinAlpha = sqrtf(1.0f - square(cosAlpha));
    const float phi = c_TWOPI * rndParam0;
    const V3 localDir{sinAlpha * lcosf(phi), sinAlpha * lsinf(phi), cosAlpha};
    V3 b0, b1;
    CoordinateSystem(n, b0, b1);
    wo = localDir.x * b0 + localDir.y * b1 + localDir.z * n;
    cosWo = Dot(normal_, wo);
    if (cosWo < c_CosEpsilon) return false;
    contrib = V3{0, 0, 0};
    pdf = 0.0f;
    if (KsWeight > 0.0f) {
        const float alpha = fmaxf(Dot(R, wo), 0.0f);
        const float weight = lpowf(alpha, expo) * c_INVTWOPI;
        const float expoConst1 = (expo + 1.0f);
        const float expoConst2 = (expo + 2.0f);
        if (weight > 1e-10f) {
            contrib = EvalTex(S, m.Ks, st) * (expoConst2 * weight);
            pdf = KsWeight * expoConst1 * weight;
        }
        revPdf = pdf;
    }
    if (KsWeight < 1.0f) {
        contrib = contrib + EvalTex(S, m.Kd, st) * c_INVPI;
        pdf += (1.0f - KsWeight) * cosWo * c_INVPI;
        revPdf += (1.0f - KsWeight) * cosWi * c_INVPI;
    }
    contrib = contrib * cosWo;
    if (pdf < 1e-10f) return false;
    contrib = contrib * inverse(pdf);
    return true;
}

// roughdielectric.cpp:22-121
LMC_D void RoughDielectricEvaluate(const DScene &S, const DMaterial &m, bool adjoint, V3 wi, V3 normal, V3 wo, V2 st, V3 &contrib, float &cosWo, float &pdf,
                                   float &revPdf) {
    const float eta = m.eta, invEta = m.invEta;
    const float cosWi = Dot(wi, normal);
    contrib = V3{0, 0, 0};
    cosWo = 0.0f;
    pdf = revPdf = 0.0f;
    if (fabsf(cosWi) < c_CosEpsilon) return;
    cosWo = Dot(wo, normal);
    if (fabsf(cosWo) < c_CosEpsilon) return;
    const bool reflect = cosWi * cosWo > 0.0f;
    const float eta_ = cosWi > 0.0f ? eta : invEta;
    const float revEta_ = cosWo > 0.0f ? eta : invEta;
    V3 H;
    if (reflect) H = Normalize(wi + wo);
    else
        H = Normalize(wi + wo * eta_);
    if (Dot(H, normal) < 0.0f) H = -H;
    const float cosHWi = Dot(wi, H);
    const float cosHWo = Dot(wo, H);
    if (fabsf(cosHWi) < c_CosEpsilon || fabsf(cosHWo) < c_CosEpsilon) return;
    if (cosHWi * cosWi <= 0.0f) return;
    if (cosHWo * cosWo <= 0.0f) return;
    V3 b0, b1;
    CoordinateSystem(normal, b0, b1);
    const V3 localH{Dot(b0, H), Dot(b1, H), Dot(normal, H)};
    const float alp = EvalTex(S, m.expOrAlpha, st).x;
    const float D = BeckmennDistributionTerm(localH, alp, alp);
    if (D <= 0.0f) return;
    const float revCosHWi = cosHWo;
    const float revCosHWo = cosHWi;
    float unusedT;
    const float F = FresnelDielectricExt(cosHWi, unusedT, eta, invEta);
    const float aCosWi = fabsf(cosWi);
    const float aCosWo = fabsf(cosWo);
    const float G = BeckmennGeometryTerm(alp, aCosWi, aCosWo);
    const float scaledAlpha = alp * (1.2f - 0.2f * sqrtf(aCosWi));
    const float scaledD = BeckmennDistributionTerm(localH, scaledAlpha, scaledAlpha);
    const float prob = localH.z * scaledD;
    if (prob < 1e-20f) {
        contrib = V3{0, 0, 0};
        return;
    }
    const float revScaledAlpha = alp * (1.2f - 0.2f * sqrtf(aCosWo));
    const float revScaledD = BeckmennDistributionTerm(localH, revScaledAlpha, revScaledAlpha);
    const float revProb = localH.z * revScaledD;
    if (reflect) {
        const float scalar = fabsf(F * D * G / (4.0f * cosWi));
        contrib = EvalTex(S, m.Ks, st) * scalar;
        pdf = fabsf(prob * F / (4.0f * cosHWo));
        revPdf = fabsf(revProb * F / (4.0f * revCosHWo));
    } else {
        const float sqrtDenom = cosHWi + eta_ * cosHWo;
        const float revSqrtDenom = revCosHWi + revEta_ * revCosHWo;
        const float factor = adjoint ? 1.0f : square(inverse(eta_));
        const float scalar = fabsf(factor * ((1.0f - F) * D * G * square(eta_) * cosHWi * cosHWo) / (cosWi * square(sqrtDenom)));
        contrib = EvalTex(S, m.Kt, st) * scalar;
        pdf = fabsf(prob * (1.0f - F) * (square(eta_) * cosHWo) / (square(sqrtDenom)));
        revPdf = fabsf(revProb * (1.0f - F) * (square(revEta_) * revCosHWo) / (square(revSqrtDenom)));
    }
}
// roughdielectric.cpp:148-301
LMC_D bool RoughDielectricSample(const DScene &S, const DMaterial &m, bool adjoint, V3 wi, V3 normal, V2 st, V2 rndParam, float uDiscrete, V3 &wo, V3 &contrib,
                                 float &cosWo, float &pdf, float &revPdf) {
    const float eta = m.eta, invEta = m.invEta;
    const float cosWi = Dot(wi, normal);
    if (fabsf(cosWi) < c_CosEpsilon) return false;
    const float alp = EvalTex(S, m.expOrAlpha, st).x;
    const float scaledAlp = alp * (1.2f - 0.2f * sqrtf(fabsf(cosWi)));
    float mPdf;
    const V3 localH = SampleMicronormal(rndParam, scaledAlp, mPdf);
    pdf = mPdf;
    V3 b0, b1;
    CoordinateSystem(normal, b0, b1);
    const V3 H = localH.x * b0 + localH.y * b1 + localH.z * normal;
    const float cosHWi = Dot(wi, H);
    if (fabsf(cosHWi) < c_CosEpsilon) return false;
    float cosThetaT = 0.0f;
    const float F = FresnelDielectricExt(cosHWi, cosThetaT, eta, invEta);
    const bool reflect = uDiscrete <= F;
    V3 refl;
    float cosHWo;
    if (reflect) {
        wo = Reflect(wi, H);
        if (F <= 0.0f || Dot(normal, wo) * Dot(normal, wi) <= 0.0f) return false;
        refl = EvalTex(S, m.Ks, st);
        cosHWo = Dot(wo, H);
        pdf = fabsf(pdf * F / (4.0f * cosHWo));
        const float revCosHWo = cosHWi;
        const float rev_dwh_dwo = inverse(4.0f * revCosHWo);
        cosWo = Dot(wo, normal);
        if (fabsf(cosWo) < c_CosEpsilon) return false;
        const float revScaledAlp = alp * (1.2f - 0.2f * sqrtf(fabsf(cosWo)));
        const float revD = BeckmennDistributionTerm(localH, revScaledAlp, revScaledAlp);
        revPdf = fabsf(F * revD * localH.z * rev_dwh_dwo);
    } else {
        wo = Refract(wi, H, cosThetaT, eta, invEta);
        if (F >= 1.0f || cosThetaT == 0.0f || Dot(normal, wo) * Dot(normal, wi) >= 0.0f) return false;
        const float eta_ = cosWi > 0.0f ? eta : invEta;
        const float factor = adjoint ? 1.0f : square(inverse(eta_));
        refl = EvalTex(S, m.Kt, st) * factor;
        cosHWo = Dot(wo, H);
        const float sqrtDenom = cosHWi + eta_ * cosHWo;
        const float dwh_dwo = (square(eta_) * cosHWo) / square(sqrtDenom);
        pdf = fabsf(pdf * (1.0f - F) * fabsf(dwh_dwo));
        cosWo = Dot(wo, normal);
        if (fabsf(cosWo) < c_CosEpsilon) return false;
        const float revEta_ = cosWo > 0.0f ? eta : invEta;
        const float revCosHWi = cosHWo;
        const float revCosHWo = cosHWi;
        const float revSqrtDenom = revCosHWi + revEta_ * revCosHWo;
        const float rev_dwh_dwo = (square(revEta_) * revCosHWo) / square(revSqrtDenom);
        const float revScaledAlp = alp * (1.2f - 0.2f * sqrtf(fabsf(cosWo)));
        const float revD = BeckmennDistributionTerm(localH, revScaledAlp, revScaledAlp);
        revPdf = fabsf((1.0f - F) * revD * localH.z * rev_dwh_dwo);
    }
    if (fabsf(cosHWo) < c_CosEpsilon) return false;
    if (pdf < 1e-20f) return false;
    if (cosHWi * cosWi <= 0.0f) return false;
    if (cosHWo * cosWo <= 0.0f) return false;
    const float aCosWi = fabsf(cosWi);
    const float aCosWo = fabsf(cosWo);
    const float D = BeckmennDistributionTerm(localH, alp, alp);
    const float G = BeckmennGeometryTerm(alp, aCosWi, aCosWo);
    const float numerator = D * G * cosHWi;
    const float denominator = mPdf * aCosWi;
    contrib = refl * fabsf(numerator / denominator);
    return true;
}

// BSDF::Evaluate / EvaluateAdjoint (bsdf.h:16-38): only the rough dielectric distinguishes the adjoint
template <bool GLOSSY>
LMC_D void BsdfEvaluate(const DScene &S, const DMaterial &m, bool adjoint, V3 wi, V3 normal, V3 wo, V2 st, V3 &contrib, float &cosWo, float &pdf,
                        float &revPdf) {
    if (!GLOSSY || m.type == BSDF_LAMBERTIAN) LambertianEvaluate(S, m, wi, normal, wo, st, contrib, cosWo, pdf, revPdf);
    else if (m.type == BSDF_PHONG)
        PhongEvaluate(S, m, wi, normal, wo, st, contrib, cosWo, pdf, revPdf);
    else
        RoughDielectricEvaluate(S, m, adjoint, wi, normal, wo, st, contrib, cosWo, pdf, revPdf);
}
// BSDF::Sample / SampleAdjoint (bsdf.h:40-72)
template <bool GLOSSY>
LMC_D bool BsdfSample(const DScene &S, const DMaterial &m, bool adjoint, V3 wi, V3 normal, V2 st, V2 rnd, float uDiscrete, V3 &wo, V3 &contrib,
                      float &cosWo, float &pdf, float &revPdf) {
    if (!GLOSSY || m.type == BSDF_LAMBERTIAN) return LambertianSample(S, m, wi, normal, st, rnd, wo, contrib, cosWo, pdf, revPdf);
    if (m.type == BSDF_PHONG) return PhongSample(S, m, wi, normal, st, rnd, wo, contrib, cosWo, pdf, revPdf);
    return RoughDielectricSample(S, m, adjoint, wi, normal, st, rnd, uDiscrete, wo, contrib, cosWo, pdf, revPdf);
}

// ---------------------------------------------------------------------------------------------- shapes
// TriangleMesh::Sample (trianglemesh.cpp:291-365, ADEpsilon<Float>() == 0)
LMC_D void SampleTriangle(const DScene &S, int tri, V2 rnd, V3 &position, V3 &normal, float &pdf) {
    const TriData &T = S.tris[tri];
    V3 p0{T.p0[0], T.p0[1], T.p0[2]}, e1{T.e1[0], T.e1[1], T.e1[2]}, e2{T.e2[0], T.e2[1], T.e2[2]};
    const float a = sqrtf((1.0f + 0.0f) - rnd.x);
    const float b1 = 1.0f - a;
    const float b2 = a * rnd.y;
    position = p0 + (e1 * b1) + (e2 * b2);
    V3 n0{T.n0[0], T.n0[1], T.n0[2]}, n1{T.n1[0], T.n1[1], T.n1[2]}, n2{T.n2[0], T.n2[1], T.n2[2]};
    normal = Normalize(n0 * (1.0f - b1 - b2) + n1 * b1 + n2 * b2);
    pdf = S.meshes[T.mesh].invTotalArea;
}

// TriangleMesh::GetSampleParam (trianglemesh.cpp:238-285): the sampling coordinates that SampleTriangle maps to `position`
LMC_D V2 TriangleSampleParam(const DScene &S, int tri, V3 position) {
    const TriData &T = S.tris[tri];
    V3 p0{T.p0[0], T.p0[1], T.p0[2]}, e1{T.e1[0], T.e1[1], T.e1[2]}, e2{T.e2[0], T.e2[1], T.e2[2]};
    const V3 e0 = position - p0;
    const float d11 = Dot(e1, e1);
    const float d12 = Dot(e1, e2);
    const float d22 = Dot(e2, e2);
    const float d01 = Dot(e0, e1);
    const float d02 = Dot(e0, e2);
    const float invDenom = inverse(d11 * d22 - d12 * d12);
    const float b1 = (d22 * d01 - d12 * d02) * invDenom;
    const float b2 = (d11 * d02 - d12 * d01) * invDenom;
    const float a = 1.0f - b1;
    return V2{(1.0f + 0.0f) - square(a), b2 / a};
}

// ---------------------------------------------------------------------------------------------- lights
LMC_D V3 EnvAtLinear(const DEnv &E, int x, int y) {  // Image3::At without range check; one-past-the-end wraps (oracle AtQ)
    long idx = (long)y * E.W + x;
    long n = (long)E.W * E.H;
    if (idx >= n) idx -= n;
    const float *p = E.image + idx * 3;
    return V3{p[0], p[1], p[2]};
}
LMC_D V3 EnvRepAt(const DEnv &E, int x, int y) {
    const float *p = E.image + ((long)Moduloi(y, E.H) * E.W + Moduloi(x, E.W)) * 3;
    return V3{p[0], p[1], p[2]};
}
// std::lower_bound(cdf, cdf + n, u): the first index whose entry is not below u (n if none).  A CDF is non-decreasing, so that
// index does not depend on which entries a search probes: eight independent probes per round (one memory round trip) shrink
// the range ninefold, against one dependent probe per halving -- 3 round trips instead of 9-10 for the env map's 257 / 513 entries.
LMC_D int LowerBoundMonotone(const float *cdf, int n, float u) {
    int lo = 0, hi = n;  // entries before lo are below u, entries from hi on are not
    while (hi - lo > 8) {
        const int span = hi - lo;
        int p[8];
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) p[j] = lo + ((j + 1) * span) / 9, v[j] = cdf[p[j]];
        int c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) c += v[j] < u ? 1 : 0;
        int nlo = lo, nhi = hi;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j == c - 1) nlo = p[j] + 1;
            if (j == c) nhi = p[j];
        }
        lo = nlo, hi = nhi;
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = cdf[min(lo + j, n - 1)];
    int c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) c += (lo + j < hi && v[j] < u) ? 1 : 0;
    return lo + c;
}
LMC_D int EnvUToIndex(const float *cdf, int size, float &u) {  // std::lower_bound, envlight.cpp:128-133
    const int lo = LowerBoundMonotone(cdf, size + 1, u);
    int index = lo - 1;
    if (index < 0) index = 0;
    if (index > size - 1) index = size - 1;
    u = (u - cdf[index]) / (cdf[index + 1] - cdf[index]);
    return index;
}
// envlight.cpp:120-171
LMC_D void EnvSampleDirection(const DScene &S, V2 rnd, int &lPrimID, V3 &dirToLight, V3 &value, float &pdf) {
    const DEnv &E = S.env;
    float u0 = rnd.x, u1 = rnd.y;
    int row = EnvUToIndex(E.cdfRows, E.H, u1);
    int col = EnvUToIndex(E.cdfCols + (long)row * (E.W + 1), E.W, u0);
    lPrimID = row * E.W + col;
    V2 tent{Tent(u0), Tent(u1)};
    float phi = (((float)col + tent.x) + 0.5f) * E.pixelSize[0];
    float theta = (((float)row + tent.y) + 0.5f) * E.pixelSize[1];
    float sinPhi = lsinf(phi), cosPhi = lcosf(phi), sinTheta = lsinf(theta), cosTheta = lcosf(theta);
    dirToLight = XformVector(E.toWorld, V3{sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta});
    float dx1 = tent.x, dx2 = 1.0f - tent.x, dy1 = tent.y, dy2 = 1.0f - tent.y;
    // the four texels and the two row weights in one round of loads (dscene.h LMC_PIN: they were four + two dependent round trips)
    V3 e00 = EnvAtLinear(E, col, row), e10 = EnvAtLinear(E, col + 1, row), e01 = EnvAtLinear(E, col, row + 1), e11 = EnvAtLinear(E, col + 1, row + 1);
    float rowWeight0 = E.rowWeights[Clampi(row, 0, E.H - 1)];
    float rowWeight1 = E.rowWeights[Clampi(row + 1, 0, E.H - 1)];
    LMC_PIN4(e00.x, e00.y, e00.z, rowWeight0);
    LMC_PIN4(e10.x, e10.y, e10.z, rowWeight1);
    LMC_PIN3(e01.x, e01.y, e01.z);
    LMC_PIN3(e11.x, e11.y, e11.z);
    V3 value1 = e00 * dx2 * dy2 + e10 * dx1 * dy2;
    V3 value2 = e01 * dx2 * dy1 + e11 * dx1 * dy1;
    value = value1 + value2;
    pdf = (Luminance(value1) * rowWeight0 + Luminance(value2) * rowWeight1) * E.normalization / fmaxf(fabsf(sinTheta), 1e-7f);
}

// Light::SampleDirect
LMC_D bool LightSampleDirect(const DScene &S, int light, V3 pos, V2 rnd, int &lPrimID, V3 &dirToLight, float &dist, V3 &contrib,
                             float &cosAtLight, float &directPdf, float &emissionPdf) {
    const DLight &L = S.lights[light];
    if (L.type == LIGHT_ENV) {  // envlight.cpp:173-193
        V3 value;
        EnvSampleDirection(S, rnd, lPrimID, dirToLight, value, directPdf);
        dist = INFINITY;
        contrib = value * inverse(directPdf);
        cosAtLight = 1.0f;
        float positionPdf = c_INVPI / square(S.bsRadius);
        emissionPdf = directPdf * positionPdf;
        return true;
    } else if (L.type == LIGHT_AREA) {  // arealight.cpp:28-60
        V3 posOnLight, normalOnLight;
        float shapePdf;
        SampleTriangle(S, S.meshes[L.mesh].triBase + lPrimID, rnd, posOnLight, normalOnLight, shapePdf);
        dirToLight = posOnLight - pos;
        float distSq = LengthSquared(dirToLight);
        dist = sqrtf(distSq);
        dirToLight = dirToLight / dist;
        cosAtLight = -Dot(dirToLight, normalOnLight);
        if (cosAtLight > c_CosEpsilon) {
            V3 em{L.radiance[0], L.radiance[1], L.radiance[2]};
            contrib = (cosAtLight / (distSq * shapePdf)) * em;
            directPdf = shapePdf * distSq / cosAtLight;
            emissionPdf = shapePdf * cosAtLight * c_INVPI;
            return true;
        }
        return false;
    } else {  // pointlight.cpp:20-55
        V3 lp{L.pos[0], L.pos[1], L.pos[2]};
        dirToLight = lp - pos;
        const float distSq = LengthSquared(dirToLight);
        directPdf = distSq;
        dist = sqrtf(distSq);
        dirToLight = dirToLight / dist;
        contrib = V3{L.intensity[0], L.intensity[1], L.intensity[2]} * inverse(distSq);
        emissionPdf = c_INVFOURPI;
        cosAtLight = 1.0f;
        lPrimID = 0;
        return true;
    }
}

// Light::Emission (env: envlight.cpp:195-222; area: arealight.cpp:62-79)
LMC_D void LightEmission(const DScene &S, int light, V3 dirToLight, V3 normalOnLight, int &lPrimID, V3 &emission, float &directPdf,
                         float &emissionPdf) {
    const DLight &L = S.lights[light];
    if (L.type == LIGHT_ENV) {
        const DEnv &E = S.env;
        V3 d = XformVector(E.toLight, dirToLight);
        float uvx = latan2f(d.x, -d.z) * c_INVTWOPI * (float)E.W - 0.5f;
        float uvy = lacosf(d.y) * c_INVPI * (float)E.H - 0.5f;
        int col = (int)floorf(uvx), row = (int)floorf(uvy);
        lPrimID = Moduloi(row, E.H) * E.W + Moduloi(col, E.W);
        float dx1 = uvx - col, dx2 = 1.0f - dx1, dy1 = uvy - row, dy2 = 1.0f - dy1;
        V3 e00 = EnvRepAt(E, col, row), e10 = EnvRepAt(E, col + 1, row), e01 = EnvRepAt(E, col, row + 1), e11 = EnvRepAt(E, col + 1, row + 1);
        float rowWeight0 = E.rowWeights[Clampi(row, 0, E.H - 1)];
        float rowWeight1 = E.rowWeights[Clampi(row + 1, 0, E.H - 1)];
        LMC_PIN4(e00.x, e00.y, e00.z, rowWeight0);  // one round of loads (see EnvSampleDirection)
        LMC_PIN4(e10.x, e10.y, e10.z, rowWeight1);
        LMC_PIN3(e01.x, e01.y, e01.z);
        LMC_PIN3(e11.x, e11.y, e11.z);
        V3 value1 = e00 * dx2 * dy2 + e10 * dx1 * dy2;
        V3 value2 = e01 * dx2 * dy1 + e11 * dx1 * dy1;
        emission = value1 + value2;
        float sinTheta = sqrtf(1.0f - square(d.y));
        directPdf = (Luminance(value1) * rowWeight0 + Luminance(value2) * rowWeight1) * E.normalization / fmaxf(fabsf(sinTheta), 1e-7f);
        float positionPdf = c_INVPI / square(S.bsRadius);
        emissionPdf = directPdf * positionPdf;
    } else {  // area
        float cosAtLight = -Dot(normalOnLight, dirToLight);
        if (cosAtLight > 0.0f) {
            emission = V3{L.radiance[0], L.radiance[1], L.radiance[2]};
            directPdf = S.meshes[L.mesh].invTotalArea;
            emissionPdf = cosAtLight * directPdf * c_INVPI;
        } else {
            emission = V3{0, 0, 0};
            directPdf = 0.0f;
            emissionPdf = 0.0f;
        }
    }
}

// Light::Emit (env: envlight.cpp:224-248; area: arealight.cpp:81-104; point: pointlight.cpp:57-72)
LMC_D void LightEmit(const DScene &S, int light, V2 rndPos, V2 rndDir, int &lPrimID, V3 &org, V3 &dir, V3 &emission, float &cosAtLight,
                     float &emissionPdf, float &directPdf) {
    const DLight &L = S.lights[light];
    if (L.type == LIGHT_ENV) {
        EnvSampleDirection(S, rndDir, lPrimID, dir, emission, directPdf);
        dir = -dir;
        V2 offset = SampleConcentricDisc(rndPos);
        V3 b0, b1;
        CoordinateSystem(dir, b0, b1);
        V3 perpOffset = offset.x * b0 + offset.y * b1;
        org = V3{S.bsCenter[0], S.bsCenter[1], S.bsCenter[2]} + (perpOffset - dir) * S.bsRadius;
        cosAtLight = 1.0f;
        float positionPdf = c_INVPI / square(S.bsRadius);
        emissionPdf = directPdf * positionPdf;
    } else if (L.type == LIGHT_AREA) {
        V3 normal;
        float shapePdf;
        SampleTriangle(S, S.meshes[L.mesh].triBase + lPrimID, rndPos, org, normal, shapePdf);
        V3 d = SampleCosHemisphere(rndDir);
        V3 b0, b1;
        CoordinateSystem(normal, b0, b1);
        dir = d.x * b0 + d.y * b1 + d.z * normal;
        emission = V3{L.radiance[0], L.radiance[1], L.radiance[2]} * (c_PI / shapePdf);
        cosAtLight = d.z;
        emissionPdf = d.z * c_INVPI * shapePdf;
        directPdf = shapePdf;
    } else {
        org = V3{L.pos[0], L.pos[1], L.pos[2]};
        float j;
        dir = SampleSphere(rndDir, j);
        emission = V3{L.intensity[0], L.intensity[1], L.intensity[2]};
        emissionPdf = c_INVFOURPI;
        cosAtLight = directPdf = 1.0f;
    }
}

// Light::SampleDiscrete: area lights pick a triangle by area (arealight.cpp:24-26), others INVALID (-1)
LMC_D int LightSampleDiscrete(const DScene &S, int light, float u) {
    const DLight &L = S.lights[light];
    if (L.type != LIGHT_AREA) return -1;
    const DMesh &M = S.meshes[L.mesh];
    return SampleDiscrete1D(S.areaFunc + M.areaOff, S.areaCdf + M.areaCdfOff, M.numTris, M.areaFuncInt, u, nullptr);
}
LMC_D bool LightIsDelta(const DScene &S, int light) { return S.lights[light].type == LIGHT_POINT; }
LMC_D bool LightIsFinite(const DScene &S, int light) { return S.lights[light].type != LIGHT_ENV; }

// ---------------------------------------------------------------------------------------------- camera
LMC_D void SamplePrimary(const DScene &S, V2 screenPos, V3 &org, V3 &dir) {  // camera.cpp:38-51
    V3 o = XformPoint(S.cam.sampleToCam, V3{screenPos.x, screenPos.y, 0.0f});
    V3 d = Normalize(o);
    org = XformPoint(S.cam.toWorld, V3{0, 0, 0});
    dir = XformVector(S.cam.toWorld, d);
}
LMC_D float PrimaryMinT(const DScene &S, V2 screenPos, float &maxT) {
    V3 o = XformPoint(S.cam.sampleToCam, V3{screenPos.x, screenPos.y, 0.0f});
    V3 d = Normalize(o);
    float invZ = inverse(d.z);
    maxT = S.cam.farClip * invZ;
    return S.cam.nearClip * invZ;
}
LMC_D bool ProjectPoint(const DScene &S, V3 p, V2 &screenPos) {  // camera.cpp:67-84
    V3 camP = XformPoint(S.cam.worldToCamera, p);
    if (camP.z < S.cam.nearClip || camP.z > S.cam.farClip) return false;
    V3 r = XformPoint(S.cam.camToSample, camP);
    if (r.x < 0.0f || r.x > 1.0f || r.y < 0.0f || r.y > 1.0f) return false;
    screenPos = V2{r.x, r.y};
    return true;
}

}  // namespace lmcd
