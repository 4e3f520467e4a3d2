// Direct-lighting pre-pass: DirectLighting() (/root/reference/src/direct.cpp:4-54) = GeneratePath
// (path.cpp:406-527) with minDepth = min(mindepth, 2), maxDepth = min(maxdepth, 2), over 16x16 pixel tiles, one RNG
// stream per tile seeded tileIndex + seedOffset.  Unidirectional helpers: HandleHitLight path.cpp:121-183,
// DirectLighting :195-294, BSDFSampling :296-386, RussianRoulette :388-404 (lens* / jacobian terms feed nothing here).
// The image the reference writes is direct / directSpp + indirect / spp (mlt.cpp:203-207).
#pragma once
#include "dpath.h"

namespace lmcd {

LMC_D float MISWeight(float pdfA, float pdfB) {  // path.cpp:23-27
    float ratioSq = square(pdfB / pdfA);
    return 1.0f / (1.0f + ratioSq);
}

// where a sample's contributions go (direct.cpp:42-45): straight to the film ...
struct FilmSink {
    Film film;
    LMC_D void operator()(V2 screenPos, V3 contrib) const { Splat(film, screenPos, contrib); }
};
// ... or into the lane's registers, to be committed in stream order by k_direct_wave.  With maxDepth <= 2 (the pre-pass) a
// sample makes at most two contributions (direct-light sample at the first hit, emitter found by the BSDF sample), both at
// the sample's own screen position.
struct LaneSink {
    V2 screenPos;
    V3 c[2];
    int n;
    LMC_D void operator()(V2 sp, V3 contrib) {
        screenPos = sp;
        if (n < 2) c[n] = contrib;
        n++;
    }
};

// pcg32_k64 read-only: a lane that evaluates a sample ahead of the committed stream position must not advance the stream's
// extension table; it reports the case instead (`crossed`, once per 2^32 draws) and counts its draws
struct SpecRng {
    uint64_t state;
    const uint32_t *tab;
    uint32_t draws;
    bool crossed;
    LMC_D uint32_t Next() {
        const uint64_t s = state;
        if ((s & 0xFFFFFFFFull) == 0ull) crossed = true;
        const uint32_t rhs = tab[(unsigned)(s & 63u)];
        state = s * PCG_MULT + PCG_INC;
        draws++;
        return PcgOutputXshRs(s) ^ rhs;
    }
    LMC_D float Uniform() {
        float r = (float)Next() * 2.3283064365386963e-10f;
        return r >= 1.0f ? 0.99999994f : r;
    }
    LMC_D void Uniform2(float &a, float &b) {
        a = Uniform();
        b = Uniform();
    }
};
// the LCG `delta` steps ahead (pcg_random.hpp:522-541, advance())
LMC_HD uint64_t PcgAdvance(uint64_t state, uint64_t delta) {
    uint64_t accMult = 1u, accPlus = 0u, curMult = PCG_MULT, curPlus = PCG_INC;
    while (delta > 0) {
        if (delta & 1) {
            accMult *= curMult;
            accPlus = accPlus * curMult + curPlus;
        }
        curPlus = (curMult + 1) * curPlus;
        curMult *= curMult;
        delta >>= 1;
    }
    return accMult * state + accPlus;
}

// one GeneratePath(scene, (x,y), minDepth, maxDepth) call
template <class Stk, class R, class Sink>
LMC_D void DirectSample(const DScene &S, Sink &sink, int px, int py, int minDepth, int maxDepth, R &rng, Stk &stk) {
    constexpr bool G = Stk::kGlossy;
    (void)rng.Uniform();  // time
    // Vector2(f(u), g(u)): gcc evaluates the second argument first
    const float uy = rng.Uniform(), ux = rng.Uniform();
    const V2 screenPos{(px + ux) / float(S.cam.width), (py + uy) / float(S.cam.height)};
    V3 org, dir;
    SamplePrimary(S, screenPos, org, dir);
    float tnear, tfar;
    tnear = PrimaryMinT(S, screenPos, tfar);
    V3 throughput{1, 1, 1};
    float lastBsdfPdf = 1.0f;
    for (int camDepth = 0;; camDepth++) {
        SurfHit hit;
        hit.tri = -1;
        hit.st = V2{0, 0};
        Isect isect;
        const bool hitSurface = IntersectSurface(S, org, dir, tnear, tfar, hit, isect, stk);
        const int light = HitLightOf(S, hitSurface, hit);
        if (light >= 0 && camDepth + 1 >= minDepth) {  // HandleHitLight (uni)
            int lPrimID = 0;
            V3 emission;
            float directPdf, emissionPdf;
            LightEmission(S, light, dir, isect.shadingNormal, lPrimID, emission, directPdf, emissionPdf);
            if (emission.x + emission.y + emission.z > 0.0f) {
                if (hitSurface) {
                    const float distSq = DistanceSquared(org, isect.position);
                    const float cosTheta = -Dot(dir, isect.shadingNormal);
                    directPdf *= (distSq / cosTheta);
                }
                V3 contrib = cmul(throughput, emission);
                if (camDepth > 0) {
                    const float lightPickProb = PickLightProb(S, light);
                    contrib = contrib * MISWeight(lastBsdfPdf, directPdf * lightPickProb);
                }
                if (Luminance(contrib) > 0.0f) sink(screenPos, contrib);
            }
            return;
        }
        if (!hitSurface || (maxDepth != -1 && camDepth + 1 >= maxDepth)) break;
        const float bsdfDiscrete = rng.Uniform();
        const V3 wi = -dir;
        const DMaterial &m = MaterialOfTri(S, hit.tri);
        if (camDepth + 2 >= minDepth) {
            float lightPickProb = 1.0f;  // DirectLightingInit, path.cpp:184-193
            const int dl = PickLight(S, rng.Uniform(), lightPickProb);
            const V2 r = RndVec2(rng);
            int lPrimID = LightSampleDiscrete(S, dl, rng.Uniform());
            V3 dirToLight, lightContrib;
            float dist, cosAtLight, directPdf, emissionPdf;
            if (LightSampleDirect(S, dl, isect.position, r, lPrimID, dirToLight, dist, lightContrib, cosAtLight, directPdf, emissionPdf) &&
                !Occluded(S, isect.position, dirToLight, dist, stk)) {
                V3 bsdfContrib;
                float cosWo, bsdfPdf, bsdfRevPdf;
                BsdfEvaluate<G>(S, m, false, wi, isect.shadingNormal, dirToLight, hit.st, bsdfContrib, cosWo, bsdfPdf, bsdfRevPdf);
                if (!IsZero(bsdfContrib)) {
                    V3 contrib = cmul(throughput, bsdfContrib);
                    contrib = cmul(contrib, lightContrib) * inverse(lightPickProb);
                    if (!LightIsDelta(S, dl)) contrib = contrib * MISWeight(directPdf * lightPickProb, bsdfPdf);
                    if (Luminance(contrib) > 0.0f) sink(screenPos, contrib);
                }
            }
        }
        const V2 rnd = RndVec2(rng);
        V3 bsdfContrib;
        float cosWo, bsdfPdfRev;
        if (!BsdfSample<G>(S, m, false, wi, isect.shadingNormal, hit.st, rnd, bsdfDiscrete, dir, bsdfContrib, cosWo, lastBsdfPdf, bsdfPdfRev)) break;
        throughput = cmul(throughput, bsdfContrib);
        org = isect.position;
        float rrWeight;
        if (!RussianRoulette(camDepth, bsdfContrib, rrWeight, throughput, rng)) break;
        tnear = c_IsectEpsilon;
        tfar = INFINITY;
    }
}

}  // namespace lmcd
