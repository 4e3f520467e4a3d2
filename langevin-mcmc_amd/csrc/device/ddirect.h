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

// one GeneratePath(scene, (x,y), minDepth, maxDepth) call; contributions go straight to the film (direct.cpp:42-45)
template <class Stk>
LMC_D void DirectSample(const DScene &S, const Film &film, int px, int py, int minDepth, int maxDepth, Rng &rng, Stk &stk) {
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
        const int light = HitLightOf(S, hitSurface, hit.tri);
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
                if (Luminance(contrib) > 0.0f) Splat(film, screenPos, contrib);
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
                    if (Luminance(contrib) > 0.0f) Splat(film, screenPos, contrib);
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
