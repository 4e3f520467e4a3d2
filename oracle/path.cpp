// ORACLE -- TEST INFRASTRUCTURE ONLY (see common.h).
// Bidirectional path sampling / perturbation / serialisation: restates the scalar half of
// /root/reference/src/path.cpp (line numbers cited per function).  RNG consumption order is the
// reference's; every early-out is kept.
#include "render.h"

namespace orc {

void Clear(Path &path) {  // path.cpp:16-21
    path.camSurfaceVertex.clear();
    path.lgtSurfaceVertex.clear();
    path.envLightInst.light = nullptr;
    path.isSubpath = false;
}

static inline Float MIS(const Float pdf) { return square(pdf); }  // path.cpp:29-32

template <bool adjoint>
static Float ShadingNormalCorrection(const Vector3 &wi, const Intersection &isect, const Vector3 &wo) {  // path.cpp:34-54
    const Float cosWi = Dot(isect.shadingNormal, wi);
    const Float cosWo = Dot(isect.shadingNormal, wo);
    Float wiDotGeoN = Dot(isect.geomNormal, wi);
    Float woDotGeoN = Dot(isect.geomNormal, wo);
    if (wiDotGeoN * cosWi <= Float(0.0) || woDotGeoN * cosWo <= Float(0.0)) return Float(0.0);
    if (adjoint) return std::fabs((woDotGeoN * cosWi) / (wiDotGeoN * cosWo));
    return Float(1.0);
}

static inline bool Intersect(const RScene *scene, const Float time, const RaySegment &raySeg, ShapeInst &shapeInst, Intersection &isect) {
    bool hit = false;  // path.cpp:91-103
    if (Intersect(scene, time, raySeg, shapeInst)) {
        if (shapeInst.obj->Intersect(shapeInst.primID, time, raySeg, isect, shapeInst.st)) hit = true;
    }
    return hit;
}

static inline const Light *GetHitLight(const RScene *scene, const bool hitSurface, const Shape *shape) {  // path.cpp:105-120
    const Light *light = nullptr;
    if (!hitSurface) {
        if (scene->envLight != nullptr) light = scene->envLight;
    }
    if (hitSurface) {
        if (shape->areaLight != nullptr) light = shape->areaLight;
    }
    return light;
}

// Evaluation order of `Vector2(uniDist(rng), uniDist(rng))`: unspecified by C++; the reference was built
// with gcc 9 (README.md:24), which evaluates function/constructor arguments right to left on x86-64.
// The oracle therefore assigns the FIRST draw to component [1] and the second to [0].  This matters only
// for bit-level trajectory parity with a reference build, which cannot be produced here (parity unpinned);
// it is a compile-time switch so both conventions can be tested.
#ifndef ORC_ARGS_LEFT_TO_RIGHT
#define ORC_ARGS_LEFT_TO_RIGHT 0
#endif
static inline Vector2 RndVec2(std::uniform_real_distribution<Float> &uniDist, RNG &rng) {
    Float first = uniDist(rng);
    Float second = uniDist(rng);
#if ORC_ARGS_LEFT_TO_RIGHT
    return Vector2(first, second);
#else
    return Vector2(second, first);
#endif
}

struct BidirPathState {  // path.cpp:529-540
    Intersection isect;
    Vector3 wi;
    Float accMISWPrev;
    Float accMISWThis;
    Vector3 throughput;
    Vector3 lensContrib;
    Float ssJacobian;
    Float lcJacobian;
    Float lastBsdfPdf;
};

static inline void EmitFromCameraInit(const RCamera *camera, const int sx, const int sy, CameraVertex &camVertex, RNG &rng) {  // path.cpp:542-552
    std::uniform_real_distribution<Float> uniDist(Float(0.0), Float(1.0));
    // Vector2(exprX, exprY) with one draw in each argument
    Float first = uniDist(rng);
    Float second = uniDist(rng);
#if ORC_ARGS_LEFT_TO_RIGHT
    Float ux = first, uy = second;
#else
    Float ux = second, uy = first;
#endif
    camVertex.screenPos = Vector2(sx == -1 ? ux : ((sx + ux) / Float(camera->pixelWidth)), sy == -1 ? uy : ((sy + uy) / Float(camera->pixelHeight)));
}

static void EmitFromCamera(const Float time, const RCamera *camera, const CameraVertex &camVertex, RaySegment &raySeg, BidirPathState &pathState) {
    RaySegment centerRaySeg;  // path.cpp:554-574
    SamplePrimary(camera, Vector2(Float(0.5), Float(0.5)), time, centerRaySeg);
    SamplePrimary(camera, camVertex.screenPos, time, raySeg);
    const Vector3 camDir = centerRaySeg.ray.dir;
    const Vector3 dir = raySeg.ray.dir;
    const Float cosAtCamera = Dot(camDir, dir);
    const Float imagePointToCameraDist = camera->dist / cosAtCamera;
    const Float imageToSolidAngleFactor = square(imagePointToCameraDist) / cosAtCamera;
    const Float cameraPdfW = imageToSolidAngleFactor;
    const Float screenPixelCount = Float(camera->pixelWidth * camera->pixelHeight);
    pathState.throughput = Vector3(Float(1.0), Float(1.0), Float(1.0));
    pathState.accMISWPrev = MIS(screenPixelCount / cameraPdfW);
    pathState.accMISWThis = Float(0.0);
    pathState.ssJacobian = Float(1.0);
    pathState.lensContrib = Vector3(Float(1.0), Float(1.0), Float(1.0));
}

static inline void EmitFromLightInit(const RScene *scene, LightVertex &lgtVertex, Float &lightPickProb, RNG &rng) {  // path.cpp:576-586
    std::uniform_real_distribution<Float> uniDist(Float(0.0), Float(1.0));
    lgtVertex.rndParamPos = RndVec2(uniDist, rng);
    lgtVertex.rndParamDir = RndVec2(uniDist, rng);
    const Light *light = PickLight(scene, uniDist(rng), lightPickProb);
    lgtVertex.lightInst.light = light;
    lgtVertex.lightInst.lPrimID = light->SampleDiscrete(uniDist(rng));
}

static void EmitFromLight(const BSphere &bSphere, const Float lightPickProb, const Float time, LightVertex &lgtVertex, Ray &ray,
                          BidirPathState &pathState) {  // path.cpp:588-618
    const Light *light = lgtVertex.lightInst.light;
    Float cosLight, emissionPdf, directPdf;
    light->Emit(bSphere, lgtVertex.rndParamPos, lgtVertex.rndParamDir, time, lgtVertex.lightInst.lPrimID, ray, pathState.throughput, cosLight,
                emissionPdf, directPdf);
    emissionPdf *= lightPickProb;
    directPdf *= lightPickProb;
    pathState.throughput *= inverse(lightPickProb);
    pathState.accMISWPrev = MIS(directPdf / emissionPdf);
    if (!light->IsDelta())
        pathState.accMISWThis = MIS(cosLight / emissionPdf);
    else
        pathState.accMISWThis = Float(0.0);
    pathState.ssJacobian = Float(1.0);
}

static inline void ConvertMIS(const int depth, const Light *light, const Ray &ray, BidirPathState &pathState) {  // path.cpp:620-631
    if (depth > 0 || (light == nullptr) || (light != nullptr && light->IsFinite())) {
        pathState.accMISWPrev *= MIS(DistanceSquared(ray.org, pathState.isect.position));
    }
    Float invCosTheta = inverse(MIS(std::fabs(Dot(ray.dir, pathState.isect.shadingNormal))));
    pathState.accMISWPrev *= invCosTheta;
    pathState.accMISWThis *= invCosTheta;
}

static void ConnectToCamera(const int lgtDepth, const RScene *scene, const RCamera *camera, const Float time, const BidirPathState &pathState,
                            const SurfaceVertex &lgtVertex, const Vector3 &prevLensContrib, const Vector3 &prevPosition,
                            std::vector<SubpathContrib> &contribs) {  // path.cpp:633-745
    RaySegment centerRaySeg;
    SamplePrimary(camera, Vector2(Float(0.5), Float(0.5)), time, centerRaySeg);
    const Vector3 camOrg = centerRaySeg.ray.org;
    const Vector3 camDir = centerRaySeg.ray.dir;
    Vector3 dirToCamera = camOrg - pathState.isect.position;
    if (-Dot(camDir, dirToCamera) <= Float(0.0)) return;
    Vector2 screenPos;
    if (!ProjectPoint(camera, pathState.isect.position, time, screenPos)) return;
    const Float distSq = LengthSquared(dirToCamera);
    const Float dist = std::sqrt(distSq);
    dirToCamera *= inverse(dist);
    if (Occluded(scene, time, Ray{pathState.isect.position, dirToCamera}, dist)) return;
    const BSDF *bsdf = lgtVertex.shapeInst.obj->bsdf;
    Vector3 bsdfContrib;
    Float cosToCamera, bsdfPdf, bsdfRevPdf;
    bsdf->EvaluateAdjoint(pathState.wi, pathState.isect.shadingNormal, dirToCamera, lgtVertex.shapeInst.st, bsdfContrib, cosToCamera, bsdfPdf,
                          bsdfRevPdf);
    if (bsdfContrib.isZero()) return;
    const Float factor = ShadingNormalCorrection<true>(pathState.wi, pathState.isect, dirToCamera);
    if (factor <= Float(0.0)) return;
    bsdfContrib *= factor;
    bool useAbsoluteParam = bsdf->Roughness(lgtVertex.shapeInst.st, lgtVertex.bsdfDiscrete) > scene->options->roughnessThreshold;
    Vector3 lensContrib = Vector3::Zero();
    if (useAbsoluteParam && lgtDepth >= 1) {
        Vector3 bsdfContrib2;
        Float cosWo, bsdfPdf2, bsdfRevPdf2;
        bsdf->Evaluate(dirToCamera, pathState.isect.shadingNormal, pathState.wi, lgtVertex.shapeInst.st, bsdfContrib2, cosWo, bsdfPdf2, bsdfRevPdf2);
        const Float distSq2 = DistanceSquared(pathState.isect.position, prevPosition);
        if (distSq2 <= Float(0.0)) {
            contribs.clear();
            return;
        } else {
            lensContrib = bsdfContrib2.cwiseProduct(prevLensContrib) * inverse(distSq2);
        }
    }
    const Float cosAtCamera = -Dot(camDir, dirToCamera);
    const Float imagePointToCameraDist = camera->dist / cosAtCamera;
    const Float imageToSolidAngleFactor = square(imagePointToCameraDist) / cosAtCamera;
    const Float imageToSurfaceFactor = imageToSolidAngleFactor * std::fabs(cosToCamera) / distSq;
    const Float screenPixelCount = Float(camera->pixelWidth * camera->pixelHeight);
    const Float cameraPdf = imageToSurfaceFactor;
    const Float wLight = MIS(cameraPdf / screenPixelCount) * (pathState.accMISWPrev + pathState.accMISWThis * MIS(bsdfRevPdf));
    const Float misWeight = inverse(wLight + Float(1.0));
    const Float surfaceToImageFactor = cosToCamera / imageToSurfaceFactor;
    Vector3 contrib = misWeight * bsdfContrib / (screenPixelCount * surfaceToImageFactor);
    contrib = contrib.cwiseProduct(pathState.throughput);
    const Float score = Luminance(contrib);
    if (score > Float(0.0)) {
        const Float lensScore = Luminance(lensContrib);
        contribs.emplace_back(SubpathContrib{1, 2 + lgtDepth, screenPos, contrib, score, score * pathState.ssJacobian, lensScore, misWeight});
    }
}

template <bool adjoint, bool perturb>
static bool BSDFSampling(const Float roughnessThreshold, const int depth, const BidirPathState &pathState, SurfaceVertex &surfVertex,
                         BidirPathState &nextPathState, Vector3 &dir, Vector3 &bsdfContrib) {  // path.cpp:747-900
    // NB: the reference calls this with nextPathState aliasing pathState in the camera loop and in
    // PerturbPathBidir; every read of pathState below happens before the aliased field is written,
    // except where noted -- the statement order is the reference's.
    const BSDF *bsdf = surfVertex.shapeInst.obj->bsdf;
    Float cosWo, bsdfPdf, bsdfRevPdf;
    surfVertex.useAbsoluteParam = (bsdf->Roughness(surfVertex.shapeInst.st, surfVertex.bsdfDiscrete) > roughnessThreshold) ? FTRUE : FFALSE;
    if (!perturb || surfVertex.useAbsoluteParam == FFALSE) {
        if (adjoint) {
            if (!bsdf->SampleAdjoint(pathState.wi, pathState.isect.shadingNormal, surfVertex.shapeInst.st, surfVertex.bsdfRndParam,
                                     surfVertex.bsdfDiscrete, dir, bsdfContrib, cosWo, bsdfPdf, bsdfRevPdf))
                return false;
        } else {
            if (!bsdf->Sample(pathState.wi, pathState.isect.shadingNormal, surfVertex.shapeInst.st, surfVertex.bsdfRndParam, surfVertex.bsdfDiscrete,
                              dir, bsdfContrib, cosWo, bsdfPdf, bsdfRevPdf))
                return false;
        }
        if (surfVertex.useAbsoluteParam == FTRUE) {
            Float jacobian;
            surfVertex.bsdfRndParam = ToSphericalCoord(dir, jacobian);
            nextPathState.lcJacobian = inverse(jacobian);
            jacobian *= bsdfPdf;
            nextPathState.ssJacobian = pathState.ssJacobian * jacobian;
        } else {
            nextPathState.lcJacobian = bsdfPdf;
        }
    } else {
        Float jacobian;
        dir = SampleSphere(surfVertex.bsdfRndParam, jacobian);
        if (adjoint)
            bsdf->EvaluateAdjoint(pathState.wi, pathState.isect.shadingNormal, dir, surfVertex.shapeInst.st, bsdfContrib, cosWo, bsdfPdf, bsdfRevPdf);
        else
            bsdf->Evaluate(pathState.wi, pathState.isect.shadingNormal, dir, surfVertex.shapeInst.st, bsdfContrib, cosWo, bsdfPdf, bsdfRevPdf);
        if (bsdfContrib.isZero() || bsdfPdf <= Float(0.0)) return false;
        bsdfContrib *= inverse(bsdfPdf);
        nextPathState.lcJacobian = inverse(jacobian);
        jacobian *= bsdfPdf;
        nextPathState.ssJacobian = pathState.ssJacobian * jacobian;
    }
    Float factor = ShadingNormalCorrection<adjoint>(pathState.wi, pathState.isect, dir);
    if (factor <= Float(0.0)) return false;
    if (surfVertex.useAbsoluteParam == FTRUE) {
        if (adjoint) {
            if (std::fabs(cosWo) < Float(1e-2) || std::fabs(Dot(dir, pathState.isect.geomNormal)) < Float(1e-2))
                nextPathState.lensContrib = Vector3::Zero();
            else
                nextPathState.lensContrib = bsdfContrib * bsdfPdf;
        } else if (depth == 0) {
            if (std::fabs(cosWo) < Float(1e-2) || std::fabs(Dot(dir, pathState.isect.geomNormal)) < Float(1e-2))
                nextPathState.lensContrib = Vector3::Zero();
            else
                nextPathState.lensContrib = pathState.lensContrib.cwiseProduct(bsdfContrib * bsdfPdf);
        } else if (depth == 1) {
            Vector3 impBsdfContrib;
            Float impCosWo, impBsdfPdf, impBsdfPdfRev;
            bsdf->EvaluateAdjoint(dir, pathState.isect.shadingNormal, pathState.wi, surfVertex.shapeInst.st, impBsdfContrib, impCosWo, impBsdfPdf,
                                  impBsdfPdfRev);
            if (std::fabs(impCosWo) < Float(1e-2) || std::fabs(Dot(pathState.wi, pathState.isect.geomNormal)) < Float(1e-2)) {
                nextPathState.lensContrib = Vector3::Zero();
            } else {
                const Float factor2 = ShadingNormalCorrection<true>(dir, pathState.isect, pathState.wi);
                impBsdfContrib *= factor2;
                nextPathState.lensContrib = pathState.lensContrib.cwiseProduct(impBsdfContrib);
            }
        }
    } else {
        if (adjoint || depth <= 1) nextPathState.lensContrib = Vector3::Zero();
    }
    bsdfContrib *= factor;
    if (adjoint) nextPathState.lensContrib *= factor;
    nextPathState.lastBsdfPdf = bsdfPdf;
    nextPathState.accMISWThis = MIS(cosWo / bsdfPdf) * (pathState.accMISWThis * MIS(bsdfRevPdf) + pathState.accMISWPrev);
    nextPathState.accMISWPrev = MIS(inverse(bsdfPdf));
    nextPathState.throughput = pathState.throughput.cwiseProduct(bsdfContrib);
    return true;
}

static void HandleHitLight(const int camDepth, const RScene *scene, const Light *light, const bool hitSurface, const Ray &ray, const Float time,
                           const Vector2 screenPos, const BidirPathState &pathState, const bool bidirMIS, LightInst &envLightInst,
                           std::vector<SubpathContrib> &contribs) {  // path.cpp:902-967
    LightPrimID lPrimID = 0;
    Vector3 emission;
    Float directPdf, emissionPdf;
    light->Emission(scene->bSphere, ray.dir, pathState.isect.shadingNormal, time, lPrimID, emission, directPdf, emissionPdf);
    if (emission.sum() > Float(0.0)) {
        Vector3 contrib = pathState.throughput.cwiseProduct(emission);
        Float misWeight = Float(1.0);
        if (camDepth > 0) {
            Float lightPickProb = PickLightProb(scene, light);
            directPdf *= lightPickProb;
            if (bidirMIS) {
                emissionPdf *= lightPickProb;
                Float wCamera = MIS(directPdf) * pathState.accMISWPrev + MIS(emissionPdf) * pathState.accMISWThis;
                misWeight = inverse(Float(1.0) + wCamera);
            } else {
                if (hitSurface) {
                    Float distSq = DistanceSquared(ray.org, pathState.isect.position);
                    Float cosTheta = -Dot(ray.dir, pathState.isect.shadingNormal);
                    directPdf *= (distSq / cosTheta);
                }
                Float ratioSq = square(directPdf / pathState.lastBsdfPdf);
                misWeight = Float(1.0) / (Float(1.0) + ratioSq);
            }
            contrib *= misWeight;
        }
        Float score = Luminance(contrib);
        if (score > Float(0.0)) {
            if (!hitSurface) envLightInst = LightInst{light, lPrimID};
            const Float lensScore = camDepth >= 2 ? Luminance(pathState.lensContrib) : Float(0.0);
            contribs.emplace_back(SubpathContrib{2 + camDepth, 0, screenPos, contrib, score, score * pathState.ssJacobian, lensScore, misWeight});
        }
    }
}

static void DirectLighting(const int camDepth, const RScene *scene, const Float time, const BidirPathState &pathState, const Vector2 screenPos,
                           const Float lightPickProb, SurfaceVertex &camVertex, const bool doOcclusion, const bool bidirMIS,
                           std::vector<SubpathContrib> &contribs) {  // path.cpp:969-1089
    const BSDF *bsdf = camVertex.shapeInst.obj->bsdf;
    LightInst &dirLightInst = camVertex.directLightInst;
    const Light *light = dirLightInst.light;
    LightPrimID &lPrimID = dirLightInst.lPrimID;
    Vector3 dirToLight, lightContrib;
    Float dist, cosAtLight, directPdf, emissionPdf;
    if (!light->SampleDirect(scene->bSphere, pathState.isect.position, pathState.isect.shadingNormal, camVertex.directLightRndParam, time, lPrimID,
                             dirToLight, dist, lightContrib, cosAtLight, directPdf, emissionPdf))
        return;
    if (doOcclusion && Occluded(scene, time, Ray{pathState.isect.position, dirToLight}, dist)) return;
    Vector3 bsdfContrib;
    Float cosToLight, bsdfPdf, bsdfRevPdf;
    bsdf->Evaluate(pathState.wi, pathState.isect.shadingNormal, dirToLight, camVertex.shapeInst.st, bsdfContrib, cosToLight, bsdfPdf, bsdfRevPdf);
    if (bsdfContrib.isZero()) return;
    const Float factor = ShadingNormalCorrection<false>(pathState.wi, pathState.isect, dirToLight);
    if (factor <= Float(0.0)) return;
    bsdfContrib *= factor;
    Vector3 lensContrib = pathState.lensContrib;
    if (camDepth == 1) {
        bool useAbsoluteParam = bsdf->Roughness(camVertex.shapeInst.st, camVertex.bsdfDiscrete) > scene->options->roughnessThreshold;
        if (useAbsoluteParam) {
            const Intersection &isect = pathState.isect;
            Vector3 impBsdfContrib;
            Float impCosWo, impBsdfPdf, impBsdfRevPdf;
            bsdf->EvaluateAdjoint(dirToLight, isect.shadingNormal, pathState.wi, camVertex.shapeInst.st, impBsdfContrib, impCosWo, impBsdfPdf,
                                  impBsdfRevPdf);
            const Float factor2 = ShadingNormalCorrection<true>(dirToLight, pathState.isect, pathState.wi);
            impBsdfContrib *= factor2;
            lensContrib = lensContrib.cwiseProduct(impBsdfContrib);
        } else {
            lensContrib = Vector3::Zero();
        }
    }
    Vector3 contrib = pathState.throughput.cwiseProduct(bsdfContrib);
    contrib = contrib.cwiseProduct(lightContrib) * inverse(lightPickProb);
    Float misWeight = Float(1.0);
    if (bidirMIS) {
        Float wLight = light->IsDelta() ? Float(0.0) : MIS(bsdfPdf / (lightPickProb * directPdf));
        Float wCamera = MIS(emissionPdf * cosToLight / (directPdf * cosAtLight)) * (pathState.accMISWPrev + pathState.accMISWThis * MIS(bsdfRevPdf));
        misWeight = inverse(wLight + Float(1.0) + wCamera);
        contrib *= misWeight;
    } else if (!light->IsDelta()) {
        Float ratioSq = square(bsdfPdf / (directPdf * lightPickProb));
        misWeight = Float(1.0) / (Float(1.0) + ratioSq);
        contrib *= misWeight;
    }
    const Float score = Luminance(contrib);
    if (score > Float(0.0)) {
        const Float lensScore = camDepth >= 1 ? Luminance(lensContrib) : Float(0.0);
        contribs.emplace_back(SubpathContrib{2 + camDepth, 1, screenPos, contrib, score, score * pathState.ssJacobian, lensScore, misWeight});
    }
}

static void ConnectVertex(const int camDepth, const int lgtDepth, const RScene *scene, const Float time, const BidirPathState &lgtPathState,
                          const SurfaceVertex &lgtVertex, const BidirPathState &camPathState, const SurfaceVertex &camVertex, const Vector2 screenPos,
                          const bool doOcclusion, std::vector<SubpathContrib> &contribs) {  // path.cpp:1091-1235
    Vector3 dirToLight = lgtPathState.isect.position - camPathState.isect.position;
    const Float distSq = LengthSquared(dirToLight);
    const Float dist = std::sqrt(distSq);
    dirToLight *= inverse(dist);
    if (doOcclusion && Occluded(scene, time, Ray{camPathState.isect.position, dirToLight}, dist)) return;
    Vector3 camBsdfFactor;
    Float cosCamera, camBsdfPdf, camBsdfRevPdf;
    const BSDF *camBSDF = camVertex.shapeInst.obj->bsdf;
    camBSDF->Evaluate(camPathState.wi, camPathState.isect.shadingNormal, dirToLight, camVertex.shapeInst.st, camBsdfFactor, cosCamera, camBsdfPdf,
                      camBsdfRevPdf);
    if (camBsdfFactor.isZero()) return;
    Float camFactor = ShadingNormalCorrection<false>(camPathState.wi, camPathState.isect, dirToLight);
    if (camFactor <= Float(0.0)) return;
    camBsdfFactor *= camFactor;
    Vector3 lgtBsdfFactor;
    Float cosLight, lgtBsdfPdf, lgtBsdfRevPdf;
    const BSDF *lgtBSDF = lgtVertex.shapeInst.obj->bsdf;
    lgtBSDF->EvaluateAdjoint(lgtPathState.wi, lgtPathState.isect.shadingNormal, -dirToLight, lgtVertex.shapeInst.st, lgtBsdfFactor, cosLight,
                             lgtBsdfPdf, lgtBsdfRevPdf);
    if (lgtBsdfFactor.isZero()) return;
    Float lgtFactor = ShadingNormalCorrection<true>(lgtPathState.wi, lgtPathState.isect, -dirToLight);
    if (lgtFactor <= Float(0.0)) return;
    lgtBsdfFactor *= lgtFactor;
    const Float geometryTerm = inverse(distSq);
    Vector3 lensContrib = camPathState.lensContrib;
    if (camDepth == 0) {
        bool camAbs = camBSDF->Roughness(camVertex.shapeInst.st, camVertex.bsdfDiscrete) > scene->options->roughnessThreshold;
        bool lgtAbs = lgtBSDF->Roughness(lgtVertex.shapeInst.st, lgtVertex.bsdfDiscrete) > scene->options->roughnessThreshold;
        if (camAbs && lgtAbs)
            lensContrib = lgtBsdfFactor.cwiseProduct(camBsdfFactor) * geometryTerm;
        else
            lensContrib = Vector3::Zero();
    } else if (camDepth == 1) {
        bool camAbs = camBSDF->Roughness(camVertex.shapeInst.st, camVertex.bsdfDiscrete) > scene->options->roughnessThreshold;
        if (camAbs) {
            const Intersection &isect = camPathState.isect;
            Vector3 impBsdfContrib;
            Float impCosWo, impBsdfPdf, impBsdfRevPdf;
            camBSDF->EvaluateAdjoint(dirToLight, isect.shadingNormal, camPathState.wi, camVertex.shapeInst.st, impBsdfContrib, impCosWo, impBsdfPdf,
                                     impBsdfRevPdf);
            const Float factor = ShadingNormalCorrection<true>(dirToLight, isect, camPathState.wi);
            impBsdfContrib *= factor;
            lensContrib = lensContrib.cwiseProduct(impBsdfContrib);
        } else {
            lensContrib = Vector3::Zero();
        }
    }
    const Float camBsdfDirPdfA = camBsdfPdf * cosLight * geometryTerm;
    const Float lgtBsdfDirPdfA = lgtBsdfPdf * cosCamera * geometryTerm;
    const Float wLight = MIS(camBsdfDirPdfA) * (lgtPathState.accMISWPrev + lgtPathState.accMISWThis * MIS(lgtBsdfRevPdf));
    const Float wCamera = MIS(lgtBsdfDirPdfA) * (camPathState.accMISWPrev + camPathState.accMISWThis * MIS(camBsdfRevPdf));
    const Float misWeight = inverse(wLight + Float(1.0) + wCamera);
    const Vector3 throughput = lgtPathState.throughput.cwiseProduct(camPathState.throughput);
    Vector3 contrib = throughput.cwiseProduct(camBsdfFactor);
    contrib = contrib.cwiseProduct(lgtBsdfFactor) * geometryTerm;
    contrib *= misWeight;
    const Float ssJacobian = lgtPathState.ssJacobian * camPathState.ssJacobian;
    const Float score = Luminance(contrib);
    if (score > Float(0.0)) {
        const Float lensScore = Luminance(lensContrib);
        contribs.emplace_back(SubpathContrib{2 + camDepth, 2 + lgtDepth, screenPos, contrib, score, score * ssJacobian, lensScore, misWeight});
    }
}

static bool RussianRoulette(const int depth, const Vector3 &bsdfContrib, Float &rrWeight, Vector3 &throughput, RNG &rng) {  // path.cpp:388-404
    std::uniform_real_distribution<Float> uniDist(Float(0.0), Float(1.0));
    Float rrProb = Float(1.0);
    if (depth >= 3) rrProb = std::min(bsdfContrib.maxCoeff(), Float(0.95));
    if (uniDist(rng) > rrProb) return false;
    rrWeight = inverse(rrProb);
    throughput *= rrWeight;
    return true;
}

void GeneratePathBidir(const RScene *scene, const int sx, const int sy, const int minDepth, const int maxDepth, Path &path,
                       std::vector<SubpathContrib> &contribs, RNG &rng) {  // path.cpp:1237-1449
    std::uniform_real_distribution<Float> uniDist(Float(0.0), Float(1.0));
    const RCamera *camera = &scene->camera;
    path.time = uniDist(rng);
    std::vector<BidirPathState> lightPathStates;
    lightPathStates.push_back(BidirPathState());
    Float lightPickProb = Float(1.0);
    EmitFromLightInit(scene, path.lgtVertex, lightPickProb, rng);
    RaySegment raySeg;
    EmitFromLight(scene->bSphere, lightPickProb, path.time, path.lgtVertex, raySeg.ray, lightPathStates[0]);
    raySeg.minT = c_IsectEpsilon;
    raySeg.maxT = std::numeric_limits<Float>::infinity();
    Vector3 prevLensContrib = Vector3::Zero();
    for (int lgtDepth = 0;; lgtDepth++) {
        path.lgtSurfaceVertex.push_back(SurfaceVertex());
        bool hitSurface = Intersect(scene, path.time, raySeg, path.lgtSurfaceVertex.back().shapeInst, lightPathStates[lgtDepth].isect);
        if (!hitSurface) {
            lightPathStates.pop_back();
            path.lgtSurfaceVertex.pop_back();
            break;
        }
        path.lgtSurfaceVertex.back().bsdfDiscrete = uniDist(rng);
        lightPathStates[lgtDepth].wi = -raySeg.ray.dir;
        ConvertMIS(lgtDepth, path.lgtVertex.lightInst.light, raySeg.ray, lightPathStates[lgtDepth]);
        if (lgtDepth + 2 >= minDepth) {
            ConnectToCamera(lgtDepth, scene, camera, path.time, lightPathStates[lgtDepth], path.lgtSurfaceVertex[lgtDepth], prevLensContrib,
                            raySeg.ray.org, contribs);
        }
        if (maxDepth != -1 && lgtDepth + 2 >= maxDepth) break;
        lightPathStates.push_back(BidirPathState());
        SurfaceVertex &surfVertex = path.lgtSurfaceVertex.back();
        surfVertex.bsdfRndParam = RndVec2(uniDist, rng);
        Vector3 bsdfContrib;
        if (!BSDFSampling<true, false>(scene->options->roughnessThreshold, lgtDepth, lightPathStates[lgtDepth], surfVertex,
                                       lightPathStates[lgtDepth + 1], raySeg.ray.dir, bsdfContrib)) {
            lightPathStates.pop_back();
            break;
        }
        if (!RussianRoulette(lgtDepth, bsdfContrib, surfVertex.rrWeight, lightPathStates[lgtDepth + 1].throughput, rng)) {
            lightPathStates.pop_back();
            break;
        }
        prevLensContrib = lightPathStates[lgtDepth + 1].lensContrib;
        raySeg.ray.org = lightPathStates[lgtDepth].isect.position;
    }

    BidirPathState camPathState;
    EmitFromCameraInit(camera, sx, sy, path.camVertex, rng);
    EmitFromCamera(path.time, camera, path.camVertex, raySeg, camPathState);
    for (int camDepth = 0;; camDepth++) {
        path.camSurfaceVertex.push_back(SurfaceVertex());
        SurfaceVertex &surfVertex = path.camSurfaceVertex.back();
        bool hitSurface = Intersect(scene, path.time, raySeg, surfVertex.shapeInst, camPathState.isect);
        camPathState.wi = -raySeg.ray.dir;
        if (hitSurface) ConvertMIS(camDepth, nullptr, raySeg.ray, camPathState);
        if (camDepth + 1 >= minDepth) {
            const Light *light = GetHitLight(scene, hitSurface, surfVertex.shapeInst.obj);
            if (light != nullptr) {
                if (scene->options->useLightCoordinateSampling && camDepth > 1 && light->GetType() == lmc::LIGHT_AREA) {  // path.cpp:1339-1360
                    // area light: the BSDF sampling coordinates of the previous vertex become the light's direct sampling coordinates
                    SurfaceVertex &prevSurfVertex = path.camSurfaceVertex[path.camSurfaceVertex.size() - 2];
                    const ShapeInst &shapeInst = surfVertex.shapeInst;
                    prevSurfVertex.bsdfRndParam = shapeInst.obj->GetSampleParam(shapeInst.primID, camPathState.isect.position, path.time);
                    Vector3 dirToPrev = camPathState.isect.position - raySeg.ray.org;
                    const Float distSq = LengthSquared(dirToPrev);
                    const Float invDistSq = inverse(distSq);
                    const Float invDist = std::sqrt(invDistSq);
                    dirToPrev *= invDist;
                    camPathState.ssJacobian *= std::fabs(Dot(dirToPrev, camPathState.isect.shadingNormal) * invDistSq) *
                                               (camPathState.lcJacobian * surfVertex.shapeInst.obj->SamplePdf());
                }
                HandleHitLight(camDepth, scene, light, hitSurface, raySeg.ray, path.time, path.camVertex.screenPos, camPathState, true,
                               path.envLightInst, contribs);
                return;
            }
        }
        if (!hitSurface || (maxDepth != -1 && camDepth + 1 >= maxDepth)) break;
        if (camDepth == 1) {
            path.lensVertexPos = camPathState.isect.position;
            const Float distSq = DistanceSquared(camPathState.isect.position, raySeg.ray.org);
            if (distSq <= Float(0.0)) {
                contribs.clear();
                return;
            } else {
                camPathState.lensContrib *= inverse(distSq);
            }
        }
        surfVertex.bsdfDiscrete = uniDist(rng);
        if (camDepth + 2 >= minDepth) {
            Float directLightPickProb = Float(1.0);
            {  // DirectLightingInit, path.cpp:184-193
                const Light *dirLight = PickLight(scene, uniDist(rng), directLightPickProb);
                surfVertex.directLightRndParam = RndVec2(uniDist, rng);
                surfVertex.directLightInst.light = dirLight;
                surfVertex.directLightInst.lPrimID = dirLight->SampleDiscrete(uniDist(rng));
            }
            DirectLighting(camDepth, scene, path.time, camPathState, path.camVertex.screenPos, directLightPickProb, surfVertex, true, true, contribs);
        }
        int maxLgtDepth = maxDepth == -1 ? ((int)lightPathStates.size() - 1) : std::min((maxDepth - camDepth - 3), ((int)lightPathStates.size() - 1));
        for (int lgtDepth = 0; lgtDepth <= maxLgtDepth; lgtDepth++) {
            if (camDepth + lgtDepth + 3 >= minDepth) {
                ConnectVertex(camDepth, lgtDepth, scene, path.time, lightPathStates[lgtDepth], path.lgtSurfaceVertex[lgtDepth], camPathState, surfVertex,
                              path.camVertex.screenPos, true, contribs);
            }
        }
        surfVertex.bsdfRndParam = RndVec2(uniDist, rng);
        Vector3 bsdfContrib;
        {
            BidirPathState cur = camPathState;  // reference passes the same object as in and out
            if (!BSDFSampling<false, false>(scene->options->roughnessThreshold, camDepth, cur, surfVertex, camPathState, raySeg.ray.dir, bsdfContrib)) break;
        }
        if (!RussianRoulette(camDepth, bsdfContrib, surfVertex.rrWeight, camPathState.throughput, rng)) break;
        raySeg.ray.org = camPathState.isect.position;
        raySeg.minT = c_IsectEpsilon;
        raySeg.maxT = std::numeric_limits<Float>::infinity();
    }
}

// GenerateSubpath, path.cpp:1451-1658 (screenPosi = (-1,-1)): ONE technique (camLength camera vertices, lgtLength light vertices), no
// Russian roulette (rrWeight = 1); the generator of the multiplexed large step (mutation_large.h:45-57, mutation_large_cache.h:58-67).
// NB the area-light re-parameterisation of the last bounce is unconditional here (path.cpp:1549-1571: no option test, unlike :1339).
void GenerateSubpath(const RScene *scene, const int camLength, const int lgtLength, const bool bidirMIS, Path &path, std::vector<SubpathContrib> &contribs,
                     RNG &rng) {
    std::uniform_real_distribution<Float> uniDist(Float(0.0), Float(1.0));
    const RCamera *camera = &scene->camera;
    path.time = uniDist(rng);
    RaySegment raySeg;
    BidirPathState lightPathState;
    if (lgtLength > 1) {
        Float lightPickProb = Float(1.0);
        EmitFromLightInit(scene, path.lgtVertex, lightPickProb, rng);
        EmitFromLight(scene->bSphere, lightPickProb, path.time, path.lgtVertex, raySeg.ray, lightPathState);
        raySeg.minT = c_IsectEpsilon;
        raySeg.maxT = std::numeric_limits<Float>::infinity();
        Vector3 prevLensContrib = Vector3::Zero();
        for (int lgtDepth = 0;; lgtDepth++) {
            path.lgtSurfaceVertex.push_back(SurfaceVertex());
            bool hitSurface = Intersect(scene, path.time, raySeg, path.lgtSurfaceVertex.back().shapeInst, lightPathState.isect);
            if (!hitSurface) return;
            path.lgtSurfaceVertex.back().bsdfDiscrete = uniDist(rng);
            lightPathState.wi = -raySeg.ray.dir;
            if (bidirMIS) ConvertMIS(lgtDepth, path.lgtVertex.lightInst.light, raySeg.ray, lightPathState);
            if (lgtDepth + 2 == lgtLength) {
                if (camLength == 1) {
                    ConnectToCamera(lgtDepth, scene, camera, path.time, lightPathState, path.lgtSurfaceVertex[lgtDepth], prevLensContrib, raySeg.ray.org, contribs);
                    return;
                }
                break;
            }
            SurfaceVertex &surfVertex = path.lgtSurfaceVertex.back();
            surfVertex.bsdfRndParam = RndVec2(uniDist, rng);
            Vector3 bsdfContrib;
            {
                BidirPathState cur = lightPathState;  // the reference passes the same object as in and out
                if (!BSDFSampling<true, false>(scene->options->roughnessThreshold, lgtDepth, cur, surfVertex, lightPathState, raySeg.ray.dir, bsdfContrib)) return;
            }
            surfVertex.rrWeight = Float(1.0);
            prevLensContrib = lightPathState.lensContrib;
            raySeg.ray.org = lightPathState.isect.position;
        }
    }
    BidirPathState camPathState;
    EmitFromCameraInit(camera, -1, -1, path.camVertex, rng);
    EmitFromCamera(path.time, camera, path.camVertex, raySeg, camPathState);
    for (int camDepth = 0;; camDepth++) {
        path.camSurfaceVertex.push_back(SurfaceVertex());
        SurfaceVertex &surfVertex = path.camSurfaceVertex.back();
        bool hitSurface = Intersect(scene, path.time, raySeg, surfVertex.shapeInst, camPathState.isect);
        camPathState.wi = -raySeg.ray.dir;
        if (bidirMIS && hitSurface) ConvertMIS(camDepth, nullptr, raySeg.ray, camPathState);
        if (camDepth + 2 >= camLength && lgtLength == 0) {
            const Light *light = GetHitLight(scene, hitSurface, surfVertex.shapeInst.obj);
            if (light != nullptr) {
                if (camDepth > 1 && light->GetType() == lmc::LIGHT_AREA) {
                    SurfaceVertex &prevSurfVertex = path.camSurfaceVertex[path.camSurfaceVertex.size() - 2];
                    const ShapeInst &shapeInst = surfVertex.shapeInst;
                    prevSurfVertex.bsdfRndParam = shapeInst.obj->GetSampleParam(shapeInst.primID, camPathState.isect.position, path.time);
                    Vector3 dirToPrev = camPathState.isect.position - raySeg.ray.org;
                    const Float distSq = LengthSquared(dirToPrev);
                    const Float invDistSq = inverse(distSq);
                    const Float invDist = std::sqrt(invDistSq);
                    dirToPrev *= invDist;
                    camPathState.ssJacobian *= std::fabs(Dot(dirToPrev, camPathState.isect.shadingNormal) * invDistSq) *
                                               (camPathState.lcJacobian * surfVertex.shapeInst.obj->SamplePdf());
                }
                HandleHitLight(camDepth, scene, light, hitSurface, raySeg.ray, path.time, path.camVertex.screenPos, camPathState, bidirMIS, path.envLightInst,
                               contribs);
            }
            return;
        }
        if (!hitSurface) return;
        if (camDepth == 1) {
            path.lensVertexPos = camPathState.isect.position;
            const Float distSq = DistanceSquared(camPathState.isect.position, raySeg.ray.org);
            if (distSq <= Float(0.0)) return;
            camPathState.lensContrib *= inverse(distSq);
        }
        surfVertex.bsdfDiscrete = uniDist(rng);
        if (camDepth + 2 == camLength) {
            if (lgtLength == 1) {
                Float directLightPickProb = Float(1.0);
                {  // DirectLightingInit, path.cpp:184-193
                    const Light *dirLight = PickLight(scene, uniDist(rng), directLightPickProb);
                    surfVertex.directLightRndParam = RndVec2(uniDist, rng);
                    surfVertex.directLightInst.light = dirLight;
                    surfVertex.directLightInst.lPrimID = dirLight->SampleDiscrete(uniDist(rng));
                }
                DirectLighting(camDepth, scene, path.time, camPathState, path.camVertex.screenPos, directLightPickProb, surfVertex, true, bidirMIS, contribs);
            } else {
                ConnectVertex(camDepth, lgtLength - 2, scene, path.time, lightPathState, path.lgtSurfaceVertex.back(), camPathState, surfVertex,
                              path.camVertex.screenPos, true, contribs);
            }
            return;
        }
        surfVertex.bsdfRndParam = RndVec2(uniDist, rng);
        Vector3 bsdfContrib;
        {
            BidirPathState cur = camPathState;
            if (!BSDFSampling<false, false>(scene->options->roughnessThreshold, camDepth, cur, surfVertex, camPathState, raySeg.ray.dir, bsdfContrib)) return;
        }
        surfVertex.rrWeight = Float(1.0);
        raySeg.ray.org = camPathState.isect.position;
        raySeg.minT = c_IsectEpsilon;
        raySeg.maxT = std::numeric_limits<Float>::infinity();
    }
}

// ---- unidirectional generator of the direct-lighting pre-pass: GeneratePath, path.cpp:406-527, with its own helpers
// HandleHitLight :121-183, DirectLighting :195-294 (no shading-normal correction, power-heuristic MISWeight :23-27),
// BSDFSampling :296-386 (perturb = false); the lens* / jacobian bookkeeping feeds nothing in this pass and is left out.
static inline Float MISWeight(const Float pdfA, const Float pdfB) {
    Float ratioSq = square(pdfB / pdfA);
    return Float(1.0) / (Float(1.0) + ratioSq);
}
void GeneratePathUni(const RScene *scene, const int sx, const int sy, const int minDepth, const int maxDepth, std::vector<SubpathContrib> &contribs,
                     RNG &rng) {
    std::uniform_real_distribution<Float> uniDist(Float(0.0), Float(1.0));
    Float time = uniDist(rng);
    const RCamera *camera = &scene->camera;
    // Vector2(f(u), g(u)): see RndVec2 for the evaluation order
    Float first = uniDist(rng), second = uniDist(rng);
#if ORC_ARGS_LEFT_TO_RIGHT
    Vector2 screenPos((sx + first) / Float(camera->pixelWidth), (sy + second) / Float(camera->pixelHeight));
#else
    Vector2 screenPos((sx + second) / Float(camera->pixelWidth), (sy + first) / Float(camera->pixelHeight));
#endif
    RaySegment raySeg;
    SamplePrimary(camera, screenPos, time, raySeg);
    Vector3 throughput(Float(1.0), Float(1.0), Float(1.0));
    Float lastBsdfPdf = Float(1.0);
    Intersection isect;
    for (int camDepth = 0;; camDepth++) {
        SurfaceVertex surfVertex;
        bool hitSurface = Intersect(scene, time, raySeg, surfVertex.shapeInst, isect);
        const Light *light = GetHitLight(scene, hitSurface, surfVertex.shapeInst.obj);
        if (light != nullptr && camDepth + 1 >= minDepth) {
            LightPrimID lPrimID = 0;
            Vector3 emission;
            Float directPdf, emissionPdf;
            light->Emission(scene->bSphere, raySeg.ray.dir, isect.shadingNormal, time, lPrimID, emission, directPdf, emissionPdf);
            if (emission.sum() > Float(0.0)) {
                if (hitSurface) {
                    Float distSq = DistanceSquared(raySeg.ray.org, isect.position);
                    Float cosTheta = -Dot(raySeg.ray.dir, isect.shadingNormal);
                    directPdf *= (distSq / cosTheta);
                }
                Vector3 contrib = throughput.cwiseProduct(emission);
                Float misWeight = Float(1.0);
                if (camDepth > 0) {
                    Float lightPickProb = PickLightProb(scene, light);
                    misWeight = MISWeight(lastBsdfPdf, directPdf * lightPickProb);
                    contrib *= misWeight;
                }
                const Float score = Luminance(contrib);
                if (score > Float(0.0)) contribs.emplace_back(SubpathContrib{2 + camDepth, 0, screenPos, contrib, score, score, Float(0.0), misWeight});
            }
            return;
        }
        if (!hitSurface || (maxDepth != -1 && camDepth + 1 >= maxDepth)) break;
        surfVertex.bsdfDiscrete = uniDist(rng);
        const Vector3 wi = -raySeg.ray.dir;
        const BSDF *bsdf = surfVertex.shapeInst.obj->bsdf;
        if (camDepth + 2 >= minDepth) {
            Float lightPickProb = Float(1.0);
            {  // DirectLightingInit, path.cpp:184-193
                const Light *dirLight = PickLight(scene, uniDist(rng), lightPickProb);
                surfVertex.directLightRndParam = RndVec2(uniDist, rng);
                surfVertex.directLightInst.light = dirLight;
                surfVertex.directLightInst.lPrimID = dirLight->SampleDiscrete(uniDist(rng));
            }
            const Light *dl = surfVertex.directLightInst.light;
            LightPrimID &lPrimID = surfVertex.directLightInst.lPrimID;
            Vector3 dirToLight, lightContrib;
            Float distToLight, cosAtLight, directPdf, emissionPdf;
            if (dl->SampleDirect(scene->bSphere, isect.position, isect.shadingNormal, surfVertex.directLightRndParam, time, lPrimID, dirToLight,
                                 distToLight, lightContrib, cosAtLight, directPdf, emissionPdf) &&
                !Occluded(scene, time, Ray{isect.position, dirToLight}, distToLight)) {
                Vector3 bsdfContrib;
                Float cosWo, bsdfPdf, bsdfRevPdf;
                bsdf->Evaluate(wi, isect.shadingNormal, dirToLight, surfVertex.shapeInst.st, bsdfContrib, cosWo, bsdfPdf, bsdfRevPdf);
                if (!bsdfContrib.isZero()) {
                    Vector3 contrib = throughput.cwiseProduct(bsdfContrib);
                    contrib = contrib.cwiseProduct(lightContrib) * inverse(lightPickProb);
                    Float misWeight = Float(1.0);
                    if (!dl->IsDelta()) {
                        misWeight = MISWeight(directPdf * lightPickProb, bsdfPdf);
                        contrib *= misWeight;
                    }
                    const Float score = Luminance(contrib);
                    if (score > Float(0.0)) contribs.emplace_back(SubpathContrib{2 + camDepth, 1, screenPos, contrib, score, score, Float(0.0), misWeight});
                }
            }
        }
        surfVertex.bsdfRndParam = RndVec2(uniDist, rng);
        Vector3 bsdfContrib;
        Float cosWo, bsdfPdfRev;
        if (!bsdf->Sample(wi, isect.shadingNormal, surfVertex.shapeInst.st, surfVertex.bsdfRndParam, surfVertex.bsdfDiscrete, raySeg.ray.dir, bsdfContrib,
                          cosWo, lastBsdfPdf, bsdfPdfRev))
            break;
        throughput = throughput.cwiseProduct(bsdfContrib);
        raySeg.ray.org = isect.position;
        if (!RussianRoulette(camDepth, bsdfContrib, surfVertex.rrWeight, throughput, rng)) break;
        raySeg.minT = c_IsectEpsilon;
        raySeg.maxT = std::numeric_limits<Float>::infinity();
    }
}

void ToSubpath(const int camDepth, const int lgtDepth, Path &path) {  // path.cpp:1660-1669
    path.camSurfaceVertex.resize(std::max(camDepth - 1, 0));
    path.lgtSurfaceVertex.resize(std::max(lgtDepth - 1, 0));
    if (lgtDepth != 0) path.envLightInst.light = nullptr;
    path.isSubpath = true;
    path.camDepth = camDepth;
    path.lgtDepth = lgtDepth;
}

static inline void Perturb(Float &value, const std::vector<Float> &offset, int &offsetId) {  // path.cpp:1671-1673
    value = Modulo(value + offset[offsetId++], Float(1.0));
}

// path.cpp:1881-1951: the last bounce towards an area light, re-sampled in the light's own coordinates
static bool LightCoordinateSampling(const int camDepth, const RScene *scene, const Float time, const SurfaceVertex &curSurfVertex,
                                    const SurfaceVertex &nextSurfVertex, const bool doOcclusion, BidirPathState &pathState, Vector3 &dir, Vector3 &bsdfContrib) {
    const ShapeInst &shapeInst = curSurfVertex.shapeInst;
    const ShapeInst &nextShapeInst = nextSurfVertex.shapeInst;
    const BSDF *bsdf = shapeInst.obj->bsdf;
    const Intersection &isect = pathState.isect;
    const Vector3 &wi = pathState.wi;
    Vector3 nextPosition, nextNormal;
    Float shapePdf = Float(0.0);
    nextShapeInst.obj->Sample(curSurfVertex.bsdfRndParam, time, nextSurfVertex.shapeInst.primID, nextPosition, nextNormal, &shapePdf);
    dir = nextPosition - isect.position;
    Float distToLightSq = LengthSquared(dir);
    Float distToLight = std::sqrt(distToLightSq);
    dir *= inverse(distToLight);
    if (doOcclusion && Occluded(scene, time, Ray{isect.position, dir}, distToLight)) return false;
    Float cosWo, bsdfPdf, bsdfRevPdf;
    bsdf->Evaluate(wi, isect.shadingNormal, dir, shapeInst.st, bsdfContrib, cosWo, bsdfPdf, bsdfRevPdf);
    if (bsdfContrib.isZero()) return false;
    bsdfContrib *= inverse(bsdfPdf);
    pathState.throughput = pathState.throughput.cwiseProduct(bsdfContrib);
    pathState.ssJacobian *= std::fabs(Dot(dir, nextNormal) * inverse(distToLightSq)) * bsdfPdf;
    pathState.accMISWThis = MIS(cosWo / bsdfPdf) * (pathState.accMISWThis * MIS(bsdfRevPdf) + pathState.accMISWPrev);
    pathState.accMISWPrev = MIS(inverse(bsdfPdf));
    // (camDepth == 1: lensContrib, :1926-1949 -- feeds PathFuncMode::Lens only)
    (void)camDepth;
    return true;
}

void PerturbPathBidir(const RScene *scene, const std::vector<Float> &offset, Path &path, std::vector<SubpathContrib> &contribs, RNG &rng) {
    // path.cpp:1953-2160
    std::normal_distribution<Float> normDist(Float(0.0), scene->options->discreteStdDev);
    const RCamera *camera = &scene->camera;
    int offsetId = 0;
    path.time = Modulo(path.time + normDist(rng), Float(1.0));
    BidirPathState lightPathState;
    if (path.lgtDepth > 1) {
        const Float lightPickProb = PickLightProb(scene, path.lgtVertex.lightInst.light);
        RaySegment raySeg;
        Perturb(path.lgtVertex.rndParamPos[0], offset, offsetId);
        Perturb(path.lgtVertex.rndParamPos[1], offset, offsetId);
        Perturb(path.lgtVertex.rndParamDir[0], offset, offsetId);
        Perturb(path.lgtVertex.rndParamDir[1], offset, offsetId);
        EmitFromLight(scene->bSphere, lightPickProb, path.time, path.lgtVertex, raySeg.ray, lightPathState);
        raySeg.minT = c_IsectEpsilon;
        raySeg.maxT = std::numeric_limits<Float>::infinity();
        Vector3 prevLensContrib = Vector3::Zero();
        for (int lgtDepth = 0; lgtDepth < (int)path.lgtSurfaceVertex.size(); lgtDepth++) {
            SurfaceVertex &surfVertex = path.lgtSurfaceVertex[lgtDepth];
            if (!Intersect(scene, path.time, raySeg, surfVertex.shapeInst, lightPathState.isect)) return;
            lightPathState.wi = -raySeg.ray.dir;
            surfVertex.bsdfDiscrete = Modulo(surfVertex.bsdfDiscrete + normDist(rng), Float(1.0));
            ConvertMIS(lgtDepth, path.lgtVertex.lightInst.light, raySeg.ray, lightPathState);
            if (lgtDepth == (int)path.lgtSurfaceVertex.size() - 1 && path.camDepth == 1) {
                ConnectToCamera(lgtDepth, scene, camera, path.time, lightPathState, path.lgtSurfaceVertex[lgtDepth], prevLensContrib, raySeg.ray.org,
                                contribs);
                return;
            }
            if (lgtDepth == (int)path.lgtSurfaceVertex.size() - 1) break;
            Perturb(surfVertex.bsdfRndParam[0], offset, offsetId);
            Perturb(surfVertex.bsdfRndParam[1], offset, offsetId);
            Vector3 bsdfContrib;
            {
                BidirPathState cur = lightPathState;
                if (!BSDFSampling<true, true>(scene->options->roughnessThreshold, lgtDepth, cur, surfVertex, lightPathState, raySeg.ray.dir, bsdfContrib))
                    return;
            }
            lightPathState.throughput *= surfVertex.rrWeight;
            prevLensContrib = lightPathState.lensContrib;
            raySeg.ray.org = lightPathState.isect.position;
        }
    }

    CameraVertex &camVertex = path.camVertex;
    Perturb(camVertex.screenPos[0], offset, offsetId);
    Perturb(camVertex.screenPos[1], offset, offsetId);
    RaySegment raySeg;
    BidirPathState camPathState;
    EmitFromCamera(path.time, camera, path.camVertex, raySeg, camPathState);
    for (int camDepth = 0; camDepth < (int)path.camSurfaceVertex.size(); camDepth++) {
        SurfaceVertex &surfVertex = path.camSurfaceVertex[camDepth];
        bool hitSurface = Intersect(scene, path.time, raySeg, surfVertex.shapeInst, camPathState.isect);
        camPathState.wi = -raySeg.ray.dir;
        if (hitSurface) ConvertMIS(camDepth, nullptr, raySeg.ray, camPathState);
        if (camDepth == (int)path.camSurfaceVertex.size() - 1 && path.lgtDepth == 0) {
            const Light *light = GetHitLight(scene, hitSurface, surfVertex.shapeInst.obj);
            if (light != nullptr) {
                HandleHitLight(camDepth, scene, light, hitSurface, raySeg.ray, path.time, path.camVertex.screenPos, camPathState, true, path.envLightInst,
                               contribs);
            }
            return;
        }
        if (!hitSurface) return;
        surfVertex.bsdfDiscrete = Modulo(surfVertex.bsdfDiscrete + normDist(rng), Float(1.0));
        if (camDepth == 1) {
            path.lensVertexPos = camPathState.isect.position;
            const Float distSq = DistanceSquared(camPathState.isect.position, raySeg.ray.org);
            if (distSq <= Float(0.0)) {
                contribs.clear();
                return;
            } else {
                camPathState.lensContrib *= inverse(distSq);
            }
        }
        if (camDepth == (int)path.camSurfaceVertex.size() - 1) {
            if (path.lgtDepth == 1) {
                const Float directLightPickProb = PickLightProb(scene, surfVertex.directLightInst.light);
                Perturb(surfVertex.directLightRndParam[0], offset, offsetId);
                Perturb(surfVertex.directLightRndParam[1], offset, offsetId);
                DirectLighting(camDepth, scene, path.time, camPathState, path.camVertex.screenPos, directLightPickProb, surfVertex, true, true, contribs);
            } else {
                ConnectVertex(camDepth, (int)path.lgtSurfaceVertex.size() - 1, scene, path.time, lightPathState, path.lgtSurfaceVertex.back(),
                              camPathState, surfVertex, path.camVertex.screenPos, true, contribs);
            }
            return;
        }
        Perturb(surfVertex.bsdfRndParam[0], offset, offsetId);
        Perturb(surfVertex.bsdfRndParam[1], offset, offsetId);
        bool useLightCoordinatesPerturb = false;  // path.cpp:2120-2128
        if (scene->options->useLightCoordinateSampling) {
            if (camDepth == int(path.camSurfaceVertex.size()) - 2 && path.lgtDepth == 0) {
                const ShapeInst &shapeInst = path.camSurfaceVertex.back().shapeInst;
                if (shapeInst.obj != nullptr && shapeInst.obj->areaLight != nullptr) useLightCoordinatesPerturb = true;
            }
        }
        Vector3 bsdfContrib;
        if (useLightCoordinatesPerturb) {
            if (!LightCoordinateSampling(camDepth, scene, path.time, surfVertex, path.camSurfaceVertex.back(), true, camPathState, raySeg.ray.dir, bsdfContrib)) return;
        } else {
            BidirPathState cur = camPathState;
            if (!BSDFSampling<false, true>(scene->options->roughnessThreshold, camDepth, cur, surfVertex, camPathState, raySeg.ray.dir, bsdfContrib)) return;
        }
        camPathState.throughput *= surfVertex.rrWeight;
        raySeg.ray.org = camPathState.isect.position;
        raySeg.minT = c_IsectEpsilon;
        raySeg.maxT = std::numeric_limits<Float>::infinity();
    }
}

// ---------------------------------------------------------------------------------------- serialisation
size_t GetPrimaryParamSize(const int camDepth, const int lightDepth) { return std::max(camDepth + lightDepth - 1, 2) * 2 + 1; }  // path.cpp:2481-2483

size_t GetVertParamSize(const int maxCamDepth, const int maxLgtDepth) {  // path.cpp:2485-2495
    const int maxDepth = maxCamDepth + maxLgtDepth;
    return maxDepth * 46 + maxDepth * 10 + 56 + maxDepth * 2 + maxDepth * 1 + 3 + 1 + 1;
}

void Serialize(const RScene *scene, const Path &path, SerializedSubpath &subPath) {  // path.cpp:2497-2586
    int primaryIdx = 0;
    subPath.primary[primaryIdx++] = path.time;
    Float *buffer = &subPath.vertParams[0];
    for (int k = 0; k < 3; k++) *buffer++ = path.lensVertexPos[k];
    if (path.lgtDepth > 1) {
        const Light *light = path.lgtVertex.lightInst.light;
        subPath.primary[primaryIdx++] = path.lgtVertex.rndParamPos[0];
        subPath.primary[primaryIdx++] = path.lgtVertex.rndParamPos[1];
        subPath.primary[primaryIdx++] = path.lgtVertex.rndParamDir[0];
        subPath.primary[primaryIdx++] = path.lgtVertex.rndParamDir[1];
        *buffer++ = PickLightProb(scene, light);
        light->Serialize(path.lgtVertex.lightInst.lPrimID, buffer);
        buffer += 56;
        for (int lgtDepth = 0; lgtDepth < (int)path.lgtSurfaceVertex.size(); lgtDepth++) {
            const SurfaceVertex &surfVertex = path.lgtSurfaceVertex[lgtDepth];
            const ShapeInst &shapeInst = surfVertex.shapeInst;
            shapeInst.obj->Serialize(shapeInst.primID, buffer);
            buffer += 46;
            *buffer++ = surfVertex.bsdfDiscrete;
            *buffer++ = surfVertex.useAbsoluteParam;
            shapeInst.obj->bsdf->Serialize(shapeInst.st, buffer);
            buffer += 10;
            if (lgtDepth == (int)path.lgtSurfaceVertex.size() - 1 && path.camDepth == 1) return;
            if (lgtDepth == (int)path.lgtSurfaceVertex.size() - 1) break;
            subPath.primary[primaryIdx++] = surfVertex.bsdfRndParam[0];
            subPath.primary[primaryIdx++] = surfVertex.bsdfRndParam[1];
            *buffer++ = surfVertex.rrWeight;
        }
    }
    const CameraVertex &camVertex = path.camVertex;
    subPath.primary[primaryIdx++] = camVertex.screenPos[0];
    subPath.primary[primaryIdx++] = camVertex.screenPos[1];
    for (int camDepth = 0; camDepth < (int)path.camSurfaceVertex.size(); camDepth++) {
        const SurfaceVertex &surfVertex = path.camSurfaceVertex[camDepth];
        const ShapeInst &shapeInst = surfVertex.shapeInst;
        if (shapeInst.obj != nullptr) {
            shapeInst.obj->Serialize(shapeInst.primID, buffer);
        } else {
            // The ray escaped to the environment light.  The reference leaves this 46-float slot untouched
            // (path.cpp:2546-2549), i.e. it holds whatever an earlier Serialize of the same MALASmallStep object
            // wrote there; the derivative program still "intersects" it, and with an all-zero slot its reverse
            // sweep turns 0 * NaN into a NaN gradient.  The value the reference produces whenever the stale slot
            // is a non-degenerate triangle does not depend on that triangle, so the oracle writes a fixed benign
            // one (DESIGN.md "escaped last vertex").
            static const Float dummy[46] = {0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1};
            memcpy(buffer, dummy, sizeof(dummy));
        }
        buffer += 46;
        if (camDepth == (int)path.camSurfaceVertex.size() - 1) {
            if (path.lgtDepth == 0) {
                if (path.envLightInst.light != nullptr) {
                    path.envLightInst.light->Serialize(path.envLightInst.lPrimID, buffer);
                    buffer += 56;
                    *buffer++ = PickLightProb(scene, path.envLightInst.light);
                } else {
                    shapeInst.obj->areaLight->Serialize(shapeInst.primID, buffer);
                    buffer += 56;
                    *buffer++ = PickLightProb(scene, shapeInst.obj->areaLight);
                }
            } else if (path.lgtDepth == 1) {
                subPath.primary[primaryIdx++] = surfVertex.directLightRndParam[0];
                subPath.primary[primaryIdx++] = surfVertex.directLightRndParam[1];
                surfVertex.directLightInst.light->Serialize(surfVertex.directLightInst.lPrimID, buffer);
                buffer += 56;
                shapeInst.obj->bsdf->Serialize(shapeInst.st, buffer);
                buffer += 10;
                *buffer++ = PickLightProb(scene, surfVertex.directLightInst.light);
            } else {
                shapeInst.obj->bsdf->Serialize(shapeInst.st, buffer);
                buffer += 10;
            }
            return;
        }
        subPath.primary[primaryIdx++] = surfVertex.bsdfRndParam[0];
        subPath.primary[primaryIdx++] = surfVertex.bsdfRndParam[1];
        *buffer++ = surfVertex.bsdfDiscrete;
        *buffer++ = surfVertex.useAbsoluteParam;
        shapeInst.obj->bsdf->Serialize(shapeInst.st, buffer);
        buffer += 10;
        *buffer++ = surfVertex.rrWeight;
    }
}

void GetPathPss(const Path &path, std::vector<Float> &pss) {  // path.cpp:2588-2632
    const int dim = GetDimension(path);
    if (!pss.size()) pss.resize(dim);
    int primaryIdx = 0;
    if (path.lgtDepth > 1) {
        pss[primaryIdx++] = path.lgtVertex.rndParamPos[0];
        pss[primaryIdx++] = path.lgtVertex.rndParamPos[1];
        pss[primaryIdx++] = path.lgtVertex.rndParamDir[0];
        pss[primaryIdx++] = path.lgtVertex.rndParamDir[1];
        for (int lgtDepth = 0; lgtDepth < (int)path.lgtSurfaceVertex.size(); lgtDepth++) {
            const SurfaceVertex &surfVertex = path.lgtSurfaceVertex[lgtDepth];
            if (lgtDepth == (int)path.lgtSurfaceVertex.size() - 1 && path.camDepth == 1) return;
            if (lgtDepth == (int)path.lgtSurfaceVertex.size() - 1) break;
            pss[primaryIdx++] = surfVertex.bsdfRndParam[0];
            pss[primaryIdx++] = surfVertex.bsdfRndParam[1];
        }
    }
    pss[primaryIdx++] = path.camVertex.screenPos[0];
    pss[primaryIdx++] = path.camVertex.screenPos[1];
    for (int camDepth = 0; camDepth < (int)path.camSurfaceVertex.size(); camDepth++) {
        const SurfaceVertex &surfVertex = path.camSurfaceVertex[camDepth];
        if (camDepth == (int)path.camSurfaceVertex.size() - 1) {
            if (path.lgtDepth == 1) {
                pss[primaryIdx++] = surfVertex.directLightRndParam[0];
                pss[primaryIdx++] = surfVertex.directLightRndParam[1];
            }
            return;
        }
        pss[primaryIdx++] = surfVertex.bsdfRndParam[0];
        pss[primaryIdx++] = surfVertex.bsdfRndParam[1];
    }
}

}  // namespace orc
