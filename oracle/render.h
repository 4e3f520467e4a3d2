// ORACLE -- TEST INFRASTRUCTURE ONLY (see common.h).
// Runtime scene objects + path types restating /root/reference/src/{scene,shape,trianglemesh,bsdf,
// lambertian,light,envlight,arealight,pointlight,camera,path}.h for the CPU oracle.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../langevin-mcmc_amd/csrc/host/scene.h"  // plain scene description (data plumbing shared with the product)
#include "common.h"
#include "rng.h"

namespace orc {

struct Ray {
    Vector3 org, dir;
};
struct RaySegment {
    Ray ray;
    Float minT, maxT;
};
struct Intersection {
    Vector3 position, shadingNormal, geomNormal;
};

// ---------------------------------------------------------------------------------------------- BVH
// Stand-in for Embree (scene.cpp:106-149).  Closest hit = smallest t in [tnear, tfar] accepted by the
// reference's own Moeller-Trumbore test (trianglemesh.cpp:30-53) with u,v >= 0, u+v <= 1; ties on t go to
// the lower global triangle id, which makes the answer independent of the tree -> bit-comparable with
// the HIP LBVH.  Embree's own hit selection is third-party and absent: parity unpinned.
struct TriAccel {
    Vector3 p0, e1, e2;
};
struct Bvh {
    struct Node {
        float bmin[3], bmax[3];
        int left, right;  // left < 0: leaf, first = right, count = -left
    };
    std::vector<Node> nodes;
    std::vector<int> triIds;  // global triangle ids in leaf order
    const std::vector<TriAccel> *tris = nullptr;
    void Build(const std::vector<TriAccel> &tris);
    // returns global triangle id or -1
    int Intersect(const Ray &ray, Float tnear, Float tfar, Float *tOut) const;
    bool Occluded(const Ray &ray, Float tnear, Float tfar) const;
};
bool TriTest(const TriAccel &tr, const Ray &ray, Float tnear, Float tfar, Float &t);

// ---------------------------------------------------------------------------------------------- scene
struct Shape;
struct Light;
struct RScene;

typedef int PrimID;
typedef PrimID LightPrimID;
const LightPrimID INVALID_LPRIM_ID = LightPrimID(-1);

struct BSDF {
    virtual ~BSDF() {}
    virtual int GetType() const = 0;
    virtual void Serialize(const Vector2 st, Float *buffer) const = 0;  // 10-float slot (bsdf.cpp:7-11)
    virtual void Evaluate(const Vector3 &wi, const Vector3 &normal, const Vector3 &wo, const Vector2 st, Vector3 &contrib, Float &cosWo,
                          Float &pdf, Float &revPdf) const = 0;
    virtual void EvaluateAdjoint(const Vector3 &wi, const Vector3 &normal, const Vector3 &wo, const Vector2 st, Vector3 &contrib,
                                 Float &cosWo, Float &pdf, Float &revPdf) const {
        Evaluate(wi, normal, wo, st, contrib, cosWo, pdf, revPdf);  // bsdf.h:27-38
    }
    virtual bool Sample(const Vector3 &wi, const Vector3 &normal, const Vector2 st, const Vector2 rndParam, const Float uDiscrete,
                        Vector3 &wo, Vector3 &contrib, Float &cosWo, Float &pdf, Float &revPdf) const = 0;
    virtual bool SampleAdjoint(const Vector3 &wi, const Vector3 &normal, const Vector2 st, const Vector2 rndParam, const Float uDiscrete,
                               Vector3 &wo, Vector3 &contrib, Float &cosWo, Float &pdf, Float &revPdf) const {
        return Sample(wi, normal, st, rndParam, uDiscrete, wo, contrib, cosWo, pdf, revPdf);
    }
    virtual Float Roughness(const Vector2 st, const Float uDiscrete) const = 0;
};

struct Shape {  // = TriangleMesh
    const lmc::Mesh *mesh = nullptr;
    const BSDF *bsdf = nullptr;
    const Light *areaLight = nullptr;
    int id = 0;
    int triBase = 0;  // global id of triangle 0
    void Serialize(const PrimID primID, Float *buffer) const;  // 46 floats
    bool Intersect(const PrimID &primID, const Float time, const RaySegment &raySeg, Intersection &isect, Vector2 &st) const;
    PrimID Sample(const Float u) const;
    void Sample(const Vector2 rndParam, const Float time, const PrimID primID, Vector3 &position, Vector3 &normal, Float *pdf) const;
    Float SamplePdf() const { return inverse(mesh->totalArea); }
    Vector2 GetSampleParam(const PrimID &primID, const Vector3 &position, const Float time) const;  // trianglemesh.cpp:255-285
};

struct ShapeInst {
    const Shape *obj = nullptr;
    PrimID primID = 0;
    Vector2 st;
};

struct BSphere {
    Vector3 center;
    Float radius;
};

struct Light {
    virtual ~Light() {}
    Float samplingWeight = 1.f;
    int id = 0;
    virtual int GetType() const = 0;
    virtual void Serialize(const LightPrimID &lPrimID, Float *buffer) const = 0;  // 56-float slot
    virtual LightPrimID SampleDiscrete(const Float uDiscrete) const { return INVALID_LPRIM_ID; }
    virtual bool SampleDirect(const BSphere &sceneSphere, const Vector3 &pos, const Vector3 &normal, const Vector2 rndParam,
                              const Float time, LightPrimID &lPrimID, Vector3 &dirToLight, Float &dist, Vector3 &contrib,
                              Float &cosAtLight, Float &directPdf, Float &emissionPdf) const = 0;
    virtual void Emission(const BSphere &sceneSphere, const Vector3 &dirToLight, const Vector3 &normalOnLight, const Float time,
                          LightPrimID &lPrimID, Vector3 &emission, Float &directPdf, Float &emissionPdf) const;
    virtual void Emit(const BSphere &sceneSphere, const Vector2 rndParamPos, const Vector2 rndParamDir, const Float time,
                      LightPrimID &lPrimID, Ray &ray, Vector3 &emission, Float &cosAtLight, Float &emissionPdf,
                      Float &directPdf) const = 0;
    virtual bool IsFinite() const = 0;
    virtual bool IsDelta() const = 0;
};

struct LightInst {
    const Light *light = nullptr;
    LightPrimID lPrimID = 0;
};

struct RCamera {
    Float sampleToCam[4][4], camToSample[4][4];
    Float toWorld[4][4], worldToCamera[4][4];  // static transforms (isMoving == 0)
    int pixelWidth, pixelHeight;
    Float nearClip, farClip, dist;
};

struct RScene {
    std::unique_ptr<lmc::Scene> desc;
    lmc::DptOptions *options = nullptr;
    RCamera camera;
    std::vector<std::unique_ptr<BSDF>> bsdfs;
    std::vector<std::unique_ptr<Shape>> objects;
    std::vector<std::unique_ptr<Light>> lights;
    const Light *envLight = nullptr;
    BSphere bSphere;
    Float lightWeightSum = 0;
    std::vector<TriAccel> tris;
    std::vector<int> triShape;  // global tri -> shape id
    Bvh bvh;
    Float sceneParams[38];
};

std::unique_ptr<RScene> BuildRScene(std::unique_ptr<lmc::Scene> desc);

bool Intersect(const RScene *scene, const Float time, const RaySegment &raySeg, ShapeInst &shapeInst);
bool Occluded(const RScene *scene, const Float time, const Ray &ray, const Float dist);
const Light *PickLight(const RScene *scene, const Float u, Float &prob);
Float PickLightProb(const RScene *scene, const Light *light);

void SamplePrimary(const RCamera *camera, const Vector2 screenPos, const Float time, RaySegment &raySeg);
bool ProjectPoint(const RCamera *camera, const Vector3 &p, const Float time, Vector2 &screenPos);

// ---------------------------------------------------------------------------------------------- path.h
struct SubpathContrib {
    int camDepth, lightDepth;
    Vector2 screenPos;
    Vector3 contrib;
    Float lsScore, ssScore, lensScore, misWeight;
};
struct CameraVertex {
    Vector2 screenPos;
};
struct SurfaceVertex {
    ShapeInst shapeInst;
    Vector2 bsdfRndParam;
    Float bsdfDiscrete = 0;
    Float useAbsoluteParam = 0;
    LightInst directLightInst;
    Vector2 directLightRndParam;
    Float rrWeight = 0;
};
struct LightVertex {
    Vector2 rndParamPos, rndParamDir;
    LightInst lightInst;
};
struct Path {
    Float time = 0;
    CameraVertex camVertex;
    std::vector<SurfaceVertex> camSurfaceVertex;
    LightVertex lgtVertex;
    std::vector<SurfaceVertex> lgtSurfaceVertex;
    LightInst envLightInst;
    Vector3 lensVertexPos;
    bool isSubpath = false;
    int camDepth = 0, lgtDepth = 0;
};
struct SerializedSubpath {
    std::vector<Float> primary, vertParams;
};

void Clear(Path &path);
void GeneratePathBidir(const RScene *scene, const int screenPosiX, const int screenPosiY, const int minDepth, const int maxDepth,
                       Path &path, std::vector<SubpathContrib> &contribs, RNG &rng);
void GeneratePathUni(const RScene *scene, const int screenPosiX, const int screenPosiY, const int minDepth, const int maxDepth,
                     std::vector<SubpathContrib> &contribs, RNG &rng);
void GenerateSubpath(const RScene *scene, const int camLength, const int lgtLength, const bool bidirMIS, Path &path, std::vector<SubpathContrib> &contribs,
                     RNG &rng);  // path.cpp:1451-1658
void ToSubpath(const int camDepth, const int lightDepth, Path &path);
void PerturbPathBidir(const RScene *scene, const std::vector<Float> &offset, Path &path, std::vector<SubpathContrib> &contribs,
                      RNG &rng);
size_t GetVertParamSize(const int maxCamDepth, const int maxLgtDepth);
size_t GetPrimaryParamSize(const int camDepth, const int lightDepth);
void Serialize(const RScene *scene, const Path &path, SerializedSubpath &subPath);
void GetPathPss(const Path &path, std::vector<Float> &pss);
inline int GetDimension(const Path &path) { return (int)GetPrimaryParamSize(path.camDepth, path.lgtDepth) - 1; }
inline int GetPathLength(const int camLength, const int lgtLength) { return camLength + lgtLength - 1; }

}  // namespace orc
