// Host-side scene description: what the reference's ParseScene() builds
// (/root/reference/src/parsescene.cpp:592-639) restated as plain arrays that both the HIP
// back end (which flattens it into device buffers) and the CPU test oracle consume.
// This is data plumbing ("on-disk formats"), not the hot path.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "imageio.h"
#include "mathx.h"

namespace lmc {

// /root/reference/src/dptoptions.h:7-34 (same names, same defaults)
struct DptOptions {
    std::string integrator = "mcmc";
    bool bidirectional = true;
    int spp = 256;
    int numInitSamples = 300000;
    int minDepth = -1;
    int maxDepth = 8;
    int directSpp = 256;
    bool h2mc = false;
    float perturbStdDev = 0.01f;
    float roughnessThreshold = 0.05f;
    float largeStepProbability = 0.05f;
    float largeStepProbScale = 1.0f;
    bool mala = false;
    float malaGN = 100.0f;
    float malaStepsize = 0.005f;
    float malaStdDev = 0.005f;
    bool sampleFromGlobalCache = false;
    int numChains = 128;
    int seedOffset = 0;
    int reportIntervalSpp = 0;
    float discreteStdDev = 0.01f;
    float uniformMixingProbability = 0.1f;
    bool useLightCoordinateSampling = false;
    bool largeStepMultiplexed = false;
};

enum BSDFType { BSDF_LAMBERTIAN = 0, BSDF_PHONG = 1, BSDF_ROUGHDIELECTRIC = 2 };  // bsdf.h:6
enum LightType { LIGHT_POINT = 0, LIGHT_AREA = 1, LIGHT_ENV = 2 };                // light.h:7

// constant or bitmap texture (texture.h, constanttexture.h, bitmaptexture.h)
struct Bitmap {
    std::string filename;
    Image3f img;
    float gamma = 1.f;  // 2.2 for 8-bit files (bitmaptexture.h:135-144)
    float avg[3] = {0, 0, 0};
};
struct TextureRef {
    int bitmap = -1;  // index into Scene::bitmaps, -1 = constant
    float value[3] = {0, 0, 0};
    float sScale = 1.f, tScale = 1.f;
};

struct Material {
    int type = BSDF_LAMBERTIAN;
    bool twoSided = false;
    TextureRef Kd, Ks, Kt;  // Lambertian: Kd; Phong: Kd, Ks; RoughDielectric: Ks, Kt
    TextureRef expOrAlpha;  // Phong exponent / dielectric alpha (channel 0)
    float eta = 1.f, invEta = 1.f;  // roughdielectric.h: eta = intIOR/extIOR
    float KsWeight = 0.f;           // phong.cpp:159-169
};

struct Mesh {
    std::vector<V3> P, N;  // static meshes only (position0/normal0); moving geometry is out of scope
    std::vector<V2> ST;    // empty => use barycentrics (trianglemesh.cpp:228-233)
    std::vector<uint32_t> idx;
    int material = -1;
    int areaLight = -1;  // index into Scene::lights
    float totalArea = 0.f;
    std::vector<float> areaCdf;  // PiecewiseConstant1D over triangle areas (size nTri+1), area lights only
    std::vector<float> areaFunc;
    float areaFuncInt = 0.f;
    V3 bmin, bmax;
    size_t numTris() const { return idx.size() / 3; }
};

// envlight.cpp:24-63
struct EnvmapSampleInfo {
    std::vector<float> cdfRows, cdfCols, rowWeights;
    float normalization = 0.f;
    float pixelSize[2] = {0, 0};
};

struct Light {
    int type = LIGHT_POINT;
    float samplingWeight = 1.f;
    V3 position{0, 0, 0}, intensity{1, 1, 1};  // point
    int mesh = -1;                             // area
    V3 radiance{1, 1, 1};
    AnimXform toWorld, toLight;  // env
    Image3f image;
    EnvmapSampleInfo sampleInfo;
};

struct Camera {
    M4 sampleToCam, camToSample;
    AnimXform camToWorld, worldToCamera;
    int width = 512, height = 512;
    float nearClip = 1e-2f, farClip = 1000.f, fov = 45.f, dist = 0.f;
};

struct Scene {
    DptOptions options;
    Camera camera;
    std::vector<Mesh> meshes;
    std::vector<Material> materials;
    std::vector<Bitmap> bitmaps;
    std::vector<Light> lights;
    int envLight = -1;
    // scene.cpp:21-28,151-158: PiecewiseConstant1D over samplingWeight
    std::vector<float> lightFunc, lightCdf;
    float lightFuncInt = 0.f, lightWeightSum = 0.f;
    V3 bsphereCenter{0, 0, 0};
    float bsphereRadius = 0.f;  // already x1000 (scene.cpp:40)
    std::string outputName = "image.exr";
    size_t numTris() const {
        size_t n = 0;
        for (auto &m : meshes) n += m.numTris();
        return n;
    }
};

struct LoadOverrides {
    bool forceDiffuse = false;  // BASELINE.json config 2: every BSDF becomes `diffuse`
    int maxDepth = 0;           // >0 overrides <dpt maxdepth>
    int numChains = 0, spp = 0, numInitSamples = 0, width = 0, height = 0, seedOffset = -1, directSpp = -1;
};

// Parses `filename` (paths inside are relative to its directory, like the reference's chdir in
// main.cpp:78-86).  Throws std::runtime_error with the reference's messages where it has them.
std::unique_ptr<Scene> ParseScene(const std::string &filename, const LoadOverrides &ov = LoadOverrides());
std::unique_ptr<Scene> ParseSceneString(const std::string &xml, const std::string &baseDir, const LoadOverrides &ov = LoadOverrides());

// PiecewiseConstant1D (distribution.h:8-60) helpers on the flattened arrays
void BuildPiecewise1D(const float *f, int n, std::vector<float> &func, std::vector<float> &cdf, float &funcInt);
int SampleDiscrete1D(const std::vector<float> &func, const std::vector<float> &cdf, float funcInt, float u, float *pdf);

// scene.cpp:160-169 + camera.cpp:30-36: the 38-float `scene` block of the path-function ABI
void SerializeSceneBlock(const Scene &scene, float out[38]);

// Texture lookup (bitmaptexture.h:72-97): periodic bilinear lookup, value^gamma via fastpow
void EvalTexture(const Scene &scene, const TextureRef &t, float s, float tt, float out[3]);

}  // namespace lmc
