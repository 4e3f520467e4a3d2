// ORACLE -- TEST INFRASTRUCTURE ONLY (see common.h).
// MLT driver state: restates /root/reference/src/{mlt.h,mutation.h,gaussian.h,global_cache.h}.
#pragma once
#include <map>
#include <memory>
#include <vector>

#include "render.h"

namespace orc {

// gaussian.h:9-19 (dense members only matter for H2MC; the LMC path uses the diagonal ones)
struct Gaussian {
    std::vector<Float> mean, covL_d, invCov_d;
    Float logDet = 0;
    bool isDiagonal = false;
    // H2MC (h2mc.cpp): dense covL / invCov, row-major dim x dim; `dense` selects them in GaussianLogPdf / GenerateSample
    bool dense = false;
    std::vector<Float> covL, invCov;
};
void IsotropicGaussian(const int dim, const Float sigma, Gaussian &gaussian);
Float GaussianLogPdf(const std::vector<Float> &offset, const Gaussian &gaussian, bool negate);
void GenerateSample(Gaussian &gaussian, std::vector<Float> &x, RNG &rng);
void ComputeGaussianMALA(const int dim, const std::vector<Float> &v1, const std::vector<Float> &v2, const Float ss, const Float shk,
                         const std::vector<Float> &M, const int t, const Float sc, Gaussian &gaussian);

struct SplatSample {
    Vector2 screenPos;
    Vector3 contrib;
};

struct MarkovState {  // mlt.h:30-39
    bool valid = false;
    SubpathContrib spContrib;
    Path path;
    Float scoreSum = 0;
    std::vector<Float> pss;
    bool gaussianInitialized = false;
    Gaussian gaussian;
    std::vector<SplatSample> toSplat;
};

// global_cache.h: constants :8-14
#define PSS_MIN_LENGTH 2
#define PSS_MAX_LENGTH 12
#define PSS_MAX_SIZE 3000
const Float PSS_QUERY_DIST = Float(0.01);
const Float PSS_REUSE_DIST = Float(0.10);
const Float CACHE_SIG = Float(0.15);   // global_cache.h:13-14
const Float CACHE_PROB = Float(0.50);

// kd-tree restating nanoflann's KDTreeSingleIndexAdaptor<L2_Simple_Adaptor> build (max leaf 10) and the
// reference-modified radiusSearch (stop after `knn` matches in traversal order, then sort by distance):
// /root/reference/nanoflann/include/nanoflann.hpp:225-262,867-1007,1291-1300,1359-1422.
struct KdTree {
    struct Node {
        int child1 = -1, child2 = -1;  // -1,-1 = leaf
        int left = 0, right = 0;       // leaf: vind range
        int divfeat = 0;
        Float divlow = 0, divhigh = 0;
    };
    int dim = 0;
    std::vector<Node> nodes;
    std::vector<int> vind;
    std::vector<Float> rootLow, rootHigh;
    const Float *pts = nullptr;  // n x dim row-major
    int n = 0;
    void Build(const Float *pts, int n, int dim);
    // returns number of matches (<= knn); idx/dist in traversal order (the reference does not sort)
    int RadiusSearch(const Float *q, Float radiusSq, int knn, int *idx, Float *dist) const;
};

struct CacheDim {  // global_cache_t<dim>, global_cache.h:33-124 (sampleCache/evalPdfCache are SURVEY.md §8f.4)
    int dim = 0;
    int data_idx = 0;
    bool is_ready = false;
    std::vector<Float> pss, v1, v2;  // PSS_MAX_SIZE x dim
    std::vector<Float> pathWeight;
    // what LargeStepCache reads (global_cache.h:21-23,42-43,47,57-58,84-90): the path and contribution of every row, the running
    // double sum of the weights, PiecewiseConstant1D over the weights once the cache is built, the Gaussian kernel's constants
    std::vector<Path> rowPath;
    std::vector<SubpathContrib> rowContrib;
    double score_sum = 0;
    std::vector<Float> distFunc, distCdf;
    Float distFuncInt = 0;
    Float inv_sigma_sq = 0, factor = 0;
    KdTree tree;
    bool push(const Float *pss_, const Float *v1_, const Float *v2_, Float weight, const Path &path, const SubpathContrib &spContrib);
    int sampleCache(Float u) const;                                             // global_cache.h:126-137: the row index
    Float evalPdfCache(const std::vector<Float> &pss_query, const Path &path) const;  // global_cache.h:139-164
    bool query(const std::vector<Float> &pss_, std::vector<Float> &v1_, std::vector<Float> &v2_) const;
};

struct GlobalCache {
    CacheDim dims[17];
    GlobalCache() {
        for (int d = 0; d < 17; d++) dims[d].dim = d;
    }
    bool isReady(int dim) const { return dim >= 2 && dim <= 16 && dims[dim].is_ready; }
};

enum class MutationType { Large, Small, H2MCSmall, MALASmall };

struct Chain {  // mutation.h:28-43
    std::vector<Float> pss, last_pss, v1, v2, g, M;
    std::vector<Float> curr_new_v1, curr_new_v2, curr_new_g;
    std::vector<Float> prop_new_v1, prop_new_v2, prop_new_g;
    Float pathWeight = 0;
    Path path;                // mutation_mala.h:90-91,185-186: the state the last Gaussian was initialised for
    SubpathContrib spContrib;
    bool buffered = false;
    Float ss = 0;
    int chainId = 0, t = 0;
    bool queried = false;
};

typedef void (*PathFuncDerv)(const Float *, const Float *, const Float *, const Float *, Float *, Float *);
typedef void (*PathFunc)(const Float *, const Float *, const Float *, const Float *, Float *);

// the reference's own generated programs (oracle/_ref/libpathref.so), resolved with dlsym like chad.cpp:1009-1016
struct PathFuncLib {
    void *handle = nullptr;
    std::map<std::pair<int, int>, PathFunc> funcMap;
    std::map<std::pair<int, int>, PathFuncDerv> dervMap;
    std::map<std::pair<int, int>, PathFuncDerv> hessMap;  // H2MC library (pathlibbidir.so): evaluate_path_bidir_<c>_<l>_static_derv(..., grad, hess)
    int maxDepth = 8;
    bool Load(const char *soPath, int maxDepth);
};

struct PendingPush {
    int dim;
    std::vector<Float> pss, v1, v2;
    Float weight;
    Path path;
    SubpathContrib spContrib;
};

struct alignas(128) StepStats {  // padded: one instance per worker thread in the MT baseline (no false sharing)
    int64_t steps = 0, largeSteps = 0, accepted = 0, gradCalls = 0, cacheQueries = 0, cacheHits = 0, resets = 0;
    double weightSum = 0;  // sum over steps of the total splat weight (1 if the current state is valid, else a): film luminance = normalization * weightSum
};

struct ChainCtx {  // per-chain objects of the ParallelFor body, mlt.cpp:60-90
    RNG rng;
    MarkovState currentState, proposalState;
    int64_t adjacentReject = 0;
    Float lastScoreSum = Float(1.0), lastScore = Float(1.0);  // LargeStep members, mutation_large.h:14-15
    Chain chain;
    int64_t numSamplesThisChain = 0;
    int64_t sampleIdx = 0;
    MutationType lastSmallType = MutationType::Small;
    std::vector<Float> *film = nullptr;  // where this chain splats (MLT::film, or a thread-private buffer in the MT bench)
    StepStats *st = nullptr;
    ChainCtx() : rng(0) {}
};

struct MLT {
    std::unique_ptr<RScene> scene;
    PathFuncLib lib;
    std::vector<MarkovState> initStates;
    std::vector<Float> lengthContrib;
    // lengthDist (mlt.h:99): PiecewiseConstant1D(lengthContrib), distribution.h:8-60
    std::vector<Float> lengthFunc, lengthCdf;
    Float lengthFuncInt = 0;
    Float LengthPmf(int length) const { return lengthFunc[length] / (lengthFuncInt * Float(lengthFunc.size())); }  // distribution.h:51-53
    Float normalization = 0;
    GlobalCache cache;
    std::vector<ChainCtx> chains;
    std::vector<Float> film;  // W*H*3, indirect buffer (un-normalised, like indirectBuffer in mlt.cpp:54)
    StepStats stats;
    int initThreads = 1;
    int64_t numInitContribs = 0;
    std::vector<int64_t> initContribSample;  // MLTInit contributions in stream order: global sample index, technique, lsScore (parity probe)
    std::vector<int> initContribCL;
    std::vector<Float> initContribLs;

    // mlt.h:41-154 with NumSystemCores() := initThreads (deterministic order: thread-major)
    Float Init(int64_t numInitSamples, int numChains, int initThreads);
    void SetupChains(int64_t numSamplesPerChain, int64_t chainsNeedExtraSamples, int chainBegin = 0, int chainEnd = -1);
    // one lock-step iteration of the per-chain loop body (mlt.cpp:91-170) for every chain that still has samples
    void StepAll();
    void StepChain(ChainCtx &c, std::vector<PendingPush> &pushes);
    Float LargeStepMutate(ChainCtx &c);
    Float LargeStepCacheMutate(ChainCtx &c);  // mutation_large_cache.h:22-141 (`samplecache` with mala)
    Float SmallStepMutate(ChainCtx &c);
    Float MALAMutate(ChainCtx &c);
    Float H2MCMutate(ChainCtx &c);  // mutation_h2mc.h:38-128
    void InitGaussianFor(ChainCtx &c, MarkovState &state, bool isProposal);
    void Splat(std::vector<Float> &film, const Vector2 screenPos, const Vector3 &contrib);
};

}  // namespace orc
