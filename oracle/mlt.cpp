// ORACLE -- TEST INFRASTRUCTURE ONLY (see common.h).
// MLT initialisation, the per-chain mutation loop, LargeStep / SmallStep / MALASmallStep, the diagonal
// Gaussian proposal and the global gradient cache.  Restates /root/reference/src/{mlt.h,mlt.cpp,
// mutation_large.h,mutation_small.h,mutation_mala.h,mala.cpp,gaussian.cpp,global_cache.h}.
//
// Scheduling: the reference runs one chain per worker thread to completion (mlt.cpp:60) and lets chains
// race on the global cache.  The oracle advances ALL chains in lock step (step-major) and applies the
// cache pushes of a step after the step, in chain-id order -- one legal interleaving of the reference,
// made deterministic; it is the contract the HIP back end is compared against.
#include "mlt.h"
#include "h2mc_serial.h"  // the oracle's serial H2MC Gaussian (Jacobi eigen-solver); the device has its own 16-lane one

#include <dlfcn.h>

#include <cstdio>
#include <stdexcept>

namespace orc {

// ============================================================================================ gaussian.cpp
void IsotropicGaussian(const int dim, const Float sigma, Gaussian &gaussian) {  // gaussian.cpp:4-22
    gaussian.isDiagonal = false;
    gaussian.dense = false;  // the dense matrices the reference fills here are diagonal: the diagonal arithmetic below is the same
    gaussian.mean.assign(dim, Float(0.0));
    gaussian.covL_d.assign(dim, sigma);
    gaussian.invCov_d.assign(dim, Float(1.0) / (sigma * sigma));
    gaussian.logDet = dim * fastlog(Float(1.0) / (sigma * sigma));
}

Float GaussianLogPdf(const std::vector<Float> &offset, const Gaussian &gaussian, bool negate) {  // gaussian.cpp:24-36
    const int dim = (int)gaussian.mean.size();
    if (gaussian.dense) return lmcd::DenseGaussianLogPdf(dim, offset.data(), negate, gaussian.mean.data(), lmcd::MatRef{const_cast<Float *>(gaussian.invCov.data()), 1}, gaussian.logDet);
    Float logPdf = dim * (-Float(0.9189385332046727));
    logPdf += Float(0.5) * gaussian.logDet;
    // d^T (invCov d); Eigen's reduction order is unpinned -- summed left to right here.
    Float q = 0;
    for (int i = 0; i < dim; i++) {
        Float d = (negate ? -offset[i] : offset[i]) - gaussian.mean[i];
        q += d * (gaussian.invCov_d[i] * d);
    }
    logPdf -= Float(0.5) * q;
    return logPdf;
}

void GenerateSample(Gaussian &gaussian, std::vector<Float> &x, RNG &rng) {  // gaussian.cpp:38-55
    std::normal_distribution<Float> normDist(Float(0.0), Float(1.0));
    for (size_t i = 0; i < x.size(); i++) x[i] = normDist(rng);
    if (gaussian.dense) {
        std::vector<Float> z(x);
        lmcd::DenseGaussianMap((int)x.size(), z.data(), gaussian.mean.data(), lmcd::MatRef{gaussian.covL.data(), 1}, x.data());
        return;
    }
    for (size_t i = 0; i < x.size(); i++) x[i] = gaussian.covL_d[i] * x[i] + gaussian.mean[i];
}

const Float PCD_MIN = Float(0.01), PCD_MAX = Float(100), MTM_MIN = Float(-5.0), MTM_MAX = Float(5.0), LS_RATIO = Float(0.1);  // mala.h:9-13

void ComputeGaussianMALA(const int dim, const std::vector<Float> &v1, const std::vector<Float> & /*v2*/, const Float ss, const Float shk,
                         const std::vector<Float> &M, const int /*t*/, const Float sc, Gaussian &gaussian) {  // mala.cpp:7-52
    gaussian.isDiagonal = true;
    gaussian.logDet = Float(0.0);
    const Float shrk = inverse(shk * shk);
    gaussian.mean.assign(dim, Float(0.0));
    gaussian.covL_d.assign(dim, Float(0.0));
    gaussian.invCov_d.assign(dim, Float(0.0));
    if (sc <= Float(1e-10)) {
        for (int i = 0; i < dim; i++) {
            gaussian.mean[i] = Float(0.0);
            gaussian.invCov_d[i] = shrk;
            gaussian.covL_d[i] = shk;
        }
        gaussian.logDet = dim * fastlog(inverse(shk * shk));
    } else {
        for (int i = 0; i < dim; i++) {
            Float cov_t = ss * ss * (M[i] + Float(1.0));
            Float invcov = inverse(cov_t) + shrk;
            Float cov = inverse(invcov);
            gaussian.invCov_d[i] = invcov;
            gaussian.covL_d[i] = std::sqrt(cov);
            gaussian.mean[i] = Clamp(v1[i], MTM_MIN, MTM_MAX) * cov / 2;
            gaussian.logDet += fastlog(invcov);
        }
    }
}

// ============================================================================================ kd-tree
namespace {
struct Interval {
    Float low, high;
};
}  // namespace

static void ComputeMinMax(const KdTree &t, const int *ind, int count, int element, Float &min_elem, Float &max_elem) {
    min_elem = t.pts[(size_t)ind[0] * t.dim + element];
    max_elem = min_elem;
    for (int i = 1; i < count; ++i) {
        Float val = t.pts[(size_t)ind[i] * t.dim + element];
        if (val < min_elem) min_elem = val;
        if (val > max_elem) max_elem = val;
    }
}

static void PlaneSplit(const KdTree &t, int *ind, const int count, int cutfeat, Float cutval, int &lim1, int &lim2) {
    auto get = [&](int i) { return t.pts[(size_t)ind[i] * t.dim + cutfeat]; };
    int left = 0, right = count - 1;
    for (;;) {
        while (left <= right && get(left) < cutval) ++left;
        while (right && left <= right && get(right) >= cutval) --right;
        if (left > right || !right) break;
        std::swap(ind[left], ind[right]);
        ++left;
        --right;
    }
    lim1 = left;
    right = count - 1;
    for (;;) {
        while (left <= right && get(left) <= cutval) ++left;
        while (right && left <= right && get(right) > cutval) --right;
        if (left > right || !right) break;
        std::swap(ind[left], ind[right]);
        ++left;
        --right;
    }
    lim2 = left;
}

static int DivideTree(KdTree &t, int left, int right, std::vector<Interval> &bbox) {
    int ni = (int)t.nodes.size();
    t.nodes.push_back(KdTree::Node());
    const int dim = t.dim;
    if ((right - left) <= 10) {  // KDTreeSingleIndexAdaptorParams(10), global_cache.h:86
        t.nodes[ni].left = left;
        t.nodes[ni].right = right;
        for (int i = 0; i < dim; ++i) bbox[i].low = bbox[i].high = t.pts[(size_t)t.vind[left] * dim + i];
        for (int k = left + 1; k < right; ++k)
            for (int i = 0; i < dim; ++i) {
                Float v = t.pts[(size_t)t.vind[k] * dim + i];
                if (bbox[i].low > v) bbox[i].low = v;
                if (bbox[i].high < v) bbox[i].high = v;
            }
        return ni;
    }
    // middleSplit_
    int *ind = &t.vind[0] + left;
    const int count = right - left;
    const Float EPS = Float(0.00001);
    Float max_span = bbox[0].high - bbox[0].low;
    for (int i = 1; i < dim; ++i) {
        Float span = bbox[i].high - bbox[i].low;
        if (span > max_span) max_span = span;
    }
    Float max_spread = -1;
    int cutfeat = 0;
    for (int i = 0; i < dim; ++i) {
        Float span = bbox[i].high - bbox[i].low;
        if (span > (1 - EPS) * max_span) {
            Float min_elem, max_elem;
            ComputeMinMax(t, ind, count, i, min_elem, max_elem);
            Float spread = max_elem - min_elem;
            if (spread > max_spread) {
                cutfeat = i;
                max_spread = spread;
            }
        }
    }
    Float split_val = (bbox[cutfeat].low + bbox[cutfeat].high) / 2;
    Float min_elem, max_elem;
    ComputeMinMax(t, ind, count, cutfeat, min_elem, max_elem);
    Float cutval;
    if (split_val < min_elem)
        cutval = min_elem;
    else if (split_val > max_elem)
        cutval = max_elem;
    else
        cutval = split_val;
    int lim1, lim2;
    PlaneSplit(t, ind, count, cutfeat, cutval, lim1, lim2);
    int idx;
    if (lim1 > count / 2)
        idx = lim1;
    else if (lim2 < count / 2)
        idx = lim2;
    else
        idx = count / 2;
    t.nodes[ni].divfeat = cutfeat;
    std::vector<Interval> left_bbox(bbox);
    left_bbox[cutfeat].high = cutval;
    int c1 = DivideTree(t, left, left + idx, left_bbox);
    std::vector<Interval> right_bbox(bbox);
    right_bbox[cutfeat].low = cutval;
    int c2 = DivideTree(t, left + idx, right, right_bbox);
    t.nodes[ni].child1 = c1;
    t.nodes[ni].child2 = c2;
    t.nodes[ni].divlow = left_bbox[cutfeat].high;
    t.nodes[ni].divhigh = right_bbox[cutfeat].low;
    for (int i = 0; i < dim; ++i) {
        bbox[i].low = std::min(left_bbox[i].low, right_bbox[i].low);
        bbox[i].high = std::max(left_bbox[i].high, right_bbox[i].high);
    }
    return ni;
}

void KdTree::Build(const Float *p, int n_, int dim_) {
    pts = p, n = n_, dim = dim_;
    nodes.clear();
    vind.resize(n);
    for (int i = 0; i < n; i++) vind[i] = i;
    std::vector<Interval> bbox(dim);
    for (int i = 0; i < dim; ++i) bbox[i].low = bbox[i].high = pts[i];
    for (int k = 1; k < n; ++k)
        for (int i = 0; i < dim; ++i) {
            Float v = pts[(size_t)k * dim + i];
            if (v < bbox[i].low) bbox[i].low = v;
            if (v > bbox[i].high) bbox[i].high = v;
        }
    DivideTree(*this, 0, n, bbox);
    rootLow.resize(dim), rootHigh.resize(dim);
    for (int i = 0; i < dim; i++) rootLow[i] = bbox[i].low, rootHigh[i] = bbox[i].high;
}

namespace {
struct RadiusSet {
    Float radius;
    int knn, count = 0;
    int *idx;
    Float *dist;
    bool addPoint(Float d, int index) {  // nanoflann.hpp:256-262
        if (d < radius) {
            idx[count] = index;
            dist[count] = d;
            count++;
        }
        if (count >= knn) return false;
        return true;
    }
};
}  // namespace

static bool SearchLevel(const KdTree &t, RadiusSet &rs, const Float *vec, int node, Float mindistsq, Float *dists) {
    const KdTree::Node &nd = t.nodes[node];
    if (nd.child1 < 0 && nd.child2 < 0) {
        Float worst_dist = rs.radius;
        for (int i = nd.left; i < nd.right; ++i) {
            const int index = t.vind[i];
            Float dist = 0;  // L2_Simple_Adaptor::evalMetric: sequential sum of squared differences
            for (int k = 0; k < t.dim; ++k) {
                const Float diff = vec[k] - t.pts[(size_t)index * t.dim + k];
                dist += diff * diff;
            }
            if (dist < worst_dist) {
                if (!rs.addPoint(dist, index)) return false;
            }
        }
        return true;
    }
    int idx = nd.divfeat;
    Float val = vec[idx];
    Float diff1 = val - nd.divlow;
    Float diff2 = val - nd.divhigh;
    int bestChild, otherChild;
    Float cut_dist;
    if ((diff1 + diff2) < 0) {
        bestChild = nd.child1;
        otherChild = nd.child2;
        cut_dist = (val - nd.divhigh) * (val - nd.divhigh);
    } else {
        bestChild = nd.child2;
        otherChild = nd.child1;
        cut_dist = (val - nd.divlow) * (val - nd.divlow);
    }
    if (!SearchLevel(t, rs, vec, bestChild, mindistsq, dists)) return false;
    Float dst = dists[idx];
    mindistsq = mindistsq + cut_dist - dst;
    dists[idx] = cut_dist;
    if (mindistsq * Float(1.0) <= rs.radius) {  // epsError = 1 + eps, eps = 0
        if (!SearchLevel(t, rs, vec, otherChild, mindistsq, dists)) return false;
    }
    dists[idx] = dst;
    return true;
}

int KdTree::RadiusSearch(const Float *q, Float radiusSq, int knn, int *idx, Float *dist) const {
    if (n == 0) return 0;
    Float dists[16];
    for (int i = 0; i < dim; i++) dists[i] = 0;
    Float distsq = 0;
    for (int i = 0; i < dim; ++i) {  // computeInitialDistances
        if (q[i] < rootLow[i]) {
            dists[i] = (q[i] - rootLow[i]) * (q[i] - rootLow[i]);
            distsq += dists[i];
        }
        if (q[i] > rootHigh[i]) {
            dists[i] = (q[i] - rootHigh[i]) * (q[i] - rootHigh[i]);
            distsq += dists[i];
        }
    }
    RadiusSet rs{radiusSq, knn, 0, idx, dist};
    SearchLevel(*this, rs, q, 0, distsq, dists);
    // No sort: the reference's nanoflann has SearchParams::sorted = false by default (nanoflann.hpp:567),
    // so matches stay in kd-tree traversal order (pinned by tests/test_oracle_pins.py).
    return rs.count;
}

// ============================================================================================ global cache
bool CacheDim::push(const Float *pss_, const Float *v1_, const Float *v2_, Float weight, const Path &path, const SubpathContrib &spContrib) {  // global_cache.h:70-94
    if (is_ready) return false;
    if (pss.empty()) {
        rowPath.resize(PSS_MAX_SIZE), rowContrib.resize(PSS_MAX_SIZE);
        inv_sigma_sq = inverse(CACHE_SIG * CACHE_SIG);  // global_cache.h:57-58; exp / log: the shared float routines (dtrans.h) stand in for libm
        factor = lmcd::lexpf(dim * (Float(0.5) * lmcd::llogf(inv_sigma_sq) - Float(0.9189385332046727)));
        pss.resize((size_t)PSS_MAX_SIZE * dim);
        v1.resize((size_t)PSS_MAX_SIZE * dim);
        v2.resize((size_t)PSS_MAX_SIZE * dim);
        pathWeight.resize(PSS_MAX_SIZE);
    }
    for (int i = 0; i < dim; i++) {
        pss[(size_t)data_idx * dim + i] = pss_[i];
        v1[(size_t)data_idx * dim + i] = v1_[i];
        v2[(size_t)data_idx * dim + i] = v2_[i];
    }
    pathWeight[data_idx] = weight;
    rowPath[data_idx] = path, rowContrib[data_idx] = spContrib;
    score_sum += weight;
    data_idx += 1;
    if (data_idx >= PSS_MAX_SIZE) {
        tree.Build(pss.data(), PSS_MAX_SIZE, dim);
        lmc::BuildPiecewise1D(pathWeight.data(), PSS_MAX_SIZE, distFunc, distCdf, distFuncInt);  // data_distrib, global_cache.h:88-89
        is_ready = true;
    }
    return true;
}

int CacheDim::sampleCache(Float u) const { return lmc::SampleDiscrete1D(distFunc, distCdf, distFuncInt, u, nullptr); }

Float CacheDim::evalPdfCache(const std::vector<Float> &pss_query, const Path &path) const {  // global_cache.h:139-164
    Float ret(0.0);
    for (int i = 0; i < PSS_MAX_SIZE; i++) {
        const SubpathContrib &spContrib = rowContrib[i];
        if (spContrib.camDepth != path.camDepth || spContrib.lightDepth != path.lgtDepth) continue;
        Float sumDistSqr = 0;
        for (int j = 0; j < dim; j++) {
            Float c = pss[(size_t)i * dim + j], q = pss_query[j];
            Float d1 = std::fabs(q - c);
            Float d2 = Float(1.0) - d1;
            Float d = std::min(d1, d2);
            sumDistSqr += d * d;
        }
        Float expo = -Float(0.5) * sumDistSqr * inv_sigma_sq;
        Float scale = Float(double(factor * pathWeight[i]) / score_sum);  // Float * Float / double -> Float
        ret += lmcd::lexpf(expo) * scale;
    }
    return ret;
}

bool CacheDim::query(const std::vector<Float> &pss_, std::vector<Float> &v1_, std::vector<Float> &v2_) const {  // global_cache.h:96-124
    if (!is_ready) return false;
    const int knn = 5;
    const Float radius = dim * (PSS_QUERY_DIST * PSS_QUERY_DIST);
    int idx[5];
    Float dist[5];
    const int nMatches = tree.RadiusSearch(pss_.data(), radius, knn, idx, dist);
    if (!nMatches) return false;
    double sum_w = 0;
    std::fill(v1_.begin(), v1_.end(), Float(0.0));
    std::fill(v2_.begin(), v2_.end(), Float(0.0));
    for (int k = 0; k < nMatches; k++) {
        int index = idx[k];
        Float d = dist[k];
        Float w = inverse(d * d + Float(1e-6));
        for (int i = 0; i < dim; i++) {
            v1_[i] += v1[(size_t)index * dim + i] * w;
            v2_[i] += v2[(size_t)index * dim + i] * w;
        }
        sum_w += w;
    }
    for (int i = 0; i < dim; i++) {
        v1_[i] /= sum_w;  // float /= double: computed in double, rounded to float
        v2_[i] /= sum_w;
    }
    return true;
}

// ============================================================================================ path-function library
bool PathFuncLib::Load(const char *soPath, int maxDepth_) {  // path.cpp:4021-4062
    maxDepth = maxDepth_;
    handle = dlopen(soPath, RTLD_LAZY);
    if (!handle) return false;
    for (int c = 1; c <= maxDepth + 1; c++)
        for (int l = 0; l <= maxDepth; l++) {
            if (c + l <= 2 || (c + l - 1) > maxDepth) continue;
            char name[128];
            snprintf(name, sizeof(name), "evaluate_path_bidir_mala_%d_%d_static", c, l);
            void *f = dlsym(handle, name);
            snprintf(name, sizeof(name), "evaluate_path_bidir_mala_%d_%d_static_derv", c, l);
            void *d = dlsym(handle, name);
            if (f && d) {
                funcMap[{c, l}] = (PathFunc)f;
                dervMap[{c, l}] = (PathFuncDerv)d;
            }
            snprintf(name, sizeof(name), "evaluate_path_bidir_%d_%d_static_derv", c, l);
            if (void *h2 = dlsym(handle, name)) hessMap[{c, l}] = (PathFuncDerv)h2;
        }
    return true;
}

// ============================================================================================ MLTInit
namespace {
struct LightMarkovState {
    int threadId;
    uint64_t rngState;  // base LCG state at the start of the sample
    uint32_t ticks;     // extension-table advances so far in this thread's stream (pcg tick, rng.h)
    int camDepth, lightDepth;
    Float lsScore;
};
}  // namespace

static RNG RngFromCheckpoint(uint64_t seed, uint64_t state, uint32_t ticks) {
    RNG r(seed);
    for (uint32_t i = 0; i < ticks; i++) r.AdvanceTable();
    r.state = state;
    return r;
}

Float MLT::Init(int64_t numInitSamples, int numChains, int initThreads_) {  // mlt.h:41-154
    initThreads = std::max(1, initThreads_);
    const RScene *sc = scene.get();
    const int64_t numSamplesPerThread = numInitSamples / initThreads;
    const int64_t threadsNeedExtraSamples = numInitSamples % initThreads;
    std::vector<LightMarkovState> mStates;
    Float totalScore(Float(0.0));
    lengthContrib.clear();
    const int minPathLength = std::max(sc->options->minDepth, 3);
    std::vector<SubpathContrib> spContribs;
    Path path;
    initContribSample.clear(), initContribCL.clear(), initContribLs.clear();
    int64_t globalSample = 0;
    for (int threadId = 0; threadId < initThreads; threadId++) {
        const uint64_t seed = (uint64_t)(threadId + sc->options->seedOffset);
        RNG rng(seed);
        uint32_t ticks = 0;
        int64_t n = numSamplesPerThread + ((threadId < threadsNeedExtraSamples) ? 1 : 0);
        for (int64_t sampleIdx = 0; sampleIdx < n; sampleIdx++) {
            spContribs.clear();
            uint64_t stateCheckpoint = rng.state;
            uint32_t ticksCheckpoint = ticks;
            uint32_t tab0 = rng.data[0] ^ rng.data[17] ^ rng.data[63];
            Clear(path);
            GeneratePathBidir(sc, -1, -1, minPathLength, sc->options->maxDepth, path, spContribs, rng);
            if ((rng.data[0] ^ rng.data[17] ^ rng.data[63]) != tab0) ticks++;  // at most one tick per sample in practice (p ~ 2^-32 per draw)
            for (const auto &spContrib : spContribs) {
                totalScore += spContrib.lsScore;
                const int pathLength = GetPathLength(spContrib.camDepth, spContrib.lightDepth);
                if (pathLength >= int(lengthContrib.size())) lengthContrib.resize(pathLength + 1, Float(0.0));
                lengthContrib[pathLength] += spContrib.lsScore;
                mStates.push_back(LightMarkovState{threadId, stateCheckpoint, ticksCheckpoint, spContrib.camDepth, spContrib.lightDepth, spContrib.lsScore});
                initContribSample.push_back(globalSample), initContribCL.push_back(spContrib.camDepth * 16 + spContrib.lightDepth), initContribLs.push_back(spContrib.lsScore);
            }
            globalSample++;
        }
    }
    numInitContribs = (int64_t)mStates.size();
    if (int(mStates.size()) < numChains)
        throw std::runtime_error("MLT initialization failed, consider using a larger number of initial samples or smaller number of chains");
    std::vector<Float> cdf(mStates.size() + 1);
    cdf[0] = Float(0.0);
    for (int i = 0; i < (int)mStates.size(); i++) cdf[i + 1] = cdf[i] + mStates[i].lsScore;
    const Float interval = cdf.back() / Float(numChains);
    std::uniform_real_distribution<Float> uniDist(Float(0.0), interval);
    RNG rng(mStates.size());
    Float pos = uniDist(rng);
    int cdfPos = 0;
    initStates.clear();
    initStates.reserve(numChains);
    for (int i = 0; i < numChains; i++) {
        // mlt.h:118-120 clamps cdfPos to size-1 inside the loop, which spins forever once pos exceeds cdf[size-1]
        // (reachable for the last chains); the oracle (and the HIP host code) stop at size-1 instead.
        while (pos > cdf[cdfPos] && cdfPos < int(mStates.size()) - 1) cdfPos++;
        initStates.push_back(MarkovState());
        MarkovState &state = initStates.back();
        state.valid = false;
        spContribs.clear();
        Clear(state.path);
        const LightMarkovState &ms = mStates[std::max(cdfPos - 1, 0)];
        RNG rngCheckpoint = RngFromCheckpoint((uint64_t)(ms.threadId + sc->options->seedOffset), ms.rngState, ms.ticks);
        GeneratePathBidir(sc, -1, -1, minPathLength, sc->options->maxDepth, state.path, spContribs, rngCheckpoint);
        state.scoreSum = Float(0.0);
        for (const auto &spContrib : spContribs) {
            state.scoreSum += spContrib.lsScore;
            if (spContrib.camDepth == ms.camDepth && spContrib.lightDepth == ms.lightDepth) state.spContrib = spContrib;
        }
        ToSubpath(state.spContrib.camDepth, state.spContrib.lightDepth, state.path);
        state.pss.clear();
        GetPathPss(state.path, state.pss);
        state.gaussianInitialized = false;
        pos += interval;
    }
    // lengthDist, mlt.h:99 (PiecewiseConstant1D over the per-length score sums; the multiplexed large step samples it)
    lmc::BuildPiecewise1D(lengthContrib.data(), (int)lengthContrib.size(), lengthFunc, lengthCdf, lengthFuncInt);
    normalization = Float(totalScore) * inverse(Float(numInitSamples));
    return normalization;
}

void MLT::SetupChains(int64_t numSamplesPerChain, int64_t chainsNeedExtraSamples, int chainBegin, int chainEnd) {  // mlt.cpp:60-90
    const int numChainsTotal = (int)initStates.size();
    if (chainEnd < 0) chainEnd = numChainsTotal;
    chains.clear();
    chains.resize(chainEnd - chainBegin);
    // chains [chainBegin, chainEnd) of the global set (multi-process sharding); ids and seeds stay global
    for (int chainId = chainBegin; chainId < chainEnd; chainId++) {
        ChainCtx &c = chains[chainId - chainBegin];
        c.rng = RNG((uint64_t)(chainId + scene->options->seedOffset));
        c.numSamplesThisChain = numSamplesPerChain + ((chainId < chainsNeedExtraSamples) ? 1 : 0);
        c.currentState = initStates[chainId];
        c.proposalState = MarkovState();
        c.chain.chainId = chainId;
        c.chain.ss = scene->options->malaStepsize;
        c.film = &film;
        c.st = &stats;
    }
    film.assign((size_t)scene->camera.pixelWidth * scene->camera.pixelHeight * 3, Float(0.0));
    stats = StepStats();
}

void MLT::Splat(std::vector<Float> &film, const Vector2 screenPos, const Vector3 &contrib) {  // image.h:66-77
    const int W = scene->camera.pixelWidth, H = scene->camera.pixelHeight;
    int ix = Clamp(int(screenPos[0] * W), 0, W - 1);
    int iy = Clamp(int(screenPos[1] * H), 0, H - 1);
    if (contrib.allFinite()) {
        Float *px = &film[((size_t)iy * W + ix) * 3];
        for (int i = 0; i < 3; i++) px[i] += contrib[i];
    }
}

// ============================================================================================ mutations
Float MLT::LargeStepMutate(ChainCtx &c) {  // mutation_large.h:31-128
    std::uniform_real_distribution<Float> uniDist(Float(0.0), Float(1.0));
    const RScene *sc = scene.get();
    MarkovState &currentState = c.currentState, &proposalState = c.proposalState;
    Float a = Float(1.0);
    std::vector<SubpathContrib> spContribs;
    Clear(proposalState.path);
    const bool multiplexed = sc->options->largeStepMultiplexed;
    if (multiplexed) {  // mutation_large.h:45-58: one technique of a length drawn from lengthDist
        int length = lmc::SampleDiscrete1D(lengthFunc, lengthCdf, lengthFuncInt, uniDist(c.rng), nullptr);
        int lgtLength = Clamp(int(uniDist(c.rng) * (length + 1)), 0, length);  // options->bidirectional
        int camLength = length - lgtLength + 1;
        GenerateSubpath(sc, camLength, lgtLength, true, proposalState.path, spContribs, c.rng);
    } else {
        GeneratePathBidir(sc, -1, -1, std::max(sc->options->minDepth, 3), sc->options->maxDepth, proposalState.path, spContribs, c.rng);
    }
    proposalState.gaussianInitialized = false;
    if (spContribs.size() > 0) {
        std::vector<Float> contribCdf;
        contribCdf.push_back(Float(0.0));
        for (const auto &spContrib : spContribs) contribCdf.push_back(contribCdf.back() + spContrib.lsScore);
        const Float scoreSum = contribCdf.back();
        const Float invSc = inverse(scoreSum);
        for (auto &v : contribCdf) v *= invSc;
        const auto it = std::upper_bound(contribCdf.begin(), contribCdf.end(), uniDist(c.rng));
        int64_t contribId = Clamp(int64_t(it - contribCdf.begin() - 1), int64_t(0), int64_t(spContribs.size() - 1));
        proposalState.spContrib = spContribs[contribId];
        proposalState.scoreSum = scoreSum;
        if (currentState.valid && multiplexed) {  // mutation_large.h:87-102
            int currentLength = GetPathLength(currentState.spContrib.camDepth, currentState.spContrib.lightDepth);
            int proposalLength = GetPathLength(proposalState.spContrib.camDepth, proposalState.spContrib.lightDepth);
            Float invProposalTechniquesPmf = Float(proposalLength) + Float(1.0);
            Float invCurrentTechniquesPmf = Float(currentLength) + Float(1.0);
            a = Clamp((invProposalTechniquesPmf * proposalState.spContrib.lsScore / LengthPmf(proposalLength)) /
                          (invCurrentTechniquesPmf * currentState.spContrib.lsScore / LengthPmf(currentLength)),
                      Float(0.0), Float(1.0));
        } else if (currentState.valid) {
            const Float probProposal = (proposalState.spContrib.lsScore / proposalState.scoreSum);
            const Float probLast = (c.lastScore / c.lastScoreSum);
            a = Clamp((proposalState.spContrib.lsScore * probLast) / (currentState.spContrib.lsScore * probProposal), Float(0.0), Float(1.0));
        }
        proposalState.toSplat.clear();
        for (const auto &spContrib : spContribs)
            proposalState.toSplat.push_back(SplatSample{spContrib.screenPos, spContrib.contrib * (normalization / scoreSum)});
    } else {
        a = Float(0.0);
    }
    return a;
}

// LargeStepCache::Mutate, mutation_large_cache.h:22-141: a large step that, once the global cache of the proposed dimension is
// built, proposes half of the time a Gaussian perturbation (sigma CACHE_SIG) of a cached path drawn by its weight, and weighs the
// two strategies against each other (MIS over the uniform multiplexed sampler and the kernel density of the cache).
Float MLT::LargeStepCacheMutate(ChainCtx &c) {
    std::uniform_real_distribution<Float> uniDist(Float(0.0), Float(1.0));
    const RScene *sc = scene.get();
    MarkovState &currentState = c.currentState, &proposalState = c.proposalState;
    if (!sc->options->largeStepMultiplexed) throw std::runtime_error("samplecache needs largestepmultiplexed (mutation_large_cache.h:33)");
    Float a = Float(1.0);
    std::vector<SubpathContrib> spContribs;
    Clear(proposalState.path);
    const int proposalLength = lmc::SampleDiscrete1D(lengthFunc, lengthCdf, lengthFuncInt, uniDist(c.rng), nullptr);
    const int proposalDim = proposalLength * 2;
    const int currentLength = GetPathLength(currentState.spContrib.camDepth, currentState.spContrib.lightDepth);
    const int currentDim = currentLength * 2;
    bool proposalCacheAvailable = proposalDim >= PSS_MIN_LENGTH && proposalDim <= PSS_MAX_LENGTH && cache.isReady(proposalDim);
    bool currentCacheAvailable = currentDim >= PSS_MIN_LENGTH && currentDim <= PSS_MAX_LENGTH && cache.isReady(currentDim);
    if (!proposalCacheAvailable || uniDist(c.rng) > CACHE_PROB) {  // uniform, as in multiplexed MLT
        int lgtLength = Clamp(int(uniDist(c.rng) * (proposalLength + 1)), 0, proposalLength);
        int camLength = proposalLength - lgtLength + 1;
        GenerateSubpath(sc, camLength, lgtLength, true, proposalState.path, spContribs, c.rng);
        if (spContribs.size() > 0) {
            ToSubpath(spContribs[0].camDepth, spContribs[0].lightDepth, proposalState.path);
            // GetPathPss only sizes an EMPTY vector (path.cpp:2591): the reference writes past the end of a pss left over from a
            // shorter state here (and at :106 below); the restatement grows the vector first, which is what those writes intend
            if ((int)proposalState.pss.size() < GetDimension(proposalState.path)) proposalState.pss.resize(GetDimension(proposalState.path));
            GetPathPss(proposalState.path, proposalState.pss);
        }
    } else {  // from the global cache
        const CacheDim &cd = cache.dims[proposalDim];
        const int idx = cd.sampleCache(uniDist(c.rng));
        proposalState.path = cd.rowPath[idx];
        const SubpathContrib &rowContrib = cd.rowContrib[idx];
        ToSubpath(rowContrib.camDepth, rowContrib.lightDepth, proposalState.path);
        std::normal_distribution<Float> normDist(Float(0.0), CACHE_SIG);
        proposalState.pss.resize(proposalDim);
        std::vector<Float> offset(2 * sc->options->maxDepth, Float(0.0));
        for (int i = 0; i < proposalDim; i++) {
            offset[i] = normDist(c.rng);
            proposalState.pss[i] = Modulo(cd.pss[(size_t)idx * proposalDim + i] + offset[i], Float(1.0));
        }
        PerturbPathBidir(sc, offset, proposalState.path, spContribs, c.rng);
    }
    proposalState.gaussianInitialized = false;
    if (spContribs.size() > 0) {
        proposalState.spContrib = spContribs[0];
        proposalState.scoreSum = spContribs[0].lsScore;
        if (currentState.valid) {
            ToSubpath(currentState.spContrib.camDepth, currentState.spContrib.lightDepth, currentState.path);
            if ((int)currentState.pss.size() < GetDimension(currentState.path)) currentState.pss.resize(GetDimension(currentState.path));
            GetPathPss(currentState.path, currentState.pss);
            const Float proposalJacobian = proposalState.spContrib.ssScore / proposalState.spContrib.lsScore;
            const Float currentJacobian = currentState.spContrib.ssScore / currentState.spContrib.lsScore;
            Float proposalTechniquePickProb = inverse(Float(proposalLength) + Float(1.0));
            Float currentTechniquePickProb = inverse(Float(currentLength) + Float(1.0));
            Float proposalUniformPdf = Float(1.0) * proposalTechniquePickProb * proposalJacobian;  // 1.0 * x: double in the reference, exact either way
            Float currentUniformPdf = Float(1.0) * currentTechniquePickProb * currentJacobian;
            Float proposalCachePdf = proposalCacheAvailable ? cache.dims[proposalDim].evalPdfCache(proposalState.pss, proposalState.path) : Float(0.0);
            Float currentCachePdf = currentCacheAvailable ? cache.dims[currentDim].evalPdfCache(currentState.pss, currentState.path) : Float(0.0);
            Float proposalPdf = !proposalCacheAvailable ? proposalUniformPdf : (1 - CACHE_PROB) * proposalUniformPdf + CACHE_PROB * proposalCachePdf;
            Float currentPdf = !currentCacheAvailable ? currentUniformPdf : (1 - CACHE_PROB) * currentUniformPdf + CACHE_PROB * currentCachePdf;
            a = Clamp(proposalState.spContrib.ssScore * currentPdf * LengthPmf(currentLength) /
                          (currentState.spContrib.ssScore * proposalPdf * LengthPmf(proposalLength)),
                      Float(0.0), Float(1.0));
        }
        proposalState.toSplat.clear();
        for (const auto &spContrib : spContribs)
            proposalState.toSplat.push_back(SplatSample{spContrib.screenPos, spContrib.contrib * (normalization / spContrib.lsScore)});
    } else {
        a = Float(0.0);
    }
    return a;
}

Float MLT::SmallStepMutate(ChainCtx &c) {  // mutation_small.h:16-56
    const RScene *sc = scene.get();
    MarkovState &currentState = c.currentState, &proposalState = c.proposalState;
    std::vector<SubpathContrib> spContribs;
    Float a = Float(1.0);
    proposalState.path = currentState.path;
    const Float stdDev = sc->options->perturbStdDev;
    std::normal_distribution<Float> normDist(Float(0.0), stdDev);
    const int dim = GetDimension(currentState.path);
    std::vector<Float> offset(2 * sc->options->maxDepth, Float(0.0));
    for (int i = 0; i < dim; i++) offset[i] = normDist(c.rng);
    PerturbPathBidir(sc, offset, proposalState.path, spContribs, c.rng);
    proposalState.gaussianInitialized = false;
    if (spContribs.size() > 0) {
        proposalState.spContrib = spContribs[0];
        a = Clamp(proposalState.spContrib.ssScore / currentState.spContrib.ssScore, Float(0.0), Float(1.0));
        proposalState.toSplat.clear();
        for (const auto &spContrib : spContribs)
            proposalState.toSplat.push_back(SplatSample{spContrib.screenPos, spContrib.contrib * (normalization / spContrib.lsScore)});
    } else {
        a = Float(0.0);
    }
    return a;
}

static bool IsFiniteVec(const std::vector<Float> &v) {
    for (Float f : v)
        if (!std::isfinite(f)) return false;
    return true;
}

// The two identical blocks of MALASmallStep::Mutate (mutation_mala.h:83-166 for the current state,
// :174-260 for the proposal): gradient + Adam-style moments while the cache for this dim is not ready,
// cache re-use / kNN query afterwards, isotropic fallback otherwise.
void MLT::InitGaussianFor(ChainCtx &c, MarkovState &state, bool isProposal) {
    const RScene *sc = scene.get();
    Chain *chain = &c.chain;
    const SubpathContrib &cspContrib = state.spContrib;
    auto funcIt = lib.dervMap.find({cspContrib.camDepth, cspContrib.lightDepth});
    const int dim = GetDimension(state.path);
    GetPathPss(state.path, chain->pss);
    chain->path = state.path;
    chain->spContrib = state.spContrib;
    chain->pathWeight = state.spContrib.lsScore;
    std::vector<Float> &new_g = isProposal ? chain->prop_new_g : chain->curr_new_g;
    std::vector<Float> &new_v1 = isProposal ? chain->prop_new_v1 : chain->curr_new_v1;
    std::vector<Float> &new_v2 = isProposal ? chain->prop_new_v2 : chain->curr_new_v2;
    if (dim >= PSS_MIN_LENGTH && dim <= PSS_MAX_LENGTH && !cache.isReady(dim) && funcIt != lib.dervMap.end()) {
        std::vector<Float> vGrad(dim, Float(0.0));
        if (cspContrib.ssScore > Float(1e-10)) {
            SerializedSubpath ssubPath;
            ssubPath.primary.assign(GetPrimaryParamSize(lib.maxDepth, lib.maxDepth), Float(0.0));
            ssubPath.vertParams.assign(GetVertParamSize(lib.maxDepth, lib.maxDepth), Float(0.0));
            Serialize(sc, state.path, ssubPath);
            funcIt->second(&cspContrib.screenPos[0], &ssubPath.primary[0], sc->sceneParams, &ssubPath.vertParams[0], &vGrad[0], NULL);
            c.st->gradCalls++;
            if (!IsFiniteVec(vGrad)) std::fill(vGrad.begin(), vGrad.end(), Float(0.0));
        }
        Float norm(0.0), drift(sc->options->malaGN);
        for (int i = 0; i < dim; i++) norm += vGrad[i] * vGrad[i];
        norm = std::sqrt(norm);
        for (int i = 0; i < dim; i++) vGrad[i] *= drift / std::max(drift, norm);
        bool first = true;
        for (int i = 0; i < dim; i++)
            if (new_v2[i] > Float(1e-10)) {
                first = false;
                break;
            }
        for (int i = 0; i < dim; i++) {
            Float g = vGrad[i];
            new_g[i] = g;
            new_v1[i] = first ? g : Float(0.9) * chain->v1[i] + Float(0.1) * g;
            new_v2[i] = first ? g * g : Float(0.999) * chain->v2[i] + Float(0.001) * g * g;
            chain->M[i] = Clamp(Float(1.0) / Float(Float(1e-3) + std::sqrt(new_v2[i])), PCD_MIN, PCD_MAX);
        }
        ComputeGaussianMALA(dim, new_v1, new_v2, chain->ss, sc->options->malaStdDev, chain->M, chain->t, cspContrib.ssScore, state.gaussian);
    } else {
        if (dim >= PSS_MIN_LENGTH && dim <= PSS_MAX_LENGTH && cache.isReady(dim)) {
            bool reuse = false;
            if (chain->queried) {
                Float dist_sqr(0.f);
                for (int i = 0; i < dim; i++) {
                    Float diff = chain->pss[i] - chain->last_pss[i];
                    dist_sqr += diff * diff;
                }
                if (dist_sqr < dim * (PSS_REUSE_DIST * PSS_REUSE_DIST)) reuse = true;
            }
            auto fromV = [&]() {
                for (int i = 0; i < dim; i++) chain->M[i] = Clamp(Float(1.0) / Float(Float(1e-3) + std::sqrt(chain->v2[i])), PCD_MIN, PCD_MAX);
                ComputeGaussianMALA(dim, chain->v1, chain->v2, chain->ss, sc->options->malaStdDev, chain->M, chain->t, cspContrib.ssScore, state.gaussian);
            };
            if (reuse) {
                fromV();
            } else {
                c.st->cacheQueries++;
                if (cache.dims[dim].query(chain->pss, chain->v1, chain->v2)) {
                    c.st->cacheHits++;
                    chain->queried = true;
                    chain->last_pss = chain->pss;
                    fromV();
                } else {
                    IsotropicGaussian(dim, sc->options->malaStdDev, state.gaussian);
                }
            }
        } else
            IsotropicGaussian(dim, sc->options->malaStdDev, state.gaussian);
    }
    state.gaussianInitialized = true;
}

Float MLT::MALAMutate(ChainCtx &c) {  // mutation_mala.h:35-278
    const RScene *sc = scene.get();
    MarkovState &currentState = c.currentState, &proposalState = c.proposalState;
    Chain *chain = &c.chain;
    std::uniform_real_distribution<Float> uniDist(Float(0.0), Float(1.0));
    if (uniDist(c.rng) < sc->options->uniformMixingProbability) {
        Float a = SmallStepMutate(c);
        c.lastSmallType = MutationType::Small;
        return a;
    }
    std::vector<SubpathContrib> spContribs;
    Float a = Float(1.0);
    c.lastSmallType = MutationType::MALASmall;
    const int dim = GetDimension(currentState.path);
    if (!chain->buffered) {
        const int maxdim = 2 * sc->options->maxDepth;
        for (std::vector<Float> *v : {&chain->M, &chain->pss, &chain->last_pss, &chain->g, &chain->curr_new_g, &chain->prop_new_g, &chain->v1,
                                      &chain->curr_new_v1, &chain->prop_new_v1, &chain->v2, &chain->curr_new_v2, &chain->prop_new_v2})
            v->assign(maxdim, Float(0.0));
        chain->buffered = true;
        chain->queried = false;
    }
    if (!currentState.gaussianInitialized) InitGaussianFor(c, currentState, false);
    std::vector<Float> offset(dim);
    GenerateSample(currentState.gaussian, offset, c.rng);
    proposalState.path = currentState.path;
    PerturbPathBidir(sc, offset, proposalState.path, spContribs, c.rng);
    if (spContribs.size() > 0) {
        proposalState.spContrib = spContribs[0];
        InitGaussianFor(c, proposalState, true);
        Float py = GaussianLogPdf(offset, currentState.gaussian, false);
        Float px = GaussianLogPdf(offset, proposalState.gaussian, true);
        a = Clamp(lmcd::lexpf(px - py) * proposalState.spContrib.ssScore / currentState.spContrib.ssScore, Float(0.0), Float(1.0));
        proposalState.toSplat.clear();
        for (const auto &spContrib : spContribs)
            proposalState.toSplat.push_back(SplatSample{spContrib.screenPos, spContrib.contrib * normalization / spContrib.lsScore});
    } else {
        a = Float(0.0);
    }
    return a;
}

Float MLT::H2MCMutate(ChainCtx &c) {  // mutation_h2mc.h:38-128
    const RScene *sc = scene.get();
    MarkovState &currentState = c.currentState, &proposalState = c.proposalState;
    std::uniform_real_distribution<Float> uniDist(Float(0.0), Float(1.0));
    if (uniDist(c.rng) < sc->options->uniformMixingProbability) {
        Float a = SmallStepMutate(c);
        c.lastSmallType = MutationType::Small;
        return a;
    }
    std::vector<SubpathContrib> spContribs;
    Float a = Float(1.0);
    c.lastSmallType = MutationType::H2MCSmall;
    const int dim = GetDimension(currentState.path);
    const lmcd::H2MCParam param = lmcd::MakeH2MCParam(sc->options->perturbStdDev);  // H2MCSmallStep(scene, maxDervDepth, perturbStdDev), mlt.cpp:76-79
    auto initGaussian = [&](MarkovState &state) {
        const SubpathContrib &csp = state.spContrib;
        auto funcIt = lib.hessMap.find({csp.camDepth, csp.lightDepth});
        const int d = GetDimension(state.path);
        if (funcIt != lib.hessMap.end()) {
            std::vector<Float> vGrad(d, Float(0.0)), vHess((size_t)d * d, Float(0.0));
            if (csp.ssScore > Float(1e-15)) {
                SerializedSubpath ssubPath;
                ssubPath.primary.assign(GetPrimaryParamSize(lib.maxDepth, lib.maxDepth), Float(0.0));
                ssubPath.vertParams.assign(GetVertParamSize(lib.maxDepth, lib.maxDepth), Float(0.0));
                Serialize(sc, state.path, ssubPath);
                funcIt->second(&csp.screenPos[0], &ssubPath.primary[0], sc->sceneParams, &ssubPath.vertParams[0], &vGrad[0], &vHess[0]);
                c.st->gradCalls++;
                if (!IsFiniteVec(vGrad) || !IsFiniteVec(vHess)) {
                    std::fill(vGrad.begin(), vGrad.end(), Float(0.0));
                    std::fill(vHess.begin(), vHess.end(), Float(0.0));
                }
            }
            Gaussian &g = state.gaussian;
            g.dense = true, g.isDiagonal = false;
            g.mean.assign(d, 0.f), g.covL.assign((size_t)d * d, 0.f), g.invCov.assign((size_t)d * d, 0.f);
            std::vector<Float> work((size_t)d * d + 4 * d);
            lmcd::ComputeGaussianH2MC(param, d, csp.ssScore, vGrad.data(), vHess.data(), g.mean.data(), lmcd::MatRef{g.covL.data(), 1}, lmcd::MatRef{g.invCov.data(), 1},
                                      g.logDet, work.data());
        } else {
            IsotropicGaussian(d, param.sigma, state.gaussian);
        }
        state.gaussianInitialized = true;
    };
    if (!currentState.gaussianInitialized) initGaussian(currentState);
    std::vector<Float> offset(dim);
    GenerateSample(currentState.gaussian, offset, c.rng);
    proposalState.path = currentState.path;
    PerturbPathBidir(sc, offset, proposalState.path, spContribs, c.rng);
    if (spContribs.size() > 0) {
        proposalState.spContrib = spContribs[0];
        initGaussian(proposalState);
        Float py = GaussianLogPdf(offset, currentState.gaussian, false);
        Float px = GaussianLogPdf(offset, proposalState.gaussian, true);
        a = Clamp(lmcd::lexpf(px - py) * proposalState.spContrib.ssScore / currentState.spContrib.ssScore, Float(0.0), Float(1.0));
        proposalState.toSplat.clear();
        for (const auto &spContrib : spContribs)
            proposalState.toSplat.push_back(SplatSample{spContrib.screenPos, spContrib.contrib * (normalization / spContrib.lsScore)});
    } else {
        a = Float(0.0);
    }
    return a;
}

// mutation.h:5-8
#define OUTLIER_WEAK_REJECT_CNT 10000
#define OUTLIER_STRONG_REJECT_CNT 1000
const Float OUTLIER_RATIO_THRESHOLD = Float(30.0);

void MLT::StepChain(ChainCtx &c, std::vector<PendingPush> &pushes) {  // body of the loop at mlt.cpp:91-170
    const RScene *sc = scene.get();
    std::uniform_real_distribution<Float> uniDist(Float(0.0), Float(1.0));
    const Float largeStepProb = sc->options->largeStepProbability;
    const int numChains = (int)initStates.size();
    MarkovState &currentState = c.currentState, &proposalState = c.proposalState;
    Chain &chain = c.chain;
    const int64_t sampleIdx = c.sampleIdx;
    Float a = Float(1.0);
    bool isLargeStep = false;
    Float lsScale = (sampleIdx > c.numSamplesThisChain * LS_RATIO) ? sc->options->largeStepProbScale : Float(1.0);
    if (!currentState.valid || uniDist(c.rng) < largeStepProb * lsScale) {
        isLargeStep = true;
        a = (sc->options->sampleFromGlobalCache && sc->options->mala) ? LargeStepCacheMutate(c) : LargeStepMutate(c);  // mlt.cpp:71-73
        c.st->largeSteps++;
    } else {
        // mlt.cpp:74-85: H2MC takes precedence over LMC, plain isotropic steps otherwise
        a = sc->options->h2mc ? H2MCMutate(c) : sc->options->mala ? MALAMutate(c) : SmallStepMutate(c);
        if (!sc->options->h2mc && !sc->options->mala) c.lastSmallType = MutationType::Small;
    }
    c.st->steps++;
    c.st->weightSum += currentState.valid ? 1.0 : (a > Float(0.0) ? (double)a : 0.0);
    if (currentState.valid && a < Float(1.0)) {
        for (const auto &splat : currentState.toSplat) Splat(*c.film, splat.screenPos, (Float(1.0) - a) * splat.contrib);
    }
    if (a > Float(0.0)) {
        for (const auto &splat : proposalState.toSplat) Splat(*c.film, splat.screenPos, a * splat.contrib);
    }
    if (a > Float(0.0) && uniDist(c.rng) <= a) {
        ToSubpath(proposalState.spContrib.camDepth, proposalState.spContrib.lightDepth, proposalState.path);
        std::swap(currentState, proposalState);
        currentState.valid = true;
        c.adjacentReject = 0;
        c.st->accepted++;
        if (isLargeStep) {
            if (chain.buffered && chain.pathWeight > Float(1e-10)) {
                int dim = GetDimension(proposalState.path);
                if (dim >= PSS_MIN_LENGTH && dim <= PSS_MAX_LENGTH && !cache.isReady(dim)) {
                    PendingPush p;
                    p.dim = dim;
                    p.pss.assign(chain.pss.begin(), chain.pss.begin() + dim);
                    p.v1.assign(chain.v1.begin(), chain.v1.begin() + dim);
                    p.v2.assign(chain.v2.begin(), chain.v2.begin() + dim);
                    p.weight = chain.pathWeight;
                    p.path = chain.path, p.spContrib = chain.spContrib;
                    pushes.push_back(std::move(p));
                }
            }
            c.lastScoreSum = currentState.scoreSum;
            c.lastScore = currentState.spContrib.lsScore;
            currentState.gaussianInitialized = false;
            chain.buffered = false;
        } else {
            if (c.lastSmallType == MutationType::MALASmall) {
                chain.g = chain.prop_new_g;
                chain.v1 = chain.prop_new_v1;
                chain.v2 = chain.prop_new_v2;
                chain.t += 1;
                chain.buffered = true;
                currentState.gaussianInitialized = true;
            }
        }
    } else {
        c.adjacentReject += 1;  // REMOVE_OUTLIERS, mlt.cpp:147-169
        bool strongReject = currentState.spContrib.lsScore > OUTLIER_RATIO_THRESHOLD * normalization;
        if (c.adjacentReject > OUTLIER_WEAK_REJECT_CNT || (strongReject && c.adjacentReject > OUTLIER_STRONG_REJECT_CNT)) {
            int _chainId = chain.chainId, cnt = 0;
            while (true) {
                currentState = initStates[_chainId];
                if (currentState.spContrib.lsScore < OUTLIER_RATIO_THRESHOLD * normalization) break;
                _chainId = (int)((_chainId + sampleIdx + cnt++) % numChains);
            }
            currentState.valid = false;
            currentState.gaussianInitialized = false;
            currentState.toSplat.clear();
            proposalState.valid = false;
            proposalState.gaussianInitialized = false;
            proposalState.toSplat.clear();
            proposalState.pss.clear();
            Clear(proposalState.path);
            chain.buffered = false;
            c.st->resets++;
        }
    }
    c.sampleIdx++;
}

void MLT::StepAll() {
    std::vector<PendingPush> pushes;
    for (auto &c : chains)
        if (c.sampleIdx < c.numSamplesThisChain) StepChain(c, pushes);
    for (auto &p : pushes) cache.dims[p.dim].push(p.pss.data(), p.v1.data(), p.v2.data(), p.weight, p.path, p.spContrib);
}

}  // namespace orc
