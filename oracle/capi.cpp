// ORACLE -- TEST INFRASTRUCTURE ONLY (see common.h).
// C entry points for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (ctypes).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <mutex>
#include <thread>

#include "mlt.h"
#include "h2mc_serial.h"

using namespace orc;

static thread_local std::string g_err;

#define ORC_TRY try {
#define ORC_CATCH(ret)                 \
    }                                  \
    catch (const std::exception &e) {  \
        g_err = e.what();              \
        return ret;                    \
    }

extern "C" {

const char *orc_last_error() { return g_err.c_str(); }

// overrides: <=0 / <0 means "keep the XML value" (see lmc::LoadOverrides)
void *orc_create(const char *xmlPath, int forceDiffuse, int maxDepth, int width, int height, int seedOffset, const char *pathrefSo) {
    ORC_TRY
    lmc::LoadOverrides ov;
    ov.forceDiffuse = forceDiffuse != 0;
    ov.maxDepth = maxDepth;
    ov.width = width;
    ov.height = height;
    ov.seedOffset = seedOffset;
    std::unique_ptr<MLT> m(new MLT);
    m->scene = BuildRScene(lmc::ParseScene(xmlPath, ov));
    if (pathrefSo && pathrefSo[0]) m->lib.Load(pathrefSo, 8);
    return m.release();
    ORC_CATCH(nullptr)
}

void orc_destroy(void *h) { delete (MLT *)h; }

int orc_info(void *h, int *out) {  // [width, height, numTris, maxDepth, numDervFuncs, numLights]
    MLT *m = (MLT *)h;
    out[0] = m->scene->camera.pixelWidth;
    out[1] = m->scene->camera.pixelHeight;
    out[2] = (int)m->scene->tris.size();
    out[3] = m->scene->options->maxDepth;
    out[4] = (int)m->lib.dervMap.size();
    out[5] = (int)m->scene->lights.size();
    return 0;
}

void orc_scene_params(void *h, float *out38) { memcpy(out38, ((MLT *)h)->scene->sceneParams, 38 * sizeof(float)); }

int orc_set_option(void *h, const char *name, double v) {
    lmc::DptOptions &o = *((MLT *)h)->scene->options;
    std::string n(name);
    if (n == "largestepprob") o.largeStepProbability = (float)v;
    else if (n == "largestepscale") o.largeStepProbScale = (float)v;
    else if (n == "mala") o.mala = v != 0;
    else if (n == "h2mc") o.h2mc = v != 0;
    else if (n == "uniformmixprob") o.uniformMixingProbability = (float)v;
    else if (n == "mala-stepsize") o.malaStepsize = (float)v;
    else if (n == "mala-gn") o.malaGN = (float)v;
    else if (n == "perturbstddev") o.perturbStdDev = (float)v;
    else if (n == "mindepth") o.minDepth = (int)v;
    else if (n == "largestepmultiplexed") o.largeStepMultiplexed = v != 0;
    else if (n == "samplecache") o.sampleFromGlobalCache = v != 0;
    else if (n == "uselightcoordinatesampling") {
        o.useLightCoordinateSampling = v != 0;
        ((MLT *)h)->scene->sceneParams[0] = v != 0 ? 1.f : 0.f;  // scene.cpp:165: the flag is the first word of the serialized scene block
    }
    else return -1;
    return 0;
}

int orc_init(void *h, long long numInitSamples, int numChains, int initThreads, float *normalization, long long *numContribs) {
    ORC_TRY
    MLT *m = (MLT *)h;
    float n = m->Init(numInitSamples, numChains, initThreads);
    if (normalization) *normalization = n;
    if (numContribs) *numContribs = m->numInitContribs;
    return 0;
    ORC_CATCH(-1)
}

// parity probe: every MLTInit contribution in stream order (global sample index, c * 16 + l, lsScore); returns the count
long long orc_init_contribs(void *h, long long cap, long long *sample, int *cl, float *ls) {
    MLT *m = (MLT *)h;
    const long long n = (long long)m->initContribCL.size();
    for (long long i = 0; i < n && i < cap; i++) sample[i] = m->initContribSample[i], cl[i] = m->initContribCL[i], ls[i] = m->initContribLs[i];
    return n;
}

// consistency probe of the two path generators: n samples of GenerateSubpath(camLength, lgtLength) on RNG(seed); sumLs / sumLsSq = sum of the
// lsScores / of their squares, count = samples with a contribution.  In expectation sumLs / n equals what technique (camLength, lgtLength) contributes
// per GeneratePathBidir sample (both are unbiased estimates of the technique's MIS-weighted integral; only one plays Russian roulette).
int orc_subpath_probe(void *h, int camLength, int lgtLength, long long n, long long seed, double *sumLs, double *sumLsSq, long long *count) {
    ORC_TRY
    MLT *m = (MLT *)h;
    RNG rng((uint64_t)seed);
    Path path;
    std::vector<SubpathContrib> sp;
    double s = 0, s2 = 0;
    long long c = 0;
    for (long long i = 0; i < n; i++) {
        sp.clear();
        Clear(path);
        GenerateSubpath(m->scene.get(), camLength, lgtLength, true, path, sp, rng);
        for (auto &x : sp) {
            if (x.camDepth != camLength || x.lightDepth != lgtLength) throw std::runtime_error("GenerateSubpath returned another technique");
            s += x.lsScore, s2 += double(x.lsScore) * x.lsScore, c++;
        }
    }
    *sumLs = s, *sumLsSq = s2, *count = c;
    return 0;
    ORC_CATCH(-1)
}

// parity probe: rows of one cache dim: pss (3000 x dim), weight (3000), per row 8 words of its path / contribution:
// camDepth, lightDepth, lsScore, ssScore, path.time, screenPos x, y, number of camera surface vertices.  Returns the rows filled.
int orc_cache_rows(void *h, int dim, float *pss, float *weight, float *info) {
    MLT *m = (MLT *)h;
    const CacheDim &cd = m->cache.dims[dim];
    for (int r = 0; r < cd.data_idx; r++) {
        for (int k = 0; k < dim; k++) pss[(size_t)r * dim + k] = cd.pss[(size_t)r * dim + k];
        weight[r] = cd.pathWeight[r];
        float *o = info + (size_t)r * 8;
        const SubpathContrib &sp = cd.rowContrib[r];
        const Path &p = cd.rowPath[r];
        o[0] = (float)sp.camDepth, o[1] = (float)sp.lightDepth, o[2] = sp.lsScore, o[3] = sp.ssScore, o[4] = p.time, o[5] = p.camVertex.screenPos[0],
        o[6] = p.camVertex.screenPos[1], o[7] = (float)p.camSurfaceVertex.size();
    }
    return cd.data_idx;
}

// parity probe: the oracle's sampleCache / evalPdfCache on its cache as it stands (same arguments as lmc_cache_probe)
int orc_cache_probe(void *h, int dim, int n, const float *u, int *row, const float *query, const int *cl, float *pdf) {
    MLT *m = (MLT *)h;
    const CacheDim &cd = m->cache.dims[dim];
    if (!cd.is_ready) return -2;
    std::vector<Float> q(dim);
    Path path;
    for (int i = 0; i < n; i++) {
        row[i] = cd.sampleCache(u[i]);
        for (int k = 0; k < dim; k++) q[k] = query[(size_t)i * dim + k];
        path.camDepth = cl[2 * i], path.lgtDepth = cl[2 * i + 1];
        pdf[i] = cd.evalPdfCache(q, path);
    }
    return 0;
}

int orc_setup_chains(void *h, long long samplesPerChain, long long chainsNeedExtra) {
    ORC_TRY((MLT *)h)->SetupChains(samplesPerChain, chainsNeedExtra);
    return 0;
    ORC_CATCH(-1)
}
int orc_setup_chains_range(void *h, long long samplesPerChain, long long chainsNeedExtra, int chainBegin, int chainEnd) {
    ORC_TRY((MLT *)h)->SetupChains(samplesPerChain, chainsNeedExtra, chainBegin, chainEnd);
    return 0;
    ORC_CATCH(-1)
}

int orc_step(void *h, int nsteps) {
    ORC_TRY
    MLT *m = (MLT *)h;
    for (int i = 0; i < nsteps; i++) m->StepAll();
    return 0;
    ORC_CATCH(-1)
}

// DirectLighting(scene, buffer), direct.cpp:4-54: 16x16 tiles, RNG(tileIndex + seedOffset), pixels row by row, directSpp
// samples each; out = un-normalised W*H*3 buffer
int orc_direct(void *h, int directSpp, float *out) {
    ORC_TRY
    MLT *m = (MLT *)h;
    const RScene *scene = m->scene.get();
    const int W = scene->camera.pixelWidth, H = scene->camera.pixelHeight;
    std::vector<Float> buf((size_t)W * H * 3, 0.f);
    if (!(scene->options->minDepth > 2 || scene->options->maxDepth < 1)) {
        const int tileSize = 16, nX = (W + tileSize - 1) / tileSize, nY = (H + tileSize - 1) / tileSize;
        for (int ty = 0; ty < nY; ty++)
            for (int tx = 0; tx < nX; tx++) {
                RNG rng(ty * nX + tx + scene->options->seedOffset);
                const int x0 = tx * tileSize, x1 = std::min(x0 + tileSize, W), y0 = ty * tileSize, y1 = std::min(y0 + tileSize, H);
                for (int y = y0; y < y1; y++)
                    for (int x = x0; x < x1; x++)
                        for (int s = 0; s < directSpp; s++) {
                            std::vector<SubpathContrib> sp;
                            GeneratePathUni(scene, x, y, std::min(scene->options->minDepth, 2), std::min(scene->options->maxDepth, 2), sp, rng);
                            for (const auto &c : sp) m->Splat(buf, c.screenPos, c.contrib);
                        }
            }
    }
    memcpy(out, buf.data(), buf.size() * sizeof(float));
    return 0;
    ORC_CATCH(-1)
}

// Plain Monte Carlo estimators of the image, the oracle side of lmc_path_trace / lmc_bidir_mc (the product's cross-check estimators, which
// bench.py's equal-time RMSE and the option tests use as ground truth: pinned here so that a biased "truth" cannot hide):
//   orc_path_trace  GeneratePath (path.cpp:406-527; the "mc" integrator's generator, pathtrace.cpp) over the depth range [minDepth, maxDepth],
//                   tile / stream structure of DirectLighting (direct.cpp:4-54); out = un-normalised W*H*3
//   orc_bidir_mc    GeneratePathBidir (path.cpp:1237-1449) samples of nThreads streams RNG(t + seedOffset), samplesPerThread each, every
//                   contribution splatted as is (they carry their MIS weights); out = un-normalised sum
int orc_path_trace(void *h, int spp, int minDepth, int maxDepth, float *out) {
    ORC_TRY
    MLT *m = (MLT *)h;
    const RScene *scene = m->scene.get();
    const int W = scene->camera.pixelWidth, H = scene->camera.pixelHeight;
    std::vector<Float> buf((size_t)W * H * 3, 0.f);
    const int tileSize = 16, nX = (W + tileSize - 1) / tileSize, nY = (H + tileSize - 1) / tileSize;
    for (int ty = 0; ty < nY; ty++)
        for (int tx = 0; tx < nX; tx++) {
            RNG rng(ty * nX + tx + scene->options->seedOffset);
            const int x0 = tx * tileSize, x1 = std::min(x0 + tileSize, W), y0 = ty * tileSize, y1 = std::min(y0 + tileSize, H);
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++)
                    for (int s = 0; s < spp; s++) {
                        std::vector<SubpathContrib> sp;
                        GeneratePathUni(scene, x, y, minDepth, maxDepth, sp, rng);
                        for (const auto &c : sp) m->Splat(buf, c.screenPos, c.contrib);
                    }
        }
    memcpy(out, buf.data(), buf.size() * sizeof(float));
    return 0;
    ORC_CATCH(-1)
}
int orc_bidir_mc(void *h, int nThreads, int samplesPerThread, float *out) {
    ORC_TRY
    MLT *m = (MLT *)h;
    const RScene *scene = m->scene.get();
    const int W = scene->camera.pixelWidth, H = scene->camera.pixelHeight;
    std::vector<Float> buf((size_t)W * H * 3, 0.f);
    std::vector<SubpathContrib> sp;
    Path path;
    for (int t = 0; t < nThreads; t++) {
        RNG rng((uint64_t)(t + scene->options->seedOffset));
        for (int s = 0; s < samplesPerThread; s++) {
            sp.clear();
            Clear(path);
            GeneratePathBidir(scene, -1, -1, std::max(scene->options->minDepth, 3), scene->options->maxDepth, path, sp, rng);
            for (const auto &c : sp) m->Splat(buf, c.screenPos, c.contrib);
        }
    }
    memcpy(out, buf.data(), buf.size() * sizeof(float));
    return 0;
    ORC_CATCH(-1)
}

void orc_film(void *h, float *out) {
    MLT *m = (MLT *)h;
    memcpy(out, m->film.data(), m->film.size() * sizeof(float));
}

void orc_stats(void *h, long long *out) {  // steps, largeSteps, accepted, gradCalls, cacheQueries, cacheHits, resets, cacheReadyMask
    MLT *m = (MLT *)h;
    out[0] = m->stats.steps, out[1] = m->stats.largeSteps, out[2] = m->stats.accepted, out[3] = m->stats.gradCalls;
    out[4] = m->stats.cacheQueries, out[5] = m->stats.cacheHits, out[6] = m->stats.resets;
    long long mask = 0;
    for (int d = 2; d <= 16; d++)
        if (m->cache.isReady(d)) mask |= 1ll << d;
    out[7] = mask;
    memcpy(&out[8], &m->stats.weightSum, 8);
}

// per-chain summary, `stride` floats each (>= 32):
// [valid, camDepth, lightDepth, lsScore, ssScore, scoreSum, time, gaussianInitialized, buffered, sampleIdx,
//  screenX, screenY, contribR, contribG, contribB, nSplats, pss[0..15]]
int orc_chain_summary(void *h, int which /*0 current, 1 init*/, float *out, int stride) {
    MLT *m = (MLT *)h;
    size_t n = which == 0 ? m->chains.size() : m->initStates.size();
    for (size_t i = 0; i < n; i++) {
        const MarkovState &s = which == 0 ? m->chains[i].currentState : m->initStates[i];
        float *o = out + i * stride;
        memset(o, 0, stride * sizeof(float));
        o[0] = s.valid, o[1] = (float)s.spContrib.camDepth, o[2] = (float)s.spContrib.lightDepth, o[3] = s.spContrib.lsScore, o[4] = s.spContrib.ssScore;
        o[5] = s.scoreSum, o[6] = s.path.time, o[7] = s.gaussianInitialized;
        if (which == 0) o[8] = m->chains[i].chain.buffered, o[9] = (float)m->chains[i].sampleIdx;
        o[10] = s.spContrib.screenPos[0], o[11] = s.spContrib.screenPos[1];
        o[12] = s.spContrib.contrib[0], o[13] = s.spContrib.contrib[1], o[14] = s.spContrib.contrib[2];
        o[15] = (float)s.toSplat.size();
        for (size_t k = 0; k < s.pss.size() && k < 16 && 16 + (int)k < stride; k++) o[16 + k] = s.pss[k];
    }
    return (int)n;
}

// Serialises init state i into the path-function ABI buffers (primary[2L+1], vertParams[V]); returns
// camDepth*16+lightDepth, or -1.  Lets tests feed identical inputs to libpathref.so and to the HIP gradient.
int orc_serialize_init_state(void *h, int i, float *primary, int primaryCap, float *vertParams, int vertCap) {
    ORC_TRY
    MLT *m = (MLT *)h;
    if (i < 0 || i >= (int)m->initStates.size()) return -1;
    const MarkovState &s = m->initStates[i];
    SerializedSubpath ss;
    ss.primary.assign(GetPrimaryParamSize(8, 8), 0.f);
    ss.vertParams.assign(GetVertParamSize(8, 8), 0.f);
    Serialize(m->scene.get(), s.path, ss);
    int np = (int)GetPrimaryParamSize(s.path.camDepth, s.path.lgtDepth);
    if (np > primaryCap) return -1;
    memcpy(primary, ss.primary.data(), np * sizeof(float));
    int nv = std::min(vertCap, (int)ss.vertParams.size());
    memcpy(vertParams, ss.vertParams.data(), nv * sizeof(float));
    return s.path.camDepth * 16 + s.path.lgtDepth;
    ORC_CATCH(-1)
}

// reference gradient / forward programs through the dlsym'd table (mutation_mala.h:101-107)
int orc_ref_eval(void *h, int c, int l, const float *primary, const float *vertParams, float *logLum, float *grad) {
    MLT *m = (MLT *)h;
    auto f = m->lib.funcMap.find({c, l});
    auto d = m->lib.dervMap.find({c, l});
    if (f == m->lib.funcMap.end() || d == m->lib.dervMap.end()) return -1;
    float lens[2] = {0, 0};
    if (logLum) f->second(lens, primary, m->scene->sceneParams, vertParams, logLum);
    if (grad) d->second(lens, primary, m->scene->sceneParams, vertParams, grad, nullptr);
    return 0;
}

// closest-hit / occlusion probes: rays = n x [ox,oy,oz,dx,dy,dz,tnear,tfar]
void orc_trace(void *h, int n, const float *rays, int *prim, float *t) {
    MLT *m = (MLT *)h;
    for (int i = 0; i < n; i++) {
        const float *r = rays + (size_t)i * 8;
        Ray ray{Vector3(r[0], r[1], r[2]), Vector3(r[3], r[4], r[5])};
        Float tt = 0;
        prim[i] = m->scene->bvh.Intersect(ray, r[6], r[7], &tt);
        t[i] = prim[i] >= 0 ? tt : 0.f;
    }
}
void orc_occluded(void *h, int n, const float *rays, int *occ) {
    MLT *m = (MLT *)h;
    for (int i = 0; i < n; i++) {
        const float *r = rays + (size_t)i * 8;
        Ray ray{Vector3(r[0], r[1], r[2]), Vector3(r[3], r[4], r[5])};
        occ[i] = m->scene->bvh.Occluded(ray, r[6], r[7]) ? 1 : 0;
    }
}
// brute force over all triangles (pins the BVHs, CPU and HIP alike)
void orc_trace_brute(void *h, int n, const float *rays, int *prim, float *t) {
    MLT *m = (MLT *)h;
    for (int i = 0; i < n; i++) {
        const float *r = rays + (size_t)i * 8;
        Ray ray{Vector3(r[0], r[1], r[2]), Vector3(r[3], r[4], r[5])};
        int best = -1;
        Float bestT = r[7];
        for (size_t k = 0; k < m->scene->tris.size(); k++) {
            Float tt;
            if (TriTest(m->scene->tris[k], ray, r[6], bestT, tt) && (best < 0 || tt < bestT)) best = (int)k, bestT = tt;
        }
        prim[i] = best;
        t[i] = best >= 0 ? bestT : 0.f;
    }
}

// ---- small pure-function probes -----------------------------------------------------------------
void orc_pcg_u32(unsigned long long seed, int n, unsigned *out) {
    RNG rng(seed);
    for (int i = 0; i < n; i++) out[i] = rng();
}
void orc_pcg_uniform(unsigned long long seed, int n, float *out) {
    RNG rng(seed);
    for (int i = 0; i < n; i++) out[i] = Uniform01(rng);
}
void orc_pcg_normal(unsigned long long seed, int n, float mean, float stddev, float *out) {
    RNG rng(seed);
    NormalPolar nd(mean, stddev);
    for (int i = 0; i < n; i++) out[i] = nd(rng);
}
void orc_pcg_mixed(unsigned long long seed, int rounds, int k, float *out) {
    RNG rng(seed);
    int o = 0;
    for (int r = 0; r < rounds; r++) {
        out[o++] = Uniform01(rng);
        out[o++] = Uniform01(rng);
        NormalPolar nd(0.f, 1.f);
        for (int i = 0; i < k; i++) out[o++] = nd(rng);
    }
}
// state such that the next draw ticks the extension table: set the base state directly
void orc_pcg_dump(unsigned long long seed, int ndraws, unsigned *out66) {
    RNG rng(seed);
    for (int i = 0; i < ndraws; i++) rng();
    memcpy(out66, &rng.state, 8);
    memcpy(out66 + 2, rng.data, 256);
}
void orc_fastlog(int n, const float *in, float *out) {
    for (int i = 0; i < n; i++) out[i] = fastlog(in[i]);
}
int orc_kd_query(int dim, int npts, const float *pts, int nq, const float *q, float radiusSq, int knn, int *outN, int *outIdx, float *outDist) {
    KdTree t;
    t.Build(pts, npts, dim);
    for (int i = 0; i < nq; i++) {
        int idx[16];
        float dist[16];
        int n = t.RadiusSearch(q + (size_t)i * dim, radiusSq, knn, idx, dist);
        outN[i] = n;
        for (int k = 0; k < knn; k++) {
            outIdx[i * knn + k] = k < n ? idx[k] : -1;
            outDist[i * knn + k] = k < n ? dist[k] : 0.f;
        }
    }
    return 0;
}
// ComputeGaussian (mala.cpp:7-52) + GaussianLogPdf probe: out = [mean(dim), covL(dim), invCov(dim), logDet, logpdf(offset)]
void orc_compute_gaussian(int dim, const float *v1, const float *M, float ss, float shk, float sc, const float *offset, float *out) {
    std::vector<Float> a(v1, v1 + dim), mm(M, M + dim), off(offset, offset + dim);
    Gaussian g;
    ComputeGaussianMALA(dim, a, a, ss, shk, mm, 0, sc, g);
    for (int i = 0; i < dim; i++) out[i] = g.mean[i], out[dim + i] = g.covL_d[i], out[2 * dim + i] = g.invCov_d[i];
    out[3 * dim] = g.logDet;
    out[3 * dim + 1] = GaussianLogPdf(off, g, false);
}

// parity / analysis probe: the points of the global cache of one dim (row-major n x dim); returns n
int orc_cache_points(void *h, int dim, float *out, int cap) {
    MLT *m = (MLT *)h;
    if (dim < 2 || dim > 16) return 0;
    const CacheDim &c = m->cache.dims[dim];
    const int n = c.is_ready ? PSS_MAX_SIZE : c.data_idx;
    for (int i = 0; i < n && i < cap; i++)
        for (int k = 0; k < dim; k++) out[(size_t)i * dim + k] = c.pss[(size_t)i * dim + k];
    return n;
}

// H2MC Gaussian of one state (h2mc.cpp:70-142 through oracle/h2mc_serial.h): out = mean[dim], covL[dim*dim],
// invCov[dim*dim], logDet
// pair j of round r of the round-robin Jacobi order (h2mc_serial.h JacobiRoundPair; the device's h2gauss.hip RoundPair is its twin)
void orc_jacobi_round_pair(int m, int r, int j, int *p, int *q) { lmcd::JacobiRoundPair(m, r, j, *p, *q); }
void orc_h2mc_gaussian(int dim, float sigma, float sc, const float *grad, const float *hess, float *out) {
    lmcd::H2MCParam p = lmcd::MakeH2MCParam(sigma);
    std::vector<float> work((size_t)dim * dim + 4 * dim), h(hess, hess + (size_t)dim * dim);
    lmcd::ComputeGaussianH2MC(p, dim, sc, grad, h.data(), out, lmcd::MatRef{out + dim, 1}, lmcd::MatRef{out + dim + dim * dim, 1}, out[dim + 2 * dim * dim], work.data());
}

// ---- CPU baseline (bench.py): the chain loop on `threads` host threads, chains handed out in contiguous
// blocks (one chain per work item as in parallel.cpp:82-142).  Cache pushes stay deferred per step and are
// applied in chain order, so it is the same lock-step algorithm as orc_step; each thread splats into a
// private film that is summed at the end (float add order differs from the 1-thread run).
double orc_bench_steps(void *h, int nsteps, int threads, long long *stepsDone) {
    MLT *m = (MLT *)h;
    threads = std::max(1, threads);
    const int n = (int)m->chains.size();
    threads = std::min(threads, std::max(1, n));
    long long before = m->stats.steps;
    std::vector<std::vector<Float>> films(threads);
    std::vector<StepStats> st(threads);
    if (threads > 1)
        for (int t = 0; t < threads; t++) {
            films[t].assign(m->film.size(), 0.f);
            int lo = (int)((long long)n * t / threads), hi = (int)((long long)n * (t + 1) / threads);
            for (int i = lo; i < hi; i++) m->chains[i].film = &films[t], m->chains[i].st = &st[t];
        }
    auto t0 = std::chrono::steady_clock::now();
    if (threads == 1) {
        for (int i = 0; i < nsteps; i++) m->StepAll();
    } else {
        // persistent worker threads (like the reference's pool, parallel.cpp:82-142) meeting at a barrier after every
        // lock-step iteration; thread 0 applies the step's cache pushes in chain order between two barriers
        std::vector<std::vector<PendingPush>> pushes(threads);
        std::atomic<int> arrived{0};
        std::atomic<int> phase{0};
        auto barrier = [&]() {
            int ph = phase.load(std::memory_order_acquire);
            if (arrived.fetch_add(1, std::memory_order_acq_rel) == threads - 1) {
                arrived.store(0, std::memory_order_relaxed);
                phase.store(ph + 1, std::memory_order_release);
            } else {
                int spins = 0;
                while (phase.load(std::memory_order_acquire) == ph)
                    if (++spins > 2000) std::this_thread::yield();
            }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++)
            pool.emplace_back([&, t]() {
                int lo = (int)((long long)n * t / threads), hi = (int)((long long)n * (t + 1) / threads);
                for (int s = 0; s < nsteps; s++) {
                    pushes[t].clear();
                    for (int i = lo; i < hi; i++)
                        if (m->chains[i].sampleIdx < m->chains[i].numSamplesThisChain) m->StepChain(m->chains[i], pushes[t]);
                    barrier();
                    if (t == 0)
                        for (int tt = 0; tt < threads; tt++)
                            for (auto &p : pushes[tt]) m->cache.dims[p.dim].push(p.pss.data(), p.v1.data(), p.v2.data(), p.weight, p.path, p.spContrib);
                    barrier();
                }
            });
        for (auto &th : pool) th.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (threads > 1) {
        for (int t = 0; t < threads; t++) {
            for (size_t i = 0; i < m->film.size(); i++) m->film[i] += films[t][i];
            m->stats.steps += st[t].steps, m->stats.largeSteps += st[t].largeSteps, m->stats.accepted += st[t].accepted;
            m->stats.gradCalls += st[t].gradCalls, m->stats.cacheQueries += st[t].cacheQueries, m->stats.cacheHits += st[t].cacheHits;
            m->stats.resets += st[t].resets;
            m->stats.weightSum += st[t].weightSum;
        }
        for (auto &c : m->chains) c.film = &m->film, c.st = &m->stats;
    }
    double sec = std::chrono::duration<double>(t1 - t0).count();
    long long done = m->stats.steps - before;
    if (stepsDone) *stepsDone = done;
    return done / sec;
}

// ---- The reference's own scheduling (mlt.cpp:60-196 over parallel.cpp:82-142): one chain per work item, each worker runs
// its chain from the first to the last mutation; accepted large steps push to the global cache at once under the dim's
// mutex (mlt.cpp:120-127), readers see is_ready without a lock (global_cache.h:66-68).  Not lock step, hence not
// reproducible run to run -- exactly like the reference.  Used for (a) the CPU baseline of bench.py, (b) the CPU leg of
// the equal-time RMSE, (c) the CPU run of the reference's shipped chain configuration (128 chains).
// maxSeconds > 0: workers stop at the deadline (throughput measurement on a bounded sample).  Returns chain-steps/s.
double orc_run_async(void *h, int threads, double maxSeconds, long long *stepsDone) {
    MLT *m = (MLT *)h;
    const int n = (int)m->chains.size();
    threads = std::max(1, std::min(threads, std::max(1, n)));
    const long long before = m->stats.steps;
    std::vector<std::vector<Float>> films(threads);
    // one cache line (and its prefetched neighbour) per worker: the counters are bumped at every mutation, and packed 56 bytes apart they
    // made the workers fight over lines -- on the GPU box's 256-thread host the whole pool ran no faster than 32 threads (5.5 M steps/s at
    // 32, 64 and 128 threads, 3.9 M at 256: scripts/cpu_baseline_scaling.py, round 4)
    struct alignas(128) PaddedStats {
        StepStats s;
    };
    std::vector<PaddedStats> stPad(threads);
    std::mutex dimMutex[17];
    std::atomic<int> next{0};
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t]() {
            films[t].assign(m->film.size(), 0.f);
            std::vector<PendingPush> pushes;
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= n) break;
                ChainCtx &c = m->chains[i];
                c.film = &films[t], c.st = &stPad[t].s;
                while (c.sampleIdx < c.numSamplesThisChain) {
                    m->StepChain(c, pushes);
                    if (!pushes.empty()) {
                        for (auto &p : pushes) {
                            std::lock_guard<std::mutex> lock(dimMutex[p.dim]);
                            m->cache.dims[p.dim].push(p.pss.data(), p.v1.data(), p.v2.data(), p.weight, p.path, p.spContrib);
                        }
                        pushes.clear();
                    }
                    if (maxSeconds > 0 && (c.sampleIdx & 1023) == 0 &&
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > maxSeconds)
                        break;
                }
                if (maxSeconds > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > maxSeconds) break;
            }
        });
    for (auto &th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; t++) {
        if (films[t].size() == m->film.size())
            for (size_t i = 0; i < m->film.size(); i++) m->film[i] += films[t][i];
        const StepStats &x = stPad[t].s;
        m->stats.steps += x.steps, m->stats.largeSteps += x.largeSteps, m->stats.accepted += x.accepted;
        m->stats.gradCalls += x.gradCalls, m->stats.cacheQueries += x.cacheQueries, m->stats.cacheHits += x.cacheHits;
        m->stats.resets += x.resets;
        m->stats.weightSum += x.weightSum;
    }
    for (auto &c : m->chains) c.film = &m->film, c.st = &m->stats;
    const double sec = std::chrono::duration<double>(t1 - t0).count();
    const long long done = m->stats.steps - before;
    if (stepsDone) *stepsDone = done;
    return done / sec;
}

}  // extern "C"
