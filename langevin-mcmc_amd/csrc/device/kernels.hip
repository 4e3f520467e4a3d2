// gfx950 kernels of the LMC hot path.  One thread = one Markov chain (or one MLT-init stream / sample);
// launches are grid-stride ("persistent thread") loops over index lists so that large steps and small
// steps run in separate, divergence-free launches.  Launch glue is in host/context.cpp.
#include "dh2coop.h"
#include "kernels.h"
#include "upload.h"

#include "ddirect.h"
#include "dstep.h"

namespace lmcd {

// ---------------------------------------------------------------------------------------------- RNG
__global__ void k_seed_rng(int n, long long firstSeed, uint64_t *state, uint32_t *tab) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        state[i] = PcgSeed((uint64_t)(firstSeed + i), tab + (size_t)i * 64);
}

// test probe: draws from RNG(seed): mode 0 raw u32, 1 uniform, 2 one normal object (mean, stddev), 3 mixed rounds
// test hook: LowerBoundMonotone (dshade.h) against std::lower_bound on the host side of the test
__global__ void k_lower_bound_probe(int n, const float *cdf, int nq, const float *u, int *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) out[i] = LowerBoundMonotone(cdf, n, u[i]);
}

__global__ void k_rng_probe(int nSeeds, const unsigned long long *seeds, int mode, int n, float mean, float stddev, uint32_t *tabScratch, uint32_t *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nSeeds) return;
    Rng rng;
    rng.tab = tabScratch + (size_t)i * 64;
    rng.state = PcgSeed(seeds[i], rng.tab);
    rng.ticks = 0;
    if (mode & 8) {  // the form the chain kernels use (dchain.h LoadChainRng): the table synthesised from the seed, the memory behind it poisoned until a tick materialises it
        for (int k = 0; k < 64; k++) rng.tab[k] = 0xDEADBEEFu;
        rng.SetSynth(seeds[i]);
        mode &= 7;
    }
    uint32_t *o = out + (size_t)i * (n + 66);
    if (mode == 0) {
        for (int k = 0; k < n; k++) o[k] = rng.Next();
    } else if (mode == 1) {
        for (int k = 0; k < n; k++) o[k] = __float_as_uint(rng.Uniform());
    } else if (mode == 2) {
        NormalDist nd(mean, stddev);
        for (int k = 0; k < n; k++) o[k] = __float_as_uint(nd(rng));
    } else {
        int k = 0;
        while (k + 9 <= n) {
            o[k++] = __float_as_uint(rng.Uniform());
            o[k++] = __float_as_uint(rng.Uniform());
            NormalDist nd(0.f, 1.f);
            for (int j = 0; j < 7; j++) o[k++] = __float_as_uint(nd(rng));
        }
    }
    // state dump after the draws: [lo, hi, table 64]
    o[n] = (uint32_t)rng.state, o[n + 1] = (uint32_t)(rng.state >> 32);
    for (int k = 0; k < 64; k++) o[n + 2 + k] = rng.Entry(k);
}

// ---------------------------------------------------------------------------------------------- probes
// counter calibration: the chain state's access pattern (one dword per lane, unit stride) over a known byte count
// parity probe of the deterministic float transcendentals (dtrans.h): mode 0 exp, 1 log, 2 pow
__global__ void k_trans_probe(int n, int mode, const float *x, const float *y, float *o) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float a = x[i], b = y[i];
        o[i] = mode == 0 ? lexpf(a) : mode == 1 ? llogf(a) : mode == 2 ? lpowf(a, b) : mode == 3 ? lsinf(a) : mode == 4 ? lcosf(a) : mode == 5 ? lacosf(a) : mode == 6 ? latan2f(a, b) : GlibcLogf(a);
    }
}

__global__ void k_stream_probe(long long n, const float *in, float *out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = in[i] + 1.0f;
}

// Layout probe: every thread reads `words` words of "its chain's state" and writes half as many back, in batches of `batch`
// independent loads (a step kernel's access pattern without the arithmetic).  mode 0: struct of arrays over all chains, word w
// of chain i at [w * N + i] (a wave's load = 256 contiguous bytes, consecutive words 4 N bytes apart); mode 1: tiles of 64
// chains, [tile][w][lane] (a wave's state = one contiguous block).
template <int BATCH>
__global__ void __launch_bounds__(64) k_layout_probe(int N, int words, int mode, const float *in, float *out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= N) return;
    const size_t wordStride = mode == 0 ? (size_t)N : 64, base = mode == 0 ? (size_t)i : (size_t)(i / 64) * 64 * words + (i % 64);
    float acc = 0.f;
    for (int w0 = 0; w0 < words; w0 += BATCH) {
        float v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; k++) v[k] = w0 + k < words ? in[base + (size_t)(w0 + k) * wordStride] : 0.f;
#pragma unroll
        for (int k = 0; k < BATCH; k++) acc += v[k];
        // dependent on the batch: the next batch's addresses wait for this one (acc is folded into the store below)
#pragma unroll
        for (int k = 0; k < BATCH; k += 2)
            if (w0 + k < words) out[base + (size_t)(w0 + k) * wordStride] = acc;
        if (acc == 123456.789f) w0 += 1;  // keeps the batches ordered without changing anything
    }
}
__global__ void k_trace(DScene S, int n, const float *rays, int *prim, float *t, int anyHit) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float *r = rays + (size_t)i * 8;
        V3 org{r[0], r[1], r[2]}, dir{r[3], r[4], r[5]};
        LocalStackT<false> stk;
        if (anyHit) {
            prim[i] = BvhOccluded(S, org, dir, r[6], r[7], stk) ? 1 : 0;
        } else {
            float tt = 0.f;
            int id = BvhIntersect(S, org, dir, r[6], r[7], tt, stk);
            prim[i] = id;
            t[i] = id >= 0 ? tt : 0.f;
        }
    }
}

__global__ void k_kd_probe(DCacheDim C, int dim, int nq, const float *q, float radiusSq, int knn, int *outN, int *outIdx, float *outDist) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int idx[8];
    float dist[8];
    float qq[MAXPSS];
    for (int k = 0; k < dim; k++) qq[k] = q[(size_t)i * dim + k];
    int n = KdRadiusSearch(C, dim, qq, radiusSq, knn, idx, dist);
    outN[i] = n;
    for (int k = 0; k < knn; k++) {
        outIdx[i * knn + k] = k < n ? idx[k] : -1;
        outDist[i * knn + k] = k < n ? dist[k] : 0.f;
    }
}

__global__ void k_gauss_probe(int n, int dim, const float *v1, const float *M, float ss, float shk, const float *sc, const float *offset, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a[MAXPSS], m[MAXPSS], off[MAXPSS];
    for (int k = 0; k < dim; k++) a[k] = v1[(size_t)i * dim + k], m[k] = M[(size_t)i * dim + k], off[k] = offset[(size_t)i * dim + k];
    Gauss g;
    ComputeGaussianMALA(dim, a, ss, shk, m, sc[i], g);
    float *o = out + (size_t)i * (3 * dim + 2);
    for (int k = 0; k < dim; k++) o[k] = g.mean[k], o[dim + k] = g.covL[k], o[2 * dim + k] = g.invCov[k];
    o[3 * dim] = g.logDet;
    o[3 * dim + 1] = GaussianLogPdf(dim, off, false, g);
}

// ---------------------------------------------------------------------------------------------- gradient batch
// lmc_grad_batch: n independent evaluations of the (c,l) path program, inputs/outputs SoA (word-major)
__global__ void k_grad_batch(int c, int l, int n, const float *primarySoA, const float *scene, const float *vertSoA, float *logLum, float *gradSoA,
                             int wantGrad) {
    __shared__ float sScene[38];
    if (threadIdx.x < 38) sScene[threadIdx.x] = scene[threadIdx.x];
    __syncthreads();
    const int L = c + l - 1 > 2 ? c + l - 1 : 2;
    const int dim = 2 * L;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float primary[2 * MAXD + 1];
        for (int k = 0; k < dim + 1; k++) primary[k] = primarySoA[(size_t)k * n + i];
        StridedIn vin{vertSoA + i, (size_t)n};
        if (wantGrad) {
            float ll, g[MAXPSS];
            PathFuncGrad(c, l, primary, sScene, vin, &ll, g);
            if (logLum) logLum[i] = ll;
            for (int k = 0; k < dim; k++) gradSoA[(size_t)k * n + i] = g[k];
        } else {
            logLum[i] = PathFuncValue(c, l, primary, sScene, vin);
        }
    }
}

// ---------------------------------------------------------------------------------------------- MLT init
// pass 1: one thread per virtual init thread (mlt.h:66-98 with NumSystemCores() := V): RNG(threadId + seedOffset),
// samples drawn back to back; records the per-sample RNG checkpoint and the number of contributions.
template <bool GLOSSY>
__global__ void k_init_pass1(DScene S, int tBegin, int nStreams, long long perThread, long long extra, long long gBase, uint32_t *tabScratch,
                             float *contribScratch, uint64_t *ckState, uint32_t *ckTicks, unsigned char *count) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nStreams) return;
    const int t = tBegin + idx;  // this rank runs the init streams [tBegin, tBegin + nStreams); its arrays start at sample gBase
    Rng rng;
    rng.tab = tabScratch + (size_t)idx * 64;
    rng.state = PcgSeed((uint64_t)(t + S.opt.seedOffset), rng.tab);
    rng.ticks = 0;
    long long n = perThread + (t < extra ? 1 : 0);
    long long base = (long long)t * perThread + (t < extra ? t : extra) - gBase;
    const int minPathLength = max(S.opt.minDepth, 3);
    DPath path;
    LocalStackT<GLOSSY> stk;
    for (long long s = 0; s < n; s++) {
        ckState[base + s] = rng.state;
        ckTicks[base + s] = rng.ticks;
        ContribSink sink{contribScratch, (size_t)nStreams, (size_t)idx, 0};
        GeneratePathBidir(S, minPathLength, S.opt.maxDepth, path, sink, rng, stk);
        count[base + s] = (unsigned char)sink.count;
    }
}

LMC_D int InitThreadOfSample(long long g, long long perThread, long long extra) {
    long long head = extra * (perThread + 1);
    if (g < head) return (int)(g / (perThread + 1));
    return (int)(extra + (g - head) / perThread);
}

LMC_D void RngFromCheckpoint(Rng &rng, uint64_t seed, uint64_t state, uint32_t ticks) {
    PcgSeed(seed, rng.tab);
    for (uint32_t k = 0; k < ticks; k++) PcgAdvanceTable(rng.tab);
    rng.state = state;
    rng.ticks = ticks;
}

// pass 2: one grid-stride thread per sample; re-runs the sample from its checkpoint and writes (c,l,lsScore) compactly
template <bool GLOSSY>
__global__ void k_init_pass2(DScene S, long long gBegin, long long numLocal, long long perThread, long long extra, int nSlots, uint32_t *tabScratch,
                             float *contribScratch, const uint64_t *ckState, const uint32_t *ckTicks, const unsigned long long *offset, unsigned char *outCL,
                             float *outLs) {
    int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nSlots) return;
    const int minPathLength = max(S.opt.minDepth, 3);
    DPath path;
    LocalStackT<GLOSSY> stk;
    for (long long k = slot; k < numLocal; k += nSlots) {  // local sample k = global sample gBegin + k
        Rng rng;
        rng.tab = tabScratch + (size_t)slot * 64;
        int t = InitThreadOfSample(gBegin + k, perThread, extra);
        RngFromCheckpoint(rng, (uint64_t)(t + S.opt.seedOffset), ckState[k], ckTicks[k]);
        ContribSink sink{contribScratch, (size_t)nSlots, (size_t)slot, 0};
        GeneratePathBidir(S, minPathLength, S.opt.maxDepth, path, sink, rng, stk);
        unsigned long long o = offset[k];  // relative to this rank's first contribution
        for (int j = 0; j < sink.count; j++) {
            Contrib c = sink.Get(j);
            outCL[o + j] = (unsigned char)(c.camDepth * 16 + c.lightDepth);
            outLs[o + j] = c.lsScore;
        }
    }
}

// regenerate the seed paths of THIS rank's chains (mlt.h:121-148) into its init-state arrays; per chain: the init sample that seeds
// it (global index: names the stream), the technique to pick, and the sample's RNG checkpoint (which may come from another rank)
template <bool GLOSSY>
__global__ void k_init_regen(DScene S, int numChains, long long perThread, long long extra, const long long *seedSample, const unsigned char *seedCL,
                             uint32_t *tabScratch, float *contribScratch, const uint64_t *seedCkState, const uint32_t *seedCkTicks, float *initPath,
                             float *initContrib, float *initScoreSum) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numChains) return;
    const long long g = seedSample[i];
    Rng rng;
    rng.tab = tabScratch + (size_t)i * 64;
    int t = InitThreadOfSample(g, perThread, extra);
    RngFromCheckpoint(rng, (uint64_t)(t + S.opt.seedOffset), seedCkState[i], seedCkTicks[i]);
    DPath path;
    ContribSink sink{contribScratch, (size_t)numChains, (size_t)i, 0};
    LocalStackT<GLOSSY> stk;
    GeneratePathBidir(S, max(S.opt.minDepth, 3), S.opt.maxDepth, path, sink, rng, stk);
    float scoreSum = 0.f;
    Contrib sel;
    sel.camDepth = sel.lightDepth = 0;
    sel.screenPos = V2{0, 0};
    sel.contrib = V3{0, 0, 0};
    sel.lsScore = sel.ssScore = 0.f;
    const int wantC = seedCL[i] >> 4, wantL = seedCL[i] & 15;
    for (int k = 0; k < sink.count; k++) {
        Contrib c = sink.Get(k);
        scoreSum += c.lsScore;
        if (c.camDepth == wantC && c.lightDepth == wantL) sel = c;
    }
    ToSubpath(sel.camDepth, sel.lightDepth, path);
    StorePath(initPath, numChains, i, path);
    StoreContrib(initContrib, numChains, i, sel);
    initScoreSum[i] = scoreSum;
}

// direct-lighting pre-pass (direct.cpp:4-54): one thread per 16x16 tile, RNG(tileIndex + seedOffset), pixels and samples
// in the reference's order so that the stream is consumed identically
template <bool GLOSSY>
__global__ void k_direct(DScene S, Film film, int directSpp, int minDepth, int maxDepth, int nXTiles, int nYTiles, uint32_t *tabScratch) {
    const int tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= nXTiles * nYTiles) return;
    const int tx = tile % nXTiles, ty = tile / nXTiles;
    Rng rng;
    rng.tab = tabScratch + (size_t)tile * 64;
    rng.state = PcgSeed((uint64_t)(tile + S.opt.seedOffset), rng.tab);
    rng.ticks = 0;
    const int x0 = tx * 16, x1 = min(x0 + 16, S.cam.width), y0 = ty * 16, y1 = min(y0 + 16, S.cam.height);
    LocalStackT<GLOSSY> stk;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++)
            for (int s = 0; s < directSpp; s++) {
                FilmSink sink{film};
                DirectSample(S, sink, x, y, minDepth, maxDepth, rng, stk);
            }
}

// The same pass with one WAVE per tile.  The reference consumes one RNG stream per tile, sample after sample, and a sample's
// number of draws depends on what its rays hit -- but inside a tile it is nearly always the same (3 when the primary ray
// escapes, 10 on a surface), so the lanes evaluate 64 consecutive samples at once, lane j from the stream position
// "committed position + j * K" (K = the draw count of the last committed sample; the LCG is jumped ahead, the extension table
// is only read).  Lane 0's position is exact by construction; lane j's is exact iff every lane before it drew exactly K
// numbers.  The wave commits that prefix -- plus the first lane that drew a different number, whose position was still right
// -- in stream order into a tile accumulator in LDS, moves the committed position, and repeats.  Same numbers, same order of
// the float sums as the one-thread-per-tile kernel above (the parity test is unchanged), 17 s -> well under a second at
// 1024 x 768 x 256 spp.  Needs maxDepth <= 2 (LaneSink) and a BVH no deeper than the LDS stack.
template <bool GLOSSY>
__global__ void __launch_bounds__(64) k_direct_wave(DScene S, Film film, int directSpp, int minDepth, int maxDepth, int nXTiles, uint32_t *tabScratch) {
    __shared__ float sTile[16 * 16 * 3];
    __shared__ int sStack[BVH_LDS_STACK * 64];
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int tx = tile % nXTiles, ty = tile / nXTiles;
    const int x0 = tx * 16, x1 = min(x0 + 16, S.cam.width), y0 = ty * 16, y1 = min(y0 + 16, S.cam.height);
    const int tw = x1 - x0, th = y1 - y0;
    for (int k = lane; k < 16 * 16 * 3; k += 64) sTile[k] = 0.f;
    uint32_t *tab = tabScratch + (size_t)tile * 64;
    uint64_t committed = 0;  // stream state at the first uncommitted sample (same value in every lane)
    if (lane == 0) committed = PcgSeed((uint64_t)(tile + S.opt.seedOffset), tab);
    committed = ((uint64_t)__shfl((unsigned)(committed >> 32), 0) << 32) | __shfl((unsigned)committed, 0);
    __syncthreads();
    LdsStackT<GLOSSY> stk{sStack + lane, 64, 0};
    const int total = tw * th * directSpp;
    int n0 = 0;
    unsigned K = 10;
    auto commit = [&](const LaneSink &ls) {  // Splat (dchain.h), into the tile accumulator when the pixel is the tile's
        for (int k = 0; k < ls.n && k < 2; k++) {
            if (!AllFinite(ls.c[k])) continue;
            const int ix = Clampi((int)(ls.screenPos.x * film.W), 0, film.W - 1), iy = Clampi((int)(ls.screenPos.y * film.H), 0, film.H - 1);
            if (ix >= x0 && ix < x1 && iy >= y0 && iy < y1) {
                float *px = sTile + ((iy - y0) * 16 + (ix - x0)) * 3;
                px[0] += ls.c[k].x, px[1] += ls.c[k].y, px[2] += ls.c[k].z;
            } else {
                float *px = film.rgb + ((size_t)iy * film.W + ix) * 3;
                unsafeAtomicAdd(px + 0, ls.c[k].x), unsafeAtomicAdd(px + 1, ls.c[k].y), unsafeAtomicAdd(px + 2, ls.c[k].z);
            }
        }
    };
    while (n0 < total) {
        const int active = min(64, total - n0);
        LaneSink ls;
        ls.n = 0;
        SpecRng rng{PcgAdvance(committed, (uint64_t)lane * K), tab, 0u, false};
        if (lane < active) {
            const int n = n0 + lane, pix = n / directSpp;
            DirectSample(S, ls, x0 + pix % tw, y0 + pix / tw, minDepth, maxDepth, rng, stk);
        }
        const unsigned long long bad = __ballot(lane < active && (rng.draws != K || rng.crossed));
        const int m = bad ? __ffsll((long long)bad) - 1 : active;  // lanes [0, m) consumed exactly K draws each
        int nCommit = m;
        uint64_t advance = (uint64_t)m * K;
        bool sequential = false;
        if (m < active) {
            if (__shfl((int)rng.crossed, m)) sequential = true;  // the table advances inside sample n0 + m: run it for real
            else {
                const unsigned drawsM = __shfl(rng.draws, m);
                nCommit = m + 1, advance += drawsM, K = drawsM;
            }
        }
        for (int j = 0; j < nCommit; j++)  // stream order
            if (lane == j) commit(ls);
        committed = PcgAdvance(committed, advance);
        n0 += nCommit;
        if (sequential) {
            __syncthreads();
            uint64_t st = 0;
            if (lane == 0) {
                Rng real;
                real.state = committed, real.tab = tab, real.ticks = 0;
                LaneSink one;
                one.n = 0;
                const int pix = n0 / directSpp;
                DirectSample(S, one, x0 + pix % tw, y0 + pix / tw, minDepth, maxDepth, real, stk);
                commit(one);
                st = real.state;
            }
            committed = ((uint64_t)__shfl((unsigned)(st >> 32), 0) << 32) | __shfl((unsigned)st, 0);
            n0 += 1;
            __syncthreads();  // the advanced table is visible to every lane
        }
    }
    __syncthreads();
    for (int k = lane; k < 16 * 16; k += 64) {
        const int ix = x0 + k % 16, iy = y0 + k / 16;
        if (ix < x1 && iy < y1) {
            float *px = film.rgb + ((size_t)iy * film.W + ix) * 3;
            unsafeAtomicAdd(px + 0, sTile[k * 3 + 0]), unsafeAtomicAdd(px + 1, sTile[k * 3 + 1]), unsafeAtomicAdd(px + 2, sTile[k * 3 + 2]);
        }
    }
}

// cross-check estimator: plain Monte Carlo over GeneratePathBidir samples (uniform screen positions), every contribution
// splatted unweighted; image = film * W * H / numSamples.  Isolates the bidirectional generator from the Markov chain.
template <bool GLOSSY>
__global__ void k_bidir_mc(DScene S, Film film, int nThreads, int samplesPerThread, uint32_t *tabScratch, float *contribScratch) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nThreads) return;
    Rng rng;
    rng.tab = tabScratch + (size_t)t * 64;
    rng.state = PcgSeed((uint64_t)(t + S.opt.seedOffset), rng.tab);
    rng.ticks = 0;
    LocalStackT<GLOSSY> stk;
    DPath path;
    for (int s = 0; s < samplesPerThread; s++) {
        ContribSink sink{contribScratch, (size_t)nThreads, (size_t)t, 0};
        GeneratePathBidir(S, max(S.opt.minDepth, 3), S.opt.maxDepth, path, sink, rng, stk);
        for (int k = 0; k < sink.count; k++) {
            Contrib c = sink.Get(k);
            Splat(film, c.screenPos, c.contrib);
        }
    }
}

// chain set-up (mlt.cpp:60-90): current state = the chain's init state, marked invalid (mlt.h:121: every chain begins with an
// unconditionally accepted large step), everything else cleared.  The init arrays hold this rank's chains only.
__global__ void k_setup_chains(ChainArrays A, int chainBegin, long long perChain, long long chainsNeedExtra) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.N) return;
    const int gid = chainBegin + i;
    DPath p;
    LoadPath(A.initPath, A.N, i, p);
    StorePath(A.curPath, A.N, i, p);
    StoreContrib(A.curContrib, A.N, i, LoadContrib(A.initContrib, A.N, i));
    A.scoreSum[i] = A.initScoreSum[i];
    A.flags[i] = 0;
    A.curSplatCount[i] = 0;
    A.pathWeight[i] = 0.f;
    A.lastScoreSum[i] = 1.0f;
    A.lastScore[i] = 1.0f;
    A.adjacentReject[i] = 0;
    A.sampleIdx[i] = 0;
    A.numSamples[i] = (int)(perChain + (gid < chainsNeedExtra ? 1 : 0));
    A.pushDim[i] = 0;
}

// kind of the very first step (only needed when chains start valid: an invalid state always takes a large step)
__global__ void k_first_kind(DScene S, const DCache *cache, ChainArrays A, StepParams P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.N) return;
    Rng rng = LoadChainRng(A, P.chainBegin, S.opt.seedOffset, i);
    QueueNext(S, *cache, A, P, i, rng);
    StoreChainRng(A, i, rng);
}

// Work lists of the next step.  Each block owns a tile of 1024 consecutive chains and reserves one contiguous range per
// list with a single atomic.
//   * large / generic lists: filled in chain-id order (ordered scan), so that the SoA chain state is read and written in
//     whole cache lines.  (Appending straight from the step kernels scrambles the order a little more every step until
//     every lane touches its own cache line: profiles/r01_b_*.)
//   * plain small steps: counting sort of the tile's entries by technique key (QueueNext), so that a wave of the lean
//     kernel retraces ONE technique -- same ray count, same terminal strategy -- instead of the 5 different ones a wave of
//     64 consecutive chains holds on average (profiles/r02_*).  A wave's chains still come from one 1024-chain tile: its
//     state accesses stay within a few cache lines per word.
// bin of a plain entry inside its tile: the full technique key (sortPlain 1, 2) or just "path length >= 5" (sortPlain 3: two
// classes, so that a wave's lanes stay almost as dense in chain order as unsorted ones)
__device__ inline int PlainSortBin(unsigned char nk, int sortPlain) { return sortPlain == 3 ? ((nk >> 2) / 6 >= 2 ? 1 : 0) : (nk >> 2); }
template <int CPT>  // chains per thread: the tile is 256 * CPT consecutive chains
__global__ void __launch_bounds__(256) k_build_lists(ChainArrays A, NextLists next, int sortPlain, unsigned leanDims) {
    __shared__ unsigned long long sWave[4];
    __shared__ int sBase[3];
    __shared__ int sHist[64], sStart[64];
    if (threadIdx.x < 64) sHist[threadIdx.x] = 0;
    __syncthreads();
    const int first = (blockIdx.x * 256 + threadIdx.x) * CPT;
    unsigned char k[CPT];
    for (int j = 0; j < CPT; j++) k[j] = 0;
    if (CPT == 4 && first + 3 < A.N) {
        const uchar4 v = *reinterpret_cast<const uchar4 *>(A.nextKind + first);
        k[0] = v.x, k[1 % CPT] = v.y, k[2 % CPT] = v.z, k[3 % CPT] = v.w;
    } else {
        for (int j = 0; j < CPT; j++)
            if (first + j < A.N) k[j] = A.nextKind[first + j];
    }
    // a generic entry whose cache became ready after the chain queued it (leanDims, context.cpp) is a plain one now; its
    // technique key gives the dimension: 2 * path length
    for (int j = 0; j < CPT; j++)
        if ((k[j] & 3) == NEXT_SMALL_GENERIC && ((leanDims >> (2 * (3 + (k[j] >> 2) / 6))) & 1u) && !((leanDims >> 31) && (k[j] >> 2) % 6 > 1))  // bit 31: lean launch without light sub-paths
            k[j] = (unsigned char)(k[j] | NEXT_SMALL_PLAIN);
    if (A.stepKind)  // relocate.hip: which launch runs the slot's chain in the step these lists are for
        for (int j = 0; j < CPT; j++)
            if (first + j < A.N) A.stepKind[first + j] = k[j] & 3;
    // four 16-bit counters packed into one word: [large | generic | plain A | plain B]; B = the "long path" class of sortPlain 3
    // (a STABLE two-way partition: inside a class the entries keep their chain order, so a wave's lanes stay dense)
    auto fieldOf = [&](unsigned char nk) {
        const int kind = nk & 3;
        return kind == NEXT_SMALL_PLAIN ? (sortPlain == 3 && PlainSortBin(nk, 3) ? 3 : 2) : kind - 1;
    };
    const bool histSort = sortPlain == 1 || sortPlain == 2;
    unsigned long long mine = 0;
    for (int j = 0; j < CPT; j++) {
        const int kind = k[j] & 3;
        if (kind) mine += 1ull << (16 * fieldOf(k[j]));
        if (kind == NEXT_SMALL_PLAIN && histSort) atomicAdd(&sHist[PlainSortBin(k[j], sortPlain)], 1);
    }
    unsigned long long incl = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    unsigned long long before = 0, total = 0;
    for (int w = 0; w < 4; w++) {
        if (w < wave) before += sWave[w];
        total += sWave[w];
    }
    const int totalA = (int)((total >> 32) & 0xffff), totalB = (int)((total >> 48) & 0xffff);
    if (threadIdx.x < 3) {
        const int n = threadIdx.x < 2 ? (int)((total >> (16 * threadIdx.x)) & 0xffff) : totalA + totalB;
        sBase[threadIdx.x] = n ? atomicAdd(&next.counts[threadIdx.x], n) : 0;
    }
    if (wave == 1) {  // exclusive prefix of the key histogram; the bins then serve as cursors
        int h = sHist[lane], inc = h;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        sStart[lane] = inc - h;
    }
    __syncthreads();
    const unsigned long long excl = before + incl - mine;
    int pos[4] = {sBase[0] + (int)(excl & 0xffff), sBase[1] + (int)((excl >> 16) & 0xffff), sBase[2] + (int)((excl >> 32) & 0xffff),
                  sBase[2] + totalA + (int)((excl >> 48) & 0xffff)};
    int *lists[4] = {next.large, next.smallGrad, next.smallPlain, next.smallPlain};
    for (int j = 0; j < CPT; j++) {
        const int kind = k[j] & 3;
        if (!kind) continue;
        if (kind == NEXT_SMALL_PLAIN && histSort) next.smallPlain[sBase[2] + atomicAdd(&sStart[PlainSortBin(k[j], sortPlain)], 1)] = first + j;
        else {
            const int f = fieldOf(k[j]);
            lists[f][pos[f]++] = first + j;
        }
    }
}

// first step: every chain starts invalid -> large step (mlt.cpp:97)
__global__ void k_init_lists(int n, int *large, int *counts) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) large[i] = i;
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[0] = n, counts[1] = 0, counts[2] = 0;
}

// Global-cache pushes of one step (mlt.cpp:120-127, global_cache.h:66-92), applied after the step in chain-id order
// (the lock-step contract of DESIGN.md).  Three small launches replace the per-dim single-block scan of round 1:
//   k_push_count   per 1024-chain tile, how many chains push to each dim (slot = (dim - 6) / 2; MLT states have L >= 3)
//   k_push_scatter every tile ranks its pushes behind those of all lower tiles and copies them into the cache rows
//   k_push_finish  adds the totals to the per-dim fill counts
// Counts of the four slots travel packed 4 x 16 bit (a tile holds 1024 chains, so a field never overflows).
LMC_D unsigned long long PushKey(int dim) { return (dim >= 6 && dim <= PSS_MAX_LENGTH && !(dim & 1)) ? 1ull << (16 * ((dim - 6) >> 1)) : 0ull; }

// One WAVE per tile (64 threads x 16 chains): the pack runs beside the small-step launches, whose waves hold every SIMD's registers -- a
// one-wave block takes the first slot that frees up, a four-wave block waited for four at once, i.e. for the tail of the hot launch
// (0.9 ms for k_push_count, profiles/r04_fill_r_*).  Block 0 also zeroes the stage's row counts (no separate fill launch, same reason).
constexpr int PUSH_PER = 16;  // chains per thread: a tile = 64 * PUSH_PER = 1024 chains
__global__ void __launch_bounds__(64) k_push_count(ChainArrays A, unsigned long long *tileCounts, int *stageCounts) {
    if (stageCounts && blockIdx.x == 0 && threadIdx.x < 16) stageCounts[threadIdx.x] = 0;
    const int first = (blockIdx.x * 64 + threadIdx.x) * PUSH_PER;  // chain ids; their slots: A.slotOf once chains are relocated (relocate.hip)
    unsigned long long mine = 0;
    for (int j = 0; j < PUSH_PER; j++)
        if (first + j < A.N) mine += PushKey(A.pushDim[A.slotOf ? A.slotOf[first + j] : first + j]);
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if (threadIdx.x == 0) tileCounts[blockIdx.x] = mine;
}

__global__ void __launch_bounds__(64) k_push_scatter(ChainArrays A, const unsigned long long *tileCounts, CachePushTargets T) {
    if (tileCounts[blockIdx.x] == 0) return;  // nothing to push in this tile (the common case once the caches fill up)
    // pushes of all lower tiles; a lower tile's field can exceed 16 bits only in the sum, so widen while adding
    unsigned long long lo = 0, hi = 0;  // slots 0,1 (32 bit each) | slots 2,3
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 64) {
        const unsigned long long v = tileCounts[b];
        lo += (v & 0xffffull) | (((v >> 16) & 0xffffull) << 32);
        hi += ((v >> 32) & 0xffffull) | (((v >> 48) & 0xffffull) << 32);
    }
    for (int off = 32; off > 0; off >>= 1) lo += __shfl_down(lo, off), hi += __shfl_down(hi, off);
    const unsigned long long sLo = __shfl(lo, 0), sHi = __shfl(hi, 0);
    const int first = (blockIdx.x * 64 + threadIdx.x) * PUSH_PER;
    int dims[PUSH_PER];
    unsigned long long mine = 0;
    for (int j = 0; j < PUSH_PER; j++) {
        dims[j] = 0;
        if (first + j < A.N) {
            dims[j] = A.pushDim[A.slotOf ? A.slotOf[first + j] : first + j];
            if (!PushKey(dims[j])) dims[j] = 0;
            mine += PushKey(dims[j]);
        }
    }
    unsigned long long incl = mine;
    const int lane = threadIdx.x;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    const unsigned long long excl = incl - mine;
    const size_t N = A.N;
    for (int j = 0; j < PUSH_PER; j++) {
        const int dim = dims[j];
        if (!dim) continue;
        const int slot = (dim - 6) >> 1;
        const long long lower = slot == 0 ? (long long)(sLo & 0xffffffffull) : slot == 1 ? (long long)(sLo >> 32) : slot == 2 ? (long long)(sHi & 0xffffffffull) : (long long)(sHi >> 32);
        int rank = (int)((excl >> (16 * slot)) & 0xffffull);
        for (int jj = 0; jj < j; jj++) rank += dims[jj] == dim ? 1 : 0;
        const long long row = (long long)T.count[slot] + lower + rank;
        const int i = A.slotOf ? A.slotOf[first + j] : first + j;
        if (row < PSS_MAX_SIZE) {
            for (int k = 0; k < dim; k++) {
                T.pss[slot][(size_t)row * dim + k] = A.pushData[(size_t)k * N + i];
                T.v1[slot][(size_t)row * dim + k] = A.pushData[(size_t)(MAXPSS + k) * N + i];
                T.v2[slot][(size_t)row * dim + k] = A.pushData[(size_t)(2 * MAXPSS + k) * N + i];
            }
            T.weight[slot][row] = A.pushData[(size_t)(3 * MAXPSS) * N + i];
            if (T.extra[slot]) {  // `samplecache`: chain.path, chain.spContrib (no step of this chain ran since the push was decided)
                float *x = T.extra[slot] + (size_t)row * CACHE_ROW_EXTRA;
                for (int k = 0; k < DPATH_WORDS; k++) x[k] = A.chPath[(size_t)k * N + i];
                for (int k = 0; k < CONTRIB_WORDS; k++) x[DPATH_WORDS + k] = A.chContrib[(size_t)k * N + i];
            }
        }
        A.pushDim[i] = 0;  // consumed: a chain that has run its last step must not be pushed again by the following steps
    }
}

__global__ void __launch_bounds__(64) k_push_finish(const unsigned long long *tileCounts, int nTiles, CachePushTargets T) {
    unsigned long long lo = 0, hi = 0;
    for (int b = threadIdx.x; b < nTiles; b += 64) {
        const unsigned long long v = tileCounts[b];
        lo += (v & 0xffffull) | (((v >> 16) & 0xffffull) << 32);
        hi += ((v >> 32) & 0xffffull) | (((v >> 48) & 0xffffull) << 32);
    }
    for (int off = 32; off > 0; off >>= 1) lo += __shfl_down(lo, off), hi += __shfl_down(hi, off);
    const unsigned long long sLo = __shfl(lo, 0), sHi = __shfl(hi, 0);
    if (threadIdx.x < CACHE_SLOTS) {
        const int slot = threadIdx.x;
        const long long add = slot == 0 ? (long long)(sLo & 0xffffffffull) : slot == 1 ? (long long)(sLo >> 32) : slot == 2 ? (long long)(sHi & 0xffffffffull) : (long long)(sHi >> 32);
        const long long n = (long long)T.count[slot] + add;
        T.count[slot] = (int)(n < PSS_MAX_SIZE ? n : PSS_MAX_SIZE);
    }
}

// ---- multi-rank form of the push: every rank first writes its pushes, in its own chain order, into a STAGE with the cache's row
// layout (k_push_count / scatter / finish above with the stage as target and a zeroed count), the ranks all-gather their stages, and
// every rank appends the gathered rows in rank order = global chain-id order.  All ranks therefore hold the same cache at every step
// -- the cache a single rank with all the chains would hold -- and an N-rank render follows the one-rank trajectories.
// One block per (slot, rank); a rank's rows land behind the cache's rows and the rows of the lower ranks, cut at PSS_MAX_SIZE.
__global__ void __launch_bounds__(64) k_push_apply(const float *gathered, size_t stageFloats, int world, PushStageLayout lay, CachePushTargets T) {
    const int slot = blockIdx.x, r = blockIdx.y;
    if (!T.pss[slot]) return;
    const int dim = 6 + 2 * slot;
    long long base = T.count[slot];
    for (int q = 0; q < r; q++) base += reinterpret_cast<const int *>(gathered + (size_t)q * stageFloats)[slot];
    const float *st = gathered + (size_t)r * stageFloats;
    const int n = reinterpret_cast<const int *>(st)[slot];
    const long long room = (long long)PSS_MAX_SIZE - base;
    const int take = (int)(room <= 0 ? 0 : (n < room ? n : room));
    const float *pss = st + lay.pss[slot], *v1 = st + lay.v1[slot], *v2 = st + lay.v2[slot], *w = st + lay.weight[slot];
    for (int e = threadIdx.x; e < take * dim; e += blockDim.x) {
        const size_t d = (size_t)base * dim + e;
        T.pss[slot][d] = pss[e], T.v1[slot][d] = v1[e], T.v2[slot][d] = v2[e];
    }
    for (int e = threadIdx.x; e < take; e += blockDim.x) T.weight[slot][base + e] = w[e];
    if (T.extra[slot]) {
        const float *x = st + lay.extra[slot];
        for (size_t e = threadIdx.x; e < (size_t)take * CACHE_ROW_EXTRA; e += blockDim.x) T.extra[slot][(size_t)base * CACHE_ROW_EXTRA + e] = x[e];
    }
}
// hostCounts: the pinned mirror of the fill counts, written from here (no copy launch behind this one; visible to the host once the
// event recorded behind the launch has completed)
__global__ void k_push_apply_finish(const float *gathered, size_t stageFloats, int world, CachePushTargets T, int *hostCounts) {
    const int slot = threadIdx.x;
    if (slot >= CACHE_SLOTS) return;
    long long n = T.count[slot];
    for (int q = 0; q < world; q++) n += reinterpret_cast<const int *>(gathered + (size_t)q * stageFloats)[slot];
    T.count[slot] = (int)(n < PSS_MAX_SIZE ? n : PSS_MAX_SIZE);
    if (hostCounts) {
        hostCounts[slot] = T.count[slot];
        __threadfence_system();
    }
}

}  // namespace lmcd

// ================================================================================================ launch glue
using namespace lmcd;

void LaunchLayoutProbe(int N, int words, int mode, int batch, const float *in, float *out, hipStream_t s) {
    if (batch >= 12) hipLaunchKernelGGL(k_layout_probe<12>, dim3((N + 63) / 64), dim3(64), 0, s, N, words, mode, in, out);
    else
        hipLaunchKernelGGL(k_layout_probe<4>, dim3((N + 63) / 64), dim3(64), 0, s, N, words, mode, in, out);
}


static inline int GridFor(long long n, int block, int maxBlocks = 2048) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > maxBlocks) g = maxBlocks;
    return (int)g;
}

void LaunchSeedRng(int n, long long firstSeed, uint64_t *state, uint32_t *tab, hipStream_t s) {
    hipLaunchKernelGGL(k_seed_rng, dim3(GridFor(n, 256)), dim3(256), 0, s, n, firstSeed, state, tab);
}
void LaunchRngProbe(int nSeeds, const unsigned long long *seeds, int mode, int n, float mean, float stddev, uint32_t *tabScratch, uint32_t *out, hipStream_t s) {
    hipLaunchKernelGGL(k_rng_probe, dim3((nSeeds + 63) / 64), dim3(64), 0, s, nSeeds, seeds, mode, n, mean, stddev, tabScratch, out);
}
void LaunchLowerBoundProbe(int n, const float *cdf, int nq, const float *u, int *out, hipStream_t s) {
    hipLaunchKernelGGL(k_lower_bound_probe, dim3((nq + 63) / 64), dim3(64), 0, s, n, cdf, nq, u, out);
}
void LaunchTrace(const DScene &S, int n, const float *rays, int *prim, float *t, int anyHit, hipStream_t s) {
    hipLaunchKernelGGL(k_trace, dim3(GridFor(n, 256)), dim3(256), 0, s, S, n, rays, prim, t, anyHit);
}
void LaunchKdProbe(const DCacheDim &C, int dim, int nq, const float *q, float radiusSq, int knn, int *outN, int *outIdx, float *outDist, hipStream_t s) {
    hipLaunchKernelGGL(k_kd_probe, dim3((nq + 63) / 64), dim3(64), 0, s, C, dim, nq, q, radiusSq, knn, outN, outIdx, outDist);
}
void LaunchGaussProbe(int n, int dim, const float *v1, const float *M, float ss, float shk, const float *sc, const float *offset, float *out, hipStream_t s) {
    hipLaunchKernelGGL(k_gauss_probe, dim3((n + 63) / 64), dim3(64), 0, s, n, dim, v1, M, ss, shk, sc, offset, out);
}
void LaunchGradBatch(int c, int l, int n, const float *primarySoA, const float *scene, const float *vertSoA, float *logLum, float *gradSoA, int wantGrad,
                     hipStream_t s) {
    hipLaunchKernelGGL(k_grad_batch, dim3(GridFor(n, 128, 4096)), dim3(128), 0, s, c, l, n, primarySoA, scene, vertSoA, logLum, gradSoA, wantGrad);
}
void LaunchInitPass1(const DScene &S, int tBegin, int nStreams, long long perThread, long long extra, long long gBase, uint32_t *tabScratch, float *contribScratch,
                     uint64_t *ckState, uint32_t *ckTicks, unsigned char *count, hipStream_t s) {
    if (nStreams <= 0) return;
    if (S.glossy) hipLaunchKernelGGL(k_init_pass1<true>, dim3((nStreams + 127) / 128), dim3(128), 0, s, S, tBegin, nStreams, perThread, extra, gBase, tabScratch, contribScratch, ckState, ckTicks, count);
    else
        hipLaunchKernelGGL(k_init_pass1<false>, dim3((nStreams + 127) / 128), dim3(128), 0, s, S, tBegin, nStreams, perThread, extra, gBase, tabScratch, contribScratch, ckState, ckTicks, count);
}
void LaunchInitPass2(const DScene &S, long long gBegin, long long numLocal, long long perThread, long long extra, int nSlots, uint32_t *tabScratch, float *contribScratch,
                     const uint64_t *ckState, const uint32_t *ckTicks, const unsigned long long *offset, unsigned char *outCL, float *outLs, hipStream_t s) {
    if (numLocal <= 0) return;
    if (S.glossy) hipLaunchKernelGGL(k_init_pass2<true>, dim3((nSlots + 127) / 128), dim3(128), 0, s, S, gBegin, numLocal, perThread, extra, nSlots, tabScratch, contribScratch, ckState,
                       ckTicks, offset, outCL, outLs);
    else
        hipLaunchKernelGGL(k_init_pass2<false>, dim3((nSlots + 127) / 128), dim3(128), 0, s, S, gBegin, numLocal, perThread, extra, nSlots, tabScratch, contribScratch, ckState,
                       ckTicks, offset, outCL, outLs);
}
void LaunchInitRegen(const DScene &S, int numChains, long long perThread, long long extra, const long long *seedSample, const unsigned char *seedCL,
                     uint32_t *tabScratch, float *contribScratch, const uint64_t *seedCkState, const uint32_t *seedCkTicks, float *initPath, float *initContrib,
                     float *initScoreSum, hipStream_t s) {
    if (S.glossy) hipLaunchKernelGGL(k_init_regen<true>, dim3((numChains + 127) / 128), dim3(128), 0, s, S, numChains, perThread, extra, seedSample, seedCL, tabScratch,
                       contribScratch, seedCkState, seedCkTicks, initPath, initContrib, initScoreSum);
    else
        hipLaunchKernelGGL(k_init_regen<false>, dim3((numChains + 127) / 128), dim3(128), 0, s, S, numChains, perThread, extra, seedSample, seedCL, tabScratch,
                       contribScratch, seedCkState, seedCkTicks, initPath, initContrib, initScoreSum);
}
void LaunchDirect(const DScene &S, const Film &film, int directSpp, int minDepth, int maxDepth, int bvhDepth, bool waveKernel, uint32_t *tabScratch, hipStream_t s) {
    const int nX = (S.cam.width + 15) / 16, nY = (S.cam.height + 15) / 16;
    if (waveKernel && maxDepth >= 0 && maxDepth <= 2 && bvhDepth <= BVH_LDS_STACK) {
        if (S.glossy) hipLaunchKernelGGL(k_direct_wave<true>, dim3(nX * nY), dim3(64), 0, s, S, film, directSpp, minDepth, maxDepth, nX, tabScratch);
        else
            hipLaunchKernelGGL(k_direct_wave<false>, dim3(nX * nY), dim3(64), 0, s, S, film, directSpp, minDepth, maxDepth, nX, tabScratch);
        return;
    }
    if (S.glossy) hipLaunchKernelGGL(k_direct<true>, dim3((nX * nY + 63) / 64), dim3(64), 0, s, S, film, directSpp, minDepth, maxDepth, nX, nY, tabScratch);
    else
        hipLaunchKernelGGL(k_direct<false>, dim3((nX * nY + 63) / 64), dim3(64), 0, s, S, film, directSpp, minDepth, maxDepth, nX, nY, tabScratch);
}
void LaunchBidirMC(const DScene &S, const Film &film, int nThreads, int samplesPerThread, uint32_t *tabScratch, float *contribScratch, hipStream_t s) {
    if (S.glossy) hipLaunchKernelGGL(k_bidir_mc<true>, dim3((nThreads + 127) / 128), dim3(128), 0, s, S, film, nThreads, samplesPerThread, tabScratch, contribScratch);
    else
        hipLaunchKernelGGL(k_bidir_mc<false>, dim3((nThreads + 127) / 128), dim3(128), 0, s, S, film, nThreads, samplesPerThread, tabScratch, contribScratch);
}
void LaunchSetupChains(const ChainArrays &A, int chainBegin, long long perChain, long long chainsNeedExtra, hipStream_t s) {
    hipLaunchKernelGGL(k_setup_chains, dim3((A.N + 255) / 256), dim3(256), 0, s, A, chainBegin, perChain, chainsNeedExtra);
}
void LaunchFirstKind(const DScene &S, const DCache *cache, const ChainArrays &A, const StepParams &P, hipStream_t s) {
    hipLaunchKernelGGL(k_first_kind, dim3((A.N + 255) / 256), dim3(256), 0, s, S, cache, A, P);
}
// ---------------------------------------------------------------------------------------------------------------------------
// Dilated grid of one cache dim (DCacheDim::gridStart / gridRows, dchain.h), built on the device from the point rows that are
// already there: count the (point, neighbour cell) pairs, scan the counts, scatter the rows.  (On the host the same build
// cost 8-26 ms per dim, lmc::BuildCacheGrid in accel.cpp, which stays as the checker.)  The order of the rows inside a cell
// depends on the atomics; the existence test ORs over a cell's rows, so it does not see it.
struct GridShape {
    int G, m, nbrs, dim, n;
    int coord[4];
};
__device__ inline bool GridNeighbourCell(const GridShape &g, const float *p, int o, int &cell) {
    cell = 0;
    bool inside = true;
    for (int k = 0; k < g.m; k++, o /= 3) {
        const int ck = CacheGridCell(p[g.coord[k]], g.G) + (o % 3) - 1;
        inside = inside && ck >= 0 && ck < g.G;
        cell = cell * g.G + ck;
    }
    return inside;
}
__global__ void __launch_bounds__(256) k_grid_count(GridShape g, const float *pts, int *start) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.n * g.nbrs) return;
    int cell;
    if (GridNeighbourCell(g, pts + (size_t)(t / g.nbrs) * g.dim, t % g.nbrs, cell)) atomicAdd(&start[cell + 1], 1);
}
constexpr int SCAN_ITEMS = 8, SCAN_TILE = 256 * SCAN_ITEMS;
// inclusive scan of v[0..n) in three launches: tiles of 2048, the tile totals (one block), the carry-in
__global__ void __launch_bounds__(256) k_scan_tiles(int *v, int n, int *tileSums) {
    __shared__ int warpSums[4];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int x[SCAN_ITEMS], run = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) x[k] = run += (base + k < n ? v[base + k] : 0);
    int incl = run;  // inclusive scan of the thread totals: wave shuffle, then the four wave totals
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63) warpSums[wave] = incl;
    __syncthreads();
    int carry = incl - run;
    for (int w = 0; w < wave; w++) carry += warpSums[w];
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n) v[base + k] = x[k] + carry;
    if (threadIdx.x == 255) tileSums[blockIdx.x] = carry + run;
}
__global__ void __launch_bounds__(256) k_scan_sums(int *tileSums, int nTiles) {  // one block; nTiles is a few thousand at most
    __shared__ int warpSums[4];
    __shared__ int carryIn;
    if (threadIdx.x == 0) carryIn = 0;
    __syncthreads();
    for (int b = 0; b < nTiles; b += 256) {
        const int i = b + threadIdx.x, val = i < nTiles ? tileSums[i] : 0;
        int incl = val;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(incl, d);
            if (lane >= d) incl += y;
        }
        if (lane == 63) warpSums[wave] = incl;
        __syncthreads();
        int carry = carryIn;
        for (int w = 0; w < wave; w++) carry += warpSums[w];
        if (i < nTiles) tileSums[i] = incl + carry;
        __syncthreads();
        if (threadIdx.x == 255) carryIn = incl + carry;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_scan_carry(int *v, int n, const int *tileSums) {
    if (blockIdx.x == 0) return;
    const int carry = tileSums[blockIdx.x - 1], base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n) v[base + k] += carry;
}
void LaunchInclusiveScan(int *v, int n, int *tileSums, hipStream_t s) {
    const int nTiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tiles, dim3(nTiles), dim3(256), 0, s, v, n, tileSums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, s, tileSums, nTiles);
    hipLaunchKernelGGL(k_scan_carry, dim3(nTiles), dim3(256), 0, s, v, n, tileSums);
}
__global__ void __launch_bounds__(256) k_grid_scatter(GridShape g, const float *pts, const int *start, int *cursor, unsigned short *idx) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.n * g.nbrs) return;
    const int row = t / g.nbrs;
    int cell;
    if (!GridNeighbourCell(g, pts + (size_t)row * g.dim, t % g.nbrs, cell)) return;
    idx[start[cell] + atomicAdd(&cursor[cell], 1)] = (unsigned short)row;
}
// occupancy word of 32 cells (start: the inclusive scan of the per-cell counts, start[c] .. start[c + 1] = cell c's range) and its number of
// non-empty cells, which the scan below turns into ranks
__global__ void __launch_bounds__(256) k_grid_words(const int *start, int cells, int nWords, uint2 *words, int *wordCount) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nWords) return;
    unsigned bits = 0;
    for (int b = 0; b < 32; b++) {
        const int c = w * 32 + b;
        if (c < cells && start[c + 1] > start[c]) bits |= 1u << b;
    }
    words[w].x = bits;
    wordCount[w] = __popc(bits);
}
// wordCount: inclusive scan of the counts -> rank of every word's first cell; the r-th non-empty cell's range start (and, behind the last, the total)
__global__ void __launch_bounds__(256) k_grid_compact(const int *start, int cells, int nWords, uint2 *words, const int *wordCountIncl, int *cellStart) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nWords) return;
    const unsigned bits = words[w].x;
    int r = wordCountIncl[w] - __popc(bits);
    words[w].y = (unsigned)r;
    for (int b = 0; b < 32; b++)
        if (bits & (1u << b)) cellStart[r++] = start[w * 32 + b];
    if (w == nWords - 1) cellStart[wordCountIncl[w]] = start[cells];
}
// scratchStart: cells + 1 ints, scratchCursor: cells ints, scratchWordCount: ceil(cells / 32) ints, tileSums: ceil((cells + 1) / 2048) + 1 ints -- build-time
// scratch, shared by the dims (the builds are stream-ordered); words: ceil(cells / 32); cellStart: min(cells, n * 3^m) + 1 ints; idx: n * 3^m
void LaunchBuildCacheGrid(const float *pts, int n, int dim, int G, int m, const int *coord, int *scratchStart, int *scratchCursor, int *scratchWordCount, int *tileSums,
                          uint2 *words, int *cellStart, unsigned short *idx, hipStream_t s) {
    GridShape g{G, m, 1, dim, n, {0, 1, 2, 3}};
    for (int k = 0; k < m && k < 4; k++) g.coord[k] = coord[k];
    size_t cells = 1;
    for (int k = 0; k < m; k++) cells *= G, g.nbrs *= 3;
    (void)hipMemsetAsync(scratchStart, 0, (cells + 1) * sizeof(int), s);
    (void)hipMemsetAsync(scratchCursor, 0, cells * sizeof(int), s);
    const int pairs = n * g.nbrs, nScan = (int)cells + 1, nWords = (int)((cells + 31) / 32);
    hipLaunchKernelGGL(k_grid_count, dim3((pairs + 255) / 256), dim3(256), 0, s, g, pts, scratchStart);
    LaunchInclusiveScan(scratchStart, nScan, tileSums, s);
    hipLaunchKernelGGL(k_grid_scatter, dim3((pairs + 255) / 256), dim3(256), 0, s, g, pts, scratchStart, scratchCursor, idx);
    hipLaunchKernelGGL(k_grid_words, dim3((nWords + 255) / 256), dim3(256), 0, s, scratchStart, (int)cells, nWords, words, scratchWordCount);
    LaunchInclusiveScan(scratchWordCount, nWords, tileSums, s);
    hipLaunchKernelGGL(k_grid_compact, dim3((nWords + 255) / 256), dim3(256), 0, s, scratchStart, (int)cells, nWords, words, scratchWordCount, cellStart);
}

void LaunchCachePush(const ChainArrays &A, const CachePushTargets &T, unsigned long long *tileCounts, int *stageCounts, hipStream_t s) {
    const int nTiles = (A.N + 1023) / 1024;
    hipLaunchKernelGGL(k_push_count, dim3(nTiles), dim3(64), 0, s, A, tileCounts, stageCounts);
    hipLaunchKernelGGL(k_push_scatter, dim3(nTiles), dim3(64), 0, s, A, tileCounts, T);
    hipLaunchKernelGGL(k_push_finish, dim3(1), dim3(64), 0, s, tileCounts, nTiles, T);
}

void LaunchCachePushApply(const float *gathered, size_t stageFloats, int world, const PushStageLayout &lay, const CachePushTargets &T, int *hostCounts, hipStream_t s) {
    hipLaunchKernelGGL(k_push_apply, dim3(CACHE_SLOTS, world), dim3(64), 0, s, gathered, stageFloats, world, lay, T);
    hipLaunchKernelGGL(k_push_apply_finish, dim3(1), dim3(64), 0, s, gathered, stageFloats, world, T, hostCounts);
}

void LaunchTransProbe(int n, int mode, const float *x, const float *y, float *o, hipStream_t s) {
    hipLaunchKernelGGL(k_trans_probe, dim3(GridFor(n, 256)), dim3(256), 0, s, n, mode, x, y, o);
}
// dst += src (the film merge of an in-process group of contexts, host/context.cpp lmc_group_film_reduce)
template <class T>
__global__ void k_add_into(T *dst, const T *src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
void LaunchAddInto(float *dst, const float *src, size_t n, hipStream_t s) { hipLaunchKernelGGL(k_add_into<float>, dim3(GridFor((long long)n, 256)), dim3(256), 0, s, dst, src, n); }
void LaunchAddIntoF64(double *dst, const double *src, size_t n, hipStream_t s) { hipLaunchKernelGGL(k_add_into<double>, dim3(GridFor((long long)n, 256)), dim3(256), 0, s, dst, src, n); }
// dst[0] = src[0] + .. + src[n - 1], left to right (the splat-weight sums of a group's members, in rank order: the same double on every member)
__global__ void k_sum_f64(double *dst, const double *src, int n) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double a = 0;
        for (int k = 0; k < n; k++) a += src[k];
        dst[0] = a;
    }
}
void LaunchSumF64(double *dst, const double *src, int n, hipStream_t s) { hipLaunchKernelGGL(k_sum_f64, dim3(1), dim3(64), 0, s, dst, src, n); }
// ---- the work lists of a pipeline stage (dh2coop.h H2Bins): counts -> offsets, then the chain ids scattered into ONE N-entry array
__global__ void __launch_bounds__(64) k_bins_scan(H2Bins bins) {
    constexpr int PER = H2_COUNT_WORDS / 64;
    const int lane = threadIdx.x;
    int c[PER], sum = 0;
    for (int k = 0; k < PER; k++) c[k] = bins.count[lane * PER + k], sum += c[k];
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    int run = incl - sum;
    for (int k = 0; k < PER; k++) {
        bins.start[lane * PER + k] = run, bins.cursor[lane * PER + k] = 0;
        run += c[k];
    }
}
__global__ void __launch_bounds__(64) k_bins_scatter(H2Bins bins, const int *list, const int *listCount) {
    const int total = *listCount, lane = threadIdx.x & 63;
    for (int j0 = blockIdx.x * 64; j0 < total; j0 += gridDim.x * 64) {  // whole waves: the ballots below need every lane
        const int j = j0 + lane;
        const int i = j < total ? list[j] : -1;
        const int t = i >= 0 ? bins.binOf[i] : -1;
        unsigned long long todo = __ballot(t >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int tl = __shfl(t, leader);
            const unsigned long long mask = __ballot(t == tl);
            int base = 0;
            if (lane == leader) base = atomicAdd(&bins.cursor[tl], __popcll(mask));
            base = __shfl(base, leader);
            if (t == tl) bins.items[bins.start[tl] + base + __popcll(mask & ((1ull << lane) - 1ull))] = i;
            todo &= ~mask;
        }
    }
}
// The generic list cut into `parts` interleaved sub-lists (groups of 64 entries go round robin: every part sees the same technique mix), each with
// its own count: the H2MC pipeline of every part then runs on a stream of its own (host/context.cpp LaunchGeneric).  sub: parts x stride entries.
__global__ void __launch_bounds__(64) k_split_list(const int *list, const int *listCount, int parts, int *sub, int stride, int *subCount) {  // one-wave blocks: they take the first free slot
    const int total = *listCount;
    for (int j = blockIdx.x * 64 + threadIdx.x; j < total; j += gridDim.x * 64) {
        const int g = j >> 6, h = g % parts;
        sub[(size_t)h * stride + (g / parts) * 64 + (j & 63)] = list[j];
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < parts) {
        const int h = threadIdx.x, groups = total >> 6, rem = total & 63;
        int n = (groups / parts + (h < groups % parts ? 1 : 0)) * 64;
        if (rem && groups % parts == h) n += rem;
        subCount[h] = n;
    }
}
void LaunchSplitList(const int *list, const int *listCount, int parts, int *sub, int stride, int *subCount, int gridBlocks, hipStream_t s) {
    hipLaunchKernelGGL(k_split_list, dim3(gridBlocks), dim3(64), 0, s, list, listCount, parts, sub, stride, subCount);
}
void LaunchBinsCompact(const H2Bins &bins, const int *list, const int *listCount, int gridBlocks, hipStream_t s) {
    hipLaunchKernelGGL(k_bins_scan, dim3(1), dim3(64), 0, s, bins);
    hipLaunchKernelGGL(k_bins_scatter, dim3(gridBlocks), dim3(64), 0, s, bins, list, listCount);
}
void LaunchStreamProbe(long long nWords, const float *in, float *out, hipStream_t s) {
    hipLaunchKernelGGL(k_stream_probe, dim3(8192), dim3(256), 0, s, nWords, in, out);
}
// Global counting sort of the generic small-step list by technique key.  The generic launches evaluate a path program per
// chain (gradient of the cache-filling launch, Hessian of H2MC): a wave whose 64 chains follow different techniques (c,l) runs
// their programs one after the other, so -- unlike the lean kernel, where the coalescing of the state loads mattered more
// (profiles/r02_b_*) -- grouping equal techniques pays (H2MC: +85 %).  Two levels: every block of SORT_CHUNK entries counts its
// keys in LDS, one small block turns the (block, key) counts into offsets, every block scatters its entries with LDS cursors.
// The order inside a (block, key) group is left to the LDS atomics: no result depends on the order of a list.
constexpr int SORT_CHUNK = 2048;
__global__ void __launch_bounds__(256) k_sort_hist(const unsigned char *nextKind, const int *in, const int *count, int *blockHist) {
    __shared__ int h[64];
    const int n = *count, base = blockIdx.x * SORT_CHUNK;
    if (base >= n) return;
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    for (int i = base + threadIdx.x; i < min(base + SORT_CHUNK, n); i += 256) atomicAdd(&h[nextKind[in[i]] >> 2], 1);
    __syncthreads();
    if (threadIdx.x < 64) blockHist[blockIdx.x * 64 + threadIdx.x] = h[threadIdx.x];
}
__global__ void __launch_bounds__(64) k_sort_scan(const int *count, int *blockHist) {  // counts -> first output index of every (block, key) group
    const int nBlocks = (*count + SORT_CHUNK - 1) / SORT_CHUNK, key = threadIdx.x;
    int total = 0;
    for (int b = 0; b < nBlocks; b++) total += blockHist[b * 64 + key];
    int incl = total;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (key >= off) incl += o;
    }
    int run = incl - total;  // entries with smaller keys
    for (int b = 0; b < nBlocks; b++) {
        const int c = blockHist[b * 64 + key];
        blockHist[b * 64 + key] = run;
        run += c;
    }
}
__global__ void __launch_bounds__(256) k_sort_scatter(const unsigned char *nextKind, const int *in, int *out, const int *count, const int *blockHist) {
    __shared__ int cursor[64];
    const int n = *count, base = blockIdx.x * SORT_CHUNK;
    if (base >= n) return;
    if (threadIdx.x < 64) cursor[threadIdx.x] = blockHist[blockIdx.x * 64 + threadIdx.x];
    __syncthreads();
    for (int i = base + threadIdx.x; i < min(base + SORT_CHUNK, n); i += 256) {
        const int chain = in[i];
        out[atomicAdd(&cursor[nextKind[chain] >> 2], 1)] = chain;
    }
}
// blockHist: 64 ints per SORT_CHUNK entries of the longest possible list (maxEntries)
void LaunchSortByTechnique(const unsigned char *nextKind, const int *in, int *out, const int *count, int *blockHist, int maxEntries, hipStream_t s) {
    const int nBlocks = (maxEntries + SORT_CHUNK - 1) / SORT_CHUNK;
    hipLaunchKernelGGL(k_sort_hist, dim3(nBlocks), dim3(256), 0, s, nextKind, in, count, blockHist);
    hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(64), 0, s, count, blockHist);
    hipLaunchKernelGGL(k_sort_scatter, dim3(nBlocks), dim3(256), 0, s, nextKind, in, out, count, blockHist);
}
void LaunchBuildLists(const ChainArrays &A, const NextLists &next, int sortPlain, unsigned leanDims, hipStream_t s) {
    // sortPlain: 0 = id order; 1 = technique sort inside 1024-chain tiles; 2 = inside 256-chain tiles (one tile = one 256-thread
    // block of the lean kernel, so the tile's cache lines are shared through that CU's L1)
    if (sortPlain == 2) hipLaunchKernelGGL(k_build_lists<1>, dim3((A.N + 255) / 256), dim3(256), 0, s, A, next, 1, leanDims);
    else
        hipLaunchKernelGGL(k_build_lists<4>, dim3((A.N + 1023) / 1024), dim3(256), 0, s, A, next, sortPlain, leanDims);
}
void LaunchInitLists(int n, int *large, int *counts, hipStream_t s) {
    hipLaunchKernelGGL(k_init_lists, dim3(GridFor(n, 256)), dim3(256), 0, s, n, large, counts);
}

// upload.h: host -> device words through a kernel (one block per 64 KB of a segment, blockIdx.y = segment)
__global__ void __launch_bounds__(256) k_upload_segments(UploadSegments U) {
    const int seg = blockIdx.y, n = U.words[seg];
    const unsigned *src = U.src[seg];
    unsigned *dst = U.dst[seg];
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) dst[k] = src[k];
}
void LaunchUploadSegments(const UploadSegments &U, hipStream_t s) {
    if (U.count <= 0) return;
    int most = 0;
    for (int k = 0; k < U.count; k++) most = max(most, U.words[k]);
    if (most <= 0) return;
    hipLaunchKernelGGL(k_upload_segments, dim3(min((most + 255) / 256, 64), U.count), dim3(256), 0, s, U);
}
