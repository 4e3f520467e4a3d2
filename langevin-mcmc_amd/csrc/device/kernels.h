// Launch entry points of kernels.hip (compiled by hipcc for gfx950); called from host/context.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include "dchain.h"
#include "dstep_params.h"

namespace lmcd {  // the H2MC launches' types (dh2coop.h, dh2mc.h): only named here, so that the LMC kernels do not depend on those headers
struct H2Arrays;
struct MalaPipe;
struct H2Bins;
struct H2MCParam;
}  // namespace lmcd

void LaunchSeedRng(int n, long long firstSeed, uint64_t *state, uint32_t *tab, hipStream_t s);
void LaunchLowerBoundProbe(int n, const float *cdf, int nq, const float *u, int *out, hipStream_t s);
void LaunchRngProbe(int nSeeds, const unsigned long long *seeds, int mode, int n, float mean, float stddev, uint32_t *tabScratch, uint32_t *out, hipStream_t s);
// single-call form of the plugin symbols (plugin.hip, step_small_h2mc.hip): in / out are device-visible pointers of host-mapped pinned buffers
void LaunchPluginGrad(int c, int l, const float *in, float *out, int wantGrad, hipStream_t s);
void LaunchPluginHess(int c, int l, const float *in, float *stage /* 655 floats of device memory */, float *out, hipStream_t s);
void LaunchCacheProbe(const lmcd::DCache *cache, int dim, int n, const float *u, int *row, const float *query, const int *cl, float *pdf, hipStream_t s);  // step_large_cache.hip
void LaunchTransProbe(int n, int mode, const float *x, const float *y, float *o, hipStream_t s);
void LaunchStreamProbe(long long nWords, const float *in, float *out, hipStream_t s);
void LaunchAddInto(float *dst, const float *src, size_t n, hipStream_t s);  // dst += src
void LaunchAddIntoF64(double *dst, const double *src, size_t n, hipStream_t s);
// a pipeline stage's work lists (dh2coop.h H2Bins): the counts turned into offsets, the chains of `list` scattered into the stage's N-entry array by bin
void LaunchBinsCompact(const lmcd::H2Bins &bins, const int *list, const int *listCount, int gridBlocks, hipStream_t s);
void LaunchSplitList(const int *list, const int *listCount, int parts, int *sub, int stride, int *subCount, int gridBlocks, hipStream_t s);
void LaunchSumF64(double *dst, const double *src, int n, hipStream_t s);  // dst[0] = sum of src[0 .. n), left to right
void LaunchTrace(const lmcd::DScene &S, int n, const float *rays, int *prim, float *t, int anyHit, hipStream_t s);
void LaunchKdProbe(const lmcd::DCacheDim &C, int dim, int nq, const float *q, float radiusSq, int knn, int *outN, int *outIdx, float *outDist, hipStream_t s);
void LaunchGaussProbe(int n, int dim, const float *v1, const float *M, float ss, float shk, const float *sc, const float *offset, float *out, hipStream_t s);
void LaunchGradBatch(int c, int l, int n, const float *primarySoA, const float *scene, const float *vertSoA, float *logLum, float *gradSoA, int wantGrad,
                     hipStream_t s);
// MLTInit, sharded by init stream: a rank runs the streams [tBegin, tBegin + nStreams); its sample arrays start at global sample gBase
void LaunchInitPass1(const lmcd::DScene &S, int tBegin, int nStreams, long long perThread, long long extra, long long gBase, uint32_t *tabScratch,
                     float *contribScratch, uint64_t *ckState, uint32_t *ckTicks, unsigned char *count, hipStream_t s);
void LaunchInitPass2(const lmcd::DScene &S, long long gBegin, long long numLocal, long long perThread, long long extra, int nSlots, uint32_t *tabScratch,
                     float *contribScratch, const uint64_t *ckState, const uint32_t *ckTicks, const unsigned long long *offset, unsigned char *outCL, float *outLs,
                     hipStream_t s);
void LaunchInitRegen(const lmcd::DScene &S, int numChains, long long perThread, long long extra, const long long *seedSample, const unsigned char *seedCL,
                     uint32_t *tabScratch, float *contribScratch, const uint64_t *seedCkState, const uint32_t *seedCkTicks, float *initPath, float *initContrib,
                     float *initScoreSum, hipStream_t s);
// direct.cpp:4-54; tabScratch: 64 words per 16x16 tile
// waveKernel: one wave per tile with speculative stream positions (kernels.hip) when maxDepth <= 2 and the BVH fits the LDS stack
void LaunchDirect(const lmcd::DScene &S, const lmcd::Film &film, int directSpp, int minDepth, int maxDepth, int bvhDepth, bool waveKernel, uint32_t *tabScratch,
                  hipStream_t s);
void LaunchBidirMC(const lmcd::DScene &S, const lmcd::Film &film, int nThreads, int samplesPerThread, uint32_t *tabScratch, float *contribScratch, hipStream_t s);
void LaunchSetupChains(const lmcd::ChainArrays &A, int chainBegin, long long perChain, long long chainsNeedExtra, hipStream_t s);
void LaunchFirstKind(const lmcd::DScene &S, const lmcd::DCache *cache, const lmcd::ChainArrays &A, const lmcd::StepParams &P, hipStream_t s);
// one of the three step launches (device/step_*.hip): chains of `list` (count read on the device) run one mutation and
// append themselves to the lists of the next step
void LaunchStepLarge(const lmcd::DScene &S, const lmcd::DCache *cache, const lmcd::ChainArrays &A, const lmcd::Film &film, const lmcd::StepParams &P,
                     const int *list, const int *listCount, const lmcd::NextLists &next, float *gradBuf, int gradStride, bool glossy, int gridBlocks, int bvhStackNeed, int blockThreads, hipStream_t s);
void LaunchStepLargeMux(const lmcd::DScene &S, const lmcd::DCache *cache, const lmcd::ChainArrays &A, const lmcd::Film &film, const lmcd::StepParams &P,
                     const int *list, const int *listCount, const lmcd::NextLists &next, float *gradBuf, int gradStride, bool glossy, int gridBlocks, int bvhStackNeed, int blockThreads, hipStream_t s);  // step_large_mux.hip
void LaunchStepLargeCache(const lmcd::DScene &S, const lmcd::DCache *cache, const lmcd::ChainArrays &A, const lmcd::Film &film, const lmcd::StepParams &P,
                     const int *list, const int *listCount, const lmcd::NextLists &next, float *gradBuf, int gradStride, bool glossy, int gridBlocks, int bvhStackNeed, int blockThreads, hipStream_t s);  // step_large_cache.hip
void LaunchStepSmallGrad(const lmcd::DScene &S, const lmcd::DCache *cache, const lmcd::ChainArrays &A, const lmcd::Film &film, const lmcd::StepParams &P,
                         const int *list, const int *listCount, const lmcd::NextLists &next, float *gradBuf, int gradStride, bool glossy, int gridBlocks, hipStream_t s);
void LaunchStepSmallLeanGrad(const lmcd::DScene &S, const lmcd::DCache *cache, const lmcd::ChainArrays &A, const lmcd::Film &film, const lmcd::StepParams &P, const int *list,
                             const int *listCount, const lmcd::NextLists &next, float *gradBuf, int gradStride, bool glossy, int gridBlocks, int blockThreads, int bvhStackNeed,
                             hipStream_t s);
void LaunchStepSmallPlain(const lmcd::DScene &S, const lmcd::DCache *cache, const lmcd::ChainArrays &A, const lmcd::Film &film, const lmcd::StepParams &P,
                          const int *list, const int *listCount, const lmcd::NextLists &next, int bvhDepth, bool glossy, int gridBlocks, int blockThreads,
                          bool profile, hipStream_t s);
// all small steps of an H2MC render: the launches of the wave-cooperative pipeline (device/dh2coop.h; step_h2_phases.hip, h2hess.hip, h2gauss.hip)
void LaunchH2Begin(const lmcd::DScene &S, const lmcd::ChainArrays &A, const lmcd::StepParams &P, const lmcd::H2Arrays &H, const int *list, const int *listCount, int gridBlocks,
                   hipStream_t s);
void LaunchH2Perturb(const lmcd::DScene &S, const lmcd::ChainArrays &A, const lmcd::StepParams &P, const lmcd::H2Arrays &H, const int *list, const int *listCount,
                     int bvhStackNeed, int gridBlocks, hipStream_t s);
void LaunchH2Finish(const lmcd::DScene &S, const lmcd::DCache *cache, const lmcd::ChainArrays &A, const lmcd::Film &film, const lmcd::StepParams &P, const lmcd::H2Arrays &H,
                    const int *list, const int *listCount, int gridBlocks, hipStream_t s);
void LaunchH2Hess(const float *rec, const lmcd::H2Bins &bins, int N, const float *scene38 /* host memory: passed by value */, float *hout, int gridBlocks, hipStream_t s);
void LaunchH2Gauss(const lmcd::H2Bins &bins, int N, const float *hout, const lmcd::H2MCParam &param, int expFlags, const int *chainFlags, int stage, float *gaussBuf,
                   const float *offsetSoA, float *px, int gridBlocks, hipStream_t s);
void LaunchH2Sample(const int *list, const int *listCount, int N, const int *chainFlags, const unsigned char *stepKind, const float *curContrib, const float *gaussBuf,
                    float sigma, float *offsetSoA, float *py, int gridBlocks, hipStream_t s);
// the gradient steps of the LMC cache-fill phase as a pipeline (device/dh2coop.h MalaPipe; step_mala_phases.hip, gradcoop.hip)
void LaunchMalaBegin(const lmcd::DScene &S, const lmcd::DCache *cache, const lmcd::ChainArrays &A, const lmcd::StepParams &P, const lmcd::MalaPipe &M, const int *list,
                     const int *listCount, int gridBlocks, hipStream_t s);
void LaunchMalaMid(const lmcd::DScene &S, const lmcd::DCache *cache, const lmcd::ChainArrays &A, const lmcd::StepParams &P, const lmcd::MalaPipe &M, const int *list,
                   const int *listCount, int bvhStackNeed, bool glossy, int gridBlocks, hipStream_t s);
void LaunchMalaFinish(const lmcd::DScene &S, const lmcd::DCache *cache, const lmcd::ChainArrays &A, const lmcd::Film &film, const lmcd::StepParams &P, const lmcd::MalaPipe &M,
                      const int *list, const int *listCount, int gridBlocks, hipStream_t s);
void LaunchMalaGrad(const float *rec, const lmcd::H2Bins &bins, int N, const float *scene38 /* host memory: passed by value */, float *gout, int gridBlocks, hipStream_t s);
// n gradient + Hessian evaluations of the (c,l) path program (the throughput form of the H2MC plugin symbols)
void LaunchHessBatch(int c, int l, int n, const float *primarySoA, const float *scene, const float *vertSoA, float *logLum, float *gradSoA, float *hessSoA,
                     hipStream_t s);
// id-ordered work lists of the next step from A.nextKind (coalescing: a wave's 64 list entries are (nearly) consecutive chains)
// sortPlain: group the plain small steps of every 1024-chain tile by technique (QueueNext's key)
void LaunchBuildLists(const lmcd::ChainArrays &A, const lmcd::NextLists &next, int sortPlain, unsigned leanDims, hipStream_t s);
void LaunchInitLists(int n, int *large, int *counts, hipStream_t s);
// pending global-cache pushes of the step just run, all dims in one pass, chain-id order; tileCounts: one word per 1024 chains
void LaunchCachePush(const lmcd::ChainArrays &A, const lmcd::CachePushTargets &T, unsigned long long *tileCounts, int *stageCounts /* zeroed first */, hipStream_t s);
// multi-rank push: append the gathered stages of all ranks (rank order) to the cache rows; see kernels.hip k_push_apply
void LaunchCachePushApply(const float *gathered, size_t stageFloats, int world, const lmcd::PushStageLayout &lay, const lmcd::CachePushTargets &T,
                          int *hostCounts /* pinned; nullptr: the fill counts stay on the device */, hipStream_t s);
// measurement aid: state-layout probe (kernels.hip k_layout_probe)
void LaunchLayoutProbe(int N, int words, int mode, int batch, const float *in, float *out, hipStream_t s);
// groups the entries of a work list by the technique key of A.nextKind (blockHist: 64 ints per 2048 entries of the longest list)
void LaunchSortByTechnique(const unsigned char *nextKind, const int *in, int *out, const int *count, int *blockHist, int maxEntries, hipStream_t s);
// inclusive scan of v[0, n) in place; tileSums: n / 2048 + 1 ints
void LaunchInclusiveScan(int *v, int n, int *tileSums, hipStream_t s);
// chain relocation (relocate.hip): the chains of the step's large-step launch whose technique changed since they were placed are sorted by technique into the slots they occupy
struct RelocBuffers {
    unsigned *placedKey;        // N: the key a slot's chain was placed under (all ones: never placed): the technique key, or with `fine` the 24-bit [technique | screen Morton] key
    int *tileCount, *tileHist;  // RelocTiles(N), 64 x RelocTiles(N): members per 1024-slot tile / per (tile, key); then their exclusive prefixes
    int *members;               // N: the slots that take part, ascending
    int *sorted;                // N: member indices by key
    int *count;                 // 2: number of members (0: the relocation was skipped, its movers exceeded `capacity`), relocations skipped so far
    float *staging;             // RelocRecordWords(maxDepth) floats per member
    int capacity;               // records the staging buffer holds: a step with more movers leaves them where they are (host/context.cpp)
    bool fine;                  // the per-step relocation and the full re-sort place by the fine key (LMC_RELOC_FINE=0: the technique alone, round 4's rule)
};
size_t RelocTiles(int N);
size_t RelocRecordWords(int maxDepth);
void LaunchRelocIota(int n, int *v, hipStream_t s);
// withoutGaussianOnly (H2MC renders): chains that hold a stored Gaussian stay where they are (the pipeline's Gaussian buffers are per slot)
void LaunchRelocate(const lmcd::ChainArrays &A, int maxDepth, const RelocBuffers &B, bool withoutGaussianOnly, hipStream_t s);
void LaunchRelocFineKey(const lmcd::ChainArrays &A, const int *leafPosOfTri, int numTris, int mode, unsigned long long *keys, const lmcd::TriData *tris, const lmcd::DMaterial *materials, hipStream_t s);
void LaunchRelocMove(const lmcd::ChainArrays &A, int maxDepth, const RelocBuffers &B, hipStream_t s, int keyMode = 0);
// the periodic full re-sort by (technique, screen Morton code): work buffers of the device radix sort
struct RelocSortBuffers {
    unsigned *keys[2];  // N each
    int *vals[2];       // N each
    int *hist;          // 256 x RelocSortBlocks(N)
    int *scanSums;      // 256 x RelocSortBlocks(N) / 2048 + 2
};
size_t RelocSortBlocks(int N);
void LaunchRelocFullSort(const lmcd::ChainArrays &A, int maxDepth, const RelocBuffers &B, const RelocSortBuffers &W, hipStream_t s);
void LaunchRelocateFine(const lmcd::ChainArrays &A, int maxDepth, const RelocBuffers &B, const RelocSortBuffers &W, bool withoutGaussianOnly, hipStream_t s);
// dilated grid of one cache dim on the device (DCacheDim::gridStart / gridRows); buffer sizes in kernels.hip
void LaunchBuildCacheGrid(const float *pts, int n, int dim, int G, int m, const int *coord, int *scratchStart, int *scratchCursor, int *scratchWordCount, int *tileSums,
                          uint2 *words, int *cellStart, unsigned short *idx, hipStream_t s);
