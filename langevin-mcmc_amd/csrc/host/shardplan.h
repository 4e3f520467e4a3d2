// Host-side planning of the sharded MLTInit (mlt.h:41-154 over the ranks of a job; context.cpp "MLTInit + chain set-up"):
// which init streams / samples a rank runs, how the padded blocks the ranks all-gather are put back into stream order, the
// sequential float sums and the equal-spaced CDF walk every rank repeats, and which chains each rank's samples seed.
// Pure C++ (no HIP): context.cpp calls these between its device phases, and the C-ABI test hooks lmc_shard_* expose them to the
// CPU tier (tests/test_dist_gloo.py drives them with world_size 2 over gloo).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace lmc {

// first global sample index of init stream t (streams t < extra run one sample more, mlt.h:60-69)
inline long long SampleBase(long long t, long long perThread, long long extra) { return t * perThread + (t < extra ? t : extra); }

struct ShardLayout {
    int t0 = 0, t1 = 0;                 // this rank's init streams [t0, t1)
    long long g0 = 0, g1 = 0;           // ... = the samples [g0, g1) in stream order
    long long maxLocalSamples = 0;      // the largest rank's sample count (block size of the count exchange)
};
ShardLayout MakeShardLayout(int world, int rank, int V, long long numInitSamples);

// counts[world][maxLocalSamples] (padded, as gathered) -> hOff[numInitSamples + 1] (first contribution of every sample, total at the
// end) and rankFirst[world + 1] (first contribution of every rank's block)
void AssembleCounts(int world, int V, long long numInitSamples, const unsigned char *padded, long long maxLocalSamples, std::vector<unsigned long long> &hOff,
                    std::vector<unsigned long long> &rankFirst);
// padded[world][maxLocalContribs] -> out[total], rank blocks concatenated = stream order
void AssembleBlocks(int world, const std::vector<unsigned long long> &rankFirst, const void *padded, unsigned long long maxLocalContribs, size_t elemSize, void *out);

// normalization = (sequential float sum of all lsScores) / numInitSamples (mlt.h:64,87,153); equal-spaced seeding (mlt.h:107-148):
// per chain the init sample that seeds it, the technique of the seeding contribution and its lsScore
void SeedWalk(long long numInitSamples, int numChains, const std::vector<unsigned long long> &hOff, const unsigned char *cl, const float *ls, float (*uniform01)(void *),
              void *rng, std::vector<long long> &seedSample, std::vector<unsigned char> &seedCL, std::vector<float> &seedLs, float &normalization);

// rank r's samples seed the chains [ownedBegin[r], ownedBegin[r + 1]) (seedSample is non-decreasing, sample ranges are contiguous)
std::vector<int> OwnedRanges(int world, int V, long long numInitSamples, const std::vector<long long> &seedSample);

}  // namespace lmc
