#include "shardplan.h"

#include <algorithm>
#include <cstring>

namespace lmc {

ShardLayout MakeShardLayout(int world, int rank, int V, long long numInitSamples) {
    const long long perThread = numInitSamples / V, extra = numInitSamples % V;
    ShardLayout L;
    L.t0 = (int)((long long)V * rank / world), L.t1 = (int)((long long)V * (rank + 1) / world);
    L.g0 = SampleBase(L.t0, perThread, extra), L.g1 = SampleBase(L.t1, perThread, extra);
    for (int r = 0; r < world; r++) {
        const long long a = SampleBase((long long)V * r / world, perThread, extra), b = SampleBase((long long)V * (r + 1) / world, perThread, extra);
        L.maxLocalSamples = std::max(L.maxLocalSamples, b - a);
    }
    return L;
}

void AssembleCounts(int world, int V, long long numInitSamples, const unsigned char *padded, long long maxLocalSamples, std::vector<unsigned long long> &hOff,
                    std::vector<unsigned long long> &rankFirst) {
    const long long perThread = numInitSamples / V, extra = numInitSamples % V;
    hOff.assign((size_t)numInitSamples + 1, 0);
    rankFirst.assign((size_t)world + 1, 0);
    unsigned long long total = 0;
    for (int r = 0; r < world; r++) {
        const long long a = SampleBase((long long)V * r / world, perThread, extra), b = SampleBase((long long)V * (r + 1) / world, perThread, extra);
        rankFirst[r] = total;
        const unsigned char *cnt = padded + (size_t)r * maxLocalSamples;
        for (long long g = a; g < b; g++) {
            hOff[g] = total;
            total += cnt[g - a];
        }
    }
    rankFirst[world] = total;
    hOff[numInitSamples] = total;
}

void AssembleBlocks(int world, const std::vector<unsigned long long> &rankFirst, const void *padded, unsigned long long maxLocalContribs, size_t elemSize, void *out) {
    for (int r = 0; r < world; r++) {
        const unsigned long long n = rankFirst[r + 1] - rankFirst[r];
        memcpy((char *)out + rankFirst[r] * elemSize, (const char *)padded + (size_t)r * maxLocalContribs * elemSize, n * elemSize);
    }
}

void SeedWalk(long long numInitSamples, int numChains, const std::vector<unsigned long long> &hOff, const unsigned char *cl, const float *ls, float (*uniform01)(void *),
              void *rng, std::vector<long long> &seedSample, std::vector<unsigned char> &seedCL, std::vector<float> &seedLs, float &normalization) {
    const unsigned long long total = hOff[numInitSamples];
    // sequential float arithmetic on the host like the reference
    float totalScore = 0.f;
    for (unsigned long long i = 0; i < total; i++) totalScore += ls[i];
    std::vector<float> cdf(total + 1);
    cdf[0] = 0.f;
    for (unsigned long long i = 0; i < total; i++) cdf[i + 1] = cdf[i] + ls[i];
    const float interval = cdf.back() / float(numChains);
    float pos = uniform01(rng) * (interval - 0.f) + 0.f;  // uniform_real_distribution<Float>(0, interval) on RNG(mStates.size()), mlt.h:114-116
    seedSample.resize(numChains), seedCL.resize(numChains), seedLs.resize(numChains);
    long long cdfPos = 0, g = 0;
    for (int i = 0; i < numChains; i++) {
        // mlt.h:118-120; the reference's clamp inside the loop never terminates once pos > cdf[size-1]: stop at size-1
        while (pos > cdf[cdfPos] && cdfPos < (long long)total - 1) cdfPos++;
        const long long m = std::max<long long>(cdfPos - 1, 0);
        while (g + 1 < numInitSamples && hOff[g + 1] <= (unsigned long long)m) g++;  // monotone: the sample that owns contribution m
        while (g > 0 && hOff[g] > (unsigned long long)m) g--;
        seedSample[i] = g;
        seedCL[i] = cl[m];
        seedLs[i] = ls[m];
        pos += interval;
    }
    normalization = totalScore * (1.0f / float(numInitSamples));
}

std::vector<int> OwnedRanges(int world, int V, long long numInitSamples, const std::vector<long long> &seedSample) {
    const long long perThread = numInitSamples / V, extra = numInitSamples % V;
    const int NT = (int)seedSample.size();
    std::vector<int> ownedBegin((size_t)world + 1, NT);
    int i = 0;
    for (int r = 0; r < world; r++) {
        const long long b = SampleBase((long long)V * (r + 1) / world, perThread, extra);
        ownedBegin[r] = i;
        while (i < NT && seedSample[i] < b) i++;
    }
    ownedBegin[world] = NT;
    return ownedBegin;
}

}  // namespace lmc
