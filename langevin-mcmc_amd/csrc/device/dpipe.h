// What the lane-per-chain phases of the two small-step pipelines share (step_h2_phases.hip: H2MC; step_mala_phases.hip: the
// gradient steps of the cache-fill phase): a state's record for the wave-cooperative derivative launches and its place in a
// stage's work lists (layout: dh2coop.h).
#pragma once
#include "dh2coop.h"
#include "step_kernel.h"

namespace lmcd {

// one chain's part in a stage (t = its bin): the bin recorded, the bin's count raised -- wave-aggregated (the chains of a wave are grouped by
// technique, so a few atomics per wave are the rule).  Every lane of the wave calls it (want = false: the chain takes no part).
LMC_D void H2Enqueue(const H2Bins &bins, int N, bool want, int t, int i) {
    unsigned long long todo = __ballot(want);
    const int lane = threadIdx.x & 63;
    if (i >= 0 && i < N) bins.binOf[i] = want ? t : -1;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int tl = __shfl(t, leader);
        const unsigned long long mask = __ballot(want && t == tl);
        if (lane == leader) atomicAdd(&bins.count[tl], __popcll(mask));
        todo &= ~mask;
    }
}

// what decides the BSDF branches the second-order program takes on this state: every surface vertex's material type and sampling mode
LMC_D unsigned H2MaterialSignature(const DScene &S, const DPath &path) {
#ifdef LMC_H2_NOSIG  // A/B build: one bin per technique
    return 0;
#endif
    unsigned h = 0;
    for (int d = 0; d < path.lgtCount; d++) h = h * 7u + (unsigned)MaterialOfTri(S, path.lgt[d].tri).type + (path.lgt[d].useAbs != 0.0f ? 3u : 0u) + 1u;
    for (int d = 0; d < path.camCount; d++)
        if (path.cam[d].tri >= 0) h = h * 7u + (unsigned)MaterialOfTri(S, path.cam[d].tri).type + (path.cam[d].useAbs != 0.0f ? 3u : 0u) + 1u;
    return h;
}

// Serialize(scene, path, ss) (path.cpp:2497-2586) into the chain's record
LMC_D void H2Serialize(const DScene &S, const DPath &path, float *rec) {
    float primary[2 * MAXD + 1];
    StridedOut o{rec + H2_REC_VP, 1, 0};
    SerializePath(S, path, primary, o);
    const int L = max(path.camDepth + path.lgtDepth - 1, 2);
    for (int k = 0; k < 2 * L + 1; k++) rec[k] = primary[k];
    rec[H2_REC_C] = __int_as_float(path.camDepth), rec[H2_REC_L] = __int_as_float(path.lgtDepth);
}

}  // namespace lmcd
