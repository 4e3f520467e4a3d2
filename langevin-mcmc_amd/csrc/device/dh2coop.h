// Shared layout of the wave-cooperative H2MC pipeline (h2hess.hip, h2gauss.hip, step_h2_phases.hip, host/context.cpp).
// An H2MC small step (H2MCSmallStep::Mutate, /root/reference/src/mutation_h2mc.h:38-128) is run as a short pipeline of launches
// instead of one thread per chain: the lane-per-chain parts (draws, path perturbation, accept / reject) stay lane-per-chain, the
// second-order path program runs with the lanes of a wave = the 2 x 2 Hessian blocks of ONE state (h2hess.hip k_h2_hess), the
// eigen-solve and the dense Gaussian with 16 lanes per state (k_h2_gauss).  Hand-off between the launches, per chain (index i):
//   rec  [i * H2_REC_WORDS ..]  the serialised state the path program reads (path.cpp:2497-2586 layout), AoS so that a wave stages
//                               it into LDS with whole-line loads: [0,17) primary | [17] c | [18] l | [H2_REC_VP ..) vertParams
//   hout [i * H2_OUT_WORDS ..]  what the program returns: [0,16) gradient | [16] logLum | [H2_OUT_HESS ..) Hessian rows (stride dim),
//                               the triangle Eigen reads (h2mc.cpp:78: the UPPER triangle of the rows as delivered)
//   gauss[buf][i * H2_GAUSS_AOS ..]  a state's proposal Gaussian (h2mc.cpp:3-142), AoS: [0,16) mean | [16] logDet | [17] kind |
//                               [32 ..) covL (n x n, stride n) | [32 + 256 ..) invCov
#pragma once
#include "dmath.h"

namespace lmcd {

constexpr int H2_REC_WORDS = 640, H2_REC_C = 17, H2_REC_L = 18, H2_REC_VP = 20;  // vertParams: V <= 238 + 59 * 6 = 592 words
constexpr int H2_OUT_WORDS = 288, H2_OUT_LOGLUM = 16, H2_OUT_HESS = 32;
constexpr int H2_GAUSS_AOS = 544, H2_GAUSS_LOGDET = 16, H2_GAUSS_COVL = 32, H2_GAUSS_INVCOV = 32 + 256;

// Exact technique index: the derivative programs exist for 1 <= c <= 9, 0 <= l <= 8, 3 <= c + l <= 9 (path.cpp:4030-4037): 42 of them
constexpr int H2_NTECH = 42;
LMC_HD int H2TechIndex(int c, int l) {  // -1: no program
    const int s = c + l;
    if (c < 1 || c > 9 || l < 0 || l > 8 || s < 3 || s > 9) return -1;
    return (s - 1) * s / 2 - 3 + l;
}
LMC_HD void H2TechOf(int t, int &c, int &l) {
    int s = 3, base = 0;
    while (base + s <= t) base += s, s++;
    l = t - base, c = s - l;
}
LMC_HD int H2TechDim(int t) {  // 2 * max(c + l - 1, 2)
    int s = 3, base = 0;
    while (base + s <= t) base += s, s++;
    return 2 * (s - 1);
}
// lanes one state occupies in the Hessian launch: the 2 x 2 blocks of the upper triangle of a dim x dim matrix
LMC_HD int H2BlocksOfDim(int dim) { return (dim / 2) * (dim / 2 + 1) / 2; }

// Work lists of one pipeline stage: a bin holds the chains whose state of ONE technique needs its Gaussian.  A technique has H2_NSIG bins, chosen
// by a signature of the state's materials (the BSDF type and the sampling mode of every surface vertex, hashed to three bits): the three to
// ten states that share a wave of the Hessian launch then take the same BSDF branches as well as the same path structure (lanes active per
// issued instruction 0.74-0.77 with one bin per technique: profiles/r04_final_h2mc_pmc_*.json).
// Layout (round 5; ADVICE r4: the bins used to be 336 arrays of N entries each, 2.8 GB at 2^20 chains): ONE array of N entries per stage.  The
// lane-per-chain launch that fills a stage records every chain's bin (binOf) and counts the bins; dpipe.h LaunchBinsCompact then turns the
// counts into offsets (start) and scatters the chain ids: bin b's entries are items[start[b] .. start[b] + count[b]).
constexpr int H2_NSIG = 8, H2_NBINS = H2_NTECH * H2_NSIG;
struct H2Bins {
    int *items;   // N: the chains of the stage, grouped by bin
    int *count;   // H2_NBINS (+ padding to a multiple of 64)
    int *start;   // H2_NBINS (+ padding): first entry of every bin
    int *cursor;  // H2_NBINS (+ padding): scatter cursors
    int *binOf;   // N: the bin of a chain that takes part in the stage, -1 otherwise (valid for the chains of the step's list)
};
constexpr int H2_COUNT_WORDS = (H2_NBINS + 63) / 64 * 64;
LMC_HD int H2BinIndex(int tech, unsigned sigHash) { return tech * H2_NSIG + (int)((sigHash ^ (sigHash >> 3) ^ (sigHash >> 6) ^ (sigHash >> 9) ^ (sigHash >> 12)) & (unsigned)(H2_NSIG - 1)); }

#if defined(__HIPCC__)
// The task table of a stage, built by every block for itself: a task = up to ipw(technique) consecutive items of one bin.  incl (LDS, H2_NBINS
// words) receives the inclusive prefix of the bins' task counts; returns the total.  One wave; lane k handles the bins 6 k .. 6 k + 5.
template <class IpwOfTech>
__device__ __forceinline__ int H2BuildTaskTable(const int *count, int *incl, IpwOfTech ipwOfTech) {
    constexpr int PER = (H2_NBINS + 63) / 64;
    const int lane = threadIdx.x & 63;
    int tasks[PER], sum = 0;
    for (int k = 0; k < PER; k++) {
        const int b = lane * PER + k;
        tasks[k] = 0;
        if (b < H2_NBINS) {
            const int ipw = ipwOfTech(b / H2_NSIG);
            tasks[k] = (count[b] + ipw - 1) / ipw;
        }
        sum += tasks[k];
    }
    int inclLane = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inclLane, off);
        if (lane >= off) inclLane += o;
    }
    int run = inclLane - sum;
    for (int k = 0; k < PER; k++) {
        const int b = lane * PER + k;
        run += tasks[k];
        if (b < H2_NBINS) incl[b] = run;
    }
    __syncthreads();
    return __shfl(inclLane, 63);
}
// the bin of task w: the first bin whose inclusive prefix exceeds w (all lanes read the same words: LDS broadcasts)
__device__ __forceinline__ int H2BinOfTask(const int *incl, int w) {
    int lo = 0, hi = H2_NBINS - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (incl[mid] > w) hi = mid;
        else
            lo = mid + 1;
    }
    return lo;
}
#endif

// per-step hand-off state of the H2MC pipeline (host/context.cpp allocates it for H2MC renders only)
enum : int { H2S_H2 = 1, H2S_DENSE = 2, H2S_OK = 4 };  // `step` bits: an H2MC (not uniform-mixing) step; dim <= 16; the re-trace carries light
enum : int { H2K_DENSE = 0, H2K_ISO_EARLYOUT = 1, H2K_ISO_NODERV = 2 };  // gauss[..][17]: which of the reference's three outcomes the record holds
struct H2Arrays {
    float *rec, *hout;    // AoS, H2_REC_WORDS / H2_OUT_WORDS per chain
    float *gauss;         // AoS, 2 buffers x N x H2_GAUSS_AOS (current / proposal, selected by F_GSEL)
    float *offset;        // MAXPSS x N (SoA): the step's normal draws z, replaced by the proposal offset
    float *py, *px;       // N: log density of the offset under the current Gaussian / of its reverse under the proposal's
    float *propContrib;   // 9 x N: the proposal's SubpathContrib
    int *step;            // N: H2S_* bits
    unsigned char *kind;  // N: 1 = k_h2_sample turns this chain's z into the offset (dense or isotropic record), 0 = k_h2_begin did
    H2Bins bins[2];       // stage 0: current states without a Gaussian; stage 1: proposals
};

// ---- the same pipeline shape for the gradient steps of the LMC cache-fill phase (step_mala_phases.hip, gradcoop.hip): MALASmallStep::Mutate
// (mutation_mala.h:38-290) cut at its two gradient evaluations; records, bins and task table as above, per chain a gradient instead of a Hessian
constexpr int MG_OUT_WORDS = 32, MG_OUT_LOGLUM = 16;  // gout [i * MG_OUT_WORDS ..]: [0,16) gradient | [16] logLum
enum : int { MS_MALA = 1, MS_OK = 2, MS_GRAD_CUR = 4, MS_GRAD_PROP = 8 };  // `step` bits: a MALA (not uniform-mixing) step; the re-trace carries light; a stage evaluated the state's gradient
struct MalaPipe {
    float *rec, *gout;   // AoS, H2_REC_WORDS / MG_OUT_WORDS per chain
    float *offset;       // MAXPSS x N (SoA): the step's normal draws z, replaced by the proposal offset
    float *py;           // N: log density of the offset under the current Gaussian
    float *propContrib;  // 9 x N: the proposal's SubpathContrib
    int *step;           // N: MS_* bits
    H2Bins bins[2];      // stage 0: current states without a Gaussian; stage 1: proposals
};

}  // namespace lmcd
