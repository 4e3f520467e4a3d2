// Chain relocation: the resident chains kept PHYSICALLY grouped by technique (c,l), so that the 64 consecutive chains of a wave
// retrace the same technique -- same number of path segments, same terminal strategy -- while their state accesses stay coalesced.
//
// Why: the lanes of the lean small-step kernel (dsmall.h) idle while the longest path of their wave finishes.  Grouping the WORK LIST by
// technique removes that (-24 % vector instructions on the Lambertian torus) but was measured slower there three times (rounds 2-4):
// the chain state is SoA [word][chain], and the lanes of a sorted wave touch one cache line each per state word instead of two lines per
// wave (L2 misses x 2.5, profiles/r03_a_ab_sorted_lists_instruction_counts.jsonl).  Moving the STATE instead costs little, because a
// chain's technique only changes when it accepts a large step (mlt.cpp:113-132: small steps perturb a path of fixed (c,l),
// path.cpp:1953-2160) or is reset to an init state (mlt.cpp:147-169) -- about a tenth of the chains per step -- and because what is
// alive of a chain right after either event is small: its new path, contribution and splats, its RNG, a dozen scalars (the MALA vectors
// are zero and no Gaussian is stored: dchain.h ClearBuffered).
//
// One relocation (after the step's launches, before the next step's work lists are built):
//   members = slots whose chain's technique key differs from the key the slot was placed under, in ascending slot order
//   the members' chains are sorted by key; the p-th chain of that order moves into the p-th member slot
// so every mover lands at the slot quantile of its key quantile.  In the stationary regime the chains that leave a technique and the
// chains that enter it balance, so the movers of a key land in the slots that movers of that key region vacated: the array stays sorted
// by key without any region bookkeeping.  The very first relocation finds every slot unplaced and is a full sort.
// The state travels slot -> staging record (AoS, one record per member) -> slot, two launches, because the moves form cycles.
// Nothing that a chain computes depends on the slot it lives in: the RNG stream moves with it, and the one use of the chain's id
// (the outlier reset, dchain.h ResetToInitState) reads it from A.chainId.  The film differs by the order of its atomics only.
#include "dstep.h"
#include "kernels.h"

namespace lmcd {
namespace {

enum : int { RW_FLAGS = 0, RW_SAMPLEIDX, RW_NUMSAMPLES, RW_ADJREJECT, RW_SPLATCOUNT, RW_CHAINID, RW_SCORESUM, RW_LASTSCORESUM, RW_LASTSCORE, RW_PATHWEIGHT, RW_NEXTKIND, RW_RNG_LO, RW_RNG_HI, RW_KEY, RW_RNG_TICKED, RW_SCALARS = 16 };
constexpr int RW_TAB = RW_SCALARS, RW_HEAD = RW_TAB + 64, RW_VERT = RW_HEAD + DPATH_HEAD_WORDS;

struct RecordLayout {
    int nV, nS;  // vertex records kept per sub-path, splats kept
    LMC_HD int Contrib() const { return RW_VERT + 2 * nV * DVERTEX_WORDS; }
    LMC_HD int Splats() const { return Contrib() + CONTRIB_WORDS; }
    LMC_HD int Vectors() const { return Splats() + nS * SPLAT_WORDS; }
    LMC_HD int Gauss() const { return Vectors() + 7 * MAXPSS; }
    LMC_HD int Words() const { return Gauss() + GAUSS_WORDS; }
};

// Order of the groups along the slots: LONGEST paths first (LMC_RELOC_ORDER=1, the default).  The work lists follow the slot order, a launch
// hands out its blocks in list order, and a wave of long paths runs several times as long as a wave of short ones: started last they are
// the tail of the launch, started first the short waves fill in behind them.
#ifndef LMC_RELOC_ORDER
#define LMC_RELOC_ORDER 1
#endif
// LMC_RELOC_TILES (A/B build): where the technique keys leave room in the 64 bins -- states without a light sub-path up to path length 6: eight
// techniques -- a technique's chains are sub-ordered by the screen tile (4 x 2) of their camera vertex, so that a wave's camera rays start out
// through the same part of the tree.  `tiles`: 8 or 0 (LaunchRelocate).
#ifndef LMC_RELOC_TILES
#define LMC_RELOC_TILES 0
#endif
LMC_D int SlotKey(const ChainArrays &A, int i, int tiles) {
    const int c = __float_as_int(A.curContrib[i]), l = __float_as_int(A.curContrib[(size_t)A.N + i]);
    int key = TechniqueKey(c, l);
#if LMC_RELOC_TILES
    if (tiles && l <= 1 && c + l - 1 <= 6) {
        const float *path = CurPathBuf(A, A.flags[i]);
        const float sx = path[(size_t)1 * A.N + i], sy = path[(size_t)2 * A.N + i];  // DPath::screen0, screen1
        const int tx = min(3, max(0, (int)(sx * 4.f))), ty = min(1, max(0, (int)(sy * 2.f)));
        key = ((max(c + l - 1, 3) - 3) * 2 + l) * 8 + ty * 4 + tx;
    }
#endif
    return LMC_RELOC_ORDER ? 63 - key : key;
}
LMC_D unsigned Part1By1(unsigned x) {
    x &= 0x0000ffffu;
    x = (x ^ (x << 8)) & 0x00ff00ffu;
    x = (x ^ (x << 4)) & 0x0f0f0f0fu;
    x = (x ^ (x << 2)) & 0x33333333u;
    x = (x ^ (x << 1)) & 0x55555555u;
    return x;
}
// the fine key of a slot's chain: [63 - technique (6 bits) | Morton code of the camera vertex's screen position (9 + 9 bits)] -- the order of the full
// re-sort and, since round 6, of the per-step relocation too (key mode -1 of the move kernels)
LMC_D unsigned FineKey(const ChainArrays &A, int i) {
    const size_t N = A.N;
    const float *path = CurPathBuf(A, A.flags[i]);
    const float sx = path[(size_t)1 * N + i], sy = path[(size_t)2 * N + i];  // DPath::screen0, screen1
    const unsigned mx = (unsigned)min(511, max(0, (int)(sx * 512.f))), my = (unsigned)min(511, max(0, (int)(sy * 512.f)));
    return ((unsigned)SlotKey(A, i, 0) << 18) | Part1By1(mx) | (Part1By1(my) << 1);
}
LMC_D int SlotKeyMode(const ChainArrays &A, int i, int mode) { return mode < 0 ? (int)FineKey(A, i) : SlotKey(A, i, mode); }
LMC_D bool VectorsMayBeNonZero(int flags) { return (flags & F_BUFFERED) && (flags & F_VDIRTY); }  // dchain.h: the invariant of the seven MALA vectors
LMC_D bool HasStoredGaussian(int flags) { return (flags & F_GAUSS) && !(flags & F_GAUSS_ISO); }

// ---- who moves, and where.  A tile = 1024 consecutive slots (one block, four slots per thread).
// Members: the slots whose chain ran a LARGE step in the step just launched (A.stepKind, written by k_build_lists) and now has another
// technique key than the one the slot was placed under.  Only those: the relocation runs on the large-step launch's stream right behind
// it, BESIDE the small-step launches, whose chains it must not touch (a small step changes (c,l) only through the outlier reset; such a
// chain is picked up at its next large step).
constexpr int RELOC_TILE = 1024;
// H2MC renders (placedKey's top bit of the launch argument `mode`): a chain that holds a stored Gaussian stays -- the dense Gaussian lives in the
// pipeline's own per-slot buffers (dh2coop.h H2Arrays::gauss), which are not moved; an accepted large step, the event that changes the
// technique, has just dropped it (dstep.h), so only chains kept by a REJECTED large step wait for their next one.
LMC_D int MemberKey(const ChainArrays &A, const unsigned *placedKey, int i, bool withoutGaussianOnly, int tiles) {  // -1: not a member
    if (i >= A.N || A.stepKind[i] != NEXT_LARGE) return -1;
    const int key = SlotKey(A, i, tiles);
    if ((unsigned)key == placedKey[i]) return -1;
    if (withoutGaussianOnly && (A.flags[i] & F_GAUSS)) return -1;
    return key;
}
// All three launches are ONE-WAVE blocks: they run beside the small-step launches, whose waves hold every SIMD's registers -- a one-wave block
// takes the first slot that frees up, a four-wave block waits for four at once (k_reloc_count as 256-thread blocks: 1.0 ms in the queue,
// profiles/r04_reloc_b_*; the same lesson as kernels.hip k_push_count).  Lane l of a tile's wave looks at slots base + 64 j + l, j = 0 .. 15.
// tileCount[t] = members of tile t; tileHist[t][k] = ... with key k
__global__ void __launch_bounds__(64) k_reloc_count(ChainArrays A, const unsigned *placedKey, int *tileCount, int *tileHist, bool noGauss, int tiles) {
    __shared__ int h[64];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RELOC_TILE + threadIdx.x;
    int total = 0;
    for (int j = 0; j < RELOC_TILE / 64; j++) {
        const int key = MemberKey(A, placedKey, base + 64 * j, noGauss, tiles);
        if (key >= 0) atomicAdd(&h[key], 1), total++;
    }
    __syncthreads();
    tileHist[blockIdx.x * 64 + threadIdx.x] = h[threadIdx.x];
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off);
    if (threadIdx.x == 0) tileCount[blockIdx.x] = total;
}
// one wave: tileCount -> first member index of every tile; tileHist -> first sorted position of every (tile, key) group; *count
// count[0] = members of this relocation -- 0 when they exceed the staging capacity: the move is skipped as a whole (nothing depends on a slot; the
// chains are picked up at their next large step) and count[1], the number of relocations skipped so far, goes up (lmc_relocation_skipped)
__global__ void __launch_bounds__(64) k_reloc_offsets(int nTiles, int *tileCount, int *tileHist, int *count, int capacity) {
    const int key = threadIdx.x;
    int total = 0;
#pragma unroll 8
    for (int t = 0; t < nTiles; t++) total += tileHist[t * 64 + key];
    int incl = total;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (key >= off) incl += o;
    }
    if (key == 63) {
        if (incl > capacity) count[0] = 0, count[1]++;
        else
            count[0] = incl;
    }
    int run = incl - total;
#pragma unroll 8
    for (int t = 0; t < nTiles; t++) {
        const int c = tileHist[t * 64 + key];
        tileHist[t * 64 + key] = run;
        run += c;
    }
    int carry = 0;
    for (int b = 0; b < nTiles; b += 64) {
        const int t = b + threadIdx.x, v = t < nTiles ? tileCount[t] : 0;
        int in = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(in, off);
            if (key >= off) in += o;
        }
        if (t < nTiles) tileCount[t] = carry + in - v;
        carry += __shfl(in, 63);
    }
}
// members[m] = slot (ascending); sorted[p] = m for the p-th chain by key (inside a (tile, key) group the order is the LDS atomics')
__global__ void __launch_bounds__(64) k_reloc_assign(ChainArrays A, const unsigned *placedKey, const int *tileStart, const int *groupStart, int *members, int *sorted, bool noGauss, int tiles) {
    __shared__ int cursor[64];
    cursor[threadIdx.x] = groupStart[blockIdx.x * 64 + threadIdx.x];
    __syncthreads();
    const int base = blockIdx.x * RELOC_TILE + threadIdx.x;
    int m0 = tileStart[blockIdx.x];
    for (int j = 0; j < RELOC_TILE / 64; j++) {
        const int key = MemberKey(A, placedKey, base + 64 * j, noGauss, tiles);
        const unsigned long long mask = __ballot(key >= 0);
        const unsigned long long below = (1ull << threadIdx.x) - 1ull;
        if (key >= 0) members[m0 + __popcll(mask & below)] = base + 64 * j;
        // a member's position inside its (tile, key) group = its rank among the members of that key, in slot order: the chain-to-slot assignment
        // is the same from run to run (it used to follow the arrival order of an LDS atomic)
        unsigned long long todo = mask;
        while (todo) {
            const int k = __shfl(key, __ffsll((long long)todo) - 1);
            const unsigned long long same = __ballot(key == k);
            if (key == k) sorted[cursor[k] + __popcll(same & below)] = m0 + __popcll(mask & below);
            __syncthreads();
            if (threadIdx.x == 0) cursor[k] += __popcll(same);
            __syncthreads();
            todo &= ~same;
        }
        m0 += __popcll(mask);
    }
}

LMC_D float *VectorBase(const ChainArrays &A, int v) {
    float *const b[7] = {A.chV1, A.chV2, A.chCurrNewV2, A.chPropNewV1, A.chPropNewV2, A.chPss, A.chLastPss};
    return b[v];
}

// member m's chain -> staging record m
__global__ void __launch_bounds__(64) k_reloc_gather(ChainArrays A, RecordLayout R, const int *members, const int *sorted, const int *count, float *staging, int capacity, int tiles) {
    const int M = *count;
    if (M > capacity) return;  // (k_reloc_offsets reports 0 members for a relocation that would not fit; the test stays as the guard of the staging buffer)
    const size_t N = A.N;
    for (int m = blockIdx.x * 64 + threadIdx.x; m < M; m += gridDim.x * 64) {
        const int i = members[m];
        float *r = staging + (size_t)m * R.Words();
        const int flags = A.flags[i];
        const uint64_t rs = A.rngState[i];
        const int nSplat = min(A.curSplatCount[i], R.nS);
        r[RW_FLAGS] = __int_as_float(flags), r[RW_SAMPLEIDX] = __int_as_float(A.sampleIdx[i]), r[RW_NUMSAMPLES] = __int_as_float(A.numSamples[i]);
        r[RW_ADJREJECT] = __int_as_float(A.adjacentReject[i]), r[RW_SPLATCOUNT] = __int_as_float(nSplat), r[RW_CHAINID] = __int_as_float(A.chainId[i]);
        r[RW_SCORESUM] = A.scoreSum[i], r[RW_LASTSCORESUM] = A.lastScoreSum[i], r[RW_LASTSCORE] = A.lastScore[i], r[RW_PATHWEIGHT] = A.pathWeight[i];
        r[RW_NEXTKIND] = __int_as_float((int)A.nextKind[i]), r[RW_RNG_LO] = __int_as_float((int)(uint32_t)rs), r[RW_RNG_HI] = __int_as_float((int)(uint32_t)(rs >> 32));
        r[RW_KEY] = __int_as_float(SlotKeyMode(A, i, tiles));
        const int ticked = A.rngTicked[i];  // the extension table travels only once the stream has ticked: until then it is a function of the chain's seed (drng.h)
        r[RW_RNG_TICKED] = __int_as_float(ticked);
        if (ticked) {
            const uint4 *tab = reinterpret_cast<const uint4 *>(A.rngTab + (size_t)i * 64);
            uint4 *rt = reinterpret_cast<uint4 *>(r + RW_TAB);
#pragma unroll 4
            for (int k = 0; k < 16; k++) rt[k] = tab[k];
        }
        const float *path = CurPathBuf(A, flags);
#pragma unroll 4
        for (int k = 0; k < DPATH_HEAD_WORDS; k++) r[RW_HEAD + k] = path[(size_t)k * N + i];
        const int camCount = min(max(__float_as_int(r[RW_HEAD + 12]), 0), R.nV), lgtCount = min(max(__float_as_int(r[RW_HEAD + 13]), 0), R.nV);  // DPath: camCount, lgtCount
        for (int v = 0; v < camCount; v++)
#pragma unroll
            for (int k = 0; k < DVERTEX_WORDS; k++) r[RW_VERT + v * DVERTEX_WORDS + k] = path[(size_t)(DPATH_HEAD_WORDS + v * DVERTEX_WORDS + k) * N + i];
        for (int v = 0; v < lgtCount; v++)
#pragma unroll
            for (int k = 0; k < DVERTEX_WORDS; k++) r[RW_VERT + (R.nV + v) * DVERTEX_WORDS + k] = path[(size_t)(DPATH_HEAD_WORDS + (MAXD + v) * DVERTEX_WORDS + k) * N + i];
#pragma unroll
        for (int k = 0; k < CONTRIB_WORDS; k++) r[R.Contrib() + k] = A.curContrib[(size_t)k * N + i];
        for (int k = 0; k < nSplat * SPLAT_WORDS; k++) r[R.Splats() + k] = A.curSplat[(size_t)k * N + i];
        if (VectorsMayBeNonZero(flags))
            for (int v = 0; v < 7; v++) {
                const float *src = VectorBase(A, v);
#pragma unroll 4
                for (int k = 0; k < MAXPSS; k++) r[R.Vectors() + v * MAXPSS + k] = src[(size_t)k * N + i];
            }
        if (HasStoredGaussian(flags)) {
            const float *G = CurGaussBuf(A, flags);
#pragma unroll 4
            for (int k = 0; k < GAUSS_WORDS; k++) r[R.Gauss() + k] = G[(size_t)k * N + i];
        }
    }
}

// staging record sorted[d] -> member slot d.  Record d still holds what the slot contained: its flags say whether the slot's MALA
// vectors have to be zeroed for an incoming chain whose vectors are zero by the invariant.
__global__ void __launch_bounds__(64) k_reloc_scatter(ChainArrays A, RecordLayout R, const int *members, const int *sorted, const int *count, const float *staging,
                                                       unsigned *placedKey, int capacity) {
    const int M = *count;
    if (M > capacity) return;
    const size_t N = A.N;
    for (int d = blockIdx.x * 64 + threadIdx.x; d < M; d += gridDim.x * 64) {
        const int i = members[d], m = sorted[d];
        const float *r = staging + (size_t)m * R.Words();
        placedKey[i] = (unsigned)__float_as_int(r[RW_KEY]);
        if (m == d) continue;  // the chain stays where it is
        const int flags = __float_as_int(r[RW_FLAGS]), oldFlags = __float_as_int(staging[(size_t)d * R.Words() + RW_FLAGS]);
        const int nSplat = __float_as_int(r[RW_SPLATCOUNT]);
        A.flags[i] = flags, A.sampleIdx[i] = __float_as_int(r[RW_SAMPLEIDX]), A.numSamples[i] = __float_as_int(r[RW_NUMSAMPLES]);
        A.adjacentReject[i] = __float_as_int(r[RW_ADJREJECT]), A.curSplatCount[i] = nSplat, A.chainId[i] = __float_as_int(r[RW_CHAINID]), A.slotOf[__float_as_int(r[RW_CHAINID])] = i;
        A.scoreSum[i] = r[RW_SCORESUM], A.lastScoreSum[i] = r[RW_LASTSCORESUM], A.lastScore[i] = r[RW_LASTSCORE], A.pathWeight[i] = r[RW_PATHWEIGHT];
        A.nextKind[i] = (unsigned char)__float_as_int(r[RW_NEXTKIND]);
        A.rngState[i] = (uint64_t)(uint32_t)__float_as_int(r[RW_RNG_LO]) | ((uint64_t)(uint32_t)__float_as_int(r[RW_RNG_HI]) << 32);
        const int ticked = __float_as_int(r[RW_RNG_TICKED]);
        A.rngTicked[i] = (unsigned char)ticked;
        if (ticked) {
            uint4 *tab = reinterpret_cast<uint4 *>(A.rngTab + (size_t)i * 64);
            const uint4 *rt = reinterpret_cast<const uint4 *>(r + RW_TAB);
#pragma unroll 4
            for (int k = 0; k < 16; k++) tab[k] = rt[k];
        }
        float *path = CurPathBuf(A, flags);
#pragma unroll 4
        for (int k = 0; k < DPATH_HEAD_WORDS; k++) path[(size_t)k * N + i] = r[RW_HEAD + k];
        const int camCount = min(max(__float_as_int(r[RW_HEAD + 12]), 0), R.nV), lgtCount = min(max(__float_as_int(r[RW_HEAD + 13]), 0), R.nV);
        for (int v = 0; v < camCount; v++)
#pragma unroll
            for (int k = 0; k < DVERTEX_WORDS; k++) path[(size_t)(DPATH_HEAD_WORDS + v * DVERTEX_WORDS + k) * N + i] = r[RW_VERT + v * DVERTEX_WORDS + k];
        for (int v = 0; v < lgtCount; v++)
#pragma unroll
            for (int k = 0; k < DVERTEX_WORDS; k++) path[(size_t)(DPATH_HEAD_WORDS + (MAXD + v) * DVERTEX_WORDS + k) * N + i] = r[RW_VERT + (R.nV + v) * DVERTEX_WORDS + k];
#pragma unroll
        for (int k = 0; k < CONTRIB_WORDS; k++) A.curContrib[(size_t)k * N + i] = r[R.Contrib() + k];
        for (int k = 0; k < nSplat * SPLAT_WORDS; k++) A.curSplat[(size_t)k * N + i] = r[R.Splats() + k];
        if (VectorsMayBeNonZero(flags)) {
            for (int v = 0; v < 7; v++) {
                float *dst = VectorBase(A, v);
#pragma unroll 4
                for (int k = 0; k < MAXPSS; k++) dst[(size_t)k * N + i] = r[R.Vectors() + v * MAXPSS + k];
            }
        } else if (VectorsMayBeNonZero(oldFlags)) {
            for (int v = 0; v < 7; v++) {
                float *dst = VectorBase(A, v);
#pragma unroll 4
                for (int k = 0; k < MAXPSS; k++) dst[(size_t)k * N + i] = 0.f;
            }
        }
        if (HasStoredGaussian(flags)) {
            float *G = CurGaussBuf(A, flags);
#pragma unroll 4
            for (int k = 0; k < GAUSS_WORDS; k++) G[(size_t)k * N + i] = r[R.Gauss() + k];
        }
    }
}

// ---- the same move, WAVE-COOPERATIVE (round 6).  The lane-per-record kernels above write (read) a staging record word by word from one lane: 64 lanes,
// 64 records 2.5 KB apart -- every store instruction touches 64 cache lines for 4 bytes each.  Harmless for the ~16 k movers of a step beside the
// step launches, not for the full re-sort, whose 2^20 records they moved in 3.3 + 1.7 ms (profiles/r06_d_*).  Here a wave owns 64 consecutive
// members: it reads 64 SoA rows of its chains (lane = chain: coalesced when the members are consecutive slots, as in the full re-sort), turns the
// 64 x 64 tile in LDS, and writes it to the 64 staging records with lane = word (256 contiguous bytes per record and instruction); the scatter is
// the mirror image.  Dead words (vertices beyond a chain's counts, splats beyond its count, vectors known to be zero, an isotropic Gaussian) are
// neither read nor written: per lane by predicate, per wave by the loop bounds (the wave's maxima).  The 16 scalar words of a record stay lane-wise.
constexpr int XP = 65;  // LDS pitch of a tile row: lane l, column c at [l * XP + c], conflict-free both ways
LMC_D int WaveMax(int v) {
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}
// One-wave blocks: the LDS operations of a wave execute in order, so a tile written by all lanes can be read back transposed without a barrier --
// WaveSync is a compiler-level fence only.  (__syncthreads() also drains the wave's outstanding global stores at every tile: with it, and with the
// loads of a tile issued eight at a time, a re-sort's two launches took 1.3 + 1.3 ms at 2^20 chains against 0.2 ms of traffic, profiles/r06_j_*.)
LMC_D void WaveSync() { __builtin_amdgcn_wave_barrier(); }
// SoA rows [0, waveRows) of the wave's chains -> record words [recBase, recBase + waveRows) of records m0 .. m0 + nValid - 1.  rowAddr(k): this lane's word of row k.
// All 64 loads of a tile are in flight together (fixed trip count, predicated).
template <class RowAddr>
LMC_D void RowsToRecords(float *tile, RowAddr rowAddr, int laneRows, int waveRows, float *staging, size_t recWords, int m0, int nValid, int recBase) {
    const int lane = threadIdx.x;
    const unsigned long long live = __ballot(laneRows > 0);
    for (int r0 = 0; r0 < waveRows; r0 += 64) {
        const int nr = min(64, waveRows - r0);
        float v[64];
#pragma unroll
        for (int w = 0; w < 64; w++) v[w] = (r0 + w < laneRows) ? *rowAddr(r0 + w) : 0.f;
#pragma unroll
        for (int w = 0; w < 64; w++) tile[w * XP + lane] = v[w];
        WaveSync();
        if (lane < nr) {
            float *dst = staging + (size_t)m0 * recWords + recBase + r0 + lane;
#pragma unroll 16
            for (int c = 0; c < nValid; c++)
                if ((live >> c) & 1ull) dst[(size_t)c * recWords] = tile[lane * XP + c];  // a record whose chain has none of these rows is not written (nor read by the scatter)
        }
        WaveSync();
    }
}
// the mirror image: srcRec[c] = the record that goes to the wave's chain c (-1: none)
template <class RowAddr>
LMC_D void RecordsToRows(float *tile, const int *srcRec, RowAddr rowAddr, int laneRows, int waveRows, const float *staging, size_t recWords, int nValid, int recBase) {
    const int lane = threadIdx.x;
    for (int r0 = 0; r0 < waveRows; r0 += 64) {
        const int nr = min(64, waveRows - r0);
        if (lane < nr) {
            float v[64];
#pragma unroll
            for (int c = 0; c < 64; c++) {
                const int rec = c < nValid ? srcRec[c] : -1;
                v[c] = rec >= 0 ? staging[(size_t)rec * recWords + recBase + r0 + lane] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < 64; c++)
                if (c < nValid && srcRec[c] >= 0) tile[lane * XP + c] = v[c];
        }
        WaveSync();
#pragma unroll 16
        for (int w = 0; w < nr; w++)
            if (r0 + w < laneRows) *rowAddr(r0 + w) = tile[w * XP + lane];
        WaveSync();
    }
}

__global__ void __launch_bounds__(64) k_reloc_gather_coop(ChainArrays A, RecordLayout R, const int *members, const int *count, float *staging, int capacity, int tiles) {
    __shared__ float tile[64 * XP];
    const int M = *count;
    if (M > capacity) return;
    const size_t N = A.N, W = (size_t)R.Words();
    const int lane = threadIdx.x;
    for (int m0 = blockIdx.x * 64; m0 < M; m0 += gridDim.x * 64) {
        const int m = m0 + lane, nValid = min(64, M - m0);
        const bool valid = m < M;
        const int i = valid ? members[m] : 0;
        int flags = 0, camCount = 0, lgtCount = 0, nSplat = 0;
        const float *path = A.curPath;
        if (valid) {
            float *r = staging + (size_t)m * W;
            flags = A.flags[i];
            const uint64_t rs = A.rngState[i];
            nSplat = min(A.curSplatCount[i], R.nS);
            r[RW_FLAGS] = __int_as_float(flags), r[RW_SAMPLEIDX] = __int_as_float(A.sampleIdx[i]), r[RW_NUMSAMPLES] = __int_as_float(A.numSamples[i]);
            r[RW_ADJREJECT] = __int_as_float(A.adjacentReject[i]), r[RW_SPLATCOUNT] = __int_as_float(nSplat), r[RW_CHAINID] = __int_as_float(A.chainId[i]);
            r[RW_SCORESUM] = A.scoreSum[i], r[RW_LASTSCORESUM] = A.lastScoreSum[i], r[RW_LASTSCORE] = A.lastScore[i], r[RW_PATHWEIGHT] = A.pathWeight[i];
            r[RW_NEXTKIND] = __int_as_float((int)A.nextKind[i]), r[RW_RNG_LO] = __int_as_float((int)(uint32_t)rs), r[RW_RNG_HI] = __int_as_float((int)(uint32_t)(rs >> 32));
            r[RW_KEY] = __int_as_float(SlotKeyMode(A, i, tiles));
            const int ticked = A.rngTicked[i];
            r[RW_RNG_TICKED] = __int_as_float(ticked);
            if (ticked) {  // once per 2^32 draws of a stream: lane-wise
                const uint4 *tab = reinterpret_cast<const uint4 *>(A.rngTab + (size_t)i * 64);
                uint4 *rt = reinterpret_cast<uint4 *>(r + RW_TAB);
#pragma unroll 4
                for (int k = 0; k < 16; k++) rt[k] = tab[k];
            }
            path = CurPathBuf(A, flags);
            camCount = min(max(__float_as_int(path[(size_t)12 * N + i]), 0), R.nV), lgtCount = min(max(__float_as_int(path[(size_t)13 * N + i]), 0), R.nV);  // DPath: camCount, lgtCount
        }
        const float *pi = path + i;
        // head + camera vertices are consecutive rows of the path, the light vertices start at row HEAD + MAXD * 12
        const int camRows = valid ? DPATH_HEAD_WORDS + camCount * DVERTEX_WORDS : 0, lgtRows = lgtCount * DVERTEX_WORDS;
        RowsToRecords(tile, [&](int k) { return pi + (size_t)k * N; }, camRows, WaveMax(camRows), staging, W, m0, nValid, RW_HEAD);
        const int wl = WaveMax(lgtRows);
        if (wl) RowsToRecords(tile, [&](int k) { return pi + (size_t)(DPATH_HEAD_WORDS + MAXD * DVERTEX_WORDS + k) * N; }, lgtRows, wl, staging, W, m0, nValid, RW_VERT + R.nV * DVERTEX_WORDS);
        // contribution + pending splats: two row ranges written back to back (Splats() = Contrib() + CONTRIB_WORDS)
        const float *ci = A.curContrib + i, *si = A.curSplat + i;
        const int csRows = valid ? CONTRIB_WORDS + nSplat * SPLAT_WORDS : 0;
        RowsToRecords(tile, [&](int k) { return k < CONTRIB_WORDS ? ci + (size_t)k * N : si + (size_t)(k - CONTRIB_WORDS) * N; }, csRows, WaveMax(csRows), staging, W, m0, nValid, R.Contrib());
        const int vecRows = valid && VectorsMayBeNonZero(flags) ? 7 * MAXPSS : 0, wv = WaveMax(vecRows);
        if (wv) RowsToRecords(tile, [&](int k) { return VectorBase(A, k / MAXPSS) + (size_t)(k % MAXPSS) * N + i; }, vecRows, wv, staging, W, m0, nValid, R.Vectors());
        const int gRows = valid && HasStoredGaussian(flags) ? GAUSS_WORDS : 0, wg = WaveMax(gRows);
        if (wg) {
            const float *gi = CurGaussBuf(A, flags) + i;
            RowsToRecords(tile, [&](int k) { return gi + (size_t)k * N; }, gRows, wg, staging, W, m0, nValid, R.Gauss());
        }
    }
}

__global__ void __launch_bounds__(64) k_reloc_scatter_coop(ChainArrays A, RecordLayout R, const int *members, const int *sorted, const int *count, const float *staging,
                                                            unsigned *placedKey, int capacity) {
    __shared__ float tile[64 * XP];
    __shared__ int srcRec[64];
    const int M = *count;
    if (M > capacity) return;
    const size_t N = A.N, W = (size_t)R.Words();
    const int lane = threadIdx.x;
    for (int d0 = blockIdx.x * 64; d0 < M; d0 += gridDim.x * 64) {
        const int d = d0 + lane, nValid = min(64, M - d0);
        const bool valid = d < M;
        const int i = valid ? members[d] : 0, m = valid ? sorted[d] : -1;
        const bool moves = valid && m != d;  // m == d: the chain stays where it is
        const float *r = staging + (size_t)(valid ? m : 0) * W;
        int flags = 0, oldFlags = 0, camCount = 0, lgtCount = 0, nSplat = 0;
        if (valid) placedKey[i] = (unsigned)__float_as_int(r[RW_KEY]);
        if (moves) {
            flags = __float_as_int(r[RW_FLAGS]), oldFlags = __float_as_int(staging[(size_t)d * W + RW_FLAGS]);
            nSplat = __float_as_int(r[RW_SPLATCOUNT]);
            A.flags[i] = flags, A.sampleIdx[i] = __float_as_int(r[RW_SAMPLEIDX]), A.numSamples[i] = __float_as_int(r[RW_NUMSAMPLES]);
            A.adjacentReject[i] = __float_as_int(r[RW_ADJREJECT]), A.curSplatCount[i] = nSplat, A.chainId[i] = __float_as_int(r[RW_CHAINID]), A.slotOf[__float_as_int(r[RW_CHAINID])] = i;
            A.scoreSum[i] = r[RW_SCORESUM], A.lastScoreSum[i] = r[RW_LASTSCORESUM], A.lastScore[i] = r[RW_LASTSCORE], A.pathWeight[i] = r[RW_PATHWEIGHT];
            A.nextKind[i] = (unsigned char)__float_as_int(r[RW_NEXTKIND]);
            A.rngState[i] = (uint64_t)(uint32_t)__float_as_int(r[RW_RNG_LO]) | ((uint64_t)(uint32_t)__float_as_int(r[RW_RNG_HI]) << 32);
            const int ticked = __float_as_int(r[RW_RNG_TICKED]);
            A.rngTicked[i] = (unsigned char)ticked;
            if (ticked) {
                uint4 *tab = reinterpret_cast<uint4 *>(A.rngTab + (size_t)i * 64);
                const uint4 *rt = reinterpret_cast<const uint4 *>(r + RW_TAB);
#pragma unroll 4
                for (int k = 0; k < 16; k++) tab[k] = rt[k];
            }
            camCount = min(max(__float_as_int(r[RW_HEAD + 12]), 0), R.nV), lgtCount = min(max(__float_as_int(r[RW_HEAD + 13]), 0), R.nV);
        }
        WaveSync();  // (the previous group's tile and srcRec are no longer read)
        srcRec[lane] = moves ? m : -1;
        WaveSync();
        float *pi = CurPathBuf(A, flags) + i;
        const int camRows = moves ? DPATH_HEAD_WORDS + camCount * DVERTEX_WORDS : 0, lgtRows = lgtCount * DVERTEX_WORDS;
        RecordsToRows(tile, srcRec, [&](int k) { return pi + (size_t)k * N; }, camRows, WaveMax(camRows), staging, W, nValid, RW_HEAD);
        const int wl = WaveMax(lgtRows);
        if (wl) RecordsToRows(tile, srcRec, [&](int k) { return pi + (size_t)(DPATH_HEAD_WORDS + MAXD * DVERTEX_WORDS + k) * N; }, lgtRows, wl, staging, W, nValid, RW_VERT + R.nV * DVERTEX_WORDS);
        float *ci = A.curContrib + i, *si = A.curSplat + i;
        const int csRows = moves ? CONTRIB_WORDS + nSplat * SPLAT_WORDS : 0;
        RecordsToRows(tile, srcRec, [&](int k) { return k < CONTRIB_WORDS ? ci + (size_t)k * N : si + (size_t)(k - CONTRIB_WORDS) * N; }, csRows, WaveMax(csRows), staging, W, nValid, R.Contrib());
        // the seven MALA vectors: the incoming chain's, or zeros over a slot whose previous chain had any (the record of a chain whose vectors are zero by
        // the invariant holds nothing there: such a lane has no source record and stores the zeros the tile is filled with)
        const bool vecIn = moves && VectorsMayBeNonZero(flags);
        const int vecRows = (vecIn || (moves && VectorsMayBeNonZero(oldFlags))) ? 7 * MAXPSS : 0, wv = WaveMax(vecRows);
        if (wv) {
            // a record whose vector words were not written by the gather (its chain's vectors are zero and no chain of its gather wave had any): write zeros
            WaveSync();
            const int keep = srcRec[lane];
            WaveSync();
            if (!vecIn) srcRec[lane] = -1;
            WaveSync();
            for (int w = 0; w < 64; w++) tile[w * XP + lane] = 0.f;  // a lane without a source record stores zeros
            WaveSync();
            RecordsToRows(tile, srcRec, [&](int k) { return VectorBase(A, k / MAXPSS) + (size_t)(k % MAXPSS) * N + i; }, vecRows, wv, staging, W, nValid, R.Vectors());
            srcRec[lane] = keep;
            WaveSync();
        }
        const int gRows = moves && HasStoredGaussian(flags) ? GAUSS_WORDS : 0, wg = WaveMax(gRows);
        if (wg) {
            float *gi = CurGaussBuf(A, flags) + i;
            RecordsToRows(tile, srcRec, [&](int k) { return gi + (size_t)k * N; }, gRows, wg, staging, W, nValid, R.Gauss());
        }
    }
}

// ---- experiment (LMC_EXP / lmc_set_option "exp_resort"): a FULL re-sort of the resident chains by a key finer than the technique, the order worked out
// on the host -- the upper bound of what any finer slot order can buy the step launches (profiles/r06_a_*).  key = [63 - technique | sub-key]:
//   mode 1: Morton code of the camera vertex's screen position (12 + 12 bits)
//   mode 2: leaf-order position of the first walked vertex's triangle (the tree's depth-first order is a space-filling order), then the second's
//   mode 3: first triangle's leaf position, then the screen Morton code
__global__ void __launch_bounds__(256) k_reloc_finekey(ChainArrays A, const int *leafPosOfTri, int numTris, int mode, unsigned long long *keys, const TriData *tris, const DMaterial *materials) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.N) return;
    const size_t N = A.N;
    const int c = __float_as_int(A.curContrib[i]), l = __float_as_int(A.curContrib[N + i]);
    const float *path = CurPathBuf(A, A.flags[i]);
    const float sx = path[(size_t)1 * N + i], sy = path[(size_t)2 * N + i];
    const unsigned mx = (unsigned)min(4095, max(0, (int)(sx * 4096.f))), my = (unsigned)min(4095, max(0, (int)(sy * 4096.f)));
    const unsigned morton = Part1By1(mx) | (Part1By1(my) << 1);
    const int camCount = max(c - 1, 0), lgtCount = max(l - 1, 0);
    auto posOf = [&](int walkIdx) -> unsigned {  // the walk's vertices: light sub-path first (dsmall.h)
        int word = -1;
        if (walkIdx < lgtCount) word = DPATH_HEAD_WORDS + (MAXD + walkIdx) * DVERTEX_WORDS;
        else if (walkIdx - lgtCount < camCount) word = DPATH_HEAD_WORDS + (walkIdx - lgtCount) * DVERTEX_WORDS;
        if (word < 0) return 0xffffu;
        const int tri = __float_as_int(path[(size_t)word * N + i]);
        return tri >= 0 && tri < numTris ? (unsigned)leafPosOfTri[tri] : 0xffffu;
    };
    unsigned long long sub = 0;
    if (mode == 1) sub = morton;
    else if (mode == 2) sub = ((unsigned long long)posOf(0) << 16) | posOf(1);
    else if (mode == 3) sub = ((unsigned long long)posOf(0) << 24) | morton;
    else if (mode == 7 || mode == 8) {  // the BSDF type of the first two walked vertices (the branches a glossy wave diverges on), then the screen position
        auto matOf = [&](int walkIdx) -> unsigned {
            int word = -1;
            if (walkIdx < lgtCount) word = DPATH_HEAD_WORDS + (MAXD + walkIdx) * DVERTEX_WORDS;
            else if (walkIdx - lgtCount < camCount) word = DPATH_HEAD_WORDS + (walkIdx - lgtCount) * DVERTEX_WORDS;
            if (word < 0) return 3u;
            const int tri = __float_as_int(path[(size_t)word * N + i]);
            return tri >= 0 && tri < numTris ? (unsigned)materials[tris[tri].material].type & 3u : 3u;
        };
        sub = mode == 7 ? ((unsigned long long)matOf(0) << 26) | ((unsigned long long)matOf(1) << 24) | morton : ((unsigned long long)matOf(1) << 24) | morton;
    }
    else if (mode == 5) sub = morton >> 18;  // 8 x 8 screen tiles
    else if (mode == 6) sub = morton >> 14;  // 32 x 32
    // mode 4: the technique alone (what a relocation without misplaced chains would give)
    keys[i] = ((unsigned long long)SlotKey(A, i, 0) << 48) | sub;
}

// ---- the full re-sort (round 6): every K steps ALL resident chains are re-placed in the order of a 24-bit key
//   [63 - technique (6 bits) | Morton code of the camera vertex's screen position (9 + 9 bits)]
// Why: (1) the per-step relocation above places a mover at the slot QUANTILE of its key quantile, which is exact only for a stationary technique
// histogram -- while the histogram drifts (start-up, the change of the large-step probability at 10 % of a chain's samples, mlt.cpp:96-97) the surplus
// of a growing technique lands scattered inside its neighbour's region and stays there: 100-200 k technique breaks along 2^20 slots, a dozen foreign
// chains per wave, each stretching its wave to the longer technique (profiles/r06_b_*: a pure order alone = lean kernel -7.6 %); (2) inside a
// technique, chains whose camera vertices are neighbours on the screen start their walks through the same nodes and leaves: the lanes of a wave
// leave the traversal loops together (a further -9 %, decaying as small steps drift and accepted large steps jump: half of it is gone after ~16 steps).
// The order is computed on the device: key kernel, three stable 8-bit counting passes over (key, slot) pairs (one-wave blocks of RS_TILE keys: digit
// histogram -> scan over [digit][block] -> ranked scatter, ranks by ballot match), then the move of the relocation with every slot a member.
constexpr int RS_TILE = 4096;
__global__ void __launch_bounds__(256) k_rs_key(ChainArrays A, unsigned *keys) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < A.N) keys[i] = FineKey(A, i);
}
// hist[digit * nBlocks + block] = keys of the block's tile with that digit
__global__ void __launch_bounds__(64) k_rs_hist(const unsigned *keys, const int *nPtr, int nMax, int shift, int *hist, int nBlocks) {
    __shared__ int h[256];
    const int n = min(*nPtr, nMax);  // the number of keys lives on the device (the per-step relocation's member count)
    for (int k = threadIdx.x; k < 256; k += 64) h[k] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE + threadIdx.x;
    for (int j = 0; j < RS_TILE / 64; j++) {
        const int i = base + 64 * j;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 256; k += 64) hist[k * nBlocks + blockIdx.x] = h[k];
}
// histIncl: the inclusive scan of hist.  Stable: inside a (block, digit) group the entries keep their order (rank by ballot match, rounds in order).
__global__ void __launch_bounds__(64) k_rs_scatter(const unsigned *keysIn, const int *valsIn, unsigned *keysOut, int *valsOut, const int *nPtr, int nMax, int shift, const int *histIncl, int nBlocks) {
    __shared__ int cursor[256];
    const int n = min(*nPtr, nMax);
    if (blockIdx.x * RS_TILE >= n) return;
    for (int k = threadIdx.x; k < 256; k += 64) {
        const int idx = k * nBlocks + blockIdx.x;
        cursor[k] = idx ? histIncl[idx - 1] : 0;
    }
    __syncthreads();
    const unsigned long long below = (1ull << threadIdx.x) - 1ull;
    const int base = blockIdx.x * RS_TILE + threadIdx.x;
    for (int j = 0; j < RS_TILE / 64; j++) {
        const int i = base + 64 * j;
        const bool valid = i < n;
        const unsigned key = valid ? keysIn[i] : 0u;
        const unsigned d = (key >> shift) & 255u;
        unsigned long long peers = __ballot(valid);  // the valid lanes with my digit
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(valid && bit);
            peers &= bit ? m : ~m;
        }
        const int rank = __popcll(peers & below);
        if (valid) {
            const int pos = cursor[d] + rank;
            keysOut[pos] = key;
            valsOut[pos] = valsIn ? valsIn[i] : i;
        }
        __syncthreads();
        if (valid && rank == 0) cursor[d] += __popcll(peers);
        __syncthreads();
    }
}
__global__ void __launch_bounds__(64) k_rs_set_count(int *count, int n) {
    if (threadIdx.x == 0) count[0] = n;
}

// ---- the per-step relocation by the FINE key (round 6; LMC_RELOC_FINE=1 -- measured and NOT the default, host/context.cpp): the movers of a step -- the chains of its large-step launch whose technique or 32 x 32 screen
// tile is no longer the one they were placed under (an accepted large step jumps, a rejected one still finds the drift of the small steps since the
// placement) -- are sorted by the full 24-bit key with the radix sort above and land, as before, in the member slots in ascending order: the quantile
// rule now also holds the Morton order inside a technique between two full re-sorts instead of letting it decay.
constexpr int RELOC_MEMBER_SHIFT = 8;  // membership compares technique + the top 5 + 5 Morton bits
LMC_D bool FineMember(const ChainArrays &A, const unsigned *placedKey, int i, bool withoutGaussianOnly, unsigned &key) {
    if (i >= A.N || A.stepKind[i] != NEXT_LARGE) return false;
    key = FineKey(A, i);
    if ((key >> RELOC_MEMBER_SHIFT) == (placedKey[i] >> RELOC_MEMBER_SHIFT)) return false;
    if (withoutGaussianOnly && (A.flags[i] & F_GAUSS)) return false;
    return true;
}
__global__ void __launch_bounds__(64) k_relf_count(ChainArrays A, const unsigned *placedKey, int *tileCount, bool noGauss) {
    const int base = blockIdx.x * RELOC_TILE + threadIdx.x;
    int total = 0;
    for (int j = 0; j < RELOC_TILE / 64; j++) {
        unsigned key;
        total += FineMember(A, placedKey, base + 64 * j, noGauss, key) ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off);
    if (threadIdx.x == 0) tileCount[blockIdx.x] = total;
}
// one wave: tileCount -> first member index of every tile; count[0] = members (0 and count[1]++ when they exceed the staging capacity)
__global__ void __launch_bounds__(64) k_relf_offsets(int nTiles, int *tileCount, int *count, int capacity) {
    int carry = 0;
    for (int b = 0; b < nTiles; b += 64) {
        const int t = b + threadIdx.x, v = t < nTiles ? tileCount[t] : 0;
        int in = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(in, off);
            if ((int)threadIdx.x >= off) in += o;
        }
        if (t < nTiles) tileCount[t] = carry + in - v;
        carry += __shfl(in, 63);
    }
    if (threadIdx.x == 0) {
        if (carry > capacity) count[0] = 0, count[1]++;
        else
            count[0] = carry;
    }
}
// members[m] = slot (ascending), mkeys[m] = its chain's fine key
__global__ void __launch_bounds__(64) k_relf_assign(ChainArrays A, const unsigned *placedKey, const int *tileStart, const int *count, int *members, unsigned *mkeys, bool noGauss) {
    if (count[0] == 0) return;  // nothing to do, or skipped
    const int base = blockIdx.x * RELOC_TILE + threadIdx.x;
    const unsigned long long below = (1ull << threadIdx.x) - 1ull;
    int m0 = tileStart[blockIdx.x];
    for (int j = 0; j < RELOC_TILE / 64; j++) {
        unsigned key = 0;
        const bool mem = FineMember(A, placedKey, base + 64 * j, noGauss, key);
        const unsigned long long mask = __ballot(mem);
        if (mem) {
            const int m = m0 + __popcll(mask & below);
            members[m] = base + 64 * j, mkeys[m] = key;
        }
        m0 += __popcll(mask);
    }
}

__global__ void __launch_bounds__(256) k_reloc_iota(int n, int *v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = i;
}

}  // namespace
}  // namespace lmcd

using namespace lmcd;

static RecordLayout MakeRecordLayout(int maxDepth) {
    RecordLayout R;
    R.nV = std::min(MAXD, std::max(maxDepth, 1));
    R.nS = std::min(MAXCONTRIB, (maxDepth + 1) * (maxDepth + 2) / 2);
    return R;
}
size_t RelocRecordWords(int maxDepth) { return (size_t)MakeRecordLayout(maxDepth).Words(); }
void LaunchRelocIota(int n, int *v, hipStream_t s) { hipLaunchKernelGGL(k_reloc_iota, dim3((n + 255) / 256), dim3(256), 0, s, n, v); }
size_t RelocTiles(int N) { return (size_t)(N + RELOC_TILE - 1) / RELOC_TILE; }

// gather + scatter of a relocation whose members / sorted / count are in place: the wave-cooperative kernels (LMC_RELOC_COOP=0: the lane-per-record ones, A/B)
static void LaunchMoveKernels(const ChainArrays &A, const RecordLayout &R, const RelocBuffers &B, int tiles, int moveBlocks, hipStream_t s) {
    static const bool coop = !(getenv("LMC_RELOC_COOP") && atoi(getenv("LMC_RELOC_COOP")) == 0);
    if (coop) {
        hipLaunchKernelGGL(k_reloc_gather_coop, dim3(moveBlocks), dim3(64), 0, s, A, R, B.members, B.count, B.staging, B.capacity, tiles);
        hipLaunchKernelGGL(k_reloc_scatter_coop, dim3(moveBlocks), dim3(64), 0, s, A, R, B.members, B.sorted, B.count, B.staging, B.placedKey, B.capacity);
    } else {
        hipLaunchKernelGGL(k_reloc_gather, dim3(moveBlocks), dim3(64), 0, s, A, R, B.members, B.sorted, B.count, B.staging, B.capacity, tiles);
        hipLaunchKernelGGL(k_reloc_scatter, dim3(moveBlocks), dim3(64), 0, s, A, R, B.members, B.sorted, B.count, B.staging, B.placedKey, B.capacity);
    }
}
void LaunchRelocFineKey(const ChainArrays &A, const int *leafPosOfTri, int numTris, int mode, unsigned long long *keys, const TriData *tris, const DMaterial *materials, hipStream_t s) {
    hipLaunchKernelGGL(k_reloc_finekey, dim3((A.N + 255) / 256), dim3(256), 0, s, A, leafPosOfTri, numTris, mode, keys, tris, materials);
}
// the move of a relocation whose members / sorted / count the caller has filled in
void LaunchRelocMove(const ChainArrays &A, int maxDepth, const RelocBuffers &B, hipStream_t s, int keyMode) {
    const RecordLayout R = MakeRecordLayout(maxDepth);
    // every chain moves and nothing runs beside this launch: one group of 64 records per one-wave block, all of them resident together (the move is a
    // chain of dependent round trips per tile: with 4096 blocks -- four groups per wave in a row -- it took 2.9 ms at 2^20 chains, profiles/r06_i_*)
    const int moveBlocks = std::min((A.N + 63) / 64, 65536);
    LaunchMoveKernels(A, R, B, keyMode, moveBlocks, s);
}

size_t RelocSortBlocks(int N) { return (size_t)(N + RS_TILE - 1) / RS_TILE; }
// every chain re-placed by (technique, screen Morton code); B.capacity must be N (the caller checks).  W: keys[2][N], vals[2][N], hist[256 x RelocSortBlocks(N)], scan tile sums
// the three stable 8-bit passes over (key, value) pairs: keys W.keys[0] -> [1] -> [0] -> [1], values iota -> W.vals[0] -> W.vals[1] -> out; *nPtr pairs (at most nMax)
static void LaunchRadixSort24(const RelocSortBuffers &W, const int *nPtr, int nMax, int *out, hipStream_t s) {
    const int nBlocks = (int)RelocSortBlocks(nMax);
    const unsigned *kin[3] = {W.keys[0], W.keys[1], W.keys[0]};
    unsigned *kout[3] = {W.keys[1], W.keys[0], W.keys[1]};
    const int *vin[3] = {nullptr, W.vals[0], W.vals[1]};
    int *vout[3] = {W.vals[0], W.vals[1], out};
    for (int pass = 0; pass < 3; pass++) {
        hipLaunchKernelGGL(k_rs_hist, dim3(nBlocks), dim3(64), 0, s, kin[pass], nPtr, nMax, 8 * pass, W.hist, nBlocks);
        LaunchInclusiveScan(W.hist, 256 * nBlocks, W.scanSums, s);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nBlocks), dim3(64), 0, s, kin[pass], vin[pass], kout[pass], vout[pass], nPtr, nMax, 8 * pass, W.hist, nBlocks);
    }
}
// every chain re-placed by (technique, screen Morton code); B.capacity must be N (the caller checks).  W: keys[2][N], vals[2][N], hist[256 x RelocSortBlocks(N)], scan tile sums
void LaunchRelocFullSort(const ChainArrays &A, int maxDepth, const RelocBuffers &B, const RelocSortBuffers &W, hipStream_t s) {
    const int N = A.N;
    hipLaunchKernelGGL(k_rs_set_count, dim3(1), dim3(64), 0, s, B.count, N);
    hipLaunchKernelGGL(k_rs_key, dim3((N + 255) / 256), dim3(256), 0, s, A, W.keys[0]);
    LaunchRadixSort24(W, B.count, N, B.sorted, s);
    hipLaunchKernelGGL(k_reloc_iota, dim3((N + 255) / 256), dim3(256), 0, s, N, B.members);
    LaunchRelocMove(A, maxDepth, B, s, B.fine ? -1 : 0);
}
// the per-step relocation by the fine key (k_relf_*): members -> radix sort of their keys -> move
void LaunchRelocateFine(const ChainArrays &A, int maxDepth, const RelocBuffers &B, const RelocSortBuffers &W, bool withoutGaussianOnly, hipStream_t s) {
    const RecordLayout R = MakeRecordLayout(maxDepth);
    const int N = A.N, nTiles = (N + RELOC_TILE - 1) / RELOC_TILE;
    hipLaunchKernelGGL(k_relf_count, dim3(nTiles), dim3(64), 0, s, A, B.placedKey, B.tileCount, withoutGaussianOnly);
    hipLaunchKernelGGL(k_relf_offsets, dim3(1), dim3(64), 0, s, nTiles, B.tileCount, B.count, B.capacity);
    hipLaunchKernelGGL(k_relf_assign, dim3(nTiles), dim3(64), 0, s, A, B.placedKey, B.tileCount, B.count, B.members, W.keys[0], withoutGaussianOnly);
    LaunchRadixSort24(W, B.count, B.capacity, B.sorted, s);
    const int moveBlocks = std::min((N + 63) / 64, 4096);  // beside the step launches: few blocks (relocate.hip header)
    LaunchMoveKernels(A, R, B, -1, moveBlocks, s);
}

void LaunchRelocate(const ChainArrays &A, int maxDepth, const RelocBuffers &B, bool withoutGaussianOnly, hipStream_t s) {
    const RecordLayout R = MakeRecordLayout(maxDepth);
    const int N = A.N, nTiles = (N + RELOC_TILE - 1) / RELOC_TILE;
    const int tiles = (LMC_RELOC_TILES && maxDepth <= 6) ? 8 : 0;
    hipLaunchKernelGGL(k_reloc_count, dim3(nTiles), dim3(64), 0, s, A, B.placedKey, B.tileCount, B.tileHist, withoutGaussianOnly, tiles);
    hipLaunchKernelGGL(k_reloc_offsets, dim3(1), dim3(64), 0, s, nTiles, B.tileCount, B.tileHist, B.count, B.capacity);
    hipLaunchKernelGGL(k_reloc_assign, dim3(nTiles), dim3(64), 0, s, A, B.placedKey, B.tileCount, B.tileHist, B.members, B.sorted, withoutGaussianOnly, tiles);
    const int moveBlocks = std::min((N + 63) / 64, 4096);
    LaunchMoveKernels(A, R, B, tiles, moveBlocks, s);
}
