// Device PCG: the reference's RNG = pcg32_k64_fast (/root/reference/src/commondef.h:63,
// pcg_random.hpp:1692 = extended<6,32,oneseq_xsh_rs_64_32,oneseq_rxs_m_xs_32_32,kdd>) and the two
// libstdc++ distributions the reference draws through:
//   uniform_real_distribution<float>  = one 32-bit draw, float(x) * 2^-32, clamped below 1
//   normal_distribution<float>        = Marsaglia polar, returns y*m first and keeps x*m for the next call
// State per chain: the 64-bit LCG word (kept in registers while a kernel runs) and the 64 x u32 extension
// table (256 B per chain, AoS in HBM; it only changes on a "tick", once per 2^32 draws -- until then it is synthesised from
// the seed and never read, see PcgJumpTable below).
#pragma once
#include <stdexcept>
#include "dmath.h"

namespace lmcd {

constexpr uint64_t PCG_MULT = 6364136223846793005ULL;
constexpr uint64_t PCG_INC = 1442695040888963407ULL;

LMC_HD uint32_t PcgOutputXshRs(uint64_t x) {  // pcg_random.hpp:787-809 (64 -> 32: opbits 3, xshift 22)
    unsigned rshift = (unsigned)(x >> 61) & 7u;
    x ^= x >> 22;
    return (uint32_t)(x >> (22 + rshift));
}
LMC_HD uint32_t PcgOutputRxsMXs(uint32_t x) {  // pcg_random.hpp:920-935
    unsigned rshift = (x >> 28) & 15u;
    x ^= x >> (4 + rshift);
    x *= 277803737u;
    x ^= x >> 22;
    return x;
}
LMC_HD uint32_t PcgUnxorshift32(uint32_t x, unsigned shift) {  // inverse of x ^= x >> shift on 32 bits
    uint32_t r = x;
    for (unsigned s = shift; s < 32; s += shift) r = x ^ (r >> shift);
    return r;
}
LMC_HD uint32_t PcgUnoutputRxsMXs(uint32_t x) {  // pcg_random.hpp:937-951
    x = PcgUnxorshift32(x, 22);
    x *= 2897767785u;
    unsigned rshift = (x >> 28) & 15u;
    x = PcgUnxorshift32(x, 4 + rshift);
    return x;
}
LMC_HD bool PcgExternalStep(uint32_t &randval, uint32_t i) {  // inside_out::external_step, pcg_random.hpp:1123-1130
    uint32_t s = PcgUnoutputRxsMXs(randval);
    s = s * 747796405u + 2891336453u + i * 2u;
    uint32_t result = PcgOutputRxsMXs(s);
    randval = result;
    return result == 0u;
}
// cold: runs once per 2^32 draws of a stream; kept out of line so that the ~40 inlined RNG call sites of a step kernel
// do not each carry the 64-entry table walk
#if defined(__HIPCC__)
__host__ __device__ inline __attribute__((noinline))
#else
inline
#endif
void PcgAdvanceTable(uint32_t *tab) {  // pcg_random.hpp:1439-1448
    bool carry = false;
    for (uint32_t i = 0; i < 64; ++i) {
        uint32_t v = tab[i];
        if (carry) carry = PcgExternalStep(v, i + 1);
        bool carry2 = PcgExternalStep(v, i + 1);
        carry = carry || carry2;
        tab[i] = v;
    }
}
// RNG(seed): engine ctor + selfinit (pcg_random.hpp:434-437,1337-1352)
LMC_HD uint64_t PcgSeed(uint64_t seed, uint32_t *tab) {
    uint64_t state = (seed + PCG_INC) * PCG_MULT + PCG_INC;
    uint32_t a = PcgOutputXshRs(state);
    state = state * PCG_MULT + PCG_INC;
    uint32_t b = PcgOutputXshRs(state);
    state = state * PCG_MULT + PCG_INC;
    uint32_t xdiff = a - b;
    for (int i = 0; i < 64; ++i) {
        tab[i] = PcgOutputXshRs(state) ^ xdiff;
        state = state * PCG_MULT + PCG_INC;
    }
    return state;
}

// ---- The extension table of a stream that has never ticked is a pure function of its seed (PcgSeed above): entry k is the XSH-RS output of the
// LCG state k steps behind S0 (the state the fill loop starts from), xored with xdiff.  A chain's stream is RNG(chainId + seedOffset)
// (mlt.cpp:61-62), so a step kernel need not READ the chain's 256-byte table at all: entry k = XshRs(A_k S0 + C_k) ^ xdiff with the 64 jump
// constants (A_k, C_k) = (MULT^k, INC (MULT^k - 1) / (MULT - 1)) -- one 16-byte constant look-up and a 64-bit multiply-add per draw instead of
// a 4-byte load from a table that, with 16 K resident chains per XCD, is 4 MB of live data by itself: the whole L2 of an XCD.  Measured
// (profiles/r05_g_*): the lean kernel read 2106 B per chain-step from the HBM side with the tables in memory, 1259 B without -- the 256-byte
// table was fetched 3.3 times per step.  A stream that ticks (once per 2^32 draws: the table then changes for good) materialises its table
// into the chain's slot of A.rngTab and reads it from there from then on (Rng::synth, A.rngTicked).
struct PcgJump {
    uint64_t a, c;
};
struct PcgJumpTable {
    PcgJump j[64];
    constexpr PcgJumpTable() : j{} {
        uint64_t a = 1, c = 0;
        for (int k = 0; k < 64; k++) {
            j[k].a = a, j[k].c = c;
            a = a * PCG_MULT;
            c = c * PCG_MULT + PCG_INC;
        }
    }
};
#if defined(__HIPCC__)
__device__ __constant__ const PcgJumpTable c_pcgJump{};
#endif

// ---- glibc's logf, restated: the `std::log(r2)` of libstdc++'s normal_distribution<float> (bits/random.tcc; the reference's gaussian.cpp:44,
// mutation_small.h:35, path.cpp:1958) is glibc's logf -- since 2.27 the table-driven DOUBLE-precision routine of ARM's optimized-routines
// (sysdeps/ieee754/flt-32/e_logf.c, e_logf_data.c: x = 2^k z, z in [0x1.66p-1, 0x1.66p0) by the top 4 mantissa bits i, r = z invc[i] - 1,
// log x = k ln2 + logc[i] + r + r^2 (A2 + A1 r + A0 r^2), one rounding to float at the end).  Plain IEEE double arithmetic, so it can be restated
// bit for bit; the device libm's logf (a float polynomial) agreed with it on 84.6 % of the polar method's arguments only.
// What is restated is the build of that source the ifunc selects on every x86-64 CPU with FMA (sysdeps/x86_64/fpu/multiarch/e_logf-fma.c = the same
// C under -mfma -mavx2): WHICH operations gcc fused there was read off this image's libm.so.6 (GLIBC 2.35-0ubuntu3.11, objdump of __logf_fma) and
// is spelled out below with explicit fma's; the constants are the bytes of its .rodata.  Pinned by tests/test_host.py against the host's logf on
// EVERY float of (0, 1] -- the polar method's whole domain (1 065 353 216 arguments).
struct LogfTable {
    double t[16][2];  // invc, logc
};
#define LMC_LOGF_TABLE_INIT                                                                                                                       \
    {{{0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2}, \
      {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},    \
      {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, \
      {0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5}, {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},                                  \
      {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3}, {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},     \
      {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}}}
#if defined(__HIPCC__)
__device__ __constant__ const LogfTable c_logfTab = LMC_LOGF_TABLE_INIT;
#endif
#if defined(LMC_RNG_JUMP_LDS) && defined(__HIP_DEVICE_COMPILE__)
// A translation unit that defines LMC_RNG_JUMP_LDS (the lean small-step launch) keeps the 1 KB of jump constants in LDS: a draw's look-up then is an LDS
// read instead of a vector-memory load -- and a vector-memory load's wait (`vmcnt` retires in order) is also a wait for every load issued before
// it, i.e. for the step's prefetched path words on their way from HBM.  The kernel fills the table once per block (PcgJumpLdsInit + barrier).
__device__ __forceinline__ PcgJump *PcgJumpLds() {
    __shared__ PcgJump t[64];
    return t;
}
__device__ __forceinline__ double2 *LogfTabLds() {  // ... and so do the 256 bytes of logf's table (one look-up per pair of normal variates)
    __shared__ double2 t[16];
    return t;
}
// Needs a block of at least 64 threads: every launcher of a kernel that calls it checks its block size on the host (RequireJumpLdsBlock).  (ADVICE r5
// asked for a strided fill instead, `for (k = threadIdx.x; k < 64; k += blockDim.x)`: with it the veach-door launches faulted at 2^18 chains and more,
// a run later the same build with this form did not -- gpurun_out r06_d .. r06_f; the loop form is not used.)
__device__ __forceinline__ void PcgJumpLdsInit() {
    if (threadIdx.x < 64) PcgJumpLds()[threadIdx.x] = c_pcgJump.j[threadIdx.x];
    if (threadIdx.x < 16) LogfTabLds()[threadIdx.x] = make_double2(c_logfTab.t[threadIdx.x][0], c_logfTab.t[threadIdx.x][1]);
    __syncthreads();
}
#endif
#if defined(__HIPCC__)
inline void RequireJumpLdsBlock(int blockThreads) {
    if (blockThreads < 64) throw std::runtime_error("a kernel that keeps the PCG jump constants in LDS needs blocks of at least 64 threads");
}
#endif
#if defined(LMC_RNG_JUMP_LDS) && defined(__HIP_DEVICE_COMPILE__)
#define LMC_RNG_JUMP_INIT() lmcd::PcgJumpLdsInit()
#else
#define LMC_RNG_JUMP_INIT() ((void)0)  // this translation unit (or the host pass) reads the constants from memory
#endif
LMC_HD PcgJump PcgJumpOf(unsigned k) {
#if defined(LMC_RNG_JUMP_LDS) && defined(__HIP_DEVICE_COMPILE__)
    return PcgJumpLds()[k];
#elif defined(__HIP_DEVICE_COMPILE__)
    return c_pcgJump.j[k];
#else
    static const PcgJumpTable t{};
    return t.j[k];
#endif
}
// S0 and xdiff of RNG(seed) (the first lines of PcgSeed); the stream's first state is 64 LCG steps behind S0
LMC_HD void PcgSeedConstants(uint64_t seed, uint64_t &s0, uint32_t &xdiff) {
    uint64_t state = (seed + PCG_INC) * PCG_MULT + PCG_INC;
    const uint32_t a = PcgOutputXshRs(state);
    state = state * PCG_MULT + PCG_INC;
    const uint32_t b = PcgOutputXshRs(state);
    state = state * PCG_MULT + PCG_INC;
    s0 = state, xdiff = a - b;
}

struct Rng {
    uint64_t state;
    uint32_t *tab;  // 64 entries, this stream's extension table in memory (contents undefined while `synth`)
    uint32_t ticks; // number of table advances seen (only used by the MLTInit checkpoints)
    // the table synthesised from the seed (SetSynth): valid until the stream's first tick
    uint64_t s0 = 0;
    uint32_t xdiff = 0;
    bool synth = false;
    LMC_HD void SetSynth(uint64_t seed) {
        PcgSeedConstants(seed, s0, xdiff);
        synth = true;
    }
    LMC_HD uint32_t Entry(unsigned k) const {
        if (synth) {
            const PcgJump j = PcgJumpOf(k);
            return PcgOutputXshRs(j.a * s0 + j.c) ^ xdiff;
        }
        return tab[k];
    }
    LMC_HD void Tick() {  // the table changes for good: from here on it lives in memory
        if (synth) {
            for (unsigned k = 0; k < 64; k++) tab[k] = Entry(k);
            synth = false;
        }
        PcgAdvanceTable(tab);
        ticks++;
    }

    LMC_HD uint32_t Next() {  // extended::operator(), pcg_random.hpp:1187-1213
        uint64_t s = state;
        if ((s & 0xFFFFFFFFull) == 0ull) Tick();
        uint32_t rhs = Entry((unsigned)(s & 63u));
        state = s * PCG_MULT + PCG_INC;
        return PcgOutputXshRs(s) ^ rhs;
    }
    LMC_HD float Uniform() {  // generate_canonical<float,24> with a 32-bit engine
        float r = (float)Next() * 2.3283064365386963e-10f;
        return r >= 1.0f ? 0.99999994f : r;
    }
    // Two consecutive draws (a first, then b) with both table look-ups in flight together: the second state is one LCG step
    // away, so its table slot is known before the first value has arrived.  Same stream as two Uniform() calls; the
    // once-per-2^32 table advance takes the sequential path.
    LMC_HD void Uniform2(float &a, float &b) {
        const uint64_t st0 = state, st1 = st0 * PCG_MULT + PCG_INC;
        if ((st0 & 0xFFFFFFFFull) == 0ull || (st1 & 0xFFFFFFFFull) == 0ull) {
            a = Uniform();
            b = Uniform();
            return;
        }
        const uint32_t r0 = Entry((unsigned)(st0 & 63u)), r1 = Entry((unsigned)(st1 & 63u));
        state = st1 * PCG_MULT + PCG_INC;
        const float fa = (float)(PcgOutputXshRs(st0) ^ r0) * 2.3283064365386963e-10f, fb = (float)(PcgOutputXshRs(st1) ^ r1) * 2.3283064365386963e-10f;
        a = fa >= 1.0f ? 0.99999994f : fa;
        b = fb >= 1.0f ? 0.99999994f : fb;
    }
};

LMC_HD void LogfTabOf(int i, double &invc, double &logc) {
#if defined(LMC_RNG_JUMP_LDS) && defined(__HIP_DEVICE_COMPILE__)
    const double2 e = LogfTabLds()[i];
    invc = e.x, logc = e.y;
#elif defined(__HIP_DEVICE_COMPILE__)
    invc = c_logfTab.t[i][0], logc = c_logfTab.t[i][1];
#else
    static const LogfTable T = LMC_LOGF_TABLE_INIT;
    invc = T.t[i][0], logc = T.t[i][1];
#endif
}
LMC_HD float GlibcLogf(float x) {
    uint32_t ix = __builtin_bit_cast(uint32_t, x);
    if (ix == 0x3f800000u) return 0.0f;  // log(1) = +0 in every rounding mode
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {  // x < 0x1p-126 or inf or nan
        if (ix * 2u == 0u) return -INFINITY;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");
        ix = __builtin_bit_cast(uint32_t, x * 0x1p23f);  // subnormal: normalised
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    double invc, logc;
    LogfTabOf(i, invc, logc);
    const double z = (double)__builtin_bit_cast(float, iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = __builtin_fma((double)k, 0x1.62e42fefa39efp-1, logc);
    const double r2 = r * r;
    double y = __builtin_fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = __builtin_fma(-0x1.00ea348b88334p-2, r2, y);
    y = __builtin_fma(y, r2, y0 + r);
    return (float)y;
}

// one normal_distribution<float> object (the saved variate lives as long as the object)
struct NormalDist {
    float mean, stddev, saved;
    bool savedAvailable;
    LMC_HD NormalDist(float m, float s) : mean(m), stddev(s), saved(0.f), savedAvailable(false) {}
    LMC_HD float operator()(Rng &rng) {
        float ret;
        if (savedAvailable) {
            savedAvailable = false;
            ret = saved;
        } else {
            float x, y, r2;
            do {
                float u0, u1;
                rng.Uniform2(u0, u1);
                x = 2.0f * u0 - 1.0f;
                y = 2.0f * u1 - 1.0f;
                r2 = x * x + y * y;
            } while (r2 > 1.0f || r2 == 0.0f);
            float mult = sqrtf(-2.0f * GlibcLogf(r2) / r2);
            saved = x * mult;
            savedAvailable = true;
            ret = y * mult;
        }
        return ret * stddev + mean;
    }
};

}  // namespace lmcd
