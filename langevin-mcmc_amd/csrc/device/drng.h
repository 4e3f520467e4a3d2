// Device PCG: the reference's RNG = pcg32_k64_fast (/root/reference/src/commondef.h:63,
// pcg_random.hpp:1692 = extended<6,32,oneseq_xsh_rs_64_32,oneseq_rxs_m_xs_32_32,kdd>) and the two
// libstdc++ distributions the reference draws through:
//   uniform_real_distribution<float>  = one 32-bit draw, float(x) * 2^-32, clamped below 1
//   normal_distribution<float>        = Marsaglia polar, returns y*m first and keeps x*m for the next call
// State per chain: the 64-bit LCG word (kept in registers while a kernel runs) and the 64 x u32 extension
// table (256 B per chain, AoS in HBM, read-mostly: it only changes on a "tick", once per 2^32 draws).
#pragma once
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

struct Rng {
    uint64_t state;
    uint32_t *tab;  // 64 entries, this chain's extension table
    uint32_t ticks; // number of table advances seen (only used by the MLTInit checkpoints)

    LMC_HD uint32_t Next() {  // extended::operator(), pcg_random.hpp:1187-1213
        uint64_t s = state;
        if ((s & 0xFFFFFFFFull) == 0ull) {
            PcgAdvanceTable(tab);
            ticks++;
        }
        uint32_t rhs = tab[(unsigned)(s & 63u)];
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
        const uint64_t s0 = state, s1 = s0 * PCG_MULT + PCG_INC;
        if ((s0 & 0xFFFFFFFFull) == 0ull || (s1 & 0xFFFFFFFFull) == 0ull) {
            a = Uniform();
            b = Uniform();
            return;
        }
        const uint32_t r0 = tab[(unsigned)(s0 & 63u)], r1 = tab[(unsigned)(s1 & 63u)];
        state = s1 * PCG_MULT + PCG_INC;
        const float fa = (float)(PcgOutputXshRs(s0) ^ r0) * 2.3283064365386963e-10f, fb = (float)(PcgOutputXshRs(s1) ^ r1) * 2.3283064365386963e-10f;
        a = fa >= 1.0f ? 0.99999994f : fa;
        b = fb >= 1.0f ? 0.99999994f : fb;
    }
};

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
            float mult = sqrtf(-2.0f * logf(r2) / r2);
            saved = x * mult;
            savedAvailable = true;
            ret = y * mult;
        }
        return ret * stddev + mean;
    }
};

}  // namespace lmcd
