// ORACLE -- TEST INFRASTRUCTURE ONLY (see common.h).
// Restatement of the reference's RNG = pcg32_k64_fast
//   = pcg_engines::ext_oneseq_xsh_rs_64_32<6,32,true>            (/root/reference/src/pcg_random.hpp:1692)
//   = extended<6,32, oneseq_xsh_rs_64_32, oneseq_rxs_m_xs_32_32, kdd=true>      (pcg_random.hpp:1150-1352,1623-1636)
// from M.E. O'Neill's published PCG family: 64-bit LCG base (XSH-RS output, output-previous) XORed with
// one entry of a 64 x u32 extension table selected by the low 6 state bits; the table itself is an
// array of 32-bit RXS-M-XS generators that step ("tick") whenever the low 32 state bits are zero.
// Pinned against the reference header itself by tests/test_rng.py through oracle/_ref/librefdrv.so.
//
// The distributions are libstdc++'s own <random> (the reference uses std::uniform_real_distribution<float>
// and std::normal_distribution<float>, e.g. mlt.cpp:63, gaussian.cpp:44) so the oracle consumes the engine
// exactly as the reference does; `Uniform01` / `NormalPolar` below restate them for documentation and are
// what the HIP device code mirrors.
#pragma once
#include <cmath>
#include <cstdint>
#include <random>

namespace orc {

struct Pcg32K64Fast {
    typedef uint32_t result_type;
    static constexpr uint64_t MULT = 6364136223846793005ULL;
    static constexpr uint64_t INC = 1442695040888963407ULL;
    uint64_t state;
    uint32_t data[64];

    static constexpr result_type min() { return 0u; }
    static constexpr result_type max() { return 0xFFFFFFFFu; }

    // XSH-RS 64->32 (pcg_random.hpp:787-809 with bits=64, xtypebits=32: opbits=3, xshift=22)
    static uint32_t OutputXshRs(uint64_t x) {
        unsigned rshift = (unsigned)(x >> 61) & 7u;
        x ^= x >> 22;
        return (uint32_t)(x >> (22 + rshift));
    }
    uint32_t Base() {  // engine::operator() with output_previous (pcg_random.hpp:382-394)
        uint64_t old = state;
        state = old * MULT + INC;
        return OutputXshRs(old);
    }

    // RXS-M-XS 32->32 and its inverse (pcg_random.hpp:920-952), multiplier constants :903-904
    static uint32_t OutputRxsMXs(uint32_t x) {
        unsigned rshift = (x >> 28) & 15u;
        x ^= x >> (4 + rshift);
        x *= 277803737u;
        x ^= x >> 22;
        return x;
    }
    static uint32_t Unxorshift(uint32_t x, unsigned bits, unsigned shift) {  // pcg_extras.hpp:256-272
        if (2 * shift >= bits) return x ^ (x >> shift);
        uint32_t lowmask1 = (uint32_t(1) << (bits - shift * 2)) - 1;
        uint32_t highmask1 = ~lowmask1;
        uint32_t top1 = x;
        uint32_t bottom1 = x & lowmask1;
        top1 ^= top1 >> shift;
        top1 &= highmask1;
        x = top1 | bottom1;
        uint32_t lowmask2 = (uint32_t(1) << (bits - shift)) - 1;
        uint32_t bottom2 = x & lowmask2;
        bottom2 = Unxorshift(bottom2, bits - shift, shift);
        bottom2 &= lowmask1;
        return top1 | bottom2;
    }
    static uint32_t UnoutputRxsMXs(uint32_t x) {
        x = Unxorshift(x, 32, 22);
        x *= 2897767785u;
        unsigned rshift = (x >> 28) & 15u;
        x = Unxorshift(x, 32, 4 + rshift);
        return x;
    }
    static bool ExternalStep(uint32_t &randval, uint32_t i) {  // inside_out::external_step, pcg_random.hpp:1123-1130
        uint32_t s = UnoutputRxsMXs(randval);
        s = s * 747796405u + 2891336453u + i * 2u;
        uint32_t result = OutputRxsMXs(s);
        randval = result;
        return result == 0u;
    }
    void AdvanceTable() {  // pcg_random.hpp:1439-1448
        bool carry = false;
        for (uint32_t i = 0; i < 64; ++i) {
            if (carry) carry = ExternalStep(data[i], i + 1);
            bool carry2 = ExternalStep(data[i], i + 1);
            carry = carry || carry2;
        }
    }

    explicit Pcg32K64Fast(uint64_t seed = 0xcafef00dd15ea5e5ULL) {
        state = (seed + INC) * MULT + INC;  // engine ctor: bump(state + increment()), pcg_random.hpp:434-437
        // selfinit, pcg_random.hpp:1337-1352 (gcc evaluates the left operand of '-' first)
        uint32_t a = Base();
        uint32_t b = Base();
        uint32_t xdiff = a - b;
        for (int i = 0; i < 64; ++i) data[i] = Base() ^ xdiff;
    }

    result_type operator()() {  // extended::operator(), pcg_random.hpp:1187-1213
        uint64_t s = state;
        unsigned index = (unsigned)(s & 63u);
        if ((s & 0xFFFFFFFFull) == 0ull) AdvanceTable();
        uint32_t rhs = data[index];
        uint32_t lhs = Base();
        return lhs ^ rhs;
    }
};

typedef Pcg32K64Fast RNG;

// libstdc++ generate_canonical<float,24> with a 32-bit engine = ONE draw: float(x) * 2^-32, and a result
// that rounds up to 1.0f is replaced by nextafterf(1,0)  (bits/random.tcc).
inline float Uniform01(RNG &rng) {
    float r = float(rng()) * 2.3283064365386963e-10f;
    if (r >= 1.0f) r = 0.99999994f;
    return r;
}

// libstdc++ normal_distribution<float>: Marsaglia polar; returns y*m first and saves x*m.
struct NormalPolar {
    float mean, stddev, saved = 0.f;
    bool savedAvailable = false;
    NormalPolar(float m, float s) : mean(m), stddev(s) {}
    float operator()(RNG &rng) {
        float ret;
        if (savedAvailable) {
            savedAvailable = false;
            ret = saved;
        } else {
            float x, y, r2;
            do {
                x = 2.0f * Uniform01(rng) - 1.0f;
                y = 2.0f * Uniform01(rng) - 1.0f;
                r2 = x * x + y * y;
            } while (r2 > 1.0f || r2 == 0.0f);
            float mult = std::sqrt(-2 * std::log(r2) / r2);
            saved = x * mult;
            savedAvailable = true;
            ret = y * mult;
        }
        return ret * stddev + mean;
    }
};

}  // namespace orc
