// Deterministic single-precision exp / log / pow for the glossy BSDFs (phong.cpp:42,109: pow; microfacet.h:17,173: exp, log) and the
// H2MC Gaussian.  The reference calls libm's float versions, whose last bit differs between libm builds and from the device libm;
// rounds 1-2 evaluated them in DOUBLE and rounded once, on both sides, which made the device pay double-precision transcendentals in
// every BSDF call (26 % of a full-material step).  These are float only: IEEE +, -, *, / (no contraction: the product
// and the oracle compile with -ffp-contract=off), floorf, int <-> float conversions and bit manipulation -- the SAME source gives the
// same bits under g++ and hipcc by construction.  Accuracy (tests/test_host.py, 4e6 arguments against float64): exp and log within
// 1 ulp, pow within 1 ulp for |y log x| <= 30 (the Phong range) and 2 ulp up to the overflow threshold.
//   log x = k ln2 + log m, m in [sqrt(1/2), sqrt(2)): log m = 2 s + 2 s^3 / 3 + ... with s = (m - 1) / (m + 1); s and the leading term
//           2 s are carried as float-float pairs (TwoSum / TwoProd error-free transforms), the series tail in plain float
//   exp y = 2^n exp(r), n = round(y / ln2), r = y - n ln2 (ln2 split so that n ln2_h is exact), degree-7 polynomial
//   pow(x, y) = exp(y log x) with log x and the product kept as float-float pairs
#pragma once
#include <cmath>
#include <cstdint>

#ifndef LMC_HD
#if defined(__HIPCC__)
#define LMC_HD __host__ __device__ inline
#else
#define LMC_HD inline
#endif
#endif

namespace lmcd {

struct FF {  // unevaluated sum h + l, |l| <= ulp(h) / 2
    float h, l;
};
LMC_HD float BitsToFloat(uint32_t u) { return __builtin_bit_cast(float, u); }
LMC_HD uint32_t FloatToBits(float f) { return __builtin_bit_cast(uint32_t, f); }
LMC_HD FF TwoSum(float a, float b) {  // a + b exactly (Knuth)
    const float s = a + b;
    const float bb = s - a;
    return FF{s, (a - (s - bb)) + (b - bb)};
}
LMC_HD FF FastTwoSum(float a, float b) {  // |a| >= |b|
    const float s = a + b;
    return FF{s, b - (s - a)};
}
LMC_HD FF TwoProd(float a, float b) {  // a * b exactly (Dekker, Veltkamp split at 12 bits; no fused multiply-add needed)
    const float p = a * b;
    const float ca = 4097.0f * a, cb = 4097.0f * b;
    const float ah = ca - (ca - a), bh = cb - (cb - b);
    const float al = a - ah, bl = b - bh;
    return FF{p, ((ah * bh - p) + ah * bl + al * bh) + al * bl};
}
constexpr float LN2_H = 0.693115234375f;           // 0x3f317000: 12 trailing zero bits, n * LN2_H is exact for |n| < 2048
constexpr float LN2_L = 3.1946183298714459e-05f;  // ln 2 - LN2_H
constexpr float INV_LN2 = 1.44269502f;

// x = 2^k m (finite, positive) with m in [sqrt(1/2), sqrt(2))
LMC_HD void SplitMantissa(float x, float &m, int &k) {
    k = 0;
    uint32_t ix = FloatToBits(x);
    if (ix < 0x00800000u) {  // subnormal
        x *= 8388608.0f;
        k = -23;
        ix = FloatToBits(x);
    }
    k += (int)(ix >> 23) - 127;
    m = BitsToFloat((ix & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) {
        m *= 0.5f;
        k++;
    }
}
// log(2^k m) as a float-float pair
LMC_HD FF LogFF(float m, int k) {
    const float num = m - 1.0f;  // exact
    const FF den = TwoSum(m, 1.0f);
    const float sh = num / den.h;
    const FF p = TwoProd(sh, den.h);
    const float sl = (((num - p.h) - p.l) - sh * den.l) / den.h;  // s = sh + sl
    const float z = sh * sh;
    const float tail = (2.0f * sh * z) * (0.333333343f + z * (0.2f + z * (0.142857149f + z * (0.111111112f + z * (0.0909090936f + z * (0.0769230798f + z * 0.0666666701f))))));
    const float kf = (float)k;
    const FF a = TwoSum(kf * LN2_H, 2.0f * sh);
    return FastTwoSum(a.h, a.l + (kf * LN2_L + (2.0f * sl + tail)));
}
// exp(yh + yl), |yl| << |yh|
LMC_HD float ExpFF(float yh, float yl) {
    if (yh != yh) return yh;
    if (yh > 88.7228394f) return INFINITY;
    if (yh < -103.972084f) return 0.0f;
    const float nf = floorf(yh * INV_LN2 + 0.5f);
    const float r = ((yh - nf * LN2_H) - nf * LN2_L) + yl;
    const float q = r * r * (0.5f + r * (0.166666672f + r * (0.0416666679f + r * (0.00833333377f + r * (0.00138888892f + r * 0.000198412701f)))));
    const float e = 1.0f + (r + q);
    int n = (int)nf;
    // 2^n in two factors so that results in the subnormal range are reached by one rounding multiplication
    if (n < -126) return (e * BitsToFloat((uint32_t)(n + 127 + 100) << 23)) * 7.8886090522101181e-31f;  // 2^-100
    if (n > 127) return (e * BitsToFloat((uint32_t)(n + 127 - 1) << 23)) * 2.0f;
    return e * BitsToFloat((uint32_t)(n + 127) << 23);
}

LMC_HD float llogf(float x) {
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    float m;
    int k;
    SplitMantissa(x, m, k);
    const FF L = LogFF(m, k);
    return L.h + L.l;
}
LMC_HD float lexpf(float x) { return ExpFF(x, 0.0f); }
// libm's conventions for the arguments that occur (finite base): pow(x, 0) = 1, pow(+-0, y), negative bases with integer exponents
LMC_HD float lpowf(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return NAN;
    float sign = 1.0f;
    if (x < 0.0f) {
        if (floorf(y) != y) return NAN;
        const float half = y * 0.5f;
        if (floorf(half) != half) sign = -1.0f;  // odd integer
        x = -x;
    }
    if (x == 0.0f) return y > 0.0f ? 0.0f * sign : INFINITY;
    if (x == INFINITY) return y > 0.0f ? INFINITY : 0.0f;
    if (x == 1.0f) return sign;
    float m;
    int k;
    SplitMantissa(x, m, k);
    const FF L = LogFF(m, k);
    const FF p = TwoProd(y, L.h);
    if (!(fabsf(p.h) < 1e30f)) return (p.h > 0.0f) ? INFINITY : 0.0f;  // |y log x| astronomically large (or the split overflowed)
    return sign * ExpFF(p.h, p.l + y * L.l);
}

}  // namespace lmcd
