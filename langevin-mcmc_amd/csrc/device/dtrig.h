// Deterministic single-precision sin / cos / acos / atan2 for the sampling code (sampling.h SampleSphere / ToSphericalCoord / SampleCosHemisphere /
// SampleConcentricDisc, phong.cpp, roughdielectric.cpp, envlight.cpp) -- the dtrans.h treatment for the trigonometric functions.
// The reference calls libm's float versions; the device libm and glibc differ in the last bit of some results, and one ulp in a scattering direction
// re-seeds a whole chain (and, through MLTInit's CDF, mlt.h:115, the seeding of many), which is why the veach-door parity used to be statistical.
// These are float only: IEEE + - * / sqrt, EXPLICIT fused multiply-adds (__builtin_fmaf: one rounding by definition, v_fma_f32 on the device, the
// fma instruction or glibc's exact fmaf on the host -- the build's -ffp-contract=off forbids every IMPLICIT fusion), rintf and int <-> float
// conversions: the same source gives the same bits under g++ and hipcc by construction.
// Accuracy against float64 (tests/test_host.py, exhaustive over the argument ranges the path code produces): see the test's stated bounds (<= 1.5 ulp).
//   sin / cos: k = rint(x 2/pi), r = x - k pi/2 with pi/2 in three float pieces (exact for |k| < 2^9; beyond |x| = 512 the reduction runs in
//              double), odd / even minimax polynomials on |r| <= pi/4
//   acos:      fdlibm's three ranges (|x| < 1/2: pi/2 - asin x;  x <= -1/2: pi - 2 asin sqrt((1+x)/2);  x >= 1/2: 2 asin sqrt((1-x)/2) with the
//              square root's rounding error carried), asin z = z + z^3 R(z^2)
//   atan2:     q = min / max of (|y|, |x|) in [0, 1], one reduction at tan(pi/8) (t = (q - 1) / (q + 1)), atan t = t + t^3 P(t^2); the octant's
//              constants as float pairs so that the last addition is the only sizeable rounding
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

LMC_HD float tfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// r = x - k pi/2 (|r| <= pi/4 + a rounding), returns k mod 4 in [0, 3]
LMC_HD int TrigReduce(float x, float &r) {
    if (!(fabsf(x) <= 512.0f)) {  // large (or non-finite) arguments: never produced by the path code (its arguments lie in [-pi, 2 pi]); reduced in double
        if (!(fabsf(x) < 3.0e9f)) {  // beyond 2^31 quarter turns (or inf / nan): the result carries no information; nan like libm's for inf / nan
            r = x - x;               // 0 for finite, nan for inf / nan
            return 0;
        }
        const double kd = rint((double)x * 0.63661977236758134308);
        const double rd = __builtin_fma(-kd, 6.12323399573676603587e-17, __builtin_fma(-kd, 1.57079632679489655800e+00, (double)x));  // pi/2 in two doubles; the first fma's product is exact
        r = (float)rd;
        return (int)((long long)kd & 3);
    }
    const float k = rintf(x * 0.636619747f);
    float t = tfma(-k, 1.57079637e+00f, x);   // pi/2 = 1.57079637 - 4.37113883e-08 - 1.71512451e-15 ...
    t = tfma(-k, -4.37113883e-08f, t);
    t = tfma(-k, -1.71512451e-15f, t);
    r = t;
    return (int)k & 3;
}
LMC_HD float SinPoly(float r) {  // sin r, |r| <= pi/4
    const float z = r * r;
    float s = tfma(z, 2.724988008e-06f, -1.984008704e-04f);
    s = tfma(z, s, 8.333331905e-03f);
    s = tfma(z, s, -1.666666716e-01f);
    return tfma(r * z, s, r);
}
LMC_HD float CosPoly(float r) {  // cos r, |r| <= pi/4
    const float z = r * r;
    float c = tfma(z, -2.730091069e-07f, 2.480059993e-05f);
    c = tfma(z, c, -1.388888806e-03f);
    c = tfma(z, c, 4.166666791e-02f);
    const float hz = 0.5f * z, w = 1.0f - hz;
    return w + (((1.0f - w) - hz) + (z * z) * c);
}
LMC_HD float dsinf(float x) {
    float r;
    const int q = TrigReduce(x, r);
    const float v = (q & 1) ? CosPoly(r) : SinPoly(r);
    if (x == 0.0f) return x;  // sin(-0) = -0 (the reduction's and the polynomial's fma give +0)
    return (q & 2) ? -v : v;
}
LMC_HD float dcosf(float x) {
    float r;
    const int q = TrigReduce(x, r);
    const float v = (q & 1) ? SinPoly(r) : CosPoly(r);
    return ((q + 1) & 2) ? -v : v;
}

// (asin z - z) / z^3 as a polynomial in w = z^2, 0 <= w <= 1/4
LMC_HD float AsinR(float w) {
    float p = tfma(w, 3.393122926e-02f, 1.700004004e-02f);
    p = tfma(w, p, 3.113304637e-02f);
    p = tfma(w, p, 4.459653795e-02f);
    p = tfma(w, p, 7.500103116e-02f);
    return tfma(w, p, 1.666666567e-01f);
}
LMC_HD float dacosf(float x) {
    const float PIO2_HI = 1.57079625e+00f, PIO2_LO = 7.54978942e-08f;  // fdlibm's split: 0x3fc90fda + 0x33a22168
    const float PI_F = 3.14159274e+00f;
    const float ax = fabsf(x);
    if (!(ax < 1.0f)) {
        if (x == 1.0f) return 0.0f;
        if (x == -1.0f) return PI_F;
        return (x - x) / (x - x);  // |x| > 1 or nan
    }
    if (ax < 0.5f) {
        if (ax < 2.98023224e-08f) return PIO2_HI + PIO2_LO;  // 2^-25: x below half an ulp of pi/2
        const float w = x * x;
        const float r = w * AsinR(w);
        return PIO2_HI - (x - (PIO2_LO - x * r));
    }
    if (x < 0.0f) {
        const float w = (1.0f + x) * 0.5f;
        const float s = sqrtf(w);
        const float r = w * AsinR(w);
        const float t = r * s - PIO2_LO;
        return 2.0f * (PIO2_HI - (s + t));
    }
    const float w = (1.0f - x) * 0.5f;
    const float s = sqrtf(w);
    const float df = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) & 0xfffff000u);
    const float c = (w - df * df) / (s + df);
    const float r = w * AsinR(w);
    const float t = r * s + c;
    return 2.0f * (df + t);
}

// (atan t - t) / t^3 as a polynomial in w = t^2, |t| <= tan(pi/8)
LMC_HD float AtanP(float w) {
    float p = tfma(w, 5.059889704e-02f, -8.629743755e-02f);
    p = tfma(w, p, 1.107213050e-01f);
    p = tfma(w, p, -1.428419799e-01f);
    p = tfma(w, p, 1.999997795e-01f);
    return tfma(w, p, -3.333333433e-01f);
}
// atan2 with libm's conventions for the arguments that occur (finite, not both zero: the callers test (0, 0) themselves); signed zeros as atan2f
LMC_HD float datan2f(float y, float x) {
    if (x != x || y != y) return x + y;
    const float ax = fabsf(x), ay = fabsf(y);
    const float PI_HI = 3.14159274e+00f, PI_LO = -8.74227766e-08f;        // pi = hi + lo
    const float PIO2_HI = 1.57079637e+00f, PIO2_LO = -4.37113883e-08f;
    const float PIO4_HI = 7.85398185e-01f, PIO4_LO = -2.18556941e-08f;
    float base_hi = 0.0f, base_lo = 0.0f, t, t_lo = 0.0f;  // atan(min / max) = base + atan(t + t_lo)
    if (ax == 0.0f && ay == 0.0f) {  // atan2f(+-0, +-0) = +-0 or +-pi
        const float v = __builtin_signbit(x) ? PI_HI : 0.0f;
        return __builtin_signbit(y) ? -v : v;
    }
    if (ax == INFINITY && ay == INFINITY) {
        t = 0.0f, base_hi = PIO4_HI, base_lo = PIO4_LO;
    } else {
        const float mn = ax < ay ? ax : ay, mx = ax < ay ? ay : ax;
        if (mx == INFINITY) {
            t = 0.0f;
        } else if (mn > 0.414213568f * mx) {  // t = (mn - mx) / (mn + mx), numerator and denominator as exact float pairs, the quotient with its remainder
            const float nh = mn - mx, nb = nh - mn, nl = (mn - (nh - nb)) + (-mx - nb);  // TwoSum(mn, -mx)
            const float dh = mn + mx, db = dh - mn, dl = (mn - (dh - db)) + (mx - db);   // TwoSum(mn, mx)
            t = nh / dh;
            t_lo = ((tfma(-t, dh, nh) + nl) - t * dl) / dh;
            base_hi = PIO4_HI, base_lo = PIO4_LO;
        } else {
            t = mn / mx;
            t_lo = tfma(-t, mx, mn) / mx;
        }
    }
    const float w = t * t;
    // a = base + t + t^3 P(w) + t_lo / (1 + w), assembled so that only the additions onto base_hi round at the result's scale
    const float corr = tfma(t * w, AtanP(w), base_lo) + t_lo * (1.0f - w);
    float a_hi = base_hi + t;                    // |t| <= 0.4143 < base_hi: Fast2Sum applies when base_hi != 0; when it is 0 the sum is exact
    float a_lo = (t - (a_hi - base_hi)) + corr;
    // octant: |y| > |x| -> pi/2 - a;  x < 0 -> pi - a;  y < 0 -> negate
    if (ay > ax) {
        const float h = PIO2_HI - a_hi;
        a_lo = ((PIO2_HI - h) - a_hi) + (PIO2_LO - a_lo);
        a_hi = h;
    }
    if (__builtin_signbit(x)) {
        const float h = PI_HI - a_hi;
        a_lo = ((PI_HI - h) - a_hi) + (PI_LO - a_lo);
        a_hi = h;
    }
    const float a = a_hi + a_lo;
    return __builtin_signbit(y) ? -a : a;
}

}  // namespace lmcd
