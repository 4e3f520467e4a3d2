// Device-side scalar / vector helpers for the gfx950 kernels.
// Semantics follow /root/reference/src/utils.h, sampling.h and commondef.h (same constants, same
// expression order) so that results track the CPU oracle to libm rounding.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LMC_HD __host__ __device__ __forceinline__
#define LMC_D __device__ __forceinline__
#else  // plain host compiler: the path program (pathfunc.h) is also built for the CPU-side C-ABI checks
#define LMC_HD inline
#define LMC_D inline
#endif

#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstddef>

#include "dtrans.h"
#include "dtrig.h"

namespace lmcd {
#if !defined(__HIPCC__)
using std::isfinite;
#endif

// commondef.h:52-60,70-79
constexpr float c_IsectEpsilon = 5e-4f;
constexpr float c_ShadowEpsilon = 5e-4f;
constexpr float c_CosEpsilon = 1e-4f;
constexpr float c_PI = 3.14159265358979323846f;
constexpr float c_INVPI = 1.0f / c_PI;
constexpr float c_TWOPI = 2.0f * c_PI;
constexpr float c_INVTWOPI = 1.0f / c_TWOPI;
constexpr float c_FOURPI = 4.0f * c_PI;
constexpr float c_INVFOURPI = 1.0f / c_FOURPI;
constexpr float c_PIOVERTWO = 0.5f * c_PI;
constexpr float c_PIOVERFOUR = 0.25f * c_PI;

struct V2 {
    float x, y;
};
struct V3 {
    float x, y, z;
};
LMC_HD V3 mk3(float a, float b, float c) { return V3{a, b, c}; }
LMC_HD V2 mk2(float a, float b) { return V2{a, b}; }
LMC_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
LMC_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
LMC_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
LMC_HD V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
LMC_HD V3 operator*(float s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
LMC_HD V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
LMC_HD V3 cmul(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
LMC_HD float inverse(float x) { return 1.0f / x; }
LMC_HD float square(float x) { return x * x; }
LMC_HD float Dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LMC_HD float LengthSquared(V3 v) { return square(v.x) + square(v.y) + square(v.z); }
LMC_HD float DistanceSquared(V3 a, V3 b) { return square(a.x - b.x) + square(a.y - b.y) + square(a.z - b.z); }
LMC_HD float Length(V3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
LMC_HD V3 Normalize(V3 v) {
    float invLen = inverse(Length(v));
    return v * invLen;
}
LMC_HD V3 Cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
LMC_HD float Luminance(V3 v) { return v.x * 0.212671f + v.y * 0.715160f + v.z * 0.072169f; }
LMC_HD bool IsZero(V3 v) { return v.x == 0.f && v.y == 0.f && v.z == 0.f; }
LMC_HD float MaxCoeff(V3 v) { return fmaxf(v.x, fmaxf(v.y, v.z)); }
LMC_HD bool AllFinite(V3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }
LMC_HD float Clampf(float v, float lb, float ub) {  // std::min(std::max(v, lb), ub) incl. its NaN behaviour
    float m = (v < lb) ? lb : v;
    return (ub < m) ? ub : m;
}
LMC_HD int Clampi(int v, int lb, int ub) { return v < lb ? lb : (v > ub ? ub : v); }

LMC_HD void CoordinateSystem(V3 n, V3 &b1, V3 &b2) {  // utils.h:237-247
    if (n.z < float(-1.0 + 1e-6)) {
        b1 = V3{0.f, -1.f, 0.f};
        b2 = V3{-1.f, 0.f, 0.f};
        return;
    }
    const float a = 1.0f / (1.0f + n.z);
    const float b = -n.x * n.y * a;
    b1 = V3{1.0f - square(n.x) * a, b, -n.x};
    b2 = V3{b, 1.0f - square(n.y) * a, -n.y};
}

LMC_HD float Tent(float s) {  // utils.h:275-281
    if (s < 0.5f) return 1.0f - sqrtf(2.0f * s);
    return sqrtf(2.0f * (s - 0.5f)) - 1.0f;
}

LMC_HD float Modulo1(float a) {  // Modulo(a, 1.0f), utils.h:358-361 (keeps the r + 1 == 1 edge case)
    float r = fmodf(a, 1.0f);
    return (r < 0.0f) ? r + 1.0f : r;
}
LMC_HD int Moduloi(int a, int b) {
    int r = a % b;
    return (r < 0) ? r + b : r;
}

// fastmath.h:364-381 (Mineiro fastapprox): integer reinterpretation + 4 float ops, bit-exact on any IEEE target
LMC_HD float fastlog(float x) {
    uint32_t vi = __builtin_bit_cast(uint32_t, x);
    float mx = __builtin_bit_cast(float, (vi & 0x007FFFFFu) | 0x3f000000u);
    float y = (float)vi;
    y *= 1.1920928955078125e-7f;
    return 0.69314718f * (y - 124.22551499f - 1.498030302f * mx - 1.72587999f / (0.3520887068f + mx));
}

// fastpow = fastpow2(p * fastlog2(x)) (fastmath.h, Mineiro fastapprox), used by the bitmap texture gamma (bitmaptexture.h:94)
LMC_HD float fastlog2(float x) {
    uint32_t vi = __builtin_bit_cast(uint32_t, x);
    float mx = __builtin_bit_cast(float, (vi & 0x007FFFFFu) | 0x3f000000u);
    float y = (float)vi;
    y *= 1.1920928955078125e-7f;
    return y - 124.22551499f - 1.498030302f * mx - 1.72587999f / (0.3520887068f + mx);
}
LMC_HD float fastpow2(float p) {
    float offset = (p < 0) ? 1.0f : 0.0f;
    float clipp = (p < -126) ? -126.0f : p;
    int w = (int)clipp;
    float z = clipp - w + offset;
    uint32_t v = (uint32_t)((1 << 23) * (clipp + 121.2740575f + 27.7280233f / (4.84252568f - z) - 1.49012907f * z));
    return __builtin_bit_cast(float, v);
}
LMC_HD float fastpow(float x, float p) { return fastpow2(p * fastlog2(x)); }

// pow / exp / log of the Phong and rough-dielectric code (phong.cpp:42,109, microfacet.h:17,173) and of the H2MC Gaussian: the
// deterministic float routines of dtrans.h (lpowf / lexpf / llogf), the same source on the device, in the CPU oracle and in the host
// build of the path program -- bit-equal by construction, within 1-2 ulp of the reference's libm calls (whose last bit differs
// between libm builds anyway).  Rounds 1-2 evaluated them in double and rounded once: 26 % of a full-material step.
// utils.h:197-210
LMC_HD V3 Reflect(V3 wi, V3 n) { return (2.0f * Dot(wi, n)) * n - wi; }
LMC_HD V3 Refract(V3 wi, V3 n, float cosThetaT, float eta, float invEta) {
    float eta_ = cosThetaT < 0.0f ? invEta : eta;
    return n * (Dot(wi, n) * eta_ + cosThetaT) - wi * eta_;
}

// Trigonometry: the deterministic float routines of dtrig.h (dsinf / dcosf / dacosf / datan2f) -- one implementation on the device, in the CPU oracle
// and in the host build of the path program, bit-equal by construction, within 1.5 ulp of float64 (the reference calls libm, whose last bit differs
// between libm builds; until round 6 the device libm was called here and the veach-door chains diverged from the oracle's by it).
// Out of line on the device: one copy per kernel image, reached by a call (inlined at every sampling site the libm versions made up a fifth of the
// lean step kernel's code; LMC_TRIG_INLINE=1 for the A/B with these much smaller routines).
#if defined(__HIPCC__) && !defined(LMC_TRIG_INLINE)
#define LMC_OUTLINE __host__ __device__ inline __attribute__((noinline))
#elif defined(__HIPCC__)
#define LMC_OUTLINE __host__ __device__ inline
#else
#define LMC_OUTLINE inline
#endif
LMC_OUTLINE float lsinf(float x) { return dsinf(x); }
LMC_OUTLINE float lcosf(float x) { return dcosf(x); }
LMC_OUTLINE float lacosf(float x) { return dacosf(x); }
LMC_OUTLINE float latan2f(float y, float x) { return datan2f(y, x); }

// sampling.h
LMC_HD V3 SampleSphere(V2 coord, float &jacobian) {
    const float scaledTheta = c_TWOPI * coord.x;
    const float scaledPhi = c_PI * coord.y;
    const float sinPhi = lsinf(scaledPhi), cosPhi = lcosf(scaledPhi);
    V3 dir{sinPhi * lcosf(scaledTheta), sinPhi * lsinf(scaledTheta), cosPhi};
    jacobian = fabsf(sinPhi) * c_TWOPI * c_PI;
    return dir;
}
LMC_HD float patan2(float y, float x) {
    if (y == 0.0f && x == 0.0f) return 0.0f;
    float r = latan2f(y, x);
    if (r < 0.0f) r += c_TWOPI;
    return r;
}
LMC_HD V2 ToSphericalCoord(V3 dir, float &jacobian) {
    float theta = patan2(dir.y, dir.x) * c_INVTWOPI;
    float phi = lacosf(dir.z);
    jacobian = fabsf(lsinf(phi)) * c_TWOPI * c_PI;
    phi *= c_INVPI;
    return V2{theta, phi};
}
LMC_HD V2 SampleConcentricDisc(V2 rnd) {
    float r1 = 2.0f * rnd.x - 1.0f, r2 = 2.0f * rnd.y - 1.0f;
    float phi, r;
    if (r1 == 0 || r2 == 0) {
        r = phi = 0;
    } else if (square(r1) > square(r2)) {
        r = r1;
        phi = c_PIOVERFOUR * (r2 / r1);
    } else {
        r = r2;
        phi = c_PIOVERTWO - (r1 / r2) * c_PIOVERFOUR;
    }
    return V2{r * lcosf(phi), r * lsinf(phi)};
}
LMC_HD V3 SampleCosHemisphere(V2 rnd) {
    float phi = c_TWOPI * rnd.x;
    float tmp = sqrtf(fmaxf(1.0f - rnd.y, 0.0f));
    return V3{lcosf(phi) * tmp, lsinf(phi) * tmp, sqrtf(fmaxf(rnd.y, 0.0f))};
}

// 4x4 row-major helpers (transform.h:48-79)
LMC_HD V3 XformPoint(const float *m, V3 p) {
    float tx = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float ty = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float tz = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float tw = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    float invW = inverse(tw);
    return V3{tx * invW, ty * invW, tz * invW};
}
LMC_HD V3 XformVector(const float *m, V3 v) {
    return V3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z};
}

}  // namespace lmcd
