// ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's LMC hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
// The product (langevin-mcmc_amd/) never includes, links or executes this code.
//
// common.h: scalar/vector helpers restating /root/reference/src/utils.h and commondef.h.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../langevin-mcmc_amd/csrc/device/dtrig.h"  // the deterministic sin / cos / acos / atan2 shared with the device build (like dtrans.h: same source, same bits)

namespace orc {

typedef float Float;

// commondef.h:52-60,70-79
const Float c_IsectEpsilon = Float(5e-4);
const Float c_ShadowEpsilon = Float(5e-4);
const Float c_CosEpsilon = Float(1e-4);
const Float FTRUE = Float(1.0);
const Float FFALSE = Float(0.0);
const Float c_PI = Float(3.14159265358979323846);
const Float c_INVPI = Float(1.0) / c_PI;
const Float c_TWOPI = Float(2.0) * c_PI;
const Float c_INVTWOPI = Float(1.0) / c_TWOPI;
const Float c_FOURPI = Float(4.0) * c_PI;
const Float c_INVFOURPI = Float(1.0) / c_FOURPI;
const Float c_PIOVERTWO = Float(0.5) * c_PI;
const Float c_PIOVERFOUR = Float(0.25) * c_PI;

struct Vector2 {
    Float v[2];
    Vector2() : v{0, 0} {}
    Vector2(Float a, Float b) : v{a, b} {}
    Float &operator[](int i) { return v[i]; }
    const Float &operator[](int i) const { return v[i]; }
};

struct Vector3 {
    Float v[3];
    Vector3() : v{0, 0, 0} {}
    Vector3(Float a, Float b, Float c) : v{a, b, c} {}
    Float &operator[](int i) { return v[i]; }
    const Float &operator[](int i) const { return v[i]; }
    static Vector3 Zero() { return Vector3(0, 0, 0); }
    bool isZero() const { return v[0] == 0 && v[1] == 0 && v[2] == 0; }
    Float sum() const { return v[0] + v[1] + v[2]; }
    Float maxCoeff() const { return std::max(v[0], std::max(v[1], v[2])); }
    Vector3 cwiseProduct(const Vector3 &o) const { return Vector3(v[0] * o.v[0], v[1] * o.v[1], v[2] * o.v[2]); }
    bool allFinite() const { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }
    Vector3 &operator*=(Float s) {
        v[0] *= s, v[1] *= s, v[2] *= s;
        return *this;
    }
    Vector3 &operator+=(const Vector3 &o) {
        v[0] += o.v[0], v[1] += o.v[1], v[2] += o.v[2];
        return *this;
    }
};
inline Vector3 operator+(const Vector3 &a, const Vector3 &b) { return Vector3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vector3 operator-(const Vector3 &a, const Vector3 &b) { return Vector3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline Vector3 operator-(const Vector3 &a) { return Vector3(-a[0], -a[1], -a[2]); }
inline Vector3 operator*(const Vector3 &a, Float s) { return Vector3(a[0] * s, a[1] * s, a[2] * s); }
inline Vector3 operator*(Float s, const Vector3 &a) { return Vector3(a[0] * s, a[1] * s, a[2] * s); }
inline Vector3 operator/(const Vector3 &a, Float s) { return Vector3(a[0] / s, a[1] / s, a[2] / s); }

inline Float inverse(Float x) { return Float(1.0) / x; }
inline Float square(Float x) { return x * x; }
inline Float Dot(const Vector3 &a, const Vector3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline Float LengthSquared(const Vector3 &v) { return square(v[0]) + square(v[1]) + square(v[2]); }
inline Float DistanceSquared(const Vector3 &a, const Vector3 &b) { return square(a[0] - b[0]) + square(a[1] - b[1]) + square(a[2] - b[2]); }
inline Float Length(const Vector3 &v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
inline Float Distance(const Vector3 &a, const Vector3 &b) { return Length(a - b); }
inline Vector3 Normalize(const Vector3 &v) {
    Float invLen = inverse(Length(v));
    return v * invLen;
}
inline Vector3 Cross(const Vector3 &a, const Vector3 &b) {
    return Vector3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
inline Vector3 Reflect(const Vector3 &wi, const Vector3 &n) { return (Float(2.0) * Dot(wi, n)) * n - wi; }
inline Vector3 Refract(const Vector3 &wi, const Vector3 &n, Float cosThetaT, Float eta, Float invEta) {
    Float eta_ = cosThetaT < Float(0.0) ? invEta : eta;
    return n * (Dot(wi, n) * eta_ + cosThetaT) - wi * eta_;
}
inline Float Luminance(const Vector3 &v) { return v[0] * Float(0.212671) + v[1] * Float(0.715160) + v[2] * Float(0.072169); }

inline void CoordinateSystem(const Vector3 &n, Vector3 &b1, Vector3 &b2) {  // utils.h:237-247
    if (n[2] < Float(-1.0 + 1e-6)) {
        b1 = Vector3(Float(0.0), Float(-1.0), Float(0.0));
        b2 = Vector3(Float(-1.0), Float(0.0), Float(0.0));
        return;
    }
    const Float a = Float(1.0) / (Float(1.0) + n[2]);
    const Float b = -n[0] * n[1] * a;
    b1 = Vector3(Float(1.0) - square(n[0]) * a, b, -n[0]);
    b2 = Vector3(b, Float(1.0) - square(n[1]) * a, -n[1]);
}

inline Float Tent(Float sample) {  // utils.h:275-281
    if (sample < Float(0.5)) return (Float(1.0) - std::sqrt(Float(2.0) * sample));
    return std::sqrt(Float(2.0) * (sample - Float(0.5))) - Float(1.0);
}

template <typename T>
T Clamp(const T v, const T lb, const T ub) {
    return std::min(std::max(v, lb), ub);
}

inline int32_t Modulo(int32_t a, int32_t b) {
    int32_t r = a % b;
    return (r < 0) ? r + b : r;
}
inline Float Modulo(Float a, Float b) {  // utils.h:358-361 (keeps the r+b == b edge case)
    Float r = std::fmod(a, b);
    return (r < 0.0) ? r + b : r;
}

// fastmath.h:364-381 (Mineiro fastapprox), used by mala.cpp:34,49 and gaussian.cpp:21
inline float fastlog2(float x) {
    union { float f; uint32_t i; } vx = {x};
    union { uint32_t i; float f; } mx = {(vx.i & 0x007FFFFF) | 0x3f000000};
    float y = vx.i;
    y *= 1.1920928955078125e-7f;
    return y - 124.22551499f - 1.498030302f * mx.f - 1.72587999f / (0.3520887068f + mx.f);
}
inline float fastlog(float x) { return 0.69314718f * fastlog2(x); }

// sampling.h
inline Vector3 SampleSphere(const Vector2 coord, Float &jacobian) {
    const Float scaledTheta = c_TWOPI * coord[0];
    const Float scaledPhi = c_PI * coord[1];
    const Float sinPhi = lmcd::dsinf(scaledPhi);
    const Float cosPhi = lmcd::dcosf(scaledPhi);
    Vector3 dir(sinPhi * lmcd::dcosf(scaledTheta), sinPhi * lmcd::dsinf(scaledTheta), cosPhi);
    jacobian = std::fabs(sinPhi) * c_TWOPI * c_PI;
    return dir;
}
inline Float patan2(Float y, Float x) {
    if (y == Float(0.0) && x == Float(0.0)) return Float(0.0);
    Float result = lmcd::datan2f(y, x);
    if (result < 0.0) result += c_TWOPI;
    return result;
}
inline Vector2 ToSphericalCoord(const Vector3 &dir, Float &jacobian) {
    Float theta = patan2(dir[1], dir[0]) * c_INVTWOPI;
    Float phi = lmcd::dacosf(dir[2]);
    jacobian = std::fabs(lmcd::dsinf(phi)) * c_TWOPI * c_PI;
    phi *= c_INVPI;
    return Vector2(theta, phi);
}
inline Vector2 SampleConcentricDisc(const Vector2 rndParam) {
    Float r1 = Float(2.0) * rndParam[0] - Float(1.0);
    Float r2 = Float(2.0) * rndParam[1] - Float(1.0);
    Float phi, r;
    if (r1 == 0 || r2 == 0) {
        r = phi = 0;
    } else if (square(r1) > square(r2)) {
        r = r1;
        phi = c_PIOVERFOUR * (r2 / r1);
    } else {
        r = r2;
        phi = c_PIOVERTWO - (r1 / r2) * c_PIOVERFOUR;
    }
    return Vector2(r * lmcd::dcosf(phi), r * lmcd::dsinf(phi));
}
inline Vector3 SampleCosHemisphere(const Vector2 rndParam) {  // ADEpsilon<Float>() == 0 (utils.h:440-443)
    Float phi = c_TWOPI * rndParam[0];
    Float tmp = std::sqrt(std::fmax(Float(1.0) - rndParam[1], Float(0.0)));
    return Vector3(lmcd::dcosf(phi) * tmp, lmcd::dsinf(phi) * tmp, std::sqrt(std::fmax(rndParam[1], Float(0.0))));
}

}  // namespace orc
