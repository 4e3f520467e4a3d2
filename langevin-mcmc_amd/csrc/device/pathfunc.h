// The path program of the reference's plugin ABI, as ONE generic function:
//   logLum = log Luminance(contribution of the (c,l) technique)   as a function of primary[1..2L]
// and its gradient by forward-mode dual numbers.  It restates, for PathFuncMode::Static, the program that
// RegisterPathFuncBidirMALA builds with chad (/root/reference/src/path.cpp:3664-3911) out of the AD twins
// path.cpp:2789-3417, camera.cpp:53-65, trianglemesh.cpp:55-77,291-365,367-473, bsdf.cpp:13-171,
// lambertian.cpp:95-151, light.cpp:12-324, envlight.cpp:250-400, arealight.cpp:106-208,
// pointlight.cpp:57-116, sampling.h, utils.h -- including their deviations from the scalar sampler
// (no rejection tests, ADEpsilon, the PointLight/!IsDelta swap in EmitFromLight,
// frozen env-map texel).  Inputs use the reference's serialized layout: `scene[38]`,
// `primary[2L+1]`, `vertParams[V]`; the latter is read through an accessor so the same code runs on a
// contiguous host buffer (C-ABI symbols, CPU tests) and on a strided per-thread slice in HBM (kernels).
// Host- and device-compilable; no reference code is generated or copied: this is hand-written.
#pragma once
#include "dmath.h"

// The per-vertex building blocks of the path program.  Inlined into the program for the value / gradient types; a translation
// unit that instantiates the second-order (nested dual) type defines LMC_PF_OUTLINE first: one out-of-line copy per block keeps
// the compile time of that instantiation to minutes (inlined at every call site it took 8 min per Hessian width).
#if defined(LMC_PF_OUTLINE) && defined(__HIPCC__)
#ifndef LMC_PF_ATTR
#define LMC_PF_ATTR  // build experiments: extra attributes of the out-of-line blocks (e.g. a register budget)
#endif
#define LMC_PF __host__ __device__ inline __attribute__((noinline)) LMC_PF_ATTR
#else
#define LMC_PF LMC_HD
#endif

// LMC_PF_CONTRACT (h2hess.hip): a * b + c of the dual-number arithmetic may fuse.  The step's Hessian has no bit-level contract with the
// oracle (the oracle takes its Hessians from the reference's own programs; the tests compare within 1e-2), and a second-order product
// is 25 multiplications + 16 additions without fusing, 25 instructions with.
#ifdef LMC_PF_CONTRACT
#pragma clang fp contract(fast)
#endif

namespace lmcd {

struct ContigIn {
    const float *p;
    LMC_HD float operator[](int k) const { return p[k]; }
};
struct StridedIn {
    const float *p;
    size_t stride;
    LMC_HD float operator[](int k) const { return p[(size_t)k * stride]; }
};

// ------------------------------------------------------------------------------------------ dual numbers
// Forward-mode AD value: v + sum_i d[i] eps_i.  The scalar type S is float for gradients (Dual<N>) and itself a dual
// number for second derivatives: DualS<1, Dual<N>> carries, next to value and gradient, the derivative of both along one
// more direction, i.e. one row of the Hessian per evaluation (the reference's Hessian programs are built the same way:
// one directional pass per row over the reverse sweep, chad.cpp:333-544).
template <int N, class S = float>
struct DualS {
    S v;
    S d[N];
};
#if defined(LMC_PF_VEC2) && defined(__HIP_DEVICE_COMPILE__)
// LMC_PF_VEC2 (h2hess.hip): the two derivative components of a Dual<2> live in ONE two-wide vector value, and its arithmetic (below, after the
// generic operators) is written on whole vectors.  gfx950 has packed FP32 instructions (v_pk_mul / v_pk_add / v_pk_fma_f32: two lanes' worth per
// issue); from the element-wise loops the compiler's SLP pass did form them, but out of whatever scalars happened to pair up, and paid for it in
// v_mov_b32 that assemble the operand pairs -- 22 k of the Hessian kernel's 80 k instructions (hipcc -S, profiles/r05_ax_*).  Same arithmetic per
// component, same order of operations.
typedef float lmc_f2 __attribute__((ext_vector_type(2)));
template <>
struct DualS<2, float> {
    float v;
    lmc_f2 d;
};
#endif
template <int N>
using Dual = DualS<N, float>;

template <class T> struct Lift;  // constant -> T
template <> struct Lift<float> { static LMC_HD float Of(float v) { return v; } };
template <int N, class S> struct Lift<DualS<N, S>> {
    static LMC_HD DualS<N, S> Of(float v) {
        DualS<N, S> r;
        r.v = Lift<S>::Of(v);
        for (int i = 0; i < N; i++) r.d[i] = Lift<S>::Of(0.f);
        return r;
    }
};
template <int N>
LMC_HD Dual<N> MakeDual(float v) {
    return Lift<Dual<N>>::Of(v);
}
LMC_HD float Val(float x) { return x; }
template <int N, class S> LMC_HD float Val(const DualS<N, S> &x) { return Val(x.v); }

#define LMC_DUAL_T template <int N, class S> LMC_HD DualS<N, S>
LMC_DUAL_T Chain1(const DualS<N, S> &a, const S &v, const S &dv) {  // f(a) with f' = dv
    DualS<N, S> r;
    r.v = v;
    for (int i = 0; i < N; i++) r.d[i] = dv * a.d[i];
    return r;
}
LMC_DUAL_T operator+(const DualS<N, S> &a, const DualS<N, S> &b) {
    DualS<N, S> r;
    r.v = a.v + b.v;
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] + b.d[i];
    return r;
}
LMC_DUAL_T operator-(const DualS<N, S> &a, const DualS<N, S> &b) {
    DualS<N, S> r;
    r.v = a.v - b.v;
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] - b.d[i];
    return r;
}
LMC_DUAL_T operator*(const DualS<N, S> &a, const DualS<N, S> &b) {
    DualS<N, S> r;
    r.v = a.v * b.v;
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
LMC_DUAL_T operator/(const DualS<N, S> &a, const DualS<N, S> &b) {
    DualS<N, S> r;
    S inv = 1.0f / b.v;
    r.v = a.v * inv;
    for (int i = 0; i < N; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
LMC_DUAL_T operator-(const DualS<N, S> &a) {
    DualS<N, S> r;
    r.v = -a.v;
    for (int i = 0; i < N; i++) r.d[i] = -a.d[i];
    return r;
}
LMC_DUAL_T operator+(const DualS<N, S> &a, float b) { DualS<N, S> r = a; r.v = r.v + b; return r; }
LMC_DUAL_T operator+(float b, const DualS<N, S> &a) { DualS<N, S> r = a; r.v = r.v + b; return r; }
LMC_DUAL_T operator-(const DualS<N, S> &a, float b) { DualS<N, S> r = a; r.v = r.v - b; return r; }
LMC_DUAL_T operator-(float b, const DualS<N, S> &a) { DualS<N, S> r = -a; r.v = r.v + b; return r; }
LMC_DUAL_T operator*(const DualS<N, S> &a, float b) {
    DualS<N, S> r;
    r.v = a.v * b;
    for (int i = 0; i < N; i++) r.d[i] = b * a.d[i];
    return r;
}
LMC_DUAL_T operator*(float b, const DualS<N, S> &a) { return a * b; }
LMC_DUAL_T operator/(const DualS<N, S> &a, float b) { float inv = 1.0f / b; return a * inv; }
LMC_DUAL_T operator/(float b, const DualS<N, S> &a) { S inv = 1.0f / a.v; return Chain1(a, S(b * inv), S(-b * inv * inv)); }
template <int N, class S> LMC_HD bool operator<(const DualS<N, S> &a, float b) { return Val(a) < b; }
template <int N, class S> LMC_HD bool operator>(const DualS<N, S> &a, float b) { return Val(a) > b; }
template <int N, class S> LMC_HD bool operator<(const DualS<N, S> &a, const DualS<N, S> &b) { return Val(a) < Val(b); }
template <int N, class S> LMC_HD bool operator>(const DualS<N, S> &a, const DualS<N, S> &b) { return Val(a) > Val(b); }

#if defined(LMC_PF_VEC2) && defined(__HIP_DEVICE_COMPILE__)
// Dual<2> on whole vectors (non-template overloads: preferred over the generic element-wise templates above)
typedef DualS<2, float> D2v;
LMC_HD D2v Chain1(const D2v &a, const float &v, const float &dv) { D2v r; r.v = v; r.d = a.d * dv; return r; }
LMC_HD D2v operator+(const D2v &a, const D2v &b) { D2v r; r.v = a.v + b.v; r.d = a.d + b.d; return r; }
LMC_HD D2v operator-(const D2v &a, const D2v &b) { D2v r; r.v = a.v - b.v; r.d = a.d - b.d; return r; }
LMC_HD D2v operator*(const D2v &a, const D2v &b) { D2v r; r.v = a.v * b.v; r.d = a.d * b.v + b.d * a.v; return r; }
LMC_HD D2v operator/(const D2v &a, const D2v &b) {
    D2v r;
    const float inv = 1.0f / b.v;
    r.v = a.v * inv;
    r.d = (a.d - b.d * r.v) * inv;
    return r;
}
LMC_HD D2v operator-(const D2v &a) { D2v r; r.v = -a.v; r.d = -a.d; return r; }
LMC_HD D2v operator*(const D2v &a, float b) { D2v r; r.v = a.v * b; r.d = a.d * b; return r; }
LMC_HD D2v operator*(float b, const D2v &a) { return a * b; }
#endif
LMC_HD float Detach(float x) { return x; }
template <int N, class S> LMC_HD DualS<N, S> Detach(const DualS<N, S> &x) { return Lift<DualS<N, S>>::Of(Val(x)); }
// chad's fabs / fmax are conditional expressions that pass their operand through (chad.h:1226-1244), and the code
// chad generates for a passed-through node ASSIGNS its adjoint (`_accX = _accR`) instead of accumulating into it: every
// contribution the reverse sweep had already gathered for X -- i.e. from the uses of X that come LATER in program order
// -- is dropped.  The reference's derivative programs therefore are not the true gradient wherever a named value goes
// through fabs()/fmax() and is used again afterwards (rough dielectric: cosWi, cos(H,wi), cosWo).  FabsW / FmaxW
// reproduce this in forward mode: the operand's later uses see a constant.
template <class T> LMC_HD T FabsW(T &x);
template <class T> LMC_HD T FmaxW(T &x, float b);

// LMC_PF_FASTMATH (h2hess.hip, together with LMC_PF_CONTRACT): sin / cos / exp / log / pow through the hardware's approximate instructions
// (v_sin_f32, v_cos_f32, v_exp_f32, v_log_f32: ~1e-6 absolute on the arguments that occur here -- angles within a few turns, BSDF
// exponents) instead of the correctly rounded device libm / the float-float routines of dtrans.h.  Only the step's Hessian launch
// is built this way: it has no bit-level contract (see LMC_PF_CONTRACT), and a second-order value evaluates each of them three times.
#if defined(LMC_PF_FASTMATH) && defined(__HIP_DEVICE_COMPILE__)
#define LMC_PF_FAST 1
#else
#define LMC_PF_FAST 0
#endif
LMC_HD float Sqrt(float x) { return sqrtf(x); }
#if LMC_PF_FAST
LMC_HD float Sin(float x) { return __sinf(x); }
LMC_HD float Cos(float x) { return __cosf(x); }
#else
#ifdef LMC_PF_LIBM_TRIG  // debugging aid: libm's versions, to tell an ill-conditioned state from a defect of dtrig.h
LMC_HD float Sin(float x) { return sinf(x); }
LMC_HD float Cos(float x) { return cosf(x); }
#else
LMC_HD float Sin(float x) { return dsinf(x); }  // dtrig.h: the same bits on the device and in the host build
LMC_HD float Cos(float x) { return dcosf(x); }
#endif
#endif
#ifdef LMC_PF_LIBM_TRIG
LMC_HD float Acos(float x) { return acosf(x); }
LMC_HD float Atan2(float y, float x) { return atan2f(y, x); }
#else
LMC_HD float Acos(float x) { return dacosf(x); }
LMC_HD float Atan2(float y, float x) { return datan2f(y, x); }
#endif
LMC_HD float Fabs(float x) { return fabsf(x); }
#if LMC_PF_FAST
LMC_HD float Log(float x) { return __logf(x); }
LMC_HD float Exp(float x) { return __expf(x); }
#else
LMC_HD float Log(float x) { return logf(x); }
LMC_HD float Exp(float x) { return expf(x); }
#endif
LMC_HD float Fmax(float a, float b) { return fmaxf(a, b); }
#if LMC_PF_FAST
LMC_HD float Pow(float a, float e) { return a > 0.0f ? __expf(e * __logf(a)) : lpowf(a, e); }  // the corner cases (zero / negative base) stay with the exact routine
LMC_HD float PowRaw(float a, float e) { return a > 0.0f ? __expf(e * __logf(a)) : lpowf(a, e); }
LMC_HD float ExpD(float x) { return __expf(x); }
LMC_HD float LogD(float x) { return x > 0.0f ? __logf(x) : llogf(x); }
#else
LMC_HD float Pow(float a, float e) { return lpowf(a, e); }
LMC_HD float PowRaw(float a, float e) { return lpowf(a, e); }  // the derivative factor of Pow (chad.h:727 emits pow(x, e - 1))
LMC_HD float ExpD(float x) { return lexpf(x); }
LMC_HD float LogD(float x) { return llogf(x); }
#endif
LMC_DUAL_T Sqrt(const DualS<N, S> &a) { S s = Sqrt(a.v); return Chain1(a, s, S(0.5f / s)); }
LMC_DUAL_T Sin(const DualS<N, S> &a) { return Chain1(a, S(Sin(a.v)), S(Cos(a.v))); }
LMC_DUAL_T Cos(const DualS<N, S> &a) { return Chain1(a, S(Cos(a.v)), S(-Sin(a.v))); }
LMC_DUAL_T Acos(const DualS<N, S> &a) { return Chain1(a, S(Acos(a.v)), S(-1.0f / Sqrt(1.0f - a.v * a.v))); }
LMC_DUAL_T Atan2(const DualS<N, S> &y, const DualS<N, S> &x) {
    DualS<N, S> r;
    r.v = Atan2(y.v, x.v);
    S inv = 1.0f / (x.v * x.v + y.v * y.v);
    for (int i = 0; i < N; i++) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * inv;
    return r;
}
#if defined(LMC_PF_VEC2) && defined(__HIP_DEVICE_COMPILE__)
LMC_HD D2v Atan2(const D2v &y, const D2v &x) {
    D2v r;
    r.v = Atan2(y.v, x.v);
    const float inv = 1.0f / (x.v * x.v + y.v * y.v);
    r.d = (y.d * x.v - x.d * y.v) * inv;
    return r;
}
#endif
LMC_DUAL_T Fabs(const DualS<N, S> &a) { return Val(a) >= 0.f ? a : -a; }  // chad.h:1226-1234: x >= 0 ? x : -x
LMC_DUAL_T Log(const DualS<N, S> &a) { return Chain1(a, S(Log(a.v)), S(1.0f / a.v)); }
LMC_DUAL_T Exp(const DualS<N, S> &a) { S e = Exp(a.v); return Chain1(a, e, e); }
LMC_DUAL_T PowRaw(const DualS<N, S> &a, float e) { return Chain1(a, S(PowRaw(a.v, e)), S(e * PowRaw(a.v, e - 1.0f))); }
LMC_DUAL_T Pow(const DualS<N, S> &a, float e) { return Chain1(a, S(Pow(a.v, e)), S(e * PowRaw(a.v, e - 1.0f))); }  // chad.h:727, exponent constant
LMC_DUAL_T ExpD(const DualS<N, S> &a) { S e = ExpD(a.v); return Chain1(a, e, e); }
LMC_DUAL_T LogD(const DualS<N, S> &a) { return Chain1(a, S(LogD(a.v)), S(1.0f / a.v)); }
LMC_DUAL_T Fmax(const DualS<N, S> &a, float b) { return Val(a) >= b ? a : Lift<DualS<N, S>>::Of(b); }  // chad.h:1236-1244: a >= b ? a : b

// Second order: the reference's Hessian programs are reverse over forward (chad.cpp:333-544: the kernel takes a direction d,
// returns g = d . grad f by a forward sweep -- exact -- and h = the reverse-mode gradient of g, emitted by the same reverse
// emitter with the same overwrite).  In the nested type DualS<1, Dual<N>> the outer level is that forward direction and the
// inner Dual<N> plays the reverse sweep: the operand's later uses lose their INNER derivatives (at both outer levels) and
// keep the outer one.  Gradient = outer derivative of the value, Hessian row = its inner gradient (PathFuncHessN).
LMC_HD float DetachW(float x) { return x; }
template <int N> LMC_HD Dual<N> DetachW(const Dual<N> &x) { return Lift<Dual<N>>::Of(x.v); }
template <int N, int M> LMC_HD DualS<N, Dual<M>> DetachW(const DualS<N, Dual<M>> &x) {
    DualS<N, Dual<M>> r;
    r.v = Lift<Dual<M>>::Of(x.v.v);
    for (int i = 0; i < N; i++) r.d[i] = Lift<Dual<M>>::Of(x.d[i].v);
    return r;
}
template <class T> LMC_HD T FabsW(T &x) {
    T r = Fabs(x);
    if (Val(x) >= 0.0f) x = DetachW(x);
    return r;
}
template <class T> LMC_HD T FmaxW(T &x, float b) {
    T r = Fmax(x, b);
    if (Val(x) >= b) x = DetachW(x);
    return r;
}

// ------------------------------------------------------------------------------------------ small vectors
template <class T>
struct V3T {
    T x, y, z;
};
template <class T> LMC_HD V3T<T> operator+(const V3T<T> &a, const V3T<T> &b) { return V3T<T>{a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> LMC_HD V3T<T> operator-(const V3T<T> &a, const V3T<T> &b) { return V3T<T>{a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> LMC_HD V3T<T> operator-(const V3T<T> &a) { return V3T<T>{-a.x, -a.y, -a.z}; }
template <class T> LMC_HD V3T<T> operator*(const V3T<T> &a, const T &s) { return V3T<T>{a.x * s, a.y * s, a.z * s}; }
template <class T> LMC_HD V3T<T> operator*(const T &s, const V3T<T> &a) { return V3T<T>{a.x * s, a.y * s, a.z * s}; }
template <class T> LMC_HD V3T<T> Cmul(const V3T<T> &a, const V3T<T> &b) { return V3T<T>{a.x * b.x, a.y * b.y, a.z * b.z}; }
template <class T> LMC_HD T DotT(const V3T<T> &a, const V3T<T> &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> LMC_HD T LenSqT(const V3T<T> &a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
template <class T> LMC_HD T DistSqT(const V3T<T> &a, const V3T<T> &b) { return LenSqT(a - b); }
template <class T> LMC_HD V3T<T> NormalizeT(const V3T<T> &a) {
    T invLen = 1.0f / Sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return a * invLen;
}
template <class T> LMC_HD V3T<T> CrossT(const V3T<T> &a, const V3T<T> &b) {
    return V3T<T>{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class T> LMC_HD V3T<T> DetachW3(const V3T<T> &a) { return V3T<T>{DetachW(a.x), DetachW(a.y), DetachW(a.z)}; }
template <class T> LMC_HD T LumT(const V3T<T> &v) { return v.x * 0.212671f + v.y * 0.715160f + v.z * 0.072169f; }

template <class T> LMC_HD V3T<T> C3(float a, float b, float c) { return V3T<T>{Lift<T>::Of(a), Lift<T>::Of(b), Lift<T>::Of(c)}; }
template <class T> LMC_HD T MISq(const T &p) { return p * p; }

template <class T>
struct PState {  // ADBidirPathState, path.cpp:2789-2797 (lensContrib only feeds PathFuncMode::Lens)
    V3T<T> position, shadingNormal, geomNormal, wi;
    T accMISWPrev, accMISWThis;
    V3T<T> throughput;
};

// rotation part of ToMatrix4x4(AnimatedTransform) for isStatic (animatedtransform.cpp:57-59, quaternion.h:13-40)
struct Rot3 {
    float m[3][3];
    float t[3];
};
template <class In>
LMC_HD Rot3 ReadXform(const In &b, int off) {  // 15-float block [isMoving, t0(3), t1(3), q0(4), q1(4)]
    Rot3 r;
    r.t[0] = b[off + 1], r.t[1] = b[off + 2], r.t[2] = b[off + 3];
    float q0 = b[off + 7], q1 = b[off + 8], q2 = b[off + 9], q3 = b[off + 10];
    float xx = q0 * q0, yy = q1 * q1, zz = q2 * q2, xy = q0 * q1, xz = q0 * q2, yz = q1 * q2, wx = q0 * q3, wy = q1 * q3, wz = q2 * q3;
    r.m[0][0] = 1.f - 2.f * (yy + zz), r.m[0][1] = 2.f * (xy - wz), r.m[0][2] = 2.f * (xz + wy);
    r.m[1][0] = 2.f * (xy + wz), r.m[1][1] = 1.f - 2.f * (xx + zz), r.m[1][2] = 2.f * (yz - wx);
    r.m[2][0] = 2.f * (xz - wy), r.m[2][1] = 2.f * (yz + wx), r.m[2][2] = 1.f - 2.f * (xx + yy);
    return r;
}
template <class T>
LMC_HD V3T<T> RotVec(const Rot3 &r, const V3T<T> &v) {
    return V3T<T>{v.x * r.m[0][0] + v.y * r.m[0][1] + v.z * r.m[0][2], v.x * r.m[1][0] + v.y * r.m[1][1] + v.z * r.m[1][2],
                  v.x * r.m[2][0] + v.y * r.m[2][1] + v.z * r.m[2][2]};
}

struct SceneBlk {  // scene.cpp:164-169
    float useLightCoord;
    float s2c[4][4];  // sampleToCam, row-major here
    Rot3 camToWorld;
    float pixelCount, camDist, bsCenter[3], bsRadius;
};
LMC_HD SceneBlk ReadScene(const float *s) {
    SceneBlk b;
    b.useLightCoord = s[0];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) b.s2c[j][i] = s[1 + i * 4 + j];  // column-major stream (utils.h:331-348)
    b.camToWorld = ReadXform(ContigIn{s}, 17);
    b.pixelCount = s[32], b.camDist = s[33];
    b.bsCenter[0] = s[34], b.bsCenter[1] = s[35], b.bsCenter[2] = s[36], b.bsRadius = s[37];
    return b;
}

// camera.cpp:53-65 (isStatic)
template <class T>
LMC_HD void SamplePrimaryT(const SceneBlk &sc, const T &sx, const T &sy, V3T<T> &org, V3T<T> &dir) {
    T tx = sx * sc.s2c[0][0] + sy * sc.s2c[0][1] + sc.s2c[0][3];
    T ty = sx * sc.s2c[1][0] + sy * sc.s2c[1][1] + sc.s2c[1][3];
    T tz = sx * sc.s2c[2][0] + sy * sc.s2c[2][1] + sc.s2c[2][3];
    T tw = sx * sc.s2c[3][0] + sy * sc.s2c[3][1] + sc.s2c[3][3];
    T invW = 1.0f / tw;
    V3T<T> o{tx * invW, ty * invW, tz * invW};
    V3T<T> d = NormalizeT(o);
    org = C3<T>(sc.camToWorld.t[0], sc.camToWorld.t[1], sc.camToWorld.t[2]);  // XformPoint(toWorld, 0) = translation (w = 1)
    dir = RotVec(sc.camToWorld, d);
}

// trianglemesh.cpp:55-77 + IntersectTriangleMesh :367-473 (isStatic); consumes the 46-float shape slot
template <class T, class In>
LMC_PF void IntersectT(const In &b, int off, const V3T<T> &org, const V3T<T> &dir, PState<T> &ps, T &st0, T &st1) {
    // [type, isMoving, p0, e1, e2, n0, n1, n2, (same at t=1), noST, st0, st1, st2, invTotalArea]
    V3T<T> p0 = C3<T>(b[off + 2], b[off + 3], b[off + 4]), e1 = C3<T>(b[off + 5], b[off + 6], b[off + 7]), e2 = C3<T>(b[off + 8], b[off + 9], b[off + 10]);
    V3T<T> n0 = C3<T>(b[off + 11], b[off + 12], b[off + 13]), n1 = C3<T>(b[off + 14], b[off + 15], b[off + 16]), n2 = C3<T>(b[off + 17], b[off + 18], b[off + 19]);
    ps.geomNormal = NormalizeT(CrossT(e1, e2));
    V3T<T> s1 = CrossT(dir, e2);
    T divisor = DotT(s1, e1);
    T invDivisor = 1.0f / divisor;
    V3T<T> s = org - p0;
    T u = DotT(s, s1) * invDivisor;
    V3T<T> s2 = CrossT(s, e1);
    T v = DotT(dir, s2) * invDivisor;
    T t = DotT(e2, s2) * invDivisor;
    T w = 1.0f - u - v;
    ps.position = org + dir * t;
    ps.shadingNormal = NormalizeT(n0 * w + n1 * u + n2 * v);
    const float noST = b[off + 38];
    if (noST == 0.0f) {  // AD reads the flag as "hasST" and uses barycentrics when it is 0 (trianglemesh.cpp:462); st is not used afterwards
        st0 = u, st1 = v;
    } else {
        st0 = (1.0f - u - v) * b[off + 39] + u * b[off + 41] + v * b[off + 43];
        st1 = (1.0f - u - v) * b[off + 40] + u * b[off + 42] + v * b[off + 44];
    }
}

// SampleDirect on a triangle, trianglemesh.cpp:291-305 with ADEpsilon = 1e-6
template <class T, class In>
LMC_PF void SampleShapeT(const In &b, int off, const T &r0, const T &r1, V3T<T> &pos, V3T<T> &nrm, float &pdf) {
    V3T<T> p0 = C3<T>(b[off + 2], b[off + 3], b[off + 4]), e1 = C3<T>(b[off + 5], b[off + 6], b[off + 7]), e2 = C3<T>(b[off + 8], b[off + 9], b[off + 10]);
    V3T<T> n0 = C3<T>(b[off + 11], b[off + 12], b[off + 13]), n1 = C3<T>(b[off + 14], b[off + 15], b[off + 16]), n2 = C3<T>(b[off + 17], b[off + 18], b[off + 19]);
    T a = Sqrt((1.0f + 1e-6f) - r0);
    T b1 = 1.0f - a;
    T b2 = a * r1;
    pos = p0 + e1 * b1 + e2 * b2;
    nrm = NormalizeT(n0 * (1.0f - b1 - b2) + n1 * b1 + n2 * b2);
    pdf = b[off + 45];
}

template <class T>
LMC_HD void CoordinateSystemT(const V3T<T> &n, V3T<T> &b1, V3T<T> &b2) {  // utils.h:249-273
    if (Val(n.z) < float(-1.0 + 1e-6)) {
        b1 = C3<T>(0.f, -1.f, 0.f);
        b2 = C3<T>(-1.f, 0.f, 0.f);
        return;
    }
    T a = 1.0f / (1.0f + n.z);
    T b = -n.x * n.y * a;
    // `b` leaves the reference's conditional through TWO outputs (b1.y and b2.x, utils.h:268-271); chad's reverse sweep assigns
    // the adjoint of each output to the expression it passes through (chad.cpp:283-284), so the second assignment replaces the
    // first: b1.y reaches the derivative programs as a constant.
    b1 = V3T<T>{1.0f - n.x * n.x * a, DetachW(b), -n.x};
    b2 = V3T<T>{b, 1.0f - n.y * n.y * a, -n.y};
}

template <class T>
LMC_HD V3T<T> SampleCosHemisphereT(const T &r0, const T &r1) {  // sampling.h:104-110 with ADEpsilon
    T phi = c_TWOPI * r0;
    T tmp = Sqrt(Fmax(1.0f - r1, 1e-6f));
    return V3T<T>{Cos(phi) * tmp, Sin(phi) * tmp, Sqrt(Fmax(r1, 1e-6f))};
}
template <class T>
LMC_HD V3T<T> SampleSphereT(const T &c0, const T &c1, T &jacobian) {  // sampling.h:6-16
    T scaledTheta = c_TWOPI * c0;
    T scaledPhi = c_PI * c1;
    T sinPhi = Sin(scaledPhi), cosPhi = Cos(scaledPhi);
    jacobian = Fabs(sinPhi) * (c_TWOPI * c_PI);
    return V3T<T>{sinPhi * Cos(scaledTheta), sinPhi * Sin(scaledTheta), cosPhi};
}
template <class T>
LMC_HD T TentT(const T &s) {  // utils.h:283-291
    if (Val(s) < 0.5f) return 1.0f - Sqrt(2.0f * s);
    return Sqrt(2.0f * (s - 0.5f)) - 1.0f;
}

template <class T>
LMC_HD T ShadingNormalCorrectionAdj(const V3T<T> &wi, const PState<T> &ps, const V3T<T> &wo) {  // path.cpp:56-70, adjoint = true
    T cosWi = DotT(ps.shadingNormal, wi), cosWo = DotT(ps.shadingNormal, wo);
    T wiDotGeoN = DotT(ps.geomNormal, wi), woDotGeoN = DotT(ps.geomNormal, wo);
    return Fabs((woDotGeoN * cosWi) / (wiDotGeoN * cosWo));
}

// ------------------------------------------------------------------------------------------ BSDFs (10-float slot)
template <class T> LMC_HD V3T<T> ReflectT(const V3T<T> &wi, const V3T<T> &n) { return n * (2.0f * DotT(wi, n)) - wi; }  // utils.h:197-200

// ---- microfacet.h AD versions
template <class T>
LMC_HD T BeckmennDT(const V3T<T> &localH, const T &alphaU, const T &alphaV) {  // microfacet.h:6-19
    T cosTheta2 = localH.z * localH.z;
    T beckmannExponent = ((localH.x * localH.x) / (alphaU * alphaU) + (localH.y * localH.y) / (alphaV * alphaV)) / cosTheta2;
    return ExpD(-beckmannExponent) / (c_PI * alphaU * alphaV * (cosTheta2 * cosTheta2));
}
template <class T>
LMC_HD T BeckmennG1T(float alpha, const T &cosTheta) {  // microfacet.h:41-63 (note the 1 + 1e-6)
    T tanTheta = Sqrt(Fabs((1.0f + 1e-6f) - cosTheta * cosTheta)) / cosTheta;
    if (Val(tanTheta) <= 0.0f) return Lift<T>::Of(1.0f);
    T a = 1.0f / (alpha * tanTheta);
    if (Val(a) >= 1.6f) return Lift<T>::Of(1.0f);
    T aSqr = a * a;
    return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
}
template <class T>
LMC_HD T FresnelDielectricExtT(T &cosThetaI_, T &cosThetaT_, float eta, float invEta) {  // microfacet.h:117-144 (its fabs wipes the operand)
    const float scale = Val(cosThetaI_) > 0.0f ? invEta : eta;
    T cosThetaTSqr = 1.0f - (1.0f - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (Val(cosThetaTSqr) <= 0.0f) {
        cosThetaT_ = Lift<T>::Of(0.0f);
        return Lift<T>::Of(1.0f);
    }
    const bool positive = Val(cosThetaI_) > 0.0f;
    T cosThetaI = FabsW(cosThetaI_);
    T cosThetaT = Sqrt(cosThetaTSqr);
    T etaCosThetaT = eta * cosThetaT, etaCosThetaI = eta * cosThetaI;
    T Rs = (cosThetaI - etaCosThetaT) / (cosThetaI + etaCosThetaT);
    T Rp = (etaCosThetaI - cosThetaT) / (etaCosThetaI + cosThetaT);
    cosThetaT_ = positive ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}
template <class T>
LMC_HD V3T<T> SampleMicronormalT(const T &r0, const T &r1, const T &alpha, T &pdfW) {  // microfacet.h:165-185, ADEpsilon = 1e-6
    T phiM = c_TWOPI * r1;
    T sinPhiM = Sin(phiM), cosPhiM = Cos(phiM);
    T alphaSqr = alpha * alpha;
    T tanThetaMSqr = alphaSqr * (-LogD(Fmax(1.0f - r0, 1e-6f)));
    T cosThetaM = 1.0f / Sqrt(1.0f + tanThetaMSqr);
    T cosThetaMSqr = cosThetaM * cosThetaM;
    pdfW = (1.0f - r0) / (c_PI * alphaSqr * cosThetaM * cosThetaMSqr);
    T sinThetaM = Sqrt(Fmax(1.0f - cosThetaMSqr, 1e-6f));
    return V3T<T>{sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM};
}

// the specular + diffuse terms shared by EvaluatePhong and SamplePhong (phong.cpp:200-259 = :313-372)
template <class T, class In>
LMC_HD void PhongTermsT(const In &b, int off, const T &alpha, const T &cosWi, const T &cosWo, V3T<T> &contrib, T &pdf, T &revPdf) {
    const float exponent = b[off + 7], KsWeight = b[off + 8];
    contrib = C3<T>(0.f, 0.f, 0.f);
    pdf = Lift<T>::Of(0.f);
    if (KsWeight > 0.0f) {
        T weight = Pow(alpha, exponent) * c_INVTWOPI;
        if (Val(weight) > 1e-10f) {
            contrib = C3<T>(b[off + 4], b[off + 5], b[off + 6]) * ((exponent + 2.0f) * weight);
            pdf = (KsWeight * (exponent + 1.0f)) * weight;
        }
    }
    revPdf = pdf;
    if (KsWeight < 1.0f) {
        const float tmp = (1.0f - KsWeight) * c_INVPI;
        contrib = contrib + C3<T>(b[off + 1] * c_INVPI, b[off + 2] * c_INVPI, b[off + 3] * c_INVPI);
        pdf = pdf + tmp * cosWo;
        revPdf = revPdf + tmp * cosWi;
    }
    contrib = contrib * cosWo;
}

// EvaluateBSDF, bsdf.cpp:13-63.  Unknown types produce zeros like the generated else-branch.
template <class T, class In>
LMC_PF void EvaluateBSDFT(bool adjoint, const In &b, int off, const V3T<T> &wi, V3T<T> &normal, const V3T<T> &wo, V3T<T> &contrib, T &cosWo,
                          T &pdf, T &revPdf) {
    const float type = b[off];
    if (type == (float)0 /*Lambertian*/) {  // lambertian.cpp:95-122
        T cosWi = DotT(normal, wi);
        V3T<T> n = normal;
        if (!(Val(cosWi) > 0.0f)) {
            n = -normal;
            cosWi = -cosWi;
        } else {
            normal = DetachW3(normal);  // passed through the two-sided conditional (lambertian.cpp:109-113): its later uses see a constant
        }
        cosWo = DotT(n, wo);
        T fwdScalar = cosWo * c_INVPI;
        contrib = C3<T>(b[off + 1], b[off + 2], b[off + 3]) * fwdScalar;
        pdf = fwdScalar;
        revPdf = cosWi * c_INVPI;
    } else if (type == (float)1 /*Phong*/) {  // phong.cpp:171-259: no two-sided test, no rejection, no fmax on alpha
        T cosWi = DotT(normal, wi);
        V3T<T> n = normal;
        if (!(Val(cosWi) > 0.0f)) {
            n = -normal;
            cosWi = -cosWi;
        } else {
            normal = DetachW3(normal);  // phong.cpp:192-196, as in the Lambertian case
        }
        cosWo = DotT(n, wo);
        T alpha = DotT(ReflectT(wi, n), wo);
        PhongTermsT(b, off, alpha, cosWi, cosWo, contrib, pdf, revPdf);
    } else if (type == (float)2 /*RoughDielectric*/) {  // roughdielectric.cpp:332-438, statement order kept (FabsW)
        const V3T<T> Ks = C3<T>(b[off + 1], b[off + 2], b[off + 3]), Kt = C3<T>(b[off + 4], b[off + 5], b[off + 6]);
        const float eta = b[off + 7], invEta = b[off + 8], alpha = b[off + 9];
        T cosWi = DotT(wi, normal);
        cosWo = DotT(wo, normal);
        const bool reflect = Val(cosWi * cosWo) > 0.0f;
        const float eta_ = Val(cosWi) > 0.0f ? eta : invEta;
        const float revEta_ = Val(cosWo) > 0.0f ? eta : invEta;
        V3T<T> H = reflect ? NormalizeT(wi + wo) : NormalizeT(wi + wo * Lift<T>::Of(eta_));
        if (Val(DotT(H, normal)) < 0.0f) H = -H;
        T cosHWi = DotT(wi, H), cosHWo = DotT(wo, H);
        V3T<T> b0, b1;
        CoordinateSystemT(normal, b0, b1);
        V3T<T> localH{DotT(b0, H), DotT(b1, H), DotT(normal, H)};
        const T alphaT = Lift<T>::Of(alpha);
        T D = BeckmennDT(localH, alphaT, alphaT);
        T unusedT;
        T F = FresnelDielectricExtT(cosHWi, unusedT, eta, invEta);  // cosHWi (= revCosHWo) is constant from here on when >= 0
        T aCosWi = FabsW(cosWi), aCosWo = FabsW(cosWo);
        T G = BeckmennG1T(alpha, aCosWi) * BeckmennG1T(alpha, aCosWo);
        T scaledAlpha = alpha * (1.2f - 0.2f * Sqrt(aCosWi));
        T prob = localH.z * BeckmennDT(localH, scaledAlpha, scaledAlpha);
        T revScaledAlpha = alpha * (1.2f - 0.2f * Sqrt(aCosWo));
        T revProb = localH.z * BeckmennDT(localH, revScaledAlpha, revScaledAlpha);
        if (reflect) {
            T scalar = Fabs(F * D * G / (4.0f * cosWi));
            contrib = Ks * scalar;
            pdf = Fabs(prob * F / (4.0f * cosHWo));
            revPdf = Fabs(revProb * F / (4.0f * cosHWi));
        } else {
            T sqrtDenom = cosHWi + eta_ * cosHWo;
            T revSqrtDenom = cosHWo + revEta_ * cosHWi;
            const float factor = adjoint ? 1.0f : (1.0f / eta_) * (1.0f / eta_);
            T scalar = Fabs(factor * ((1.0f - F) * D * G * (eta_ * eta_) * cosHWi * cosHWo) / (cosWi * (sqrtDenom * sqrtDenom)));
            contrib = Kt * scalar;
            pdf = Fabs(prob * (1.0f - F) * ((eta_ * eta_) * cosHWo) / (sqrtDenom * sqrtDenom));
            revPdf = Fabs(revProb * (1.0f - F) * ((revEta_ * revEta_) * cosHWi) / (revSqrtDenom * revSqrtDenom));
        }
    } else {
        contrib = C3<T>(0.f, 0.f, 0.f);
        cosWo = pdf = revPdf = Lift<T>::Of(0.f);
    }
}
// SampleBSDF, bsdf.cpp:65-171 (fixDiscrete = false)
template <class T, class In>
LMC_PF void SampleBSDFT(bool adjoint, const In &b, int off, const V3T<T> &wi, V3T<T> &normal, const T &r0, const T &r1, float uDiscrete,
                        V3T<T> &wo, V3T<T> &contrib, T &cosWo, T &pdf, T &revPdf) {
    const float type = b[off];
    if (type == (float)0) {  // lambertian.cpp:124-151
        T cosWi = DotT(wi, normal);
        V3T<T> n = normal;
        if (!(Val(cosWi) > 0.0f)) {
            n = -normal;
            cosWi = -cosWi;
        } else {
            normal = DetachW3(normal);  // lambertian.cpp:140-144
        }
        V3T<T> b0, b1;
        CoordinateSystemT(n, b0, b1);
        V3T<T> ret = SampleCosHemisphereT(r0, r1);
        wo = b0 * ret.x + b1 * ret.y + n * ret.z;
        cosWo = ret.z;
        pdf = ret.z * c_INVPI;
        contrib = C3<T>(b[off + 1], b[off + 2], b[off + 3]);
        revPdf = cosWi * c_INVPI;
    } else if (type == (float)1) {  // phong.cpp:261-393: lobe chosen by uDiscrete (the scalar sampler uses rndParam[0])
        const float exponent = b[off + 7], KsWeight = b[off + 8];
        T cosWi = DotT(normal, wi);
        V3T<T> n = normal;
        if (!(Val(cosWi) > 0.0f)) {
            n = -normal;
            cosWi = -cosWi;
        } else {
            normal = DetachW3(normal);  // phong.cpp:284-288
        }
        V3T<T> R = ReflectT(wi, n);
        V3T<T> b0, b1;
        if (uDiscrete > KsWeight) {
            V3T<T> localDir = SampleCosHemisphereT(r0, r1);
            CoordinateSystemT(n, b0, b1);
            wo = b0 * localDir.x + b1 * localDir.y + n * localDir.z;
        } else {
            const float power = 1.0f / (exponent + 1.0f);
            T cosAlpha = Pow(r1, power);
            T sinAlpha = Sqrt(Fmax(1.0f - cosAlpha * cosAlpha, 1e-6f));
            T phi = c_TWOPI * r0;
            CoordinateSystemT(R, b0, b1);
            wo = b0 * (sinAlpha * Cos(phi)) + b1 * (sinAlpha * Sin(phi)) + R * cosAlpha;
        }
        cosWo = DotT(n, wo);
        T alpha = DotT(R, wo);
        PhongTermsT(b, off, alpha, cosWi, cosWo, contrib, pdf, revPdf);
        contrib = contrib * (1.0f / pdf);
    } else if (type == (float)2) {  // roughdielectric.cpp:440-528, statement order kept (FabsW)
        const V3T<T> Ks = C3<T>(b[off + 1], b[off + 2], b[off + 3]), Kt = C3<T>(b[off + 4], b[off + 5], b[off + 6]);
        const float eta = b[off + 7], invEta = b[off + 8], alpha = b[off + 9];
        T cosWi = DotT(wi, normal);
        T scaledAlpha = alpha * (1.2f - 0.2f * Sqrt(FabsW(cosWi)));
        T mPdf;
        V3T<T> localH = SampleMicronormalT(r0, r1, scaledAlpha, mPdf);
        V3T<T> b0, b1;
        CoordinateSystemT(normal, b0, b1);
        V3T<T> H = b0 * localH.x + b1 * localH.y + normal * localH.z;
        T cosHWi = DotT(wi, H);
        T cosThetaT;
        T F = FresnelDielectricExtT(cosHWi, cosThetaT, eta, invEta);
        V3T<T> refl;
        if (uDiscrete <= Val(F)) {
            wo = ReflectT(wi, H);
            refl = Ks;
            T cosHWo = DotT(wo, H);
            pdf = Fabs(mPdf * F / (4.0f * cosHWo));
            T rev_dwh_dwo = 1.0f / (4.0f * cosHWi);
            cosWo = DotT(wo, normal);
            T revScaledAlp = alpha * (1.2f - 0.2f * Sqrt(FabsW(cosWo)));
            T revD = BeckmennDT(localH, revScaledAlp, revScaledAlp);
            revPdf = Fabs(F * revD * localH.z * rev_dwh_dwo);
        } else {
            const float etaR = Val(cosThetaT) < 0.0f ? invEta : eta;  // Refract, utils.h:202-210
            wo = H * (DotT(wi, H) * etaR + cosThetaT) - wi * Lift<T>::Of(etaR);
            const float eta_ = Val(cosWi) > 0.0f ? eta : invEta;
            const float factor = adjoint ? 1.0f : (1.0f / eta_) * (1.0f / eta_);
            refl = Kt * Lift<T>::Of(factor);
            T cosHWo = DotT(wo, H);
            T sqrtDenom = cosHWi + eta_ * cosHWo;
            T dwh_dwo = ((eta_ * eta_) * cosHWo) / (sqrtDenom * sqrtDenom);
            pdf = Fabs(mPdf * (1.0f - F) * Fabs(dwh_dwo));
            cosWo = DotT(wo, normal);
            const float revEta_ = Val(cosWo) > 0.0f ? eta : invEta;
            T revSqrtDenom = cosHWo + revEta_ * cosHWi;
            T rev_dwh_dwo = ((revEta_ * revEta_) * cosHWi) / (revSqrtDenom * revSqrtDenom);
            T revScaledAlp = alpha * (1.2f - 0.2f * Sqrt(FabsW(cosWo)));
            T revD = BeckmennDT(localH, revScaledAlp, revScaledAlp);
            revPdf = Fabs((1.0f - F) * revD * localH.z * rev_dwh_dwo);
        }
        T aCosWi = FabsW(cosWi), aCosWo = FabsW(cosWo);
        const T alphaT = Lift<T>::Of(alpha);
        T D = BeckmennDT(localH, alphaT, alphaT);
        T G = BeckmennG1T(alpha, aCosWi) * BeckmennG1T(alpha, aCosWo);
        T numerator = D * G * cosHWi;
        T denominator = mPdf * aCosWi;
        contrib = refl * Fabs(numerator / denominator);
    } else {
        wo = contrib = C3<T>(0.f, 0.f, 0.f);
        cosWo = pdf = revPdf = Lift<T>::Of(0.f);
    }
}

// BSDFSampling<adjoint, fixedDiscrete=false>, path.cpp:2962-3135.  doLightCoord: the caller's compile-time half of the reference's
// doLightCoordinateSampling (maxLightDepth == 0 && camDepth == maxCamDepth - 3, path.cpp:3872); the branch is taken when the scene
// block says uselightcoordinatesampling (scene[0] == 1) and the NEXT vertex's light slot is an area light (path.cpp:2979-3025): the
// direction comes from the point the two primary samples select on the next vertex's triangle, in the light's own sampling
// coordinates, with the area -> solid-angle Jacobian |cos| / d^2 / shapePdf
template <bool adjoint, class T, class In>
LMC_HD void BSDFSamplingT(const In &b, int off, const T &r0, const T &r1, float bsdfDiscrete, float useAbs, PState<T> &ps, V3T<T> &dir, bool doLightCoord = false,
                          float useLightCoord = 0.0f) {
    V3T<T> bsdfContrib;
    T cosWo, bsdfPdf, bsdfRevPdf, jacobian;
    const int nextShapeOff = off + 10 + 1, nextLightOff = nextShapeOff + 46;  // behind this vertex's BSDF slot and rrWeight
    if (doLightCoord && useLightCoord == 1.0f && b[nextLightOff] == 1.0f /* LightType::AreaLight */) {
        V3T<T> nextPosition, nextNormal;
        float shapePdf;
        SampleShapeT(b, nextShapeOff, r0, r1, nextPosition, nextNormal, shapePdf);
        dir = nextPosition - ps.position;
        T distSq = LenSqT(dir);
        T invDistSq = 1.0f / distSq;
        T invDist = Sqrt(invDistSq);
        dir = dir * invDist;
        EvaluateBSDFT(false, b, off, ps.wi, ps.shadingNormal, dir, bsdfContrib, cosWo, bsdfPdf, bsdfRevPdf);
        jacobian = Fabs(DotT(nextNormal, dir) * invDistSq) / shapePdf;
    } else if (useAbs == 0.0f) {
        SampleBSDFT(adjoint, b, off, ps.wi, ps.shadingNormal, r0, r1, bsdfDiscrete, dir, bsdfContrib, cosWo, bsdfPdf, bsdfRevPdf);
        jacobian = Lift<T>::Of(1.0f);
    } else {
        dir = SampleSphereT(r0, r1, jacobian);
        EvaluateBSDFT(adjoint, b, off, ps.wi, ps.shadingNormal, dir, bsdfContrib, cosWo, bsdfPdf, bsdfRevPdf);
    }
    if (adjoint) {
        T factor = ShadingNormalCorrectionAdj(ps.wi, ps, dir);
        bsdfContrib = bsdfContrib * factor;
    }
    bsdfContrib = bsdfContrib * jacobian;
    ps.accMISWThis = MISq(cosWo / bsdfPdf) * (ps.accMISWThis * MISq(bsdfRevPdf) + ps.accMISWPrev);
    ps.accMISWPrev = MISq(1.0f / bsdfPdf);
    ps.throughput = Cmul(ps.throughput, bsdfContrib);
}

// ------------------------------------------------------------------------------------------ lights (56-float slot)
struct EnvBlk {
    Rot3 toWorld, toLight;
    float cdfCol0, cdfCol1, cdfRow0, cdfRow1, col, row, pix0, pix1;
    float img[4][3];
    float rowWeight0, rowWeight1, normalization;
};
template <class In>
LMC_HD EnvBlk ReadEnv(const In &b, int off) {  // after the type float: 15 + 15 + 8 + 12 + 3 (envlight.cpp:270-288)
    EnvBlk e;
    e.toWorld = ReadXform(b, off + 1);
    e.toLight = ReadXform(b, off + 16);
    int o = off + 31;
    e.cdfCol0 = b[o + 0], e.cdfCol1 = b[o + 1], e.cdfRow0 = b[o + 2], e.cdfRow1 = b[o + 3], e.col = b[o + 4], e.row = b[o + 5], e.pix0 = b[o + 6], e.pix1 = b[o + 7];
    for (int k = 0; k < 4; k++)
        for (int c = 0; c < 3; c++) e.img[k][c] = b[o + 8 + k * 3 + c];
    e.rowWeight0 = b[o + 20], e.rowWeight1 = b[o + 21], e.normalization = b[o + 22];
    return e;
}
template <class T>
LMC_HD void EnvSampleDirectionT(const EnvBlk &e, const T &r0, const T &r1, V3T<T> &dirToLight, V3T<T> &value, T &pdf) {  // envlight.cpp:290-322
    T u0 = (r0 - e.cdfCol0) / (e.cdfCol1 - e.cdfCol0);
    T u1 = (r1 - e.cdfRow0) / (e.cdfRow1 - e.cdfRow0);
    T tent0 = TentT(u0), tent1 = TentT(u1);
    T phi = ((e.col + tent0) + 0.5f) * e.pix0;
    T theta = ((e.row + tent1) + 0.5f) * e.pix1;
    T sinPhi = Sin(phi), cosPhi = Cos(phi), sinTheta = Sin(theta), cosTheta = Cos(theta);
    dirToLight = RotVec(e.toWorld, V3T<T>{sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta});
    T dx1 = tent0, dx2 = 1.0f - tent0, dy1 = tent1, dy2 = 1.0f - tent1;
    V3T<T> value1 = C3<T>(e.img[0][0], e.img[0][1], e.img[0][2]) * dx2 * dy2 + C3<T>(e.img[1][0], e.img[1][1], e.img[1][2]) * dx1 * dy2;
    V3T<T> value2 = C3<T>(e.img[2][0], e.img[2][1], e.img[2][2]) * dx2 * dy1 + C3<T>(e.img[3][0], e.img[3][1], e.img[3][2]) * dx1 * dy1;
    value = value1 + value2;
    pdf = (LumT(value1) * e.rowWeight0 + LumT(value2) * e.rowWeight1) * e.normalization / Fmax(Fabs(sinTheta), 1e-7f);
}

// SampleDirect dispatcher, light.cpp:12-138
template <class T, class In>
LMC_PF void SampleDirectT(const In &b, int off, const SceneBlk &sc, const V3T<T> &pos, const T &r0, const T &r1, V3T<T> &dirToLight, V3T<T> &lightContrib,
                          T &cosAtLight, T &directPdf, T &emissionPdf) {
    const float type = b[off];
    if (type == 0.0f) {  // point, pointlight.cpp:20-31,74-93
        V3T<T> lightPos = C3<T>(b[off + 1], b[off + 2], b[off + 3]), emission = C3<T>(b[off + 4], b[off + 5], b[off + 6]);
        dirToLight = lightPos - pos;
        T distSq = LenSqT(dirToLight);
        directPdf = distSq;
        T dist = Sqrt(distSq);
        dirToLight = dirToLight * (1.0f / dist);
        lightContrib = emission * (1.0f / distSq);
        emissionPdf = Lift<T>::Of(c_INVFOURPI);
        cosAtLight = Lift<T>::Of(1.0f);
    } else if (type == 1.0f) {  // area, arealight.cpp:106-135
        V3T<T> posOnLight, normalOnLight;
        float shapePdf;
        SampleShapeT(b, off + 1, r0, r1, posOnLight, normalOnLight, shapePdf);
        V3T<T> emission = C3<T>(b[off + 47], b[off + 48], b[off + 49]);
        dirToLight = posOnLight - pos;
        T distSq = LenSqT(dirToLight);
        T dist = Sqrt(distSq);
        dirToLight = dirToLight * (1.0f / dist);
        cosAtLight = -DotT(dirToLight, normalOnLight);
        directPdf = shapePdf * distSq / cosAtLight;
        lightContrib = emission * (1.0f / directPdf);
        emissionPdf = shapePdf * cosAtLight * c_INVPI;
    } else if (type == 2.0f) {  // env, envlight.cpp:324-346
        EnvBlk e = ReadEnv(b, off);
        V3T<T> value;
        EnvSampleDirectionT(e, r0, r1, dirToLight, value, directPdf);
        lightContrib = value * (1.0f / directPdf);
        cosAtLight = Lift<T>::Of(1.0f);
        float positionPdf = c_INVPI / (sc.bsRadius * sc.bsRadius);
        emissionPdf = directPdf * positionPdf;
    } else {
        dirToLight = lightContrib = C3<T>(0, 0, 0);
        cosAtLight = directPdf = emissionPdf = Lift<T>::Of(0.f);
    }
}

// Emission dispatcher, light.cpp:140-185
template <class T, class In>
LMC_PF void EmissionT(const In &b, int off, const SceneBlk &sc, const V3T<T> &dirToLight, const V3T<T> &normalOnLight, V3T<T> &emission, T &directPdf,
                      T &emissionPdf) {
    const float type = b[off];
    if (type == 1.0f) {  // arealight.cpp:157-174
        float shapePdf = b[off + 1 + 45];
        emission = C3<T>(b[off + 47], b[off + 48], b[off + 49]);
        T cosAtLight = -DotT(normalOnLight, dirToLight);
        directPdf = Lift<T>::Of(shapePdf);
        emissionPdf = cosAtLight * shapePdf * c_INVPI;
    } else if (type == 2.0f) {  // envlight.cpp:348-377
        EnvBlk e = ReadEnv(b, off);
        V3T<T> d = RotVec(e.toLight, dirToLight);
        T uv0 = Atan2(d.x, -d.z) / e.pix0 - 0.5f;
        T uv1 = Acos(d.y) / e.pix1 - 0.5f;
        T dx1 = uv0 - e.col, dx2 = 1.0f - dx1, dy1 = uv1 - e.row, dy2 = 1.0f - dy1;
        V3T<T> value1 = C3<T>(e.img[0][0], e.img[0][1], e.img[0][2]) * dx2 * dy2 + C3<T>(e.img[1][0], e.img[1][1], e.img[1][2]) * dx1 * dy2;
        V3T<T> value2 = C3<T>(e.img[2][0], e.img[2][1], e.img[2][2]) * dx2 * dy1 + C3<T>(e.img[3][0], e.img[3][1], e.img[3][2]) * dx1 * dy1;
        emission = value1 + value2;
        T sinTheta = Sqrt(Fmax(1.0f - d.y * d.y, 1e-6f));
        directPdf = (LumT(value1) * e.rowWeight0 + LumT(value2) * e.rowWeight1) * e.normalization / Fmax(Fabs(sinTheta), 1e-7f);
        float positionPdf = c_INVPI / (sc.bsRadius * sc.bsRadius);
        emissionPdf = directPdf * positionPdf;
    } else {
        emission = C3<T>(0, 0, 0);
        directPdf = emissionPdf = Lift<T>::Of(0.f);
    }
}

template <class T>
LMC_HD void SampleConcentricDiscT(const T &p0, const T &p1, T &ox, T &oy) {  // sampling.h:65-97
    T r1 = 2.0f * p0 - 1.0f, r2 = 2.0f * p1 - 1.0f;
    T r, phi;
    if (Val(r1) == 0.0f || Val(r2) == 0.0f) {
        r = Lift<T>::Of(0.f), phi = Lift<T>::Of(0.f);
    } else if (Val(r1) * Val(r1) > Val(r2) * Val(r2)) {
        r = r1;
        phi = c_PIOVERFOUR * (r2 / r1);
    } else {
        r = r2;
        phi = c_PIOVERTWO - (r1 / r2) * c_PIOVERFOUR;
    }
    ox = r * Cos(phi), oy = r * Sin(phi);
}

// Emit dispatcher, light.cpp:187-324
template <class T, class In>
LMC_PF void EmitT(const In &b, int off, const SceneBlk &sc, const T &p0, const T &p1, const T &d0, const T &d1, V3T<T> &org, V3T<T> &dir, V3T<T> &emission,
                  T &cosAtLight, T &emissionPdf, T &directPdf) {
    const float type = b[off];
    if (type == 0.0f) {  // pointlight.cpp:95-116
        org = C3<T>(b[off + 1], b[off + 2], b[off + 3]);
        T j;
        dir = SampleSphereT(d0, d1, j);
        emission = C3<T>(b[off + 4], b[off + 5], b[off + 6]);
        emissionPdf = Lift<T>::Of(c_INVFOURPI);
        cosAtLight = directPdf = Lift<T>::Of(1.0f);
    } else if (type == 1.0f) {  // arealight.cpp:176-208
        V3T<T> normalOnLight;
        float shapePdf;
        SampleShapeT(b, off + 1, p0, p1, org, normalOnLight, shapePdf);
        V3T<T> d = SampleCosHemisphereT(d0, d1);
        V3T<T> b0, b1;
        CoordinateSystemT(normalOnLight, b0, b1);
        dir = b0 * d.x + b1 * d.y + normalOnLight * d.z;
        emission = C3<T>(b[off + 47], b[off + 48], b[off + 49]) * Lift<T>::Of(c_PI / shapePdf);
        cosAtLight = d.z;
        emissionPdf = d.z * c_INVPI * shapePdf;
        directPdf = Lift<T>::Of(shapePdf);
    } else if (type == 2.0f) {  // envlight.cpp:379-400
        EnvBlk e = ReadEnv(b, off);
        EnvSampleDirectionT(e, d0, d1, dir, emission, directPdf);
        dir = -dir;
        T ox, oy;
        SampleConcentricDiscT(p0, p1, ox, oy);
        V3T<T> b0, b1;
        CoordinateSystemT(dir, b0, b1);
        V3T<T> perp = b0 * ox + b1 * oy;
        org = C3<T>(sc.bsCenter[0], sc.bsCenter[1], sc.bsCenter[2]) + (perp - dir) * Lift<T>::Of(sc.bsRadius);
        cosAtLight = Lift<T>::Of(1.0f);
        float positionPdf = c_INVPI / (sc.bsRadius * sc.bsRadius);
        emissionPdf = directPdf * positionPdf;
    } else {
        org = dir = emission = C3<T>(0, 0, 0);
        cosAtLight = emissionPdf = directPdf = Lift<T>::Of(0.f);
    }
}

// ------------------------------------------------------------------------------------------ the program
// RegisterPathFuncBidirMALA, PathFuncMode::Static (path.cpp:3664-3911).  Returns log Luminance(contrib).
// `primary` is read through an accessor (primary(k) = the k-th primary sample as a T, [0] = time, inactive): an array of lifted values
// for the per-lane forms below, a lane's own seeding of the float samples for the wave-cooperative Hessian (h2hess.hip), which
// thereby never holds 17 second-order values in indexable (= private) memory.
// LCLASS: what the caller knows about the technique at compile time: -1 nothing; 0 no light sub-path (l <= 1: the light half and
// ConnectVertex are compiled out, and with them the light state that would stay live across the camera loop); 1 no camera sub-path
// (c == 1, light tracing: the camera half is compiled out); 2 both (c >= 2 and l >= 2)
template <class T, class In, class Prim, int LCLASS = -1>
LMC_HD T PathProgramP(int maxCamDepth, int maxLightDepth, const Prim &primary, const float *scene, const In &vp) {
    const SceneBlk sc = ReadScene(scene);
    int buf = 3;  // lensVertexPos
    int pi = 1;
    int lgtBSDFOff = -1;
    PState<T> lps, cps;
    V3T<T> contrib = C3<T>(0, 0, 0);
    if (LCLASS != 0 && maxLightDepth > 1) {
        const float lightPickProb = vp[buf++];
        const int lightOff = buf;
        const float lightType = vp[lightOff];
        V3T<T> org, dir;
        T rp0 = primary(pi++), rp1 = primary(pi++), rd0 = primary(pi++), rd1 = primary(pi++);
        {  // EmitFromLight, path.cpp:2799-2841
            T cosLight, emissionPdf, directPdf;
            EmitT(vp, lightOff, sc, rp0, rp1, rd0, rd1, org, dir, lps.throughput, cosLight, emissionPdf, directPdf);
            buf += 56;
            emissionPdf = emissionPdf * lightPickProb;
            directPdf = directPdf * lightPickProb;
            lps.throughput = lps.throughput * Lift<T>::Of(1.0f / lightPickProb);
            lps.accMISWPrev = MISq(directPdf / emissionPdf);
            // NB: the generated program has the IsDelta test inverted w.r.t. path.cpp:611-615 (path.cpp:2833-2838)
            lps.accMISWThis = (lightType == 0.0f) ? MISq(cosLight / emissionPdf) : Lift<T>::Of(0.0f);
        }
        for (int lgtDepth = 0; lgtDepth < maxLightDepth - 1; lgtDepth++) {
            T st0, st1;
            IntersectT(vp, buf, org, dir, lps, st0, st1);
            buf += 46;
            const float bsdfDiscrete = vp[buf++], useAbs = vp[buf++];
            lps.wi = -dir;
            if (lgtDepth == 0) {  // ConvertMISLightEmit, path.cpp:2843-2856
                T invCosTheta = 1.0f / MISq(Fabs(DotT(dir, lps.shadingNormal)));
                if (lightType == 2.0f) {
                    lps.accMISWPrev = lps.accMISWPrev * invCosTheta;
                } else {
                    lps.accMISWPrev = lps.accMISWPrev * (invCosTheta * MISq(DistSqT(org, lps.position)));
                }
                lps.accMISWThis = lps.accMISWThis * invCosTheta;
            } else {  // ConvertMIS, path.cpp:2876-2882
                lps.accMISWPrev = lps.accMISWPrev * MISq(DistSqT(org, lps.position));
                T invCosTheta = 1.0f / MISq(Fabs(DotT(dir, lps.shadingNormal)));
                lps.accMISWPrev = lps.accMISWPrev * invCosTheta;
                lps.accMISWThis = lps.accMISWThis * invCosTheta;
            }
            if (lgtDepth == maxLightDepth - 2) {
                if (maxCamDepth == 1) {  // ConnectToCamera, path.cpp:2884-2960
                    V3T<T> camOrg, camDir;
                    SamplePrimaryT(sc, Lift<T>::Of(0.5f), Lift<T>::Of(0.5f), camOrg, camDir);
                    V3T<T> dirToCamera = camOrg - lps.position;
                    T distSq = LenSqT(dirToCamera);
                    T dist = Sqrt(distSq);
                    dirToCamera = dirToCamera * (1.0f / dist);
                    V3T<T> bsdfContrib;
                    T cosToCamera, bsdfPdf, bsdfRevPdf;
                    EvaluateBSDFT(true, vp, buf, lps.wi, lps.shadingNormal, dirToCamera, bsdfContrib, cosToCamera, bsdfPdf, bsdfRevPdf);
                    T factor = ShadingNormalCorrectionAdj(lps.wi, lps, dirToCamera);
                    bsdfContrib = bsdfContrib * factor;
                    T invCosAtCamera = -(1.0f / DotT(camDir, dirToCamera));
                    T imagePointToCameraDist = sc.camDist * invCosAtCamera;
                    T imageToSolidAngleFactor = imagePointToCameraDist * imagePointToCameraDist * invCosAtCamera;
                    T imageToSurfaceFactor = imageToSolidAngleFactor * FabsW(cosToCamera) / distSq;  // cosToCamera's later use (surfaceToImageFactor) sees a constant
                    T wLight = MISq(imageToSurfaceFactor / sc.pixelCount) * (lps.accMISWPrev + lps.accMISWThis * MISq(bsdfRevPdf));
                    T misWeight = 1.0f / (wLight + 1.0f);
                    T surfaceToImageFactor = cosToCamera / imageToSurfaceFactor;
                    V3T<T> c = bsdfContrib * (misWeight / (sc.pixelCount * surfaceToImageFactor));
                    lps.throughput = Cmul(c, lps.throughput);
                    contrib = lps.throughput;
                }
                lgtBSDFOff = buf;
                buf += 10;
                break;
            }
            T r0 = primary(pi++), r1 = primary(pi++);
            BSDFSamplingT<true>(vp, buf, r0, r1, bsdfDiscrete, useAbs, lps, dir);
            buf += 10;
            const float rrWeight = vp[buf++];
            lps.throughput = lps.throughput * Lift<T>::Of(rrWeight);
            org = lps.position;
        }
    }
    if (LCLASS != 1 && maxCamDepth > 1) {
        T sx = primary(pi++), sy = primary(pi++);
        V3T<T> org, dir;
        {  // EmitFromCamera, path.cpp:3138-3170
            V3T<T> cOrg, camDir;
            SamplePrimaryT(sc, Lift<T>::Of(0.5f), Lift<T>::Of(0.5f), cOrg, camDir);
            SamplePrimaryT(sc, sx, sy, org, dir);
            T cosAtCamera = DotT(camDir, dir);
            T imagePointToCameraDist = sc.camDist / cosAtCamera;
            T cameraPdf = imagePointToCameraDist * imagePointToCameraDist / cosAtCamera;
            cps.throughput = C3<T>(1, 1, 1);
            cps.accMISWPrev = MISq(sc.pixelCount / cameraPdf);
            cps.accMISWThis = Lift<T>::Of(0.0f);
        }
        for (int camDepth = 0; camDepth < maxCamDepth - 1; camDepth++) {
            T st0, st1;
            IntersectT(vp, buf, org, dir, cps, st0, st1);
            buf += 46;
            cps.wi = -dir;
            if (camDepth == maxCamDepth - 2 && maxLightDepth == 0) {
                const float lightType = vp[buf];
                {  // ConvertMISLightHit, path.cpp:2858-2874
                    if (lightType != 2.0f) {
                        T distSq = MISq(DistSqT(org, cps.position));
                        T invCosTheta = 1.0f / MISq(Fabs(DotT(dir, cps.shadingNormal)));
                        cps.accMISWPrev = cps.accMISWPrev * (invCosTheta * distSq);
                        cps.accMISWThis = cps.accMISWThis * invCosTheta;
                    }
                }
                {  // HandleHitLight, path.cpp:3172-3200
                    V3T<T> emission;
                    T directPdf, emissionPdf;
                    EmissionT(vp, buf, sc, dir, cps.shadingNormal, emission, directPdf, emissionPdf);
                    cps.throughput = Cmul(cps.throughput, emission);
                    const float lightPickProb = vp[buf + 56];
                    directPdf = directPdf * lightPickProb;
                    emissionPdf = emissionPdf * lightPickProb;
                    T wCamera = MISq(directPdf) * cps.accMISWPrev + MISq(emissionPdf) * cps.accMISWThis;
                    T misWeight = 1.0f / (1.0f + wCamera);
                    cps.throughput = cps.throughput * misWeight;
                }
                contrib = cps.throughput;
                break;
            }
            {  // ConvertMIS
                cps.accMISWPrev = cps.accMISWPrev * MISq(DistSqT(org, cps.position));
                T invCosTheta = 1.0f / MISq(Fabs(DotT(dir, cps.shadingNormal)));
                cps.accMISWPrev = cps.accMISWPrev * invCosTheta;
                cps.accMISWThis = cps.accMISWThis * invCosTheta;
            }
            if (camDepth == maxCamDepth - 2) {
                if (maxLightDepth == 1) {  // DirectLighting, path.cpp:3202-3290
                    T r0 = primary(pi++), r1 = primary(pi++);
                    const float lightType = vp[buf];
                    V3T<T> dirToLight, lightContrib;
                    T cosAtLight, directPdf, emissionPdf;
                    SampleDirectT(vp, buf, sc, cps.position, r0, r1, dirToLight, lightContrib, cosAtLight, directPdf, emissionPdf);
                    buf += 56;
                    V3T<T> bsdfContrib;
                    T cosToLight, bsdfPdf, bsdfRevPdf;
                    EvaluateBSDFT(false, vp, buf, cps.wi, cps.shadingNormal, dirToLight, bsdfContrib, cosToLight, bsdfPdf, bsdfRevPdf);
                    buf += 10;
                    const float lightPickProb = vp[buf++];
                    cps.throughput = Cmul(cps.throughput, bsdfContrib);
                    cps.throughput = Cmul(cps.throughput, lightContrib) * Lift<T>::Of(1.0f / lightPickProb);
                    T wLight = (lightType == 0.0f) ? Lift<T>::Of(0.0f) : MISq(bsdfPdf / (lightPickProb * directPdf));
                    T wCamera = MISq(emissionPdf * cosToLight / (directPdf * cosAtLight)) * (cps.accMISWPrev + cps.accMISWThis * MISq(bsdfRevPdf));
                    T misWeight = 1.0f / (wLight + 1.0f + wCamera);
                    cps.throughput = cps.throughput * misWeight;
                } else if (LCLASS != 0) {  // ConnectVertex, path.cpp:3292-3379
                    V3T<T> dirToLight = lps.position - cps.position;
                    T distSq = LenSqT(dirToLight);
                    T dist = Sqrt(distSq);
                    dirToLight = dirToLight * (1.0f / dist);
                    V3T<T> camBsdfFactor, lgtBsdfFactor;
                    T cosCamera, camBsdfPdf, camBsdfRevPdf, cosLight, lgtBsdfPdf, lgtBsdfRevPdf;
                    EvaluateBSDFT(false, vp, buf, cps.wi, cps.shadingNormal, dirToLight, camBsdfFactor, cosCamera, camBsdfPdf, camBsdfRevPdf);
                    EvaluateBSDFT(true, vp, lgtBSDFOff, lps.wi, lps.shadingNormal, -dirToLight, lgtBsdfFactor, cosLight, lgtBsdfPdf, lgtBsdfRevPdf);
                    T lgtFactor = ShadingNormalCorrectionAdj(lps.wi, lps, -dirToLight);
                    lgtBsdfFactor = lgtBsdfFactor * lgtFactor;
                    T geometryTerm = 1.0f / distSq;
                    T camBsdfDirPdfA = camBsdfPdf * cosLight * geometryTerm;
                    T lgtBsdfDirPdfA = lgtBsdfPdf * cosCamera * geometryTerm;
                    T wLight = MISq(camBsdfDirPdfA) * (lps.accMISWPrev + lps.accMISWThis * MISq(lgtBsdfRevPdf));
                    T wCamera = MISq(lgtBsdfDirPdfA) * (cps.accMISWPrev + cps.accMISWThis * MISq(camBsdfRevPdf));
                    T misWeight = 1.0f / (wLight + 1.0f + wCamera);
                    cps.throughput = Cmul(lps.throughput, cps.throughput);
                    cps.throughput = Cmul(cps.throughput, camBsdfFactor);
                    cps.throughput = Cmul(cps.throughput, lgtBsdfFactor) * (geometryTerm * misWeight);
                }
                contrib = cps.throughput;
                break;
            }
            T r0 = primary(pi++), r1 = primary(pi++);
            const float bsdfDiscrete = vp[buf++], useAbs = vp[buf++];
            BSDFSamplingT<false>(vp, buf, r0, r1, bsdfDiscrete, useAbs, cps, dir, maxLightDepth == 0 && camDepth == maxCamDepth - 3, sc.useLightCoord);
            buf += 10;
            const float rrWeight = vp[buf++];
            cps.throughput = cps.throughput * Lift<T>::Of(rrWeight);
            org = cps.position;
        }
    }
    return Log(LumT(contrib));
}
template <class T>
struct PrimArray {
    const T *p;
    LMC_HD const T &operator()(int k) const { return p[k]; }
};
template <class T, class In>
LMC_HD T PathProgram(int maxCamDepth, int maxLightDepth, const T *primary /* [2L+1], [0] = time (inactive) */, const float *scene, const In &vp) {
    return PathProgramP<T, In, PrimArray<T>>(maxCamDepth, maxLightDepth, PrimArray<T>{primary}, scene, vp);
}

// value only: evaluate_path_bidir_mala_<c>_<l>_static
template <class In>
LMC_HD float PathFuncValue(int c, int l, const float *primary, const float *scene, const In &vp) {
    return PathProgram<float, In>(c, l, primary, scene, vp);
}

template <int N, class In>
LMC_HD void PathFuncGradN(int c, int l, const float *primary, const float *scene, const In &vp, float *logLum, float *grad) {
    const int dim = 2 * (c + l - 1 > 2 ? c + l - 1 : 2);
    Dual<N> p[2 * 8 + 1];
    p[0] = MakeDual<N>(primary[0]);
    for (int k = 0; k < dim; k++) {
        p[k + 1] = MakeDual<N>(primary[k + 1]);
        if (k < N) p[k + 1].d[k] = 1.0f;
    }
    Dual<N> r = PathProgram<Dual<N>, In>(c, l, p, scene, vp);
    if (logLum) *logLum = r.v;
    for (int k = 0; k < dim && k < N; k++) grad[k] = r.d[k];
}

// evaluate_path_bidir_mala_<c>_<l>_static_derv: d logLum / d primary[1..2L]
template <class In>
LMC_HD void PathFuncGrad(int c, int l, const float *primary, const float *scene, const In &vp, float *logLum, float *grad) {
    const int dim = 2 * (c + l - 1 > 2 ? c + l - 1 : 2);
    if (dim <= 8) PathFuncGradN<8>(c, l, primary, scene, vp, logLum, grad);
    else if (dim <= 12) PathFuncGradN<12>(c, l, primary, scene, vp, logLum, grad);
    else PathFuncGradN<16>(c, l, primary, scene, vp, logLum, grad);
}

// evaluate_path_bidir_<c>_<l>_static_derv (H2MC library, pathlibbidir.so): gradient and Hessian of logLum with respect to
// primary[1..2L]; row i of the Hessian at hess[i * 2L] (path.h:122-123, mutation_h2mc.h:76-79).  One pass per row and per
// chunk of HC columns with a dual number whose scalar is itself a Dual<HC> (value, HC gradient components, and their
// derivative along direction i).  Chunking keeps the working set of the second-order type at 2 (HC + 1) floats per value
// whatever the dimension: the 16-wide form needed 22 KB of private memory per lane and faulted on gfx950, the 8-wide one
// 12.6 KB; one instantiation also halves the compile time.
#ifndef LMC_HESS_CHUNK
#define LMC_HESS_CHUNK 8
#endif
constexpr int HC = LMC_HESS_CHUNK;
// one pass: row i, columns [c0, c0 + HC)
template <class In>
LMC_HD void PathFuncHessPass(int c, int l, const float *primary, const float *scene, const In &vp, int i, int c0, float *logLum, float *grad, float *hess,
                             bool firstOfRow) {
    const int dim = 2 * (c + l - 1 > 2 ? c + l - 1 : 2);
    typedef DualS<1, Dual<HC>> T2;
    T2 p[2 * 8 + 1];
    p[0] = Lift<T2>::Of(primary[0]);
    for (int k = 0; k < dim; k++) {
        p[k + 1] = Lift<T2>::Of(primary[k + 1]);
        if (k >= c0 && k < c0 + HC) p[k + 1].v.d[k - c0] = 1.0f;
        if (k == i) p[k + 1].d[0].v = 1.0f;
    }
    T2 r = PathProgram<T2, In>(c, l, p, scene, vp);
    if (i == 0 && firstOfRow && logLum) *logLum = r.v.v;
    if (firstOfRow && grad) grad[i] = r.d[0].v;  // the forward directional derivative: exact (the reference's `g`)
    for (int k = c0; k < dim && k < c0 + HC; k++) hess[i * dim + k] = r.d[0].d[k - c0];
}
template <class In>
LMC_HD void PathFuncHess(int c, int l, const float *primary, const float *scene, const In &vp, float *logLum, float *grad, float *hess) {
    const int dim = 2 * (c + l - 1 > 2 ? c + l - 1 : 2);
    for (int i = 0; i < dim; i++)
        for (int c0 = 0; c0 < dim; c0 += HC) PathFuncHessPass(c, l, primary, scene, vp, i, c0, logLum, grad, hess, c0 == 0);
}

// The chain loop only differentiates states with dim <= PSS_MAX_LENGTH = 12 (mutation_mala.h:94-96): no Dual<16> copy of
// the program in the step kernel, and one non-inlined copy per kernel instead of one per call site (compile time).
// The gradient in blocks of B components: ceil(dim / B) passes of the program in Dual<B>.  Forward-mode components never mix, so
// the result is bit-identical to one Dual<dim> pass; the primal is recomputed per pass (dim 12, B 4: 15 units of arithmetic
// instead of 13) in exchange for a third of the live state.
template <int B, class In>
LMC_HD void PathFuncGradBlocked(int c, int l, const float *primary, const float *scene, const In &vp, float *logLum, float *grad) {
    const int dim = 2 * (c + l - 1 > 2 ? c + l - 1 : 2);
    for (int b0 = 0; b0 < dim; b0 += B) {
        Dual<B> p[2 * 8 + 1];
        p[0] = MakeDual<B>(primary[0]);
        for (int k = 0; k < dim; k++) {
            p[k + 1] = MakeDual<B>(primary[k + 1]);
            if (k >= b0 && k < b0 + B) p[k + 1].d[k - b0] = 1.0f;
        }
        Dual<B> r = PathProgram<Dual<B>, In>(c, l, p, scene, vp);
        if (logLum && b0 == 0) *logLum = r.v;
        for (int k = b0; k < dim && k < b0 + B; k++) grad[k] = r.d[k - b0];
    }
}
// Inside the step kernels the blocked form with B = 2 is the default since round 3: the cache-filling launch runs beside the hot
// launch, and what it costs there is the SIMD slots its waves hold, not its arithmetic -- one pass in Dual<12> keeps 7 KB of
// private memory per lane alive for 2.3 ms per wave-step.  Driver window 290.5 -> 306.9 M chain-steps/s, a complete 256-step run
// 350.5 -> 354.0 M (profiles/r03_p_ab_gradient_block.jsonl; B = 1 / 2 / 3 / 4 / 6: 305 / 307 / 299 / 300-315 / 306).
#ifndef LMC_GRAD_BLOCK
#define LMC_GRAD_BLOCK 2  // 0: one pass in Dual<8> / Dual<12>
#endif
#ifdef __HIPCC__
template <class In>
__device__ __noinline__ void PathFuncGradUpTo12(int c, int l, const float *primary, const float *scene, const In &vp, float *logLum, float *grad) {
#if LMC_GRAD_BLOCK > 0
    PathFuncGradBlocked<LMC_GRAD_BLOCK>(c, l, primary, scene, vp, logLum, grad);
#else
    const int dim = 2 * (c + l - 1 > 2 ? c + l - 1 : 2);
    if (dim <= 8) PathFuncGradN<8>(c, l, primary, scene, vp, logLum, grad);
    else
        PathFuncGradN<12>(c, l, primary, scene, vp, logLum, grad);
#endif
}
#endif

}  // namespace lmcd
#ifdef LMC_PF_CONTRACT
#pragma clang fp contract(off)
#endif
