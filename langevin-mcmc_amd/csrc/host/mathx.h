// Host-side small linear algebra for scene set-up (not on the hot path).
// Mirrors the conventions of the reference's transform.cpp / quaternion.cpp / animatedtransform.cpp
// (/root/reference/src/transform.cpp:5-90, quaternion.h:13-40, quaternion.cpp:4-37,
//  animatedtransform.cpp:10-44) so that the 38-float scene block and the 15-float
// AnimatedTransform blocks of the path-function ABI come out in the reference's layout.
#pragma once
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace lmc {

struct V2 {
    float x, y;
};
struct V3 {
    float x, y, z;
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 normalize(V3 a) {
    float inv = 1.0f / length(a);
    return a * inv;
}

struct M4 {
    float m[4][4];  // m[row][col]
    static M4 identity() {
        M4 r;
        memset(&r, 0, sizeof(r));
        r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.f;
        return r;
    }
};
inline M4 operator*(const M4 &a, const M4 &b) {
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float s = 0.f;
            for (int k = 0; k < 4; k++) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
inline bool operator!=(const M4 &a, const M4 &b) { return memcmp(&a, &b, sizeof(M4)) != 0; }

inline M4 Scale(V3 s) {
    M4 r = M4::identity();
    r.m[0][0] = s.x;
    r.m[1][1] = s.y;
    r.m[2][2] = s.z;
    return r;
}
inline M4 Translate(V3 d) {
    M4 r = M4::identity();
    r.m[0][3] = d.x;
    r.m[1][3] = d.y;
    r.m[2][3] = d.z;
    return r;
}
inline float Radians(float deg) { return (float(3.14159265358979323846) / 180.f) * deg; }
inline M4 Rotate(float angle, V3 axis) {
    V3 a = normalize(axis);
    float s = std::sin(Radians(angle)), c = std::cos(Radians(angle));
    M4 r = M4::identity();
    r.m[0][0] = a.x * a.x + (1.f - a.x * a.x) * c;
    r.m[0][1] = a.x * a.y * (1.f - c) - a.z * s;
    r.m[0][2] = a.x * a.z * (1.f - c) + a.y * s;
    r.m[1][0] = a.x * a.y * (1.f - c) + a.z * s;
    r.m[1][1] = a.y * a.y + (1.f - a.y * a.y) * c;
    r.m[1][2] = a.y * a.z * (1.f - c) - a.x * s;
    r.m[2][0] = a.x * a.z * (1.f - c) - a.y * s;
    r.m[2][1] = a.y * a.z * (1.f - c) + a.x * s;
    r.m[2][2] = a.z * a.z + (1.f - a.z * a.z) * c;
    return r;
}
inline M4 LookAt(V3 pos, V3 look, V3 up) {
    V3 dir = normalize(look - pos);
    if (length(cross(normalize(up), dir)) == 0) throw std::runtime_error("[Lookat] up vector and viewing direction are parallel");
    V3 left = normalize(cross(normalize(up), dir));
    V3 newUp = cross(dir, left);
    M4 r = M4::identity();
    r.m[0][0] = left.x, r.m[1][0] = left.y, r.m[2][0] = left.z;
    r.m[0][1] = newUp.x, r.m[1][1] = newUp.y, r.m[2][1] = newUp.z;
    r.m[0][2] = dir.x, r.m[1][2] = dir.y, r.m[2][2] = dir.z;
    r.m[0][3] = pos.x, r.m[1][3] = pos.y, r.m[2][3] = pos.z;
    return r;
}
inline M4 Perspective(float fov, float n, float f) {
    float recip = 1.f / (f - n);
    float cot = 1.f / (float)std::tan((double)Radians(fov / 2.0f));
    M4 r;
    memset(&r, 0, sizeof(r));
    r.m[0][0] = cot;
    r.m[1][1] = cot;
    r.m[2][2] = f * recip;
    r.m[2][3] = -n * f * recip;
    r.m[3][2] = 1.f;
    return r;
}
// general 4x4 inverse (Gauss-Jordan in double, rounded to float)
inline M4 Inverse(const M4 &a) {
    double w[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            w[i][j] = a.m[i][j];
            w[i][j + 4] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; c++) {
        int p = c;
        for (int r = c + 1; r < 4; r++)
            if (std::fabs(w[r][c]) > std::fabs(w[p][c])) p = r;
        if (w[p][c] == 0.0) throw std::runtime_error("singular matrix");
        if (p != c)
            for (int j = 0; j < 8; j++) std::swap(w[p][j], w[c][j]);
        double inv = 1.0 / w[c][c];
        for (int j = 0; j < 8; j++) w[c][j] *= inv;
        for (int r = 0; r < 4; r++)
            if (r != c) {
                double f = w[r][c];
                for (int j = 0; j < 8; j++) w[r][j] -= f * w[c][j];
            }
    }
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[i][j] = (float)w[i][j + 4];
    return r;
}
inline V3 XformPoint(const M4 &x, V3 p) {
    float tx = x.m[0][0] * p.x + x.m[0][1] * p.y + x.m[0][2] * p.z + x.m[0][3];
    float ty = x.m[1][0] * p.x + x.m[1][1] * p.y + x.m[1][2] * p.z + x.m[1][3];
    float tz = x.m[2][0] * p.x + x.m[2][1] * p.y + x.m[2][2] * p.z + x.m[2][3];
    float tw = x.m[3][0] * p.x + x.m[3][1] * p.y + x.m[3][2] * p.z + x.m[3][3];
    float inv = 1.0f / tw;
    return {tx * inv, ty * inv, tz * inv};
}
inline V3 XformVector(const M4 &x, V3 v) {
    return {x.m[0][0] * v.x + x.m[0][1] * v.y + x.m[0][2] * v.z, x.m[1][0] * v.x + x.m[1][1] * v.y + x.m[1][2] * v.z,
            x.m[2][0] * v.x + x.m[2][1] * v.y + x.m[2][2] * v.z};
}
inline V3 XformNormal(const M4 &inv, V3 v) {
    return {inv.m[0][0] * v.x + inv.m[1][0] * v.y + inv.m[2][0] * v.z, inv.m[0][1] * v.x + inv.m[1][1] * v.y + inv.m[2][1] * v.z,
            inv.m[0][2] * v.x + inv.m[1][2] * v.y + inv.m[2][2] * v.z};
}

// translation + unit quaternion (x,y,z,w) at two times: the reference's AnimatedTransform
// (animatedtransform.h:11-35), serialized as 15 floats [isMoving, t0(3), t1(3), q0(4), q1(4)].
struct AnimXform {
    float isMoving;
    float t[2][3];
    float q[2][4];
};

inline M4 QuatToM4(const float q[4]) {  // quaternion.h:13-40 (returns the transpose of the literal table)
    float xx = q[0] * q[0], yy = q[1] * q[1], zz = q[2] * q[2];
    float xy = q[0] * q[1], xz = q[0] * q[2], yz = q[1] * q[2];
    float wx = q[0] * q[3], wy = q[1] * q[3], wz = q[2] * q[3];
    M4 m = M4::identity();
    m.m[0][0] = 1.f - 2.f * (yy + zz);
    m.m[1][0] = 2.f * (xy + wz);
    m.m[2][0] = 2.f * (xz - wy);
    m.m[0][1] = 2.f * (xy - wz);
    m.m[1][1] = 1.f - 2.f * (xx + zz);
    m.m[2][1] = 2.f * (yz + wx);
    m.m[0][2] = 2.f * (xz + wy);
    m.m[1][2] = 2.f * (yz - wx);
    m.m[2][2] = 1.f - 2.f * (xx + yy);
    return m;
}

inline void MakeQuaternion(const M4 &m, float q[4]) {  // quaternion.cpp:4-37
    float trace = m.m[0][0] + m.m[1][1] + m.m[2][2];
    if (trace > 1e-7f) {
        float s = (float)std::sqrt((double)trace + 1.0);
        q[3] = s / 2.f;
        s = 0.5f / s;
        q[0] = (m.m[2][1] - m.m[1][2]) * s;
        q[1] = (m.m[0][2] - m.m[2][0]) * s;
        q[2] = (m.m[1][0] - m.m[0][1]) * s;
    } else {
        const int nxt[3] = {1, 2, 0};
        float _q[3];
        int i = 0;
        if (m.m[1][1] > m.m[0][0]) i = 1;
        if (m.m[2][2] > m.m[i][i]) i = 2;
        int j = nxt[i], k = nxt[j];
        float s = std::sqrt((m.m[i][i] - (m.m[j][j] + m.m[k][k])) + 1.f);
        _q[i] = s * 0.5f;
        if (s != 0.f) s = 0.5f / s;
        q[3] = (m.m[k][j] - m.m[j][k]) * s;
        _q[j] = (m.m[j][i] + m.m[i][j]) * s;
        _q[k] = (m.m[k][i] + m.m[i][k]) * s;
        q[0] = _q[0], q[1] = _q[1], q[2] = _q[2];
    }
}

// Polar decomposition A = Q P of the upper-left 3x3 by Newton iteration (the reference uses an
// Eigen JacobiSVD, animatedtransform.cpp:10-31; Q = U V^T is the same matrix).  Throws on scaling.
inline void Decompose(const M4 &m, float t[3], float q[4]) {
    double Q[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Q[i][j] = m.m[i][j];
    for (int it = 0; it < 64; it++) {
        // Qn = (Q + Q^-T)/2
        double c[3][3];
        c[0][0] = Q[1][1] * Q[2][2] - Q[1][2] * Q[2][1];
        c[0][1] = Q[1][2] * Q[2][0] - Q[1][0] * Q[2][2];
        c[0][2] = Q[1][0] * Q[2][1] - Q[1][1] * Q[2][0];
        c[1][0] = Q[0][2] * Q[2][1] - Q[0][1] * Q[2][2];
        c[1][1] = Q[0][0] * Q[2][2] - Q[0][2] * Q[2][0];
        c[1][2] = Q[0][1] * Q[2][0] - Q[0][0] * Q[2][1];
        c[2][0] = Q[0][1] * Q[1][2] - Q[0][2] * Q[1][1];
        c[2][1] = Q[0][2] * Q[1][0] - Q[0][0] * Q[1][2];
        c[2][2] = Q[0][0] * Q[1][1] - Q[0][1] * Q[1][0];
        double det = Q[0][0] * c[0][0] + Q[0][1] * c[0][1] + Q[0][2] * c[0][2];
        if (det == 0.0) throw std::runtime_error("Scaling in animation");
        double diff = 0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double n = 0.5 * (Q[i][j] + c[i][j] / det);  // cofactor/det = inverse transpose
                diff += std::fabs(n - Q[i][j]);
                Q[i][j] = n;
            }
        if (diff < 1e-14) break;
    }
    // P = Q^T A must be identity (no scale / shear)
    for (int i = 0; i < 3; i++) {
        double p = 0;
        for (int k = 0; k < 3; k++) p += Q[k][i] * m.m[k][i];
        if (std::fabs(p - 1.0) > 1e-5) throw std::runtime_error("Scaling in animation");
    }
    M4 Q4 = M4::identity();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Q4.m[i][j] = (float)Q[i][j];
    MakeQuaternion(Q4, q);
    t[0] = m.m[0][3], t[1] = m.m[1][3], t[2] = m.m[2][3];
}

inline AnimXform MakeAnimXform(const M4 &m0, const M4 &m1) {  // animatedtransform.h:57-75
    AnimXform r;
    r.isMoving = (m0 != m1) ? 1.f : 0.f;
    Decompose(m0, r.t[0], r.q[0]);
    if (r.isMoving == 1.f)
        Decompose(m1, r.t[1], r.q[1]);
    else {
        memcpy(r.t[1], r.t[0], sizeof(r.t[0]));
        memcpy(r.q[1], r.q[0], sizeof(r.q[0]));
    }
    return r;
}

inline AnimXform Invert(const AnimXform &x) {  // animatedtransform.h:83-101
    AnimXform r;
    r.isMoving = x.isMoving;
    for (int k = 0; k < 2; k++) {
        r.q[k][0] = -x.q[k][0], r.q[k][1] = -x.q[k][1], r.q[k][2] = -x.q[k][2], r.q[k][3] = x.q[k][3];
        M4 rot = QuatToM4(r.q[k]);
        V3 t = XformVector(rot, V3{x.t[k][0], x.t[k][1], x.t[k][2]});
        r.t[k][0] = -t.x, r.t[k][1] = -t.y, r.t[k][2] = -t.z;
    }
    return r;
}

// static interpolation (isMoving == 0): Translate(t0) * ToMatrix(q0)   (animatedtransform.cpp:33-44)
inline M4 ToM4(const AnimXform &x) { return Translate(V3{x.t[0][0], x.t[0][1], x.t[0][2]}) * QuatToM4(x.q[0]); }

}  // namespace lmc
