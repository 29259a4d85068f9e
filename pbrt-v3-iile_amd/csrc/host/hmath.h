// hmath.h — host-side float vector / matrix helpers for scene preparation.
//
// Scene preparation has to reproduce the reference's float arithmetic
// operation-for-operation: the world-space vertex
// positions, the camera matrices and the BVH all feed bit-exact comparisons
// further down. Every helper therefore states which reference expression it
// follows. Citations are relative to /root/reference/src.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace iile {

struct V3 {
    float x = 0, y = 0, z = 0;
    V3() {}
    V3(float x, float y, float z) : x(x), y(y), z(z) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
// core/geometry.h:229-232 (Vector3::operator*): s * component
inline V3 operator*(float s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, float s) { return V3(s * a.x, s * a.y, s * a.z); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length_sq(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline float length(V3 a) { return std::sqrt(length_sq(a)); }
// core/geometry.h:242-246: division multiplies by a float reciprocal
inline V3 div(V3 a, float f) {
    float inv = 1.f / f;
    return V3(a.x * inv, a.y * inv, a.z * inv);
}
inline V3 normalize(V3 a) { return div(a, length(a)); }
// core/geometry.h:957-963: cross product evaluated in double
inline V3 cross(V3 a, V3 b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return V3(float((ay * bz) - (az * by)), float((az * bx) - (ax * bz)),
              float((ax * by) - (ay * bx)));
}

constexpr float kPi = 3.14159265358979323846f;  // core/pbrt.h:202
// core/pbrt.h:321
inline float radians(float deg) { return (kPi / 180) * deg; }
template <typename T>
inline T clampT(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Bounds3 {
    V3 pmin, pmax;
    Bounds3() {  // core/geometry.h:752-757
        float lo = std::numeric_limits<float>::lowest(), hi = std::numeric_limits<float>::max();
        pmin = V3(hi, hi, hi);
        pmax = V3(lo, lo, lo);
    }
    explicit Bounds3(V3 p) : pmin(p), pmax(p) {}
    Bounds3(V3 a, V3 b)
        : pmin(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)),
          pmax(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)) {}
    V3 diagonal() const { return pmax - pmin; }
    float surface_area() const {  // core/geometry.h:782-785
        V3 d = diagonal();
        return 2 * (d.x * d.y + d.x * d.z + d.y * d.z);
    }
    int maximum_extent() const {  // core/geometry.h:790-798
        V3 d = diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        return d.y > d.z ? 1 : 2;
    }
    V3 offset(V3 p) const {  // core/geometry.h:804-810
        V3 o = p - pmin;
        if (pmax.x > pmin.x) o.x /= pmax.x - pmin.x;
        if (pmax.y > pmin.y) o.y /= pmax.y - pmin.y;
        if (pmax.z > pmin.z) o.z /= pmax.z - pmin.z;
        return o;
    }
};
inline Bounds3 bunion(const Bounds3 &b, V3 p) {
    Bounds3 r;
    r.pmin = V3(std::min(b.pmin.x, p.x), std::min(b.pmin.y, p.y), std::min(b.pmin.z, p.z));
    r.pmax = V3(std::max(b.pmax.x, p.x), std::max(b.pmax.y, p.y), std::max(b.pmax.z, p.z));
    return r;
}
inline Bounds3 bunion(const Bounds3 &a, const Bounds3 &b) {
    Bounds3 r;
    r.pmin = V3(std::min(a.pmin.x, b.pmin.x), std::min(a.pmin.y, b.pmin.y), std::min(a.pmin.z, b.pmin.z));
    r.pmax = V3(std::max(a.pmax.x, b.pmax.x), std::max(a.pmax.y, b.pmax.y), std::max(a.pmax.z, b.pmax.z));
    return r;
}

struct Mat4 {
    float m[4][4];
    Mat4() {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) m[i][j] = (i == j) ? 1.f : 0.f;
    }
    Mat4(float a00, float a01, float a02, float a03, float a10, float a11, float a12, float a13,
         float a20, float a21, float a22, float a23, float a30, float a31, float a32, float a33) {
        float v[16] = {a00, a01, a02, a03, a10, a11, a12, a13, a20, a21, a22, a23, a30, a31, a32, a33};
        std::memcpy(m, v, sizeof(v));
    }
};
// core/transform.h:86-93
inline Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] +
                        a.m[i][3] * b.m[3][j];
    return r;
}
inline Mat4 transpose(const Mat4 &a) {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i];
    return r;
}
// Gauss-Jordan with full pivoting, core/transform.cpp:82-141. The pivot
// reciprocal is formed in double and rounded (`Float pivinv = 1. / x`).
bool invert(const Mat4 &in, Mat4 *out);

struct Xform {
    Mat4 m, inv;
    Xform() {}
    explicit Xform(const Mat4 &mm) : m(mm) { invert(mm, &inv); }
    Xform(const Mat4 &mm, const Mat4 &ii) : m(mm), inv(ii) {}
    // core/transform.cpp:251-253
    Xform operator*(const Xform &t2) const { return Xform(mul(m, t2.m), mul(t2.inv, inv)); }
    // core/transform.h:217-232 (point, with projective divide)
    V3 point(V3 p) const {
        float x = p.x, y = p.y, z = p.z;
        float xp = m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z + m.m[0][3];
        float yp = m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z + m.m[1][3];
        float zp = m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z + m.m[2][3];
        float wp = m.m[3][0] * x + m.m[3][1] * y + m.m[3][2] * z + m.m[3][3];
        if (wp == 1) return V3(xp, yp, zp);
        return div(V3(xp, yp, zp), wp);
    }
    // core/transform.h:235-241
    V3 vector(V3 v) const {
        float x = v.x, y = v.y, z = v.z;
        return V3(m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z,
                  m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z,
                  m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z);
    }
    // core/transform.h:243-249 (inverse transpose)
    V3 normal(V3 n) const {
        float x = n.x, y = n.y, z = n.z;
        return V3(inv.m[0][0] * x + inv.m[1][0] * y + inv.m[2][0] * z,
                  inv.m[0][1] * x + inv.m[1][1] * y + inv.m[2][1] * z,
                  inv.m[0][2] * x + inv.m[1][2] * y + inv.m[2][2] * z);
    }
    // core/transform.cpp:236-249
    Bounds3 bounds(const Bounds3 &b) const {
        Bounds3 r(point(V3(b.pmin.x, b.pmin.y, b.pmin.z)));
        r = bunion(r, point(V3(b.pmax.x, b.pmin.y, b.pmin.z)));
        r = bunion(r, point(V3(b.pmin.x, b.pmax.y, b.pmin.z)));
        r = bunion(r, point(V3(b.pmin.x, b.pmin.y, b.pmax.z)));
        r = bunion(r, point(V3(b.pmin.x, b.pmax.y, b.pmax.z)));
        r = bunion(r, point(V3(b.pmax.x, b.pmax.y, b.pmin.z)));
        r = bunion(r, point(V3(b.pmax.x, b.pmin.y, b.pmax.z)));
        r = bunion(r, point(V3(b.pmax.x, b.pmax.y, b.pmax.z)));
        return r;
    }
    // core/transform.cpp:255-260
    bool swaps_handedness() const {
        float det = m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[1][2] * m.m[2][1]) -
                    m.m[0][1] * (m.m[1][0] * m.m[2][2] - m.m[1][2] * m.m[2][0]) +
                    m.m[0][2] * (m.m[1][0] * m.m[2][1] - m.m[1][1] * m.m[2][0]);
        return det < 0;
    }
};
inline Xform inverse(const Xform &t) { return Xform(t.inv, t.m); }

Xform xf_translate(V3 d);
Xform xf_scale(float x, float y, float z);
Xform xf_rotate(float theta, V3 axis);
bool xf_lookat(V3 pos, V3 look, V3 up, Xform *out);
Xform xf_perspective(float fov, float n, float f);

}  // namespace iile
