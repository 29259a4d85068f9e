// oracle_path.cpp — CPU restatement of the reference's path-tracing hot path.
//
// *** TEST INFRASTRUCTURE ***  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library. The product
// (libiile_gpu.so) never links, calls or falls back to it.
//
// What it restates (citations relative to /root/reference/src), scalar, one
// sample at a time, float arithmetic in the reference's evaluation order:
//   SamplerIntegrator::Render tile loop        core/integrator.cpp:227-339
//   PathIntegrator::Li                         integrators/path.cpp:64-194
//   UniformSampleOneLight / EstimateDirect     core/integrator.cpp:85-215
//   BVHAccel::Intersect / IntersectP           accelerators/bvh.cpp:662-738
//   Bounds3::IntersectP                        core/geometry.h:1411-1438
//   Triangle::Intersect / IntersectP           shapes/triangle.cpp:188-544
//   Sphere::Intersect / IntersectP / Sample / Pdf   shapes/sphere.cpp:49-306
//   EFloat, Quadratic                          core/efloat.h
//   BSDF, Lambertian, MicrofacetReflection, TrowbridgeReitz, FrDielectric
//                                              core/reflection.{h,cpp}, core/microfacet.cpp
//   DiffuseAreaLight, VisibilityTester, SpawnRay*, OffsetRayOrigin
//   HaltonSampler / GlobalSampler / radical inverses
//                                              samplers/halton.cpp, core/sampler.cpp, core/lowdiscrepancy.cpp
//   PerspectiveCamera::GenerateRayDifferential cameras/perspective.cpp:100-149
//   FilmTile::AddSample / Film::MergeFilmTile  core/film.h:153-193, core/film.cpp:135-148
//
// Pinning (see DESIGN.md "Oracle"): the reference cannot be built in this
// image without a stand-in for the absent glog submodule, so there is no
// oracle/_ref. The restatement is pinned against the reference's own
// known-answer tests and against outputs of the reference recorded in
// SURVEY.md §6/§8c (Halton values, ray / triangle-test / node counts, image
// mean) — tests/test_oracle_pins.py.
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// ----------------------------------------------------------------------------
// constants (core/pbrt.h:196-208, core/rng.h:53)
constexpr float Pi = 3.14159265358979323846f;
constexpr float InvPi = 0.31830988618379067154f;
constexpr float Inv2Pi = 0.15915494309189533577f;
constexpr float PiOver2 = 1.57079632679489661923f;
constexpr float PiOver4 = 0.78539816339744830961f;
constexpr float Infinity = std::numeric_limits<float>::infinity();
constexpr float MachineEpsilon = std::numeric_limits<float>::epsilon() * 0.5f;
constexpr float ShadowEpsilon = 0.0001f;
constexpr float OneMinusEpsilon = 0x1.fffffep-1f;
inline float gamma_n(int n) { return (n * MachineEpsilon) / (1 - n * MachineEpsilon); }

inline uint32_t f2b(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
inline float b2f(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// core/pbrt.h:238-262
inline float next_up(float v) {
    if (std::isinf(v) && v > 0.) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = f2b(v);
    if (v >= 0)
        ++ui;
    else
        --ui;
    return b2f(ui);
}
inline float next_down(float v) {
    if (std::isinf(v) && v < 0.) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = f2b(v);
    if (v > 0)
        --ui;
    else
        ++ui;
    return b2f(ui);
}
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ----------------------------------------------------------------------------
// trigonometry: libm (reference behaviour) or the portable evaluation shared
// with the HIP kernels.
void portable_sincos(double x, double *s, double *c) {
    // Cody-Waite reduction by pi/2 (two-term) + degree-13/12 minimax polynomials
    // on [-pi/4, pi/4] (classic fdlibm coefficients). Plain IEEE double
    // operations, no FMA contraction: restated verbatim in the device code.
    const double k = std::nearbyint(x * 6.36619772367581382433e-01);
    const double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    const double z = r * r;
    const double ps =
        r + r * z *
                (-1.66666666666666324348e-01 +
                 z * (8.33333333332248946124e-03 +
                      z * (-1.98412698298579493134e-04 +
                           z * (2.75573137070700676789e-06 +
                                z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))));
    const double pc =
        (1.0 - 0.5 * z) +
        z * z *
            (4.16666666666666019037e-02 +
             z * (-1.38888888888741095749e-03 +
                  z * (2.48015872894767294178e-05 +
                       z * (-2.75573143513906633035e-07 +
                            z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    switch (int(k) & 3) {
    case 0: *s = ps; *c = pc; break;
    case 1: *s = pc; *c = -ps; break;
    case 2: *s = -ps; *c = -pc; break;
    default: *s = -pc; *c = ps; break;
    }
}
double portable_acos(double x) {
    // rational approximation of asin on [0, 0.5] (fdlibm e_acos.c structure)
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                 pi = 3.14159265358979311600e+00;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const double ax = std::fabs(x);
    if (ax >= 1.0) {
        if (x == 1.0) return 0.0;
        if (x == -1.0) return pi + 2.0 * pio2_lo;
        return std::numeric_limits<double>::quiet_NaN();
    }
    if (ax < 0.5) {
        if (ax < 6.938893903907228e-18) return pio2_hi + pio2_lo;
        const double z = x * x;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (x < 0) {
        const double z = (1.0 + x) * 0.5;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double s = std::sqrt(z);
        const double r = p / q;
        const double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    } else {
        const double z = (1.0 - x) * 0.5;
        const double s = std::sqrt(z);
        uint64_t bits;
        std::memcpy(&bits, &s, 8);
        bits &= 0xffffffff00000000ULL;
        double df;
        std::memcpy(&df, &bits, 8);
        const double c = (z - df * df) / (s + df);
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        const double w = r * s + c;
        return 2.0 * (df + w);
    }
}

// atan / atan2 with the structure and coefficients of fdlibm's s_atan.c / e_atan2.c (double);
// results are rounded once to float by the callers.
double portable_atan(double x) {
    static const double atanhi[] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01,
                                    1.57079632679489655800e+00};
    static const double atanlo[] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17,
                                    6.12323399573676603587e-17};
    static const double aT[] = {3.33333333333329318027e-01,  -1.99999999998764832476e-01, 1.42857142725034663711e-01,
                                -1.11111104054623557880e-01, 9.09088713343650656196e-02,  -7.69187620504482999495e-02,
                                6.66107313738753120669e-02,  -5.83357013379057348645e-02, 4.97687799461593236017e-02,
                                -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    const bool neg = std::signbit(x);
    double ax = std::fabs(x);
    int id;
    if (!(ax < 7.378697629483821e19)) {  // |x| >= 2^66 (or NaN)
        if (x != x) return x + x;
        return neg ? -(atanhi[3] + atanlo[3]) : (atanhi[3] + atanlo[3]);
    }
    if (ax < 0.4375) {
        if (ax < 1.862645149230957e-09) return x;  // |x| < 2^-29
        id = -1;
        ax = x;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) {
            id = 0;
            ax = (2.0 * ax - 1.0) / (2.0 + ax);
        } else {
            id = 1;
            ax = (ax - 1.0) / (ax + 1.0);
        }
    } else if (ax < 2.4375) {
        id = 2;
        ax = (ax - 1.5) / (1.0 + 1.5 * ax);
    } else {
        id = 3;
        ax = -1.0 / ax;
    }
    const double z = ax * ax, w = z * z;
    const double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return ax - ax * (s1 + s2);
    const double r = atanhi[id] - ((ax * (s1 + s2) - atanlo[id]) - ax);
    return neg ? -r : r;
}
double portable_atan2(double y, double x) {
    const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16, pi_o_2 = 1.5707963267948965580E+00,
                 pi_o_4 = 7.8539816339744827900E-01, tiny = 1.0e-300;
    if (x != x || y != y) return x + y;
    if (x == 1.0) return portable_atan(y);
    const int m = (std::signbit(y) ? 1 : 0) | (std::signbit(x) ? 2 : 0);
    if (y == 0) {
        switch (m) {
        case 0:
        case 1: return y;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (x == 0) return std::signbit(y) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (std::isinf(x)) {
        if (std::isinf(y)) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0 * pi_o_4 + tiny;
            default: return -3.0 * pi_o_4 - tiny;
            }
        }
        switch (m) {
        case 0: return 0.0;
        case 1: return -0.0;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (std::isinf(y)) return std::signbit(y) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int ey, ex;
    std::frexp(y, &ey);
    std::frexp(x, &ex);
    const int k = ey - ex;
    double z;
    if (k > 60)
        z = pi_o_2 + 0.5 * pi_lo;
    else if (std::signbit(x) && k < -60)
        z = 0.0;
    else
        z = portable_atan(std::fabs(y / x));
    switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// Natural logarithm in double: x = 2^k * m with sqrt(1/2) < m <= sqrt(2), log(m) from the atanh series in
// s = (m - 1) / (m + 1) with the classic minimax coefficients. Error well below a float ulp; restated
// operation for operation on the device (dmath.h log_d). x must be a positive finite normal double.
double portable_log(double x) {
    if (std::isnan(x) || x < 0) return std::numeric_limits<double>::quiet_NaN();
    if (x == 0) return -std::numeric_limits<double>::infinity();
    if (std::isinf(x)) return x;
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    uint64_t bits;
    std::memcpy(&bits, &x, 8);
    int k = int((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m;
    std::memcpy(&m, &bits, 8);
    if (m > 1.4142135623730951) {
        m *= 0.5;
        k += 1;
    }
    const double f = m - 1.0;
    const double sq = f / (2.0 + f);
    const double z = sq * sq;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = double(k);
    return dk * ln2_hi - ((hfsq - (sq * (hfsq + R) + dk * ln2_lo)) - f);
}

struct Trig {
    int mode;
    float log_f(float x) const {
        if (mode == ORACLE_TRIG_LIBM) return std::log(x);
        return float(portable_log(double(x)));
    }
    float sin_f(float x) const {
        if (mode == ORACLE_TRIG_LIBM) return std::sin(x);
        double s, c;
        portable_sincos(double(x), &s, &c);
        return float(s);
    }
    float cos_f(float x) const {
        if (mode == ORACLE_TRIG_LIBM) return std::cos(x);
        double s, c;
        portable_sincos(double(x), &s, &c);
        return float(c);
    }
    double sin_d(double x) const {
        if (mode == ORACLE_TRIG_LIBM) return ::sin(x);
        double s, c;
        portable_sincos(x, &s, &c);
        return s;
    }
    double cos_d(double x) const {
        if (mode == ORACLE_TRIG_LIBM) return ::cos(x);
        double s, c;
        portable_sincos(x, &s, &c);
        return c;
    }
    float acos_f(float x) const {
        if (mode == ORACLE_TRIG_LIBM) return std::acos(x);
        return float(portable_acos(double(x)));
    }
    float atan2_f(float y, float x) const {
        if (mode == ORACLE_TRIG_LIBM) return std::atan2(y, x);
        return float(portable_atan2(double(y), double(x)));
    }
};

// ----------------------------------------------------------------------------
// vectors (core/geometry.h)
struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float x, float y, float z) : x(x), y(y), z(z) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(float s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, float s) { return V3(s * a.x, s * a.y, s * a.z); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float absdot(V3 a, V3 b) { return std::abs(dot(a, b)); }
inline float length_sq(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline float length(V3 a) { return std::sqrt(length_sq(a)); }
inline V3 vdiv(V3 a, float f) {  // geometry.h:242-246: multiply by float reciprocal
    float inv = 1.f / f;
    return V3(a.x * inv, a.y * inv, a.z * inv);
}
inline V3 normalize(V3 a) { return vdiv(a, length(a)); }
inline V3 vabs(V3 a) { return V3(std::abs(a.x), std::abs(a.y), std::abs(a.z)); }
inline V3 cross(V3 a, V3 b) {  // geometry.h:957-963: evaluated in double
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return V3(float((ay * bz) - (az * by)), float((az * bx) - (ax * bz)), float((ax * by) - (ay * bx)));
}
inline V3 faceforward(V3 n, V3 v) { return (dot(n, v) < 0.f) ? -n : n; }
inline int max_dimension(V3 v) { return (v.x > v.y) ? ((v.x > v.z) ? 0 : 2) : ((v.y > v.z) ? 1 : 2); }
inline float max_component(V3 v) { return std::max(v.x, std::max(v.y, v.z)); }
inline V3 permute(V3 v, int x, int y, int z) { return V3(v[x], v[y], v[z]); }
// geometry.h:1020-1027
inline void coordinate_system(V3 v1, V3 *v2, V3 *v3) {
    if (std::abs(v1.x) > std::abs(v1.y))
        *v2 = vdiv(V3(-v1.z, 0, v1.x), std::sqrt(v1.x * v1.x + v1.z * v1.z));
    else
        *v2 = vdiv(V3(0, v1.z, -v1.y), std::sqrt(v1.y * v1.y + v1.z * v1.z));
    *v3 = cross(v1, *v2);
}

struct Rgb {
    float c[3];
    Rgb(float v = 0.f) { c[0] = c[1] = c[2] = v; }
    Rgb(float r, float g, float b) {
        c[0] = r;
        c[1] = g;
        c[2] = b;
    }
    bool is_black() const { return c[0] == 0. && c[1] == 0. && c[2] == 0.; }
    float y() const { return 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]; }
    float max_component() const { return std::max(std::max(c[0], c[1]), c[2]); }
    bool has_nans() const { return std::isnan(c[0]) || std::isnan(c[1]) || std::isnan(c[2]); }
};
inline Rgb operator+(Rgb a, Rgb b) { return Rgb(a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2]); }
inline Rgb operator*(Rgb a, Rgb b) { return Rgb(a.c[0] * b.c[0], a.c[1] * b.c[1], a.c[2] * b.c[2]); }
inline Rgb operator*(Rgb a, float s) { return Rgb(a.c[0] * s, a.c[1] * s, a.c[2] * s); }
inline Rgb operator/(Rgb a, float s) { return Rgb(a.c[0] / s, a.c[1] / s, a.c[2] / s); }

struct Ray {
    V3 o, d;
    float tmax;
};

// ----------------------------------------------------------------------------
// transforms (core/transform.h:217-410), m row-major
struct M4 {
    const float *a;
    float operator()(int r, int c) const { return a[4 * r + c]; }
};
inline V3 xf_point(M4 m, V3 p) {
    float x = p.x, y = p.y, z = p.z;
    float xp = m(0, 0) * x + m(0, 1) * y + m(0, 2) * z + m(0, 3);
    float yp = m(1, 0) * x + m(1, 1) * y + m(1, 2) * z + m(1, 3);
    float zp = m(2, 0) * x + m(2, 1) * y + m(2, 2) * z + m(2, 3);
    float wp = m(3, 0) * x + m(3, 1) * y + m(3, 2) * z + m(3, 3);
    if (wp == 1) return V3(xp, yp, zp);
    return vdiv(V3(xp, yp, zp), wp);
}
inline V3 xf_point_err(M4 m, V3 p, V3 *err) {  // transform.h:278-300
    float x = p.x, y = p.y, z = p.z;
    float xp = m(0, 0) * x + m(0, 1) * y + m(0, 2) * z + m(0, 3);
    float yp = m(1, 0) * x + m(1, 1) * y + m(1, 2) * z + m(1, 3);
    float zp = m(2, 0) * x + m(2, 1) * y + m(2, 2) * z + m(2, 3);
    float wp = m(3, 0) * x + m(3, 1) * y + m(3, 2) * z + m(3, 3);
    float xs = (std::abs(m(0, 0) * x) + std::abs(m(0, 1) * y) + std::abs(m(0, 2) * z) + std::abs(m(0, 3)));
    float ys = (std::abs(m(1, 0) * x) + std::abs(m(1, 1) * y) + std::abs(m(1, 2) * z) + std::abs(m(1, 3)));
    float zs = (std::abs(m(2, 0) * x) + std::abs(m(2, 1) * y) + std::abs(m(2, 2) * z) + std::abs(m(2, 3)));
    *err = gamma_n(3) * V3(xs, ys, zs);
    if (wp == 1) return V3(xp, yp, zp);
    return vdiv(V3(xp, yp, zp), wp);
}
inline V3 xf_point_err2(M4 m, V3 pt, V3 pe, V3 *err) {  // transform.h:302-331
    float x = pt.x, y = pt.y, z = pt.z;
    float xp = m(0, 0) * x + m(0, 1) * y + m(0, 2) * z + m(0, 3);
    float yp = m(1, 0) * x + m(1, 1) * y + m(1, 2) * z + m(1, 3);
    float zp = m(2, 0) * x + m(2, 1) * y + m(2, 2) * z + m(2, 3);
    float wp = m(3, 0) * x + m(3, 1) * y + m(3, 2) * z + m(3, 3);
    err->x = (gamma_n(3) + 1.f) * (std::abs(m(0, 0)) * pe.x + std::abs(m(0, 1)) * pe.y + std::abs(m(0, 2)) * pe.z) +
             gamma_n(3) * (std::abs(m(0, 0) * x) + std::abs(m(0, 1) * y) + std::abs(m(0, 2) * z) + std::abs(m(0, 3)));
    err->y = (gamma_n(3) + 1.f) * (std::abs(m(1, 0)) * pe.x + std::abs(m(1, 1)) * pe.y + std::abs(m(1, 2)) * pe.z) +
             gamma_n(3) * (std::abs(m(1, 0) * x) + std::abs(m(1, 1) * y) + std::abs(m(1, 2) * z) + std::abs(m(1, 3)));
    err->z = (gamma_n(3) + 1.f) * (std::abs(m(2, 0)) * pe.x + std::abs(m(2, 1)) * pe.y + std::abs(m(2, 2)) * pe.z) +
             gamma_n(3) * (std::abs(m(2, 0) * x) + std::abs(m(2, 1) * y) + std::abs(m(2, 2) * z) + std::abs(m(2, 3)));
    if (wp == 1.) return V3(xp, yp, zp);
    return vdiv(V3(xp, yp, zp), wp);
}
inline V3 xf_vector(M4 m, V3 v) {
    float x = v.x, y = v.y, z = v.z;
    return V3(m(0, 0) * x + m(0, 1) * y + m(0, 2) * z, m(1, 0) * x + m(1, 1) * y + m(1, 2) * z,
              m(2, 0) * x + m(2, 1) * y + m(2, 2) * z);
}
inline V3 xf_vector_err(M4 m, V3 v, V3 *err) {  // transform.h:333-349
    float x = v.x, y = v.y, z = v.z;
    err->x = gamma_n(3) * (std::abs(m(0, 0) * v.x) + std::abs(m(0, 1) * v.y) + std::abs(m(0, 2) * v.z));
    err->y = gamma_n(3) * (std::abs(m(1, 0) * v.x) + std::abs(m(1, 1) * v.y) + std::abs(m(1, 2) * v.z));
    err->z = gamma_n(3) * (std::abs(m(2, 0) * v.x) + std::abs(m(2, 1) * v.y) + std::abs(m(2, 2) * v.z));
    return V3(m(0, 0) * x + m(0, 1) * y + m(0, 2) * z, m(1, 0) * x + m(1, 1) * y + m(1, 2) * z,
              m(2, 0) * x + m(2, 1) * y + m(2, 2) * z);
}
inline V3 xf_normal(M4 minv, V3 n) {  // transform.h:243-249: transpose of the inverse
    float x = n.x, y = n.y, z = n.z;
    return V3(minv(0, 0) * x + minv(1, 0) * y + minv(2, 0) * z, minv(0, 1) * x + minv(1, 1) * y + minv(2, 1) * z,
              minv(0, 2) * x + minv(1, 2) * y + minv(2, 2) * z);
}

// ----------------------------------------------------------------------------
// EFloat (core/efloat.h), NDEBUG flavour (no long double shadow value)
struct EFloat {
    float v, low, high;
    EFloat() : v(0), low(0), high(0) {}
    EFloat(float v_, float err = 0.f) : v(v_) {
        if (err == 0.)
            low = high = v_;
        else {
            low = next_down(v_ - err);
            high = next_up(v_ + err);
        }
    }
};
inline EFloat operator+(EFloat a, EFloat b) {
    EFloat r;
    r.v = a.v + b.v;
    r.low = next_down(a.low + b.low);
    r.high = next_up(a.high + b.high);
    return r;
}
inline EFloat operator-(EFloat a, EFloat b) {
    EFloat r;
    r.v = a.v - b.v;
    r.low = next_down(a.low - b.high);
    r.high = next_up(a.high - b.low);
    return r;
}
inline EFloat operator*(EFloat a, EFloat b) {
    EFloat r;
    r.v = a.v * b.v;
    float prod[4] = {a.low * b.low, a.high * b.low, a.low * b.high, a.high * b.high};
    r.low = next_down(std::min(std::min(prod[0], prod[1]), std::min(prod[2], prod[3])));
    r.high = next_up(std::max(std::max(prod[0], prod[1]), std::max(prod[2], prod[3])));
    return r;
}
inline EFloat operator/(EFloat a, EFloat b) {
    EFloat r;
    r.v = a.v / b.v;
    if (b.low < 0 && b.high > 0) {
        r.low = -Infinity;
        r.high = Infinity;
    } else {
        float div[4] = {a.low / b.low, a.high / b.low, a.low / b.high, a.high / b.high};
        r.low = next_down(std::min(std::min(div[0], div[1]), std::min(div[2], div[3])));
        r.high = next_up(std::max(std::max(div[0], div[1]), std::max(div[2], div[3])));
    }
    return r;
}
// efloat.h:267-285
inline bool ef_quadratic(EFloat A, EFloat B, EFloat C, EFloat *t0, EFloat *t1) {
    double discrim = (double)B.v * (double)B.v - 4. * (double)A.v * (double)C.v;
    if (discrim < 0.) return false;
    double root = std::sqrt(discrim);
    EFloat froot(float(root), float(MachineEpsilon * root));
    EFloat q;
    if (B.v < 0)
        q = EFloat(-.5f) * (B - froot);
    else
        q = EFloat(-.5f) * (B + froot);
    *t0 = q / A;
    *t1 = C / q;
    if (t0->v > t1->v) std::swap(*t0, *t1);
    return true;
}

// ----------------------------------------------------------------------------
// counters (one per worker thread; padded so neighbours never share a cache line)
struct alignas(128) Counters {
    uint64_t camera_rays = 0, regular_rays = 0, shadow_rays = 0, tri_tests = 0, tri_hits = 0,
             sphere_tests = 0, nodes_closest = 0, nodes_any = 0, nee_evals = 0, zero_radiance = 0;
    uint64_t path_length[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int max_stack = 0;
    void add(const Counters &o) {
        camera_rays += o.camera_rays;
        regular_rays += o.regular_rays;
        shadow_rays += o.shadow_rays;
        tri_tests += o.tri_tests;
        tri_hits += o.tri_hits;
        sphere_tests += o.sphere_tests;
        nodes_closest += o.nodes_closest;
        nodes_any += o.nodes_any;
        nee_evals += o.nee_evals;
        zero_radiance += o.zero_radiance;
        for (int i = 0; i < 8; ++i) path_length[i] += o.path_length[i];
        max_stack = std::max(max_stack, o.max_stack);
    }
};

// What Li needs of a SurfaceInteraction (core/interaction.h)
struct Isect {
    int prim = -1;
    float t = 0, b0 = 0, b1 = 0, b2 = 0;
    V3 p, perr, n, wo;
    V3 sn;     // shading.n
    V3 sdpdu;  // shading.dpdu
    // what texture lookups need (triangles only): uv, dpdu, dpdv and, after compute_differentials, du/dx ...
    float uv[2] = {0, 0};
    V3 dpdu, dpdv;
    float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0;
    V3 dpdx, dpdy;  // interaction.cpp:117-118 (zero without differentials): the direct pass's reflected-ray differentials use them
    // bump mapping: the rest of the shading geometry (shading.dpdv, shading.dndu / dndv) and the orientation flag
    V3 sdpdv, dndu, dndv;
    bool flip = false;
};

// the auxiliary rays of a RayDifferential (geometry.h:890-925)
struct RayDiff {
    bool has = false;
    V3 rxo, ryo, rxd, ryd;
};

// Inverse(Matrix4x4), transform.cpp:82-141: Gauss-Jordan elimination with full pivoting
static bool invert4(const float in[16], float out[16]) {
    int indxc[4], indxr[4];
    int ipiv[4] = {0, 0, 0, 0};
    float a[4][4];
    std::memcpy(a, in, sizeof(a));
    for (int i = 0; i < 4; i++) {
        int irow = 0, icol = 0;
        float big = 0.f;
        for (int j = 0; j < 4; j++) {
            if (ipiv[j] != 1) {
                for (int k = 0; k < 4; k++) {
                    if (ipiv[k] == 0) {
                        if (std::abs(a[j][k]) >= big) {
                            big = float(std::abs(a[j][k]));
                            irow = j;
                            icol = k;
                        }
                    } else if (ipiv[k] > 1)
                        return false;
                }
            }
        }
        ++ipiv[icol];
        if (irow != icol)
            for (int k = 0; k < 4; ++k) std::swap(a[irow][k], a[icol][k]);
        indxr[i] = irow;
        indxc[i] = icol;
        if (a[icol][icol] == 0.f) return false;
        float pivinv = float(1. / a[icol][icol]);  // `Float pivinv = 1. / minv[icol][icol]`: a double divide
        a[icol][icol] = 1.;
        for (int j = 0; j < 4; j++) a[icol][j] *= pivinv;
        for (int j = 0; j < 4; j++) {
            if (j != icol) {
                float save = a[j][icol];
                a[j][icol] = 0;
                for (int k = 0; k < 4; k++) a[j][k] -= a[icol][k] * save;
            }
        }
    }
    for (int j = 3; j >= 0; j--) {
        if (indxr[j] != indxc[j])
            for (int k = 0; k < 4; k++) std::swap(a[k][indxr[j]], a[k][indxc[j]]);
    }
    std::memcpy(out, a, sizeof(a));
    return true;
}

// CreateHemisphericCamera (hemispheric.cpp:109-160): the probe's CameraToWorld = LookAt(pos, pos + dir, up)'s
// cameraToWorld (transform.cpp:203-236), and its WorldToCamera = Transform(Inverse(cameraToWorld)) whose own inverse
// (a second numerical inversion) is what transforms normals (transform.h:243-249)
struct ProbeCam {
    float c2w[16];
    float w2c_minv[16];  // mInv of WorldToCamera
    int hemi_size;
};
static bool make_probe_camera(const float pos[3], const float dir[3], int hemi_size, ProbeCam *cam) {
    const V3 up = (dir[0] == 0.0 && dir[1] == 0.0) ? V3(0.f, 1.f, 0.f) : V3(0.f, 0.f, 1.f);
    const V3 p(pos[0], pos[1], pos[2]);
    const V3 look(pos[0] + dir[0], pos[1] + dir[1], pos[2] + dir[2]);
    float m[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // Matrix4x4() is the identity ...
    m[0] = m[5] = m[10] = m[15] = 1;
    m[3] = p.x;
    m[7] = p.y;
    m[11] = p.z;
    m[15] = 1;
    V3 d = normalize(look - p);
    if (length(cross(normalize(up), d)) == 0) return false;
    V3 right = normalize(cross(normalize(up), d));
    V3 new_up = cross(d, right);
    m[0] = right.x, m[4] = right.y, m[8] = right.z, m[12] = 0.f;
    m[1] = new_up.x, m[5] = new_up.y, m[9] = new_up.z, m[13] = 0.f;
    m[2] = d.x, m[6] = d.y, m[10] = d.z, m[14] = 0.f;
    std::memcpy(cam->c2w, m, sizeof(m));
    float inv[16];
    if (!invert4(m, inv)) return false;           // cameraTransform->GetInverseMatrix()
    if (!invert4(inv, cam->w2c_minv)) return false;  // WorldToCamera = Transform(that): its mInv
    cam->hemi_size = hemi_size;
    return true;
}

struct Oracle {
    const iile_scene_desc &S;
    Trig trig;
    Counters *ctr;
    const ProbeCam *probe = nullptr;  // probe pass: HemisphericCamera + IISPTdIntegrator::Li
    float probe_aux[4] = {0, 0, 0, -1.f};  // camera-space normal and distance of the last camera ray's first hit
    Oracle(const iile_scene_desc &s, int mode, Counters *c) : S(s), trig{mode}, ctr(c) {}

    // Camera::GenerateRayDifferential (camera.cpp:60-96) for the hemispheric camera: the rays through film points
    // shifted by eps = 0.05 in x and in y, differenced (the render loop's ScaleDifferentials is by 1 / sqrt(1 spp))
    RayDiff probe_differentials(float pfx, float pfy, const Ray &r) const {
        RayDiff rd;
        const float eps = .05f;  // `for (Float eps : {.05, -.05})`: the first shift always succeeds (weight 1)
        Ray rx = probe_ray(pfx + eps, pfy), ry = probe_ray(pfx, pfy + eps);
        rd.has = true;
        rd.rxo = r.o + vdiv(rx.o - r.o, eps);
        rd.rxd = r.d + vdiv(rx.d - r.d, eps);
        rd.ryo = r.o + vdiv(ry.o - r.o, eps);
        rd.ryd = r.d + vdiv(ry.d - r.d, eps);
        const float sc = 1 / std::sqrt(float(S.halton.spp));
        rd.rxo = r.o + (rd.rxo - r.o) * sc;
        rd.ryo = r.o + (rd.ryo - r.o) * sc;
        rd.rxd = r.d + (rd.rxd - r.d) * sc;
        rd.ryd = r.d + (rd.ryd - r.d) * sc;
        return rd;
    }
    // HemisphericCamera::GenerateRay (hemispheric.cpp:15-41) + Transform::operator()(Ray) (transform.h:251-264)
    Ray probe_ray(float pfx, float pfy) const {
        float theta = Pi * pfy / float(probe->hemi_size);
        float phi = Pi * pfx / float(probe->hemi_size);
        V3 dir(trig.sin_f(theta) * trig.cos_f(phi), trig.cos_f(theta), trig.sin_f(theta) * trig.sin_f(phi));
        M4 m{probe->c2w};
        V3 oerr;
        V3 o = xf_point_err(m, V3(0, 0, 0), &oerr);
        V3 d = xf_vector(m, dir);
        float len2 = length_sq(d);
        float tmax = Infinity;
        if (len2 > 0) {
            float dt = dot(vabs(d), oerr) / len2;
            o = o + d * dt;
            tmax -= dt;
        }
        return Ray{o, d, tmax};
    }

    // ------------------------------------------------------------------------
    // Halton (samplers/halton.cpp:96-127, core/lowdiscrepancy.cpp:389-427)
    static uint64_t inverse_radical_inverse(int base, uint64_t inverse, int n_digits) {
        uint64_t index = 0;
        for (int i = 0; i < n_digits; ++i) {
            uint64_t digit = inverse % base;
            inverse /= base;
            index = index * base + digit;
        }
        return index;
    }
    int64_t halton_index(int px, int py, int64_t sample_num) const {
        const iile_halton &h = S.halton;
        int64_t offset = 0;
        if (h.sample_stride > 1) {
            auto mod = [](int a, int b) {
                int r = a - (a / b) * b;
                return r < 0 ? r + b : r;
            };
            int pm[2] = {mod(px, 128), mod(py, 128)};
            for (int i = 0; i < 2; ++i) {
                uint64_t dim_offset = inverse_radical_inverse(i == 0 ? 2 : 3, uint64_t(pm[i]), h.base_exponents[i]);
                offset += dim_offset * uint64_t(h.sample_stride / h.base_scales[i]) * uint64_t(h.mult_inverse[i]);
            }
            offset %= h.sample_stride;
        }
        return offset + sample_num * h.sample_stride;
    }
    static uint32_t reverse_bits32(uint32_t n) {
        n = (n << 16) | (n >> 16);
        n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8);
        n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4);
        n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2);
        n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1);
        return n;
    }
    static float radical_inverse(int base_index, int base, uint64_t a) {
        if (base_index == 0) {
            uint64_t n0 = reverse_bits32(uint32_t(a));
            uint64_t n1 = reverse_bits32(uint32_t(a >> 32));
            uint64_t rev = (n0 << 32) | n1;
            return float(double(rev) * 0x1p-64);
        }
        const float inv_base = 1.f / float(base);
        uint64_t reversed = 0;
        float inv_base_n = 1;
        while (a) {
            uint64_t next = a / base;
            uint64_t digit = a - next * base;
            reversed = reversed * base + digit;
            inv_base_n *= inv_base;
            a = next;
        }
        return std::min(float(reversed) * inv_base_n, OneMinusEpsilon);
    }
    static float scrambled_radical_inverse(int base, const uint16_t *perm, uint64_t a) {
        const float inv_base = 1.f / float(base);
        uint64_t reversed = 0;
        float inv_base_n = 1;
        while (a) {
            uint64_t next = a / base;
            uint64_t digit = a - next * base;
            reversed = reversed * base + perm[digit];
            inv_base_n *= inv_base;
            a = next;
        }
        return std::min(inv_base_n * (float(reversed) + inv_base * perm[0] / (1 - inv_base)), OneMinusEpsilon);
    }
    // ------------------------------------------------------------------------
    // Sobol' (samplers/sobol.cpp:42-59, core/lowdiscrepancy.h:229-274). The generator matrices are scene data built on
    // the host (iile_sobol: 32 columns per dimension, sample indices below 2^32); tests/test_sobol.py holds them against
    // the reference's tables.
    // SobolIntervalToIndex, lowdiscrepancy.h:229-252
    uint64_t sobol_interval_to_index(uint32_t m, uint64_t frame, int px, int py) const {
        if (m == 0) return 0;
        const iile_sobol &sb = S.sobol;
        const uint32_t m2 = m << 1;
        uint64_t index = uint64_t(frame) << m2;
        uint64_t delta = 0;
        for (int c = 0; frame; frame >>= 1, ++c)
            if (frame & 1) delta ^= sb.vdc[c];  // Add flipped column m + c + 1.
        uint64_t b = (((uint64_t)((uint32_t)px) << m) | ((uint32_t)py)) ^ delta;  // flipped b
        for (int c = 0; b; b >>= 1, ++c)
            if (b & 1) index ^= sb.vdc_inv[c];  // Add column 2 * m - c.
        return index;
    }
    // SobolSampleFloat, lowdiscrepancy.h:262-274 (scramble 0)
    float sobol_sample_float(int64_t a, int dimension) const {
        uint32_t v = 0;
        for (int i = dimension * 32; a != 0; a >>= 1, i++)
            if (a & 1) v ^= S.sobol.matrices32[i];
        return std::min(v * 0x1p-32f /* 1/2^32 */, OneMinusEpsilon);
    }
    // SobolSampler::GetIndexForSample / SampleDimension, sobol.cpp:42-59
    int64_t sobol_index(int px, int py, int64_t sample_num) const {
        return int64_t(sobol_interval_to_index(uint32_t(S.sobol.log2_resolution), uint64_t(sample_num), px - S.film.samp_x0, py - S.film.samp_y0));
    }
    float sobol_sample_dimension(int64_t index, int dim, int px, int py) const {
        float s = sobol_sample_float(index, dim);
        // Remap Sobol' dimensions used for pixel samples
        if (dim == 0 || dim == 1) {
            const int pmin = dim == 0 ? S.film.samp_x0 : S.film.samp_y0, cur = dim == 0 ? px : py;
            s = s * S.sobol.resolution + pmin;
            s = std::min(std::max(s - cur, 0.f), OneMinusEpsilon);  // Clamp(s - currentPixel[dim], 0, OneMinusEpsilon)
        }
        return s;
    }
    // the scene's sampler: GetIndexForSample / SampleDimension of HaltonSampler or SobolSampler
    int64_t sample_index(int px, int py, int64_t sample_num) const {
        return S.sobol.enabled && !probe ? sobol_index(px, py, sample_num) : halton_index(px, py, sample_num);
    }
    float sample_dimension(int64_t index, int dim, int px = 0, int py = 0) const {
        if (S.sobol.enabled && !probe) return sobol_sample_dimension(index, dim, px, py);
        const iile_halton &h = S.halton;
        if (h.sample_at_pixel_center && (dim == 0 || dim == 1)) return 0.5f;  // halton.cpp:119
        if (dim == 0) return radical_inverse(0, 2, uint64_t(index >> h.base_exponents[0]));
        if (dim == 1) return radical_inverse(1, 3, uint64_t(index / h.base_scales[1]));
        return scrambled_radical_inverse(h.primes[dim], h.perms + h.prime_sums[dim], uint64_t(index));
    }
    // GlobalSampler::Get1D/Get2D with no sample arrays requested
    // (core/sampler.cpp:180-195: arrayStartDim == arrayEndDim == 5, nothing is skipped)
    struct Sampler {
        const Oracle *o;
        int64_t index;
        int dim;
        int px, py;  // currentPixel (SobolSampler's dimensions 0 and 1 are relative to it)
        float get1d() { return o->sample_dimension(index, dim++, px, py); }
        void get2d(float *u) {
            u[0] = o->sample_dimension(index, dim, px, py);
            u[1] = o->sample_dimension(index, dim + 1, px, py);
            dim += 2;
        }
    };

    // ------------------------------------------------------------------------
    // sampling warps (core/sampling.cpp:113-130, core/sampling.h:159-163)
    void concentric_sample_disk(const float *u, float *dx, float *dy) const {
        float ox = 2.f * u[0] - 1, oy = 2.f * u[1] - 1;
        if (ox == 0 && oy == 0) {
            *dx = 0;
            *dy = 0;
            return;
        }
        float theta, r;
        if (std::abs(ox) > std::abs(oy)) {
            r = ox;
            theta = PiOver4 * (oy / ox);
        } else {
            r = oy;
            theta = PiOver2 - PiOver4 * (ox / oy);
        }
        *dx = r * trig.cos_f(theta);
        *dy = r * trig.sin_f(theta);
    }
    V3 cosine_sample_hemisphere(const float *u) const {
        float dx, dy;
        concentric_sample_disk(u, &dx, &dy);
        float z = std::sqrt(std::max(0.f, 1 - dx * dx - dy * dy));
        return V3(dx, dy, z);
    }

    // ------------------------------------------------------------------------
    // camera (cameras/perspective.cpp:100-149; transform.h:251-264)
    // with `rd`: GenerateRayDifferential (perspective.cpp:100-149) followed by the render loop's
    // ScaleDifferentials(1 / sqrt(spp)) (integrator.cpp:284-285, geometry.h:908-913)
    Ray camera_ray(float pfx, float pfy, const float *plens, RayDiff *rd = nullptr, bool unit_diff_scale = false,
                   float diff_scale_override = 0.f) const {
        const iile_camera &c = S.camera;
        V3 pcam = xf_point(M4{c.raster_to_camera}, V3(pfx, pfy, 0));
        V3 dir = normalize(V3(pcam.x, pcam.y, pcam.z));
        Ray r{V3(0, 0, 0), dir, Infinity};
        if (c.lens_radius > 0) {
            float lx, ly;
            concentric_sample_disk(plens, &lx, &ly);
            lx = c.lens_radius * lx;
            ly = c.lens_radius * ly;
            float ft = c.focal_distance / r.d.z;
            V3 pfocus = r.o + r.d * ft;
            r.o = V3(lx, ly, 0);
            r.d = normalize(pfocus - r.o);
        }
        V3 rxo, ryo, rxd, ryd;
        if (rd) {
            const V3 dxc(c.dx_camera[0], c.dx_camera[1], c.dx_camera[2]), dyc(c.dy_camera[0], c.dy_camera[1], c.dy_camera[2]);
            if (c.lens_radius > 0) {
                float lx, ly;
                concentric_sample_disk(plens, &lx, &ly);
                lx = c.lens_radius * lx;
                ly = c.lens_radius * ly;
                V3 dx = normalize(pcam + dxc);
                float ft = c.focal_distance / dx.z;
                V3 pfocus = V3(0, 0, 0) + (ft * dx);
                rxo = V3(lx, ly, 0);
                rxd = normalize(pfocus - rxo);
                V3 dy = normalize(pcam + dyc);
                ft = c.focal_distance / dy.z;
                pfocus = V3(0, 0, 0) + (ft * dy);
                ryo = V3(lx, ly, 0);
                ryd = normalize(pfocus - ryo);
            } else {
                rxo = ryo = r.o;
                rxd = normalize(pcam + dxc);
                ryd = normalize(pcam + dyc);
            }
        }
        M4 m{c.camera_to_world};
        V3 oerr;
        V3 o = xf_point_err(m, r.o, &oerr);
        V3 d = xf_vector(m, r.d);
        float len2 = length_sq(d);
        float tmax = r.tmax;
        if (len2 > 0) {
            float dt = dot(vabs(d), oerr) / len2;
            o = o + d * dt;
            tmax -= dt;
        }
        if (rd) {  // Transform::operator()(RayDifferential), transform.h:265-274
            rxo = xf_point(m, rxo);
            ryo = xf_point(m, ryo);
            rxd = xf_vector(m, rxd);
            ryd = xf_vector(m, ryd);
            // ScaleDifferentials(1 / sqrt(samplesPerPixel)) in the render loop (integrator.cpp:284-285); the IISPT runner
            // scales by 1.0 (iisptrenderrunner.cpp:272)
            const float sc = diff_scale_override > 0 ? diff_scale_override : (unit_diff_scale ? 1.f : 1 / std::sqrt(float(S.halton.spp)));
            rd->has = true;
            rd->rxo = o + (rxo - o) * sc;
            rd->ryo = o + (ryo - o) * sc;
            rd->rxd = d + (rxd - d) * sc;
            rd->ryd = d + (ryd - d) * sc;
        }
        return Ray{o, d, tmax};
    }

    // SurfaceInteraction::ComputeDifferentials (interaction.cpp:103-149); only du/dx ... are kept
    // (dpdx, dpdy feed bump mapping and the non-uv mappings only)
    static bool solve_2x2(const float A[2][2], const float B[2], float *x0, float *x1) {  // transform.cpp:41-49
        float det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
        if (std::abs(det) < 1e-10f) return false;
        *x0 = (A[1][1] * B[0] - A[0][1] * B[1]) / det;
        *x1 = (A[0][0] * B[1] - A[1][0] * B[0]) / det;
        if (std::isnan(*x0) || std::isnan(*x1)) return false;
        return true;
    }
    static void compute_differentials(Isect *is, const RayDiff &rd) {
        is->dudx = is->dvdx = is->dudy = is->dvdy = 0;
        is->dpdx = is->dpdy = V3(0, 0, 0);
        if (!rd.has) return;
        const V3 n = is->n, p = is->p;
        float d = dot(n, p);
        float tx = -(dot(n, rd.rxo) - d) / dot(n, rd.rxd);
        if (std::isinf(tx) || std::isnan(tx)) return;
        V3 px = rd.rxo + tx * rd.rxd;
        float ty = -(dot(n, rd.ryo) - d) / dot(n, rd.ryd);
        if (std::isinf(ty) || std::isnan(ty)) return;
        V3 py = rd.ryo + ty * rd.ryd;
        is->dpdx = px - p;
        is->dpdy = py - p;
        int dim[2];
        if (std::abs(n.x) > std::abs(n.y) && std::abs(n.x) > std::abs(n.z)) {
            dim[0] = 1;
            dim[1] = 2;
        } else if (std::abs(n.y) > std::abs(n.z)) {
            dim[0] = 0;
            dim[1] = 2;
        } else {
            dim[0] = 0;
            dim[1] = 1;
        }
        float A[2][2] = {{is->dpdu[dim[0]], is->dpdv[dim[0]]}, {is->dpdu[dim[1]], is->dpdv[dim[1]]}};
        float Bx[2] = {px[dim[0]] - p[dim[0]], px[dim[1]] - p[dim[1]]};
        float By[2] = {py[dim[0]] - p[dim[0]], py[dim[1]] - p[dim[1]]};
        if (!solve_2x2(A, Bx, &is->dudx, &is->dvdx)) is->dudx = is->dvdx = 0;
        if (!solve_2x2(A, By, &is->dudy, &is->dvdy)) is->dudy = is->dvdy = 0;
    }

    // ------------------------------------------------------------------------
    // ImageTexture<RGBSpectrum, Spectrum>::Evaluate over UVMapping2D (imagemap.h:87-94, texture.cpp:93-99)
    // and MIPMap<RGBSpectrum>::Lookup / triangle / EWA / Texel (mipmap.h:210-355) on the host-built pyramid
    Rgb tex_texel(const iile_texture &t, int level, int s, int tt) const {
        const int w = t.level_w[level], h = t.level_h[level];
        auto mod = [](int a, int b) {
            int r = a - (a / b) * b;
            return r < 0 ? r + b : r;
        };
        switch (t.wrap) {
        case IILE_WRAP_REPEAT:
            s = mod(s, w);
            tt = mod(tt, h);
            break;
        case IILE_WRAP_CLAMP:
            s = s < 0 ? 0 : (s > w - 1 ? w - 1 : s);
            tt = tt < 0 ? 0 : (tt > h - 1 ? h - 1 : tt);
            break;
        default:
            if (s < 0 || s >= w || tt < 0 || tt >= h) return Rgb(0.f);
        }
        const float *c = S.texels + 3 * (t.level_offset[level] + int64_t(tt) * w + s);
        return Rgb(c[0], c[1], c[2]);
    }
    Rgb tex_triangle(const iile_texture &t, int level, const float st[2]) const {
        level = level < 0 ? 0 : (level > t.n_levels - 1 ? t.n_levels - 1 : level);
        float s = st[0] * t.level_w[level] - 0.5f;
        float tt = st[1] * t.level_h[level] - 0.5f;
        int s0 = int(std::floor(s)), t0 = int(std::floor(tt));
        float ds = s - s0, dt = tt - t0;
        return tex_texel(t, level, s0, t0) * ((1 - ds) * (1 - dt)) + tex_texel(t, level, s0, t0 + 1) * ((1 - ds) * dt) +
               tex_texel(t, level, s0 + 1, t0) * (ds * (1 - dt)) + tex_texel(t, level, s0 + 1, t0 + 1) * (ds * dt);
    }
    float log2_f(float x) const {  // pbrt.h:325-328
        const float inv_log2 = 1.442695040888963387004650940071f;
        return trig.log_f(x) * inv_log2;
    }
    static Rgb lerp_rgb(float t, Rgb a, Rgb b) { return a * (1 - t) + b * t; }
    Rgb tex_lookup_width(const iile_texture &t, const float st[2], float width) const {  // mipmap.h:233-250
        float level = t.n_levels - 1 + log2_f(std::max(width, 1e-8f));
        if (level < 0) return tex_triangle(t, 0, st);
        if (level >= t.n_levels - 1) return tex_texel(t, t.n_levels - 1, 0, 0);
        int il = int(std::floor(level));
        float delta = level - il;
        return lerp_rgb(delta, tex_triangle(t, il, st), tex_triangle(t, il + 1, st));
    }
    Rgb tex_ewa(const iile_texture &t, int level, const float st_in[2], const float d0_in[2], const float d1_in[2]) const {
        if (level >= t.n_levels) return tex_texel(t, t.n_levels - 1, 0, 0);
        const int w = t.level_w[level], h = t.level_h[level];
        float st[2] = {st_in[0] * w - 0.5f, st_in[1] * h - 0.5f};
        float d0[2] = {d0_in[0] * w, d0_in[1] * h}, d1[2] = {d1_in[0] * w, d1_in[1] * h};
        float A = d0[1] * d0[1] + d1[1] * d1[1] + 1;
        float B = -2 * (d0[0] * d0[1] + d1[0] * d1[1]);
        float C = d0[0] * d0[0] + d1[0] * d1[0] + 1;
        float invF = 1 / (A * C - B * B * 0.25f);
        A *= invF;
        B *= invF;
        C *= invF;
        float det = -B * B + 4 * A * C;
        float inv_det = 1 / det;
        float u_sqrt = std::sqrt(det * C), v_sqrt = std::sqrt(A * det);
        int s0 = int(std::ceil(st[0] - 2 * inv_det * u_sqrt));
        int s1 = int(std::floor(st[0] + 2 * inv_det * u_sqrt));
        int t0 = int(std::ceil(st[1] - 2 * inv_det * v_sqrt));
        int t1 = int(std::floor(st[1] + 2 * inv_det * v_sqrt));
        Rgb sum(0.f);
        float sum_wts = 0;
        for (int it = t0; it <= t1; ++it) {
            float tt = it - st[1];
            for (int is = s0; is <= s1; ++is) {
                float ss = is - st[0];
                float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
                if (r2 < 1) {
                    int index = std::min(int(r2 * IILE_EWA_LUT_SIZE), IILE_EWA_LUT_SIZE - 1);
                    float weight = S.ewa_lut[index];
                    sum = sum + tex_texel(t, level, is, it) * weight;
                    sum_wts += weight;
                }
            }
        }
        return sum / sum_wts;
    }
    Rgb tex_evaluate(int tex, const Isect &is) const {
        const iile_texture &t = S.textures[tex];
        float d0[2] = {t.su * is.dudx, t.sv * is.dvdx}, d1[2] = {t.su * is.dudy, t.sv * is.dvdy};
        const float st[2] = {t.su * is.uv[0] + t.du, t.sv * is.uv[1] + t.dv};
        if (t.trilinear) {
            float width = std::max(std::max(std::abs(d0[0]), std::abs(d0[1])), std::max(std::abs(d1[0]), std::abs(d1[1])));
            return tex_lookup_width(t, st, 2 * width);
        }
        if (d0[0] * d0[0] + d0[1] * d0[1] < d1[0] * d1[0] + d1[1] * d1[1]) {
            std::swap(d0[0], d1[0]);
            std::swap(d0[1], d1[1]);
        }
        float major = std::sqrt(d0[0] * d0[0] + d0[1] * d0[1]);
        float minor = std::sqrt(d1[0] * d1[0] + d1[1] * d1[1]);
        if (minor * t.max_aniso < major && minor > 0) {
            float scale = major / (minor * t.max_aniso);
            d1[0] *= scale;
            d1[1] *= scale;
            minor *= scale;
        }
        if (minor == 0) return tex_triangle(t, 0, st);
        float lod = std::max(0.f, t.n_levels - 1.f + log2_f(minor));
        int ilod = int(std::floor(lod));
        return lerp_rgb(lod - ilod, tex_ewa(t, ilod, st, d0, d1), tex_ewa(t, ilod + 1, st, d0, d1));
    }

    // ------------------------------------------------------------------------
    // OffsetRayOrigin / SpawnRay / SpawnRayTo (geometry.h:1440-1460, interaction.h:64-78)
    static V3 offset_ray_origin(V3 p, V3 perr, V3 n, V3 w) {
        float d = dot(vabs(n), perr);
        V3 offset = d * n;
        if (dot(w, n) < 0) offset = -offset;
        V3 po = p + offset;
        for (int i = 0; i < 3; ++i) {
            if (offset[i] > 0)
                po[i] = next_up(po[i]);
            else if (offset[i] < 0)
                po[i] = next_down(po[i]);
        }
        return po;
    }
    static Ray spawn_ray(const Isect &it, V3 d) { return Ray{offset_ray_origin(it.p, it.perr, it.n, d), d, Infinity}; }

    // ------------------------------------------------------------------------
    // Triangle (shapes/triangle.cpp:188-403, 405-544)
    // alpha_mode: 0 = testAlphaTexture false (Shape::Pdf's Intersect, shape.cpp:72-87); 1 = Triangle::Intersect
    // (alphaMask, triangle.cpp:325-331); 2 = Triangle::IntersectP (alphaMask and shadowAlphaMask, :509-541)
    bool triangle_test(const Ray &ray, int prim, float *t_out, float *b0o, float *b1o, float *b2o, int alpha_mode = 0) const {
        ++ctr->tri_tests;
        const float *tp = S.tri_p + 9 * size_t(prim);
        V3 p0(tp[0], tp[1], tp[2]), p1(tp[3], tp[4], tp[5]), p2(tp[6], tp[7], tp[8]);
        V3 p0t = p0 - ray.o, p1t = p1 - ray.o, p2t = p2 - ray.o;
        int kz = max_dimension(vabs(ray.d));
        int kx = kz + 1;
        if (kx == 3) kx = 0;
        int ky = kx + 1;
        if (ky == 3) ky = 0;
        V3 d = permute(ray.d, kx, ky, kz);
        p0t = permute(p0t, kx, ky, kz);
        p1t = permute(p1t, kx, ky, kz);
        p2t = permute(p2t, kx, ky, kz);
        float Sx = -d.x / d.z, Sy = -d.y / d.z, Sz = 1.f / d.z;
        p0t.x += Sx * p0t.z;
        p0t.y += Sy * p0t.z;
        p1t.x += Sx * p1t.z;
        p1t.y += Sy * p1t.z;
        p2t.x += Sx * p2t.z;
        p2t.y += Sy * p2t.z;
        float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
        float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
        float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
        if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
            double p2txp1ty = (double)p2t.x * (double)p1t.y;
            double p2typ1tx = (double)p2t.y * (double)p1t.x;
            e0 = (float)(p2typ1tx - p2txp1ty);
            double p0txp2ty = (double)p0t.x * (double)p2t.y;
            double p0typ2tx = (double)p0t.y * (double)p2t.x;
            e1 = (float)(p0typ2tx - p0txp2ty);
            double p1txp0ty = (double)p1t.x * (double)p0t.y;
            double p1typ0tx = (double)p1t.y * (double)p0t.x;
            e2 = (float)(p1typ0tx - p1txp0ty);
        }
        if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
        float det = e0 + e1 + e2;
        if (det == 0) return false;
        p0t.z *= Sz;
        p1t.z *= Sz;
        p2t.z *= Sz;
        float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
        if (det < 0 && (t_scaled >= 0 || t_scaled < ray.tmax * det))
            return false;
        else if (det > 0 && (t_scaled <= 0 || t_scaled > ray.tmax * det))
            return false;
        float inv_det = 1 / det;
        float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
        float t = t_scaled * inv_det;
        float max_zt = max_component(vabs(V3(p0t.z, p1t.z, p2t.z)));
        float delta_z = gamma_n(3) * max_zt;
        float max_xt = max_component(vabs(V3(p0t.x, p1t.x, p2t.x)));
        float max_yt = max_component(vabs(V3(p0t.y, p1t.y, p2t.y)));
        float delta_x = gamma_n(5) * (max_xt + max_zt);
        float delta_y = gamma_n(5) * (max_yt + max_zt);
        float delta_e = 2 * (gamma_n(2) * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
        float max_e = max_component(vabs(V3(e0, e1, e2)));
        float delta_t = 3 * (gamma_n(3) * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * std::abs(inv_det);
        if (t <= delta_t) return false;
        if (alpha_mode != 0 && (S.prim_flags[prim] & IILE_PRIM_HAS_ALPHA) && S.prim_alpha) {
            // isectLocal carries uvHit and zero differentials: an ImageTexture filters bilinearly at level 0
            float uv[3][2] = {{0, 0}, {1, 0}, {1, 1}};
            if (S.prim_flags[prim] & IILE_PRIM_HAS_UV) {
                const float *u = S.tri_uv + 6 * size_t(prim);
                for (int i = 0; i < 3; ++i) {
                    uv[i][0] = u[2 * i];
                    uv[i][1] = u[2 * i + 1];
                }
            }
            Isect local;
            local.uv[0] = b0 * uv[0][0] + b1 * uv[1][0] + b2 * uv[2][0];
            local.uv[1] = b0 * uv[0][1] + b1 * uv[1][1] + b2 * uv[2][1];
            for (int k = 0; k < alpha_mode; ++k) {
                const int mask = S.prim_alpha[2 * size_t(prim) + k];
                if (mask == IILE_ALPHA_ZERO) return false;
                if (mask >= 0 && tex_evaluate(mask, local).c[0] == 0) return false;
            }
        }
        *t_out = t;
        *b0o = b0;
        *b1o = b1;
        *b2o = b2;
        ++ctr->tri_hits;
        return true;
    }
    // the SurfaceInteraction part of Triangle::Intersect (triangle.cpp:277-400)
    void triangle_interaction(const Ray &ray, int prim, float b0, float b1, float b2, Isect *is) const {
        const float *tp = S.tri_p + 9 * size_t(prim);
        V3 p0(tp[0], tp[1], tp[2]), p1(tp[3], tp[4], tp[5]), p2(tp[6], tp[7], tp[8]);
        const uint32_t flags = S.prim_flags[prim];
        float uv[3][2] = {{0, 0}, {1, 0}, {1, 1}};  // triangle.h:98-108
        if (flags & IILE_PRIM_HAS_UV) {
            const float *u = S.tri_uv + 6 * size_t(prim);
            for (int i = 0; i < 3; ++i) {
                uv[i][0] = u[2 * i];
                uv[i][1] = u[2 * i + 1];
            }
        }
        float duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]};
        float duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
        V3 dp02 = p0 - p2, dp12 = p1 - p2;
        float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
        bool degenerate = std::abs(determinant) < 1e-8;
        V3 dpdu, dpdv;
        if (!degenerate) {
            float invdet = 1 / determinant;
            dpdu = (duv12[1] * dp02 - duv02[1] * dp12) * invdet;
            dpdv = (-duv12[0] * dp02 + duv02[0] * dp12) * invdet;
        }
        if (degenerate || length_sq(cross(dpdu, dpdv)) == 0)
            coordinate_system(normalize(cross(p2 - p0, p1 - p0)), &dpdu, &dpdv);
        float xs = (std::abs(b0 * p0.x) + std::abs(b1 * p1.x) + std::abs(b2 * p2.x));
        float ys = (std::abs(b0 * p0.y) + std::abs(b1 * p1.y) + std::abs(b2 * p2.y));
        float zs = (std::abs(b0 * p0.z) + std::abs(b1 * p1.z) + std::abs(b2 * p2.z));
        is->perr = gamma_n(7) * V3(xs, ys, zs);
        is->p = b0 * p0 + b1 * p1 + b2 * p2;
        is->uv[0] = b0 * uv[0][0] + b1 * uv[1][0] + b2 * uv[2][0];  // uvHit, triangle.cpp:318
        is->uv[1] = b0 * uv[0][1] + b1 * uv[1][1] + b2 * uv[2][1];
        is->dpdu = dpdu;
        is->dpdv = dpdv;
        is->wo = normalize(-ray.d);  // Interaction ctor normalises wo, interaction.h:60
        V3 n = normalize(cross(dp02, dp12));
        const bool flip = (flags & IILE_PRIM_FLIP) != 0;
        if (flags & IILE_PRIM_HAS_NORMALS) {
            const float *nn = S.tri_n + 9 * size_t(prim);
            V3 n0(nn[0], nn[1], nn[2]), n1(nn[3], nn[4], nn[5]), n2(nn[6], nn[7], nn[8]);
            V3 ns = (b0 * n0 + b1 * n1 + b2 * n2);
            if (length_sq(ns) > 0)
                ns = normalize(ns);
            else
                ns = n;
            V3 ss = normalize(dpdu);
            V3 ts = cross(ss, ns);
            if (length_sq(ts) > 0.f) {
                ts = normalize(ts);
                ss = cross(ts, ns);
            } else
                coordinate_system(ns, &ss, &ts);
            // dndu, dndv of the interpolated normal, triangle.cpp:374-392
            V3 dn1 = n0 - n2, dn2 = n1 - n2;
            if (std::abs(determinant) < 1e-8)
                is->dndu = is->dndv = V3(0, 0, 0);
            else {
                float inv_det = 1 / determinant;
                is->dndu = (duv12[1] * dn1 - duv02[1] * dn2) * inv_det;
                is->dndv = (-duv12[0] * dn1 + duv02[0] * dn2) * inv_det;
            }
            // SetShadingGeometry(ss, ts, ..., true), interaction.cpp:72-92
            V3 sn = normalize(cross(ss, ts));
            if (flip) sn = -sn;
            n = faceforward(n, sn);
            is->sn = sn;
            is->sdpdu = ss;
            is->sdpdv = ts;
            n = faceforward(n, is->sn);  // triangle.cpp:396-397
        } else {
            if (flip) n = -n;  // triangle.cpp:398-399
            is->sn = n;
            is->sdpdu = dpdu;
            is->sdpdv = dpdv;
            is->dndu = is->dndv = V3(0, 0, 0);
        }
        is->flip = flip;
        is->n = n;
    }

    // ------------------------------------------------------------------------
    // Sphere (shapes/sphere.cpp:49-215)
    bool sphere_test(const Ray &r, const iile_sphere &sp, Ray *obj_ray, float *t_hit, V3 *phit_out) const {
        ++ctr->sphere_tests;
        M4 w2o{sp.o2w_inv};
        V3 oerr, derr;
        // Transform::operator()(Ray, oError, dError), transform.h:382-394 (tMax kept)
        V3 o = xf_point_err(w2o, r.o, &oerr);
        V3 d = xf_vector_err(w2o, r.d, &derr);
        float len2 = length_sq(d);
        if (len2 > 0) {
            float dt = dot(vabs(d), oerr) / len2;
            o = o + d * dt;
        }
        Ray ray{o, d, r.tmax};
        EFloat ox(ray.o.x, oerr.x), oy(ray.o.y, oerr.y), oz(ray.o.z, oerr.z);
        EFloat dx(ray.d.x, derr.x), dy(ray.d.y, derr.y), dz(ray.d.z, derr.z);
        EFloat a = dx * dx + dy * dy + dz * dz;
        EFloat b = EFloat(2.f) * (dx * ox + dy * oy + dz * oz);
        EFloat c = ox * ox + oy * oy + oz * oz - EFloat(sp.radius) * EFloat(sp.radius);
        EFloat t0, t1;
        if (!ef_quadratic(a, b, c, &t0, &t1)) return false;
        if (t0.high > ray.tmax || t1.low <= 0) return false;
        EFloat ts = t0;
        if (ts.low <= 0) {
            ts = t1;
            if (ts.high > ray.tmax) return false;
        }
        auto refine = [&](float t) {
            V3 ph = ray.o + ray.d * t;
            float scale = sp.radius / length(ph);  // Distance(pHit, (0,0,0))
            ph = V3(ph.x * scale, ph.y * scale, ph.z * scale);
            if (ph.x == 0 && ph.y == 0) ph.x = 1e-5f * sp.radius;
            return ph;
        };
        V3 ph = refine(ts.v);
        // (the trig mode's atan2: the device evaluates the same operations; a full sphere's phi never exceeds phiMax = Radians(360) in either)
        float phi = trig.atan2_f(ph.y, ph.x);
        if (phi < 0) phi += 2 * Pi;
        if ((sp.zmin > -sp.radius && ph.z < sp.zmin) || (sp.zmax < sp.radius && ph.z > sp.zmax) ||
            phi > sp.phi_max) {
            if (ts.v == t1.v) return false;
            if (t1.high > ray.tmax) return false;
            ts = t1;
            ph = refine(ts.v);
            phi = trig.atan2_f(ph.y, ph.x);
            if (phi < 0) phi += 2 * Pi;
            if ((sp.zmin > -sp.radius && ph.z < sp.zmin) || (sp.zmax < sp.radius && ph.z > sp.zmax) ||
                phi > sp.phi_max)
                return false;
        }
        *obj_ray = ray;
        *t_hit = ts.v;
        *phit_out = ph;
        return true;
    }
    // the SurfaceInteraction part of Sphere::Intersect (sphere.cpp:104-155)
    // followed by Transform::operator()(SurfaceInteraction) (transform.cpp:262-297)
    void sphere_interaction(const iile_sphere &sp, const Ray &obj_ray, V3 ph, Isect *is) const {
        float theta = trig.acos_f(clampf(ph.z / sp.radius, -1, 1));
        float z_radius = std::sqrt(ph.x * ph.x + ph.y * ph.y);
        float inv_z_radius = 1 / z_radius;
        float cos_phi = ph.x * inv_z_radius;
        float sin_phi = ph.y * inv_z_radius;
        V3 dpdu(-sp.phi_max * ph.y, sp.phi_max * ph.x, 0);
        V3 dpdv = (sp.theta_max - sp.theta_min) * V3(ph.z * cos_phi, ph.z * sin_phi, -sp.radius * trig.sin_f(theta));
        V3 perr = gamma_n(5) * vabs(ph);
        // SurfaceInteraction ctor, interaction.cpp:44-70
        V3 n = normalize(cross(dpdu, dpdv));
        V3 sn = n;
        if (sp.reverse_orientation ^ sp.swaps_handedness) {
            n = n * -1.f;
            sn = sn * -1.f;
        }
        V3 wo = normalize(-obj_ray.d);
        // object -> world
        M4 m{sp.o2w}, mi{sp.o2w_inv};
        is->p = xf_point_err2(m, ph, perr, &is->perr);
        is->n = normalize(xf_normal(mi, n));
        is->wo = normalize(xf_vector(m, wo));
        V3 snw = normalize(xf_normal(mi, sn));
        is->sdpdu = xf_vector(m, dpdu);
        is->sn = faceforward(snw, is->n);
        // dndu / dndv from the fundamental forms (sphere.cpp:122-143) and, with dpdu / dpdv, their object-to-world images
        // (transform.cpp:275-283: vectors by the matrix, Normal3f by the inverse transpose). Only the direct pass's
        // reflected-ray differentials read them (directprogressiveintegrator.cpp:165-184), Material::Bump and — with uv below — the texture lookups on a sphere.
        const float dt = sp.theta_max - sp.theta_min;
        const V3 d2Pduu = (-sp.phi_max * sp.phi_max) * V3(ph.x, ph.y, 0);
        const V3 d2Pduv = (dt * ph.z * sp.phi_max) * V3(-sin_phi, cos_phi, 0.f);
        const V3 d2Pdvv = (-dt * dt) * V3(ph.x, ph.y, ph.z);
        const float E = dot(dpdu, dpdu), F = dot(dpdu, dpdv), G = dot(dpdv, dpdv);
        const V3 N = normalize(cross(dpdu, dpdv));
        const float e = dot(N, d2Pduu), f = dot(N, d2Pduv), g = dot(N, d2Pdvv);
        const float inv_egf2 = 1 / (E * G - F * F);
        const V3 dndu = ((f * F - e * G) * inv_egf2) * dpdu + ((e * F - f * E) * inv_egf2) * dpdv;
        const V3 dndv = ((g * F - f * G) * inv_egf2) * dpdu + ((f * F - g * E) * inv_egf2) * dpdv;
        is->dpdu = is->sdpdu;
        is->dpdv = is->sdpdv = xf_vector(m, dpdv);
        is->dndu = xf_normal(mi, dndu);
        is->dndv = xf_normal(mi, dndv);
        // Point2f(u, v) of the hit (sphere.cpp:107-109): u = phi / phiMax, v = (theta - thetaMin) / (thetaMax - thetaMin); phi as
        // Sphere::Intersect computes it from the refined hit point
        float phi = trig.atan2_f(ph.y, ph.x);
        if (phi < 0) phi += 2 * Pi;
        is->uv[0] = phi / sp.phi_max;
        is->uv[1] = (theta - sp.theta_min) / (sp.theta_max - sp.theta_min);
        is->flip = sp.reverse_orientation ^ sp.swaps_handedness;
    }

    // ------------------------------------------------------------------------
    // BVH traversal (accelerators/bvh.cpp:662-738, geometry.h:1411-1438)
    static bool slab(const iile_bvh_node &nd, const Ray &ray, V3 inv_dir, const int neg[3]) {
        const float *bmin = nd.bmin, *bmax = nd.bmax;
        auto b = [&](int hi, int axis) { return hi ? bmax[axis] : bmin[axis]; };
        float tmin = (b(neg[0], 0) - ray.o.x) * inv_dir.x;
        float tmax = (b(1 - neg[0], 0) - ray.o.x) * inv_dir.x;
        float tymin = (b(neg[1], 1) - ray.o.y) * inv_dir.y;
        float tymax = (b(1 - neg[1], 1) - ray.o.y) * inv_dir.y;
        tmax *= 1 + 2 * gamma_n(3);
        tymax *= 1 + 2 * gamma_n(3);
        if (tmin > tymax || tymin > tmax) return false;
        if (tymin > tmin) tmin = tymin;
        if (tymax < tmax) tmax = tymax;
        float tzmin = (b(neg[2], 2) - ray.o.z) * inv_dir.z;
        float tzmax = (b(1 - neg[2], 2) - ray.o.z) * inv_dir.z;
        tzmax *= 1 + 2 * gamma_n(3);
        if (tmin > tzmax || tzmin > tmax) return false;
        if (tzmin > tmin) tmin = tzmin;
        if (tzmax < tmax) tmax = tzmax;
        return (tmin < ray.tmax) && (tmax > 0);
    }
    // closest hit; fills prim/t/b* and, if want_isect, the interaction
    bool intersect(Ray ray, Isect *is, bool want_isect = true) const {
        ++ctr->regular_rays;
        if (S.n_nodes == 0) return false;
        bool hit = false;
        V3 inv_dir(1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z);
        int neg[3] = {inv_dir.x < 0, inv_dir.y < 0, inv_dir.z < 0};
        int to_visit = 0, cur = 0;
        int stack[64];
        bool hit_is_sphere = false;
        Ray sph_obj_ray{};
        V3 sph_ph;
        int sph_index = -1;
        while (true) {
            const iile_bvh_node &nd = S.nodes[cur];
            ++ctr->nodes_closest;
            if (slab(nd, ray, inv_dir, neg)) {
                if (nd.nprims > 0) {
                    for (int i = 0; i < nd.nprims; ++i) {
                        int prim = nd.offset + i;
                        if (S.prim_flags[prim] & IILE_PRIM_SPHERE) {
                            Ray orr;
                            float t;
                            V3 ph;
                            const iile_sphere &sp = S.spheres[S.prim_shape[prim]];
                            if (sphere_test(ray, sp, &orr, &t, &ph)) {
                                hit = true;
                                ray.tmax = t;
                                is->prim = prim;
                                is->t = t;
                                is->b0 = is->b1 = is->b2 = 0;
                                hit_is_sphere = true;
                                sph_obj_ray = orr;
                                sph_ph = ph;
                                sph_index = S.prim_shape[prim];
                            }
                        } else {
                            float t, b0, b1, b2;
                            if (triangle_test(ray, prim, &t, &b0, &b1, &b2, 1)) {
                                hit = true;
                                ray.tmax = t;
                                is->prim = prim;
                                is->t = t;
                                is->b0 = b0;
                                is->b1 = b1;
                                is->b2 = b2;
                                hit_is_sphere = false;
                            }
                        }
                    }
                    if (to_visit == 0) break;
                    cur = stack[--to_visit];
                } else {
                    if (neg[nd.axis]) {
                        stack[to_visit++] = cur + 1;
                        cur = nd.offset;
                    } else {
                        stack[to_visit++] = nd.offset;
                        cur = cur + 1;
                    }
                    ctr->max_stack = std::max(ctr->max_stack, to_visit);
                }
            } else {
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            }
        }
        if (hit && want_isect) {
            if (hit_is_sphere)
                sphere_interaction(S.spheres[sph_index], sph_obj_ray, sph_ph, is);
            else
                triangle_interaction(ray, is->prim, is->b0, is->b1, is->b2, is);
        }
        return hit;
    }
    // one primitive alone (the reference's shape-level Intersect / IntersectP), for the property tests
    bool prim_intersects(const Ray &ray, int prim) const {
        if (S.prim_flags[prim] & IILE_PRIM_SPHERE) {
            Ray orr;
            float t;
            V3 ph;
            return sphere_test(ray, S.spheres[S.prim_shape[prim]], &orr, &t, &ph);
        }
        float t, b0, b1, b2;
        return triangle_test(ray, prim, &t, &b0, &b1, &b2, 1);
    }
    bool intersect_p(const Ray &ray) const {
        ++ctr->shadow_rays;
        if (S.n_nodes == 0) return false;
        V3 inv_dir(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
        int neg[3] = {inv_dir.x < 0, inv_dir.y < 0, inv_dir.z < 0};
        int stack[64];
        int to_visit = 0, cur = 0;
        while (true) {
            const iile_bvh_node &nd = S.nodes[cur];
            ++ctr->nodes_any;
            if (slab(nd, ray, inv_dir, neg)) {
                if (nd.nprims > 0) {
                    for (int i = 0; i < nd.nprims; ++i) {
                        int prim = nd.offset + i;
                        if (S.prim_flags[prim] & IILE_PRIM_SPHERE) {
                            Ray orr;
                            float t;
                            V3 ph;
                            if (sphere_test(ray, S.spheres[S.prim_shape[prim]], &orr, &t, &ph)) return true;
                        } else {
                            float t, b0, b1, b2;
                            if (triangle_test(ray, prim, &t, &b0, &b1, &b2, 2)) return true;
                        }
                    }
                    if (to_visit == 0) break;
                    cur = stack[--to_visit];
                } else {
                    if (neg[nd.axis]) {
                        stack[to_visit++] = cur + 1;
                        cur = nd.offset;
                    } else {
                        stack[to_visit++] = nd.offset;
                        cur = cur + 1;
                    }
                    ctr->max_stack = std::max(ctr->max_stack, to_visit);
                }
            } else {
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            }
        }
        return false;
    }

    // ------------------------------------------------------------------------
    // BSDF (core/reflection.{h,cpp}, core/microfacet.cpp)
    struct Bsdf {
        V3 ns, ng, ss, ts;
        int n_lobes = 0;    // nBxDFs; BxDF order: Lambertian, microfacet, specular reflection
        bool has_lambert = false, has_micro = false, has_spec = false;
        bool oren_nayar = false;  // the diffuse lobe is OrenNayar(kd, sigma) instead of LambertianReflection
        float on_a = 1, on_b = 0;
        Rgb kd, ks, kr;
        float alpha = 0, alpha_y = 0;   // TrowbridgeReitzDistribution(alphax, alphay)
        float micro_eta_i = 1.5f, micro_eta_t = 1.f;  // FresnelDielectric of the microfacet lobe
        bool spec_noop = true;                        // FresnelNoOp (mirror) or FresnelDielectric(1, spec_eta)
        float spec_eta = 1.f;
        bool spec_glass = false;                      // the specular lobe is FresnelSpecular(kr, kt, 1, spec_eta)
        Rgb kt;
        // UberMaterial's two SpecularTransmission lobes (uber.cpp:53-61, 94-99): the pass-through of a surface that is not opaque —
        // SpecularTransmission(1 - opacity, 1, 1), the FIRST lobe of the BSDF — and SpecularTransmission(opacity Kt, 1, eta), the LAST
        bool has_t0 = false, has_t1 = false;
        Rgb t0, t1;
        float t1_eta = 1.f;
        // rough glass (glass.cpp:66-90, uroughness / vroughness != 0): MicrofacetReflection(kr -> ks, FresnelDielectric(1, eta)) is the
        // microfacet lobe above; MicrofacetTransmission(kt, distrib, 1, mt_eta, Radiance) — BSDF_TRANSMISSION | BSDF_GLOSSY: not specular
        bool has_mtrans = false;
        float mt_eta = 1.f;
        float eta = 1.f;                              // BSDF::eta (path.cpp:152)
        int n_nonspec() const { return (has_lambert ? 1 : 0) + (has_micro ? 1 : 0) + (has_mtrans ? 1 : 0); }
        V3 to_local(V3 v) const { return V3(dot(v, ss), dot(v, ts), dot(v, ns)); }
        V3 to_world(V3 v) const {
            return V3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z,
                      ss.z * v.x + ts.z * v.y + ns.z * v.z);
        }
    };
    // {Matte,Plastic,Uber,Mirror}Material::ComputeScatteringFunctions (matte.cpp:45-62,
    // plastic.cpp:45-70, uber.cpp:45-100 with opacity 1 and Kt 0, mirror.cpp:44-55)
    // Material::Bump (material.cpp:45-86) with an ImageTexture<Float, Float> displacement, then
    // SetShadingGeometry(dpdu, dpdv, dndu, dndv, false) (interaction.cpp:72-92)
    void bump(int tex, Isect *is) const {
        Isect ev = *is;
        float du = .5f * (std::abs(is->dudx) + std::abs(is->dudy));
        if (du == 0) du = .0005f;
        ev.uv[0] = is->uv[0] + du;
        ev.uv[1] = is->uv[1] + 0.f;
        float u_displace = tex_evaluate(tex, ev).c[0];
        float dv = .5f * (std::abs(is->dvdx) + std::abs(is->dvdy));
        if (dv == 0) dv = .0005f;
        ev.uv[0] = is->uv[0] + 0.f;
        ev.uv[1] = is->uv[1] + dv;
        float v_displace = tex_evaluate(tex, ev).c[0];
        float displace = tex_evaluate(tex, *is).c[0];
        V3 dpdu = is->sdpdu + (u_displace - displace) / du * is->sn + displace * is->dndu;
        V3 dpdv = is->sdpdv + (v_displace - displace) / dv * is->sn + displace * is->dndv;
        V3 sn = normalize(cross(dpdu, dpdv));
        if (is->flip) sn = -sn;
        sn = faceforward(sn, is->n);
        is->sn = sn;
        is->sdpdu = dpdu;
        is->sdpdv = dpdv;
    }
    Bsdf make_bsdf(const Isect &is) const { return make_bsdf_of(S.materials[S.prim_material[is.prim]], is); }
    Bsdf make_bsdf_of(const iile_material &m, const Isect &is) const {
        Bsdf b;
        b.ns = is.sn;
        b.ng = is.n;
        b.ss = normalize(is.sdpdu);
        b.ts = cross(b.ns, b.ss);
        auto clamp0 = [](const float *c) {
            return Rgb(clampf(c[0], 0, Infinity), clampf(c[1], 0, Infinity), clampf(c[2], 0, Infinity));
        };
        // a parameter given as an image texture is looked up at the hit (Texture::Evaluate(*si))
        auto param = [&](const float *constant, int tex) {
            if (tex < 0) return clamp0(constant);
            Rgb v = tex_evaluate(tex, is) * Rgb(constant[0], constant[1], constant[2]);  // ScaleTexture: tex1 * tex2 (x 1 if plain)
            return clamp0(v.c);
        };
        // UberMaterial (uber.cpp:53-61): op = opacity.Clamp(), t = (-op + Spectrum(1.f)).Clamp(); a surface that is not opaque gets
        // BSDF(*si, 1.f) with SpecularTransmission(t, 1.f, 1.f, mode) as its first lobe, and every other coefficient is op * K.Clamp()
        Rgb op(1.f);
        if (m.type == IILE_MAT_UBER) {
            op = param(m.opacity, m.opacity_tex);   // opacity->Evaluate(*si).Clamp(), uber.cpp:53
            const float tt[3] = {-op.c[0] + 1.f, -op.c[1] + 1.f, -op.c[2] + 1.f};
            const Rgb t = clamp0(tt);
            if (!t.is_black()) {
                b.has_t0 = true;
                b.t0 = t;
                ++b.n_lobes;
            }
        }
        Rgb kd = param(m.kd, m.kd_tex);
        if (m.type == IILE_MAT_UBER) kd = op * kd;
        if (!kd.is_black()) {
            b.has_lambert = true;
            b.kd = kd;
            ++b.n_lobes;
            if (m.type == IILE_MAT_MATTE && m.sigma_tex >= 0) {  // sigma->Evaluate(*si), matte.cpp:56-61; OrenNayar ctor, reflection.h:414-420
                const float sig = clampf(tex_evaluate(m.sigma_tex, is).c[0], 0.f, 90.f);
                if (sig != 0) {
                    const float sg = (Pi / 180) * sig;
                    const float sigma2 = sg * sg;
                    b.oren_nayar = true;
                    b.on_a = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                    b.on_b = 0.45f * sigma2 / (sigma2 + 0.09f);
                }
            } else if (m.type == IILE_MAT_MATTE && m.sigma != 0) {  // matte.cpp:56-61
                b.oren_nayar = true;
                b.on_a = m.on_a;
                b.on_b = m.on_b;
            }
        }
        if (m.type == IILE_MAT_PLASTIC || m.type == IILE_MAT_UBER) {
            Rgb ks = param(m.ks, m.ks_tex);
            if (m.type == IILE_MAT_UBER) ks = op * ks;
            if (!ks.is_black()) {
                b.has_micro = true;
                b.ks = ks;
                b.alpha = m.alpha;
                if (m.rough_tex >= 0) {  // roughness->Evaluate(*si), then RoughnessToAlpha (microfacet.h:123-128)
                    float rough = tex_evaluate(m.rough_tex, is).c[0];
                    if (m.remap_roughness) {
                        rough = std::max(rough, 1e-3f);
                        const float x = trig.log_f(rough);
                        rough = 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
                    }
                    b.alpha = rough;
                }
                // uber: roughv = vroughness (a number or a float image) or roughu (uber.cpp:73-86); plastic: one roughness (plastic.cpp:60-64)
                b.alpha_y = b.alpha;
                if (m.type == IILE_MAT_UBER && m.rough_tex_v == -1) b.alpha_y = m.alpha_v;
                if (m.type == IILE_MAT_UBER && m.rough_tex_v >= 0) {
                    float rough = tex_evaluate(m.rough_tex_v, is).c[0];
                    if (m.remap_roughness) {
                        rough = std::max(rough, 1e-3f);
                        const float x = trig.log_f(rough);
                        rough = 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
                    }
                    b.alpha_y = rough;
                }
                if (m.type == IILE_MAT_UBER) {  // FresnelDielectric(1.f, e), uber.cpp:70
                    b.micro_eta_i = 1.f;
                    b.micro_eta_t = m.eta;
                }
                ++b.n_lobes;
            }
        }
        if (m.type == IILE_MAT_UBER || m.type == IILE_MAT_MIRROR) {
            Rgb kr = param(m.kr, m.kr_tex);
            if (m.type == IILE_MAT_UBER) kr = op * kr;
            if (!kr.is_black()) {
                b.has_spec = true;
                b.kr = kr;
                b.spec_noop = m.type == IILE_MAT_MIRROR;
                b.spec_eta = m.eta;
                ++b.n_lobes;
            }
            if (m.type == IILE_MAT_UBER) {
                b.eta = b.has_t0 ? 1.f : m.eta;  // BSDF(*si, 1.f) / BSDF(*si, e), uber.cpp:56-61
                Rgb kt = op * param(m.kt, m.kt_tex);   // uber.cpp:94-99
                if (!kt.is_black()) {
                    b.has_t1 = true;
                    b.t1 = kt;
                    b.t1_eta = m.eta;
                    ++b.n_lobes;
                }
            }
        }
        if (m.type == IILE_MAT_GLASS && (m.roughness != 0 || m.roughness_v != 0)) {  // glass.cpp:63-90: a rough dielectric
            b.eta = m.eta;
            Rgb R = param(m.kr, m.kr_tex), T = param(m.kt, m.kt_tex);
            if (!R.is_black()) {  // MicrofacetReflection(R, distrib, FresnelDielectric(1, eta))
                b.has_micro = true;
                b.ks = R;
                b.alpha = m.alpha;
                b.alpha_y = m.alpha_v;
                b.micro_eta_i = 1.f;
                b.micro_eta_t = m.eta;
                ++b.n_lobes;
            }
            if (!T.is_black()) {  // MicrofacetTransmission(T, distrib, 1, eta, mode)
                b.has_mtrans = true;
                b.kt = T;
                b.alpha = m.alpha;
                b.alpha_y = m.alpha_v;
                b.mt_eta = m.eta;
                ++b.n_lobes;
            }
        } else if (m.type == IILE_MAT_GLASS) {  // glass.cpp:45-66 with isSpecular && allowMultipleLobes
            b.eta = m.eta;
            Rgb R = param(m.kr, m.kr_tex), T = param(m.kt, m.kt_tex);
            if (!(R.is_black() && T.is_black())) {
                b.has_spec = true;
                b.spec_glass = true;
                b.spec_noop = false;  // (the direct pass's SpecularReflection(R, FresnelDielectric(1, eta)) reads it)
                b.kr = R;
                b.kt = T;
                b.spec_eta = m.eta;
                ++b.n_lobes;
            }
        }
        return b;
    }
    // trig helpers, reflection.h:56-84
    static float cos2_theta(V3 w) { return w.z * w.z; }
    static float sin2_theta(V3 w) { return std::max(0.f, 1.f - cos2_theta(w)); }
    static float sin_theta(V3 w) { return std::sqrt(sin2_theta(w)); }
    static float tan_theta(V3 w) { return sin_theta(w) / w.z; }
    static float tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
    static float cos_phi(V3 w) {
        float st = sin_theta(w);
        return (st == 0) ? 1 : clampf(w.x / st, -1, 1);
    }
    static float sin_phi(V3 w) {
        float st = sin_theta(w);
        return (st == 0) ? 0 : clampf(w.y / st, -1, 1);
    }
    static float cos2_phi(V3 w) { return cos_phi(w) * cos_phi(w); }
    static float sin2_phi(V3 w) { return sin_phi(w) * sin_phi(w); }
    static bool same_hemisphere(V3 a, V3 b) { return a.z * b.z > 0; }
    // reflection.cpp:47-68
    static float fr_dielectric(float cos_i, float eta_i, float eta_t) {
        cos_i = clampf(cos_i, -1, 1);
        bool entering = cos_i > 0.f;
        if (!entering) {
            std::swap(eta_i, eta_t);
            cos_i = std::abs(cos_i);
        }
        float sin_i = std::sqrt(std::max(0.f, 1 - cos_i * cos_i));
        float sin_t = eta_i / eta_t * sin_i;
        if (sin_t >= 1) return 1;
        float cos_t = std::sqrt(std::max(0.f, 1 - sin_t * sin_t));
        float r_parl = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
        float r_perp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
        return (r_parl * r_parl + r_perp * r_perp) / 2;
    }
    // TrowbridgeReitzDistribution, microfacet.cpp:155-163, 176-184
    static float tr_d(V3 wh, float ax, float ay) {
        float t2 = tan2_theta(wh);
        if (std::isinf(t2)) return 0.;
        const float cos4 = cos2_theta(wh) * cos2_theta(wh);
        float e = (cos2_phi(wh) / (ax * ax) + sin2_phi(wh) / (ay * ay)) * t2;
        return 1 / (Pi * ax * ay * cos4 * (1 + e) * (1 + e));
    }
    static float tr_lambda(V3 w, float ax, float ay) {
        float abs_tan = std::abs(tan_theta(w));
        if (std::isinf(abs_tan)) return 0.;
        float alpha = std::sqrt(cos2_phi(w) * ax * ax + sin2_phi(w) * ay * ay);
        float a2t2 = (alpha * abs_tan) * (alpha * abs_tan);
        return (-1 + std::sqrt(1.f + a2t2)) / 2;
    }
    static float tr_g1(V3 w, float ax, float ay) { return 1 / (1 + tr_lambda(w, ax, ay)); }
    static float tr_g(V3 wo, V3 wi, float ax, float ay) { return 1 / (1 + tr_lambda(wo, ax, ay) + tr_lambda(wi, ax, ay)); }
    // MicrofacetDistribution::Pdf with sampleVisibleArea, microfacet.cpp:338-344
    static float tr_pdf(V3 wo, V3 wh, float ax, float ay) { return tr_d(wh, ax, ay) * tr_g1(wo, ax, ay) * absdot(wo, wh) / std::abs(wo.z); }
    // TrowbridgeReitzSample11, microfacet.cpp:238-283 (unqualified sqrt/cos/sin:
    // double overloads, see DESIGN.md "Double precision islands")
    void tr_sample11(float cos_theta, float U1, float U2, float *slope_x, float *slope_y) const {
        if (cos_theta > .9999) {
            float r = float(std::sqrt(double(U1 / (1 - U1))));
            float phi = float(6.28318530718 * U2);
            *slope_x = float(r * trig.cos_d(phi));
            *slope_y = float(r * trig.sin_d(phi));
            return;
        }
        float sin_t = std::sqrt(std::max(0.f, 1.f - cos_theta * cos_theta));
        float tan_t = sin_t / cos_theta;
        float a = 1 / tan_t;
        float G1 = 2 / (1 + std::sqrt(1.f + 1.f / (a * a)));
        float A = 2 * U1 / G1 - 1;
        float tmp = 1.f / (A * A - 1.f);
        if (tmp > 1e10) tmp = 1e10;
        float B = tan_t;
        float D = std::sqrt(std::max(float(B * B * tmp * tmp - (A * A - B * B) * tmp), 0.f));
        float slope_x_1 = B * tmp - D;
        float slope_x_2 = B * tmp + D;
        *slope_x = (A < 0 || slope_x_2 > 1.f / tan_t) ? slope_x_1 : slope_x_2;
        float Sg;
        if (U2 > 0.5f) {
            Sg = 1.f;
            U2 = 2.f * (U2 - .5f);
        } else {
            Sg = -1.f;
            U2 = 2.f * (.5f - U2);
        }
        float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
                  (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
        *slope_y = Sg * z * std::sqrt(1.f + *slope_x * *slope_x);
    }
    // TrowbridgeReitzSample + Sample_wh (visible area), microfacet.cpp:285-336
    V3 tr_sample_wh(V3 wo, const float *u, float ax, float ay) const {
        bool flip = wo.z < 0;
        V3 wi = flip ? -wo : wo;
        V3 ws = normalize(V3(ax * wi.x, ay * wi.y, wi.z));   // 1. stretch wi
        float sx, sy;
        tr_sample11(ws.z, u[0], u[1], &sx, &sy);             // 2. simulate P22_{wi}(x_slope, y_slope, 1, 1)
        float tmp = cos_phi(ws) * sx - sin_phi(ws) * sy;     // 3. rotate
        sy = sin_phi(ws) * sx + cos_phi(ws) * sy;
        sx = tmp;
        sx = ax * sx;                                        // 4. unstretch
        sy = ay * sy;
        V3 wh = normalize(V3(-sx, -sy, 1.));                 // 5. compute normal
        if (flip) wh = -wh;
        return wh;
    }
    // MicrofacetReflection::f, reflection.cpp:226-236 with FresnelDielectric(1.5, 1) (plastic) or (1, e) (uber)
    static Rgb micro_f(const Bsdf &b, V3 wo, V3 wi) {
        float cos_o = std::abs(wo.z), cos_i = std::abs(wi.z);
        V3 wh = wi + wo;
        if (cos_i == 0 || cos_o == 0) return Rgb(0.);
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Rgb(0.);
        wh = normalize(wh);
        Rgb F(fr_dielectric(dot(wi, wh), b.micro_eta_i, b.micro_eta_t));
        return b.ks * tr_d(wh, b.alpha, b.alpha_y) * tr_g(wo, wi, b.alpha, b.alpha_y) * F / (4 * cos_i * cos_o);
    }
    static float micro_pdf(const Bsdf &b, V3 wo, V3 wi) {  // reflection.cpp:419-423
        if (!same_hemisphere(wo, wi)) return 0;
        V3 wh = normalize(wo + wi);
        return tr_pdf(wo, wh, b.alpha, b.alpha_y) / (4 * dot(wo, wh));
    }
    // Refract, reflection.h:96-108
    static bool refract(V3 wi, V3 n, float eta, V3 *wt) {
        float cos_i = dot(n, wi);
        float sin2_i = std::max(0.f, 1 - cos_i * cos_i);
        float sin2_t = eta * eta * sin2_i;
        if (sin2_t >= 1) return false;
        float cos_t = std::sqrt(1 - sin2_t);
        *wt = eta * -wi + (eta * cos_i - cos_t) * n;
        return true;
    }
    // MicrofacetTransmission::f, reflection.cpp:244-266 (etaA = 1, mode == Radiance)
    static Rgb mtrans_f(const Bsdf &b, V3 wo, V3 wi) {
        if (same_hemisphere(wo, wi)) return Rgb(0.f);  // transmission only
        float cos_o = wo.z, cos_i = wi.z;
        if (cos_i == 0 || cos_o == 0) return Rgb(0.f);
        const float eta_a = 1.f, eta_b = b.mt_eta;
        float eta = wo.z > 0 ? (eta_b / eta_a) : (eta_a / eta_b);
        V3 wh = normalize(wo + wi * eta);
        if (wh.z < 0) wh = -wh;
        float F = fr_dielectric(dot(wo, wh), eta_a, eta_b);
        float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        float factor = 1 / eta;
        Rgb one_minus_f(1.f - F);
        return one_minus_f * b.kt *
               std::abs(tr_d(wh, b.alpha, b.alpha_y) * tr_g(wo, wi, b.alpha, b.alpha_y) * eta * eta * absdot(wi, wh) * absdot(wo, wh) * factor * factor /
                        (cos_i * cos_o * sqrt_denom * sqrt_denom));
    }
    // MicrofacetTransmission::Pdf, reflection.cpp:435-447
    static float mtrans_pdf(const Bsdf &b, V3 wo, V3 wi) {
        if (same_hemisphere(wo, wi)) return 0;
        const float eta_a = 1.f, eta_b = b.mt_eta;
        float eta = wo.z > 0 ? (eta_b / eta_a) : (eta_a / eta_b);
        V3 wh = normalize(wo + wi * eta);
        float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        float dwh_dwi = std::abs((eta * eta * dot(wi, wh)) / (sqrt_denom * sqrt_denom));
        return tr_pdf(wo, wh, b.alpha, b.alpha_y) * dwh_dwi;
    }
    // LambertianReflection::f (reflection.cpp:178-180) or OrenNayar::f (reflection.cpp:197-219)
    static Rgb diffuse_f(const Bsdf &b, V3 wo, V3 wi) {
        if (!b.oren_nayar) return b.kd * InvPi;
        float sin_i = sin_theta(wi), sin_o = sin_theta(wo);
        float max_cos = 0;
        if (sin_i > 1e-4 && sin_o > 1e-4) {
            float sin_phi_i = sin_phi(wi), cos_phi_i = cos_phi(wi);
            float sin_phi_o = sin_phi(wo), cos_phi_o = cos_phi(wo);
            float d_cos = cos_phi_i * cos_phi_o + sin_phi_i * sin_phi_o;
            max_cos = std::max(0.f, d_cos);
        }
        float sin_alpha, tan_beta;
        if (std::abs(wi.z) > std::abs(wo.z)) {
            sin_alpha = sin_o;
            tan_beta = sin_i / std::abs(wi.z);
        } else {
            sin_alpha = sin_i;
            tan_beta = sin_o / std::abs(wo.z);
        }
        return b.kd * InvPi * (b.on_a + b.on_b * max_cos * sin_alpha * tan_beta);
    }
    static float lambert_pdf(V3 wo, V3 wi) { return same_hemisphere(wo, wi) ? std::abs(wi.z) * InvPi : 0; }

    // BSDF::f, reflection.cpp:686-699 (all lobes are BSDF_REFLECTION, non-specular)
    static Rgb bsdf_f(const Bsdf &b, V3 woW, V3 wiW) {
        V3 wi = b.to_local(wiW), wo = b.to_local(woW);
        if (wo.z == 0) return Rgb(0.);
        bool reflect = dot(wiW, b.ng) * dot(woW, b.ng) > 0;
        Rgb f(0.f);
        if (reflect) {
            if (b.has_lambert) f = f + diffuse_f(b, wo, wi);
            if (b.has_micro) f = f + micro_f(b, wo, wi);
        } else if (b.has_mtrans) {  // `(!reflect && (bxdfs[i]->type & BSDF_TRANSMISSION))`
            f = f + mtrans_f(b, wo, wi);
        }
        return f;
    }
    // BSDF::Pdf, reflection.cpp:786-801
    static float bsdf_pdf(const Bsdf &b, V3 woW, V3 wiW) {
        if (b.n_lobes == 0) return 0.f;
        V3 wo = b.to_local(woW), wi = b.to_local(wiW);
        if (wo.z == 0) return 0.;
        float pdf = 0.f;
        int matching = 0;
        if (b.has_lambert) {
            ++matching;
            pdf += lambert_pdf(wo, wi);
        }
        if (b.has_micro) {
            ++matching;
            pdf += micro_pdf(b, wo, wi);
        }
        if (b.has_mtrans) {
            ++matching;
            pdf += mtrans_pdf(b, wo, wi);
        }
        return matching > 0 ? pdf / matching : 0.f;
    }
    // BSDF::Sample_f, reflection.cpp:719-784. Returns f; *pdf is left untouched
    // on the early `wo.z == 0` return exactly as in the reference.
    Rgb bsdf_sample_f(const Bsdf &b, V3 woW, V3 *wiW, const float *u, float *pdf, bool allow_specular = false,
                      bool *sampled_specular = nullptr, bool *sampled_transmission = nullptr) const {
        // `type` is BSDF_ALL (allow_specular) or BSDF_ALL & ~BSDF_SPECULAR
        if (sampled_specular) *sampled_specular = false;
        if (sampled_transmission) *sampled_transmission = false;
        int matching = allow_specular ? b.n_lobes : b.n_nonspec();
        if (matching == 0) {
            *pdf = 0;
            return Rgb(0);
        }
        int comp = std::min((int)std::floor(u[0] * matching), matching - 1);
        // BxDF order: [uber's pass-through], Lambertian, microfacet, specular reflection, [uber's Kt lobe] (plastic.cpp:53-69, uber.cpp:53-99)
        int pick = -1, count = comp;  // 0 Lambertian, 1 microfacet, 2 specular, 3 / 4 uber's SpecularTransmission lobes
        if (b.has_t0 && allow_specular && count-- == 0) pick = 3;
        if (pick < 0 && b.has_lambert && count-- == 0) pick = 0;
        if (pick < 0 && b.has_micro && count-- == 0) pick = 1;
        if (pick < 0 && b.has_mtrans && count-- == 0) pick = 5;   // rough glass: MicrofacetTransmission behind MicrofacetReflection (glass.cpp:74-90)
        if (pick < 0 && b.has_spec && allow_specular && count-- == 0) pick = 2;
        if (pick < 0 && b.has_t1 && allow_specular && count-- == 0) pick = 4;
        float ur[2] = {std::min(u[0] * matching - comp, OneMinusEpsilon), u[1]};
        V3 wi, wo = b.to_local(woW);
        if (wo.z == 0) return Rgb(0.);
        *pdf = 0;
        Rgb f;
        if (pick == 0) {  // BxDF::Sample_f, reflection.cpp:378-385
            wi = cosine_sample_hemisphere(ur);
            if (wo.z < 0) wi.z *= -1;
            *pdf = lambert_pdf(wo, wi);
            f = diffuse_f(b, wo, wi);
        } else if (pick == 1) {  // MicrofacetReflection::Sample_f, reflection.cpp:405-417
            // `if (wo.z == 0) return 0.` is unreachable here
            V3 wh = tr_sample_wh(wo, ur, b.alpha, b.alpha_y);
            wi = -wo + 2 * dot(wo, wh) * wh;  // Reflect(), reflection.h:86-88
            if (!same_hemisphere(wo, wi))
                f = Rgb(0.f);
            else {
                *pdf = tr_pdf(wo, wh, b.alpha, b.alpha_y) / (4 * dot(wo, wh));
                f = micro_f(b, wo, wi);
            }
        } else if (pick == 5) {  // MicrofacetTransmission::Sample_f, reflection.cpp:425-433
            V3 wh = tr_sample_wh(wo, ur, b.alpha, b.alpha_y);
            const float eta_a = 1.f, eta_b = b.mt_eta;
            float eta = wo.z > 0 ? (eta_a / eta_b) : (eta_b / eta_a);
            if (!refract(wo, wh, eta, &wi)) return Rgb(0);  // `return 0`, pdf stays 0
            *pdf = mtrans_pdf(b, wo, wi);
            f = mtrans_f(b, wo, wi);
        } else if (pick >= 3) {  // SpecularTransmission::Sample_f, reflection.cpp:154-170 (mode == Radiance)
            const float eta_a = 1.f, eta_b = pick == 3 ? 1.f : b.t1_eta;
            const bool entering = wo.z > 0;
            const float eta_i = entering ? eta_a : eta_b, eta_t = entering ? eta_b : eta_a;
            // Refract(wo, Faceforward(Normal3f(0, 0, 1), wo), etaI / etaT, wi), reflection.h:96-108
            V3 n = (wo.z < 0.f) ? -V3(0, 0, 1) : V3(0, 0, 1);
            const float eta = eta_i / eta_t;
            const float cos_i = dot(n, wo);
            const float sin2_i = std::max(0.f, 1 - cos_i * cos_i);
            const float sin2_t = eta * eta * sin2_i;
            if (sin2_t >= 1) return Rgb(0);  // `return 0`, pdf stays 0
            const float cos_t = std::sqrt(1 - sin2_t);
            wi = eta * -wo + (eta * cos_i - cos_t) * n;
            *pdf = 1;
            Rgb ft = (pick == 3 ? b.t0 : b.t1) * (1.f - fr_dielectric(wi.z, eta_a, eta_b));
            ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
            f = ft / std::abs(wi.z);
            if (sampled_specular) *sampled_specular = true;
            if (sampled_transmission) *sampled_transmission = true;
        } else if (b.spec_glass) {  // FresnelSpecular::Sample_f, reflection.cpp:629-672 (mode == Radiance)
            const float eta_a = 1.f, eta_b = b.spec_eta;
            float F = fr_dielectric(wo.z, eta_a, eta_b);
            if (ur[0] < F) {
                wi = V3(-wo.x, -wo.y, wo.z);
                *pdf = F;
                f = F * b.kr / std::abs(wi.z);
            } else {
                bool entering = wo.z > 0;
                float eta_i = entering ? eta_a : eta_b, eta_t = entering ? eta_b : eta_a;
                // Refract(wo, Faceforward(Normal3f(0, 0, 1), wo), etaI / etaT, wi), reflection.h:96-108
                V3 n = (wo.z < 0.f) ? -V3(0, 0, 1) : V3(0, 0, 1);  // -n carries negative zeros, as there
                float eta = eta_i / eta_t;
                float cos_i = dot(n, wo);
                float sin2_i = std::max(0.f, 1 - cos_i * cos_i);
                float sin2_t = eta * eta * sin2_i;
                if (sin2_t >= 1) return Rgb(0);  // total internal reflection: `return 0`, pdf stays 0
                float cos_t = std::sqrt(1 - sin2_t);
                wi = eta * -wo + (eta * cos_i - cos_t) * n;
                Rgb ft = b.kt * (1 - F);
                ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
                *pdf = 1 - F;
                f = ft / std::abs(wi.z);
                if (sampled_transmission) *sampled_transmission = true;
            }
            if (sampled_specular) *sampled_specular = true;
        } else {  // SpecularReflection::Sample_f, reflection.cpp:136-143
            wi = V3(-wo.x, -wo.y, wo.z);
            *pdf = 1;
            Rgb fr = b.spec_noop ? Rgb(1.f) : Rgb(fr_dielectric(wi.z, 1.f, b.spec_eta));
            f = fr * b.kr / std::abs(wi.z);
            if (sampled_specular) *sampled_specular = true;
        }
        if (*pdf == 0) {
            if (sampled_specular) *sampled_specular = false;
            if (sampled_transmission) *sampled_transmission = false;
            return Rgb(0);
        }
        *wiW = b.to_world(wi);
        const bool specular = pick >= 2 && pick != 5;
        if (!specular && matching > 1) {  // a specular lobe's Pdf() is 0
            if (pick != 0 && b.has_lambert) *pdf += lambert_pdf(wo, wi);
            if (pick != 1 && b.has_micro) *pdf += micro_pdf(b, wo, wi);
            if (pick != 5 && b.has_mtrans) *pdf += mtrans_pdf(b, wo, wi);
        }
        if (matching > 1) *pdf /= matching;
        if (!specular && matching > 1) {  // a specular lobe's f() is 0
            bool reflect = dot(*wiW, b.ng) * dot(woW, b.ng) > 0;
            f = Rgb(0.);
            if (reflect) {
                if (b.has_lambert) f = f + diffuse_f(b, wo, wi);
                if (b.has_micro) f = f + micro_f(b, wo, wi);
            } else if (b.has_mtrans) {
                f = f + mtrans_f(b, wo, wi);
            }
        }
        return f;
    }

    // ------------------------------------------------------------------------
    // sphere as emitter (shapes/sphere.cpp:219-306, core/shape.cpp:72-87)
    struct LightSample {
        V3 p, perr, n;
    };
    static V3 sphere_center(const iile_sphere &sp) { return xf_point(M4{sp.o2w}, V3(0, 0, 0)); }
    LightSample sphere_sample_area(const iile_sphere &sp, const float *u, float *pdf) const {
        // UniformSampleSphere, sampling.cpp:98-103
        float z = 1 - 2 * u[0];
        float r = std::sqrt(std::max(0.f, 1.f - z * z));
        float phi = 2 * Pi * u[1];
        V3 us(r * trig.cos_f(phi), r * trig.sin_f(phi), z);
        V3 pobj = V3(0, 0, 0) + sp.radius * us;
        LightSample it;
        it.n = normalize(xf_normal(M4{sp.o2w_inv}, V3(pobj.x, pobj.y, pobj.z)));
        if (sp.reverse_orientation) it.n = it.n * -1.f;
        float scale = sp.radius / length(pobj);
        pobj = V3(pobj.x * scale, pobj.y * scale, pobj.z * scale);
        V3 pobj_err = gamma_n(5) * vabs(pobj);
        it.p = xf_point_err2(M4{sp.o2w}, pobj, pobj_err, &it.perr);
        float area = sp.phi_max * sp.radius * (sp.zmax - sp.zmin);
        *pdf = 1 / area;
        return it;
    }
    LightSample sphere_sample(const iile_sphere &sp, const Isect &ref, const float *u, float *pdf) const {
        V3 pc = sphere_center(sp);
        V3 porigin = offset_ray_origin(ref.p, ref.perr, ref.n, pc - ref.p);
        if (length_sq(porigin - pc) <= sp.radius * sp.radius) {
            LightSample intr = sphere_sample_area(sp, u, pdf);
            V3 wi = intr.p - ref.p;
            if (length_sq(wi) == 0)
                *pdf = 0;
            else {
                wi = normalize(wi);
                *pdf *= length_sq(ref.p - intr.p) / absdot(intr.n, -wi);
            }
            if (std::isinf(*pdf)) *pdf = 0.f;
            return intr;
        }
        V3 wc = normalize(pc - ref.p);
        V3 wcx, wcy;
        coordinate_system(wc, &wcx, &wcy);
        float sin_tmax2 = sp.radius * sp.radius / length_sq(ref.p - pc);
        float cos_tmax = std::sqrt(std::max(0.f, 1 - sin_tmax2));
        float cos_t = (1 - u[0]) + u[0] * cos_tmax;
        float sin_t = std::sqrt(std::max(0.f, 1 - cos_t * cos_t));
        float phi = u[1] * 2 * Pi;
        float dc = length(ref.p - pc);
        float ds = dc * cos_t - std::sqrt(std::max(0.f, sp.radius * sp.radius - dc * dc * sin_t * sin_t));
        float cos_a = (dc * dc + sp.radius * sp.radius - ds * ds) / (2 * dc * sp.radius);
        float sin_a = std::sqrt(std::max(0.f, 1 - cos_a * cos_a));
        // SphericalDirection(sinAlpha, cosAlpha, phi, -wcX, -wcY, -wc), geometry.h:1467-1472
        V3 nw = sin_a * trig.cos_f(phi) * (-wcx) + sin_a * trig.sin_f(phi) * (-wcy) + cos_a * (-wc);
        V3 pw = pc + sp.radius * V3(nw.x, nw.y, nw.z);
        LightSample it;
        it.p = pw;
        it.perr = gamma_n(5) * vabs(pw);
        it.n = nw;
        if (sp.reverse_orientation) it.n = it.n * -1.f;
        *pdf = 1 / (2 * Pi * (1 - cos_tmax));
        return it;
    }
    float sphere_pdf(const iile_sphere &sp, const Isect &ref, V3 wi) const {
        V3 pc = sphere_center(sp);
        V3 porigin = offset_ray_origin(ref.p, ref.perr, ref.n, pc - ref.p);
        if (length_sq(porigin - pc) <= sp.radius * sp.radius) {
            // Shape::Pdf, shape.cpp:72-87: intersect the shape alone (not a scene ray)
            Ray ray = spawn_ray(ref, wi);
            Ray orr;
            float t;
            V3 ph;
            uint64_t keep = ctr->sphere_tests;
            bool ok = sphere_test(ray, sp, &orr, &t, &ph);
            ctr->sphere_tests = keep;
            if (!ok) return 0;
            Isect li;
            sphere_interaction(sp, orr, ph, &li);
            float area = sp.phi_max * sp.radius * (sp.zmax - sp.zmin);
            float pdf = length_sq(ref.p - li.p) / (absdot(li.n, -wi) * area);
            if (std::isinf(pdf)) pdf = 0.f;
            return pdf;
        }
        float sin_tmax2 = sp.radius * sp.radius / length_sq(ref.p - pc);
        float cos_tmax = std::sqrt(std::max(0.f, 1 - sin_tmax2));
        return 1 / (2 * Pi * (1 - cos_tmax));
    }
    // DiffuseAreaLight::L, lights/diffuse.h:56-58
    static Rgb light_L(const iile_light &lt, V3 n, V3 w) {
        return (lt.two_sided || dot(n, w) > 0) ? Rgb(lt.lemit[0], lt.lemit[1], lt.lemit[2]) : Rgb(0.f);
    }
    Rgb isect_le(const Isect &is, V3 w) const {  // interaction.cpp:151-154
        int l = S.prim_light[is.prim];
        return l >= 0 ? light_L(S.lights[l], is.n, w) : Rgb(0.f);
    }

    // ------------------------------------------------------------------------
    // EstimateDirect for one area light with MIS, core/integrator.cpp:108-215
    static float power_heuristic(int nf, float fpdf, int ng, float gpdf) {
        float f = nf * fpdf, g = ng * gpdf;
        return (f * f) / (f * f + g * g);
    }
    // ------------------------------------------------------------------------
    // Triangle emitter (shapes/triangle.cpp:546-579) through the generic Shape::Sample(ref, u) /
    // Shape::Pdf(ref, wi) (core/shape.cpp:56-87), and the sphere / triangle dispatch
    float triangle_area(int prim) const {
        const float *tp = S.tri_p + 9 * size_t(prim);
        V3 p0(tp[0], tp[1], tp[2]), p1(tp[3], tp[4], tp[5]), p2(tp[6], tp[7], tp[8]);
        return float(0.5 * length(cross(p1 - p0, p2 - p0)));
    }
    LightSample triangle_sample_area(int prim, const float *u, float *pdf) const {
        float su0 = std::sqrt(u[0]);  // UniformSampleTriangle, sampling.cpp:154-157
        float b0 = 1 - su0, b1 = u[1] * su0;
        const float *tp = S.tri_p + 9 * size_t(prim);
        V3 p0(tp[0], tp[1], tp[2]), p1(tp[3], tp[4], tp[5]), p2(tp[6], tp[7], tp[8]);
        const uint32_t flags = S.prim_flags[prim];
        LightSample it;
        it.p = b0 * p0 + b1 * p1 + (1 - b0 - b1) * p2;
        it.n = normalize(cross(p1 - p0, p2 - p0));
        if (flags & IILE_PRIM_HAS_NORMALS) {
            const float *nn = S.tri_n + 9 * size_t(prim);
            V3 n0(nn[0], nn[1], nn[2]), n1(nn[3], nn[4], nn[5]), n2(nn[6], nn[7], nn[8]);
            V3 ns = b0 * n0 + b1 * n1 + (1 - b0 - b1) * n2;
            it.n = faceforward(it.n, ns);
        } else if (flags & IILE_PRIM_FLIP)
            it.n = it.n * -1.f;
        V3 abs_sum = vabs(b0 * p0) + vabs(b1 * p1) + vabs((1 - b0 - b1) * p2);
        it.perr = gamma_n(6) * abs_sum;
        *pdf = 1 / triangle_area(prim);
        return it;
    }
    LightSample shape_sample(const iile_light &lt, const Isect &ref, const float *u, float *pdf) const {
        if (lt.type == IILE_LIGHT_DIFFUSE_AREA) return sphere_sample(S.spheres[lt.sphere], ref, u, pdf);
        // Shape::Sample(ref, u, pdf), shape.cpp:56-70
        LightSample intr = triangle_sample_area(lt.prim, u, pdf);
        V3 wi = intr.p - ref.p;
        if (length_sq(wi) == 0)
            *pdf = 0;
        else {
            wi = normalize(wi);
            *pdf *= length_sq(ref.p - intr.p) / absdot(intr.n, -wi);
            if (std::isinf(*pdf)) *pdf = 0.f;
        }
        return intr;
    }
    float shape_pdf(const iile_light &lt, const Isect &ref, V3 wi) const {
        if (lt.type == IILE_LIGHT_DIFFUSE_AREA) return sphere_pdf(S.spheres[lt.sphere], ref, wi);
        // Shape::Pdf(ref, wi), shape.cpp:72-87: intersect the shape alone (Triangle::Intersect counts
        // its tests and hits like any other call)
        Ray ray = spawn_ray(ref, wi);
        float t, b0, b1, b2;
        if (!triangle_test(ray, lt.prim, &t, &b0, &b1, &b2)) return 0;
        Isect li;
        triangle_interaction(ray, lt.prim, b0, b1, b2, &li);
        float pdf = length_sq(ref.p - li.p) / (absdot(li.n, -wi) * triangle_area(lt.prim));
        if (std::isinf(pdf)) pdf = 0.f;
        return pdf;
    }

    // ------------------------------------------------------------------------
    // SpatialLightDistribution (core/lightdistrib.cpp:91-299), the path integrator's default
    // "spatial" strategy whenever the scene has more than one light (lightdistrib.cpp:47-66).
    // The reference fills a hash table lazily; a voxel's distribution is a pure function of its
    // index, so a per-instance cache gives the same values.
    struct LightDist {  // Distribution1D, sampling.h:55-109
        int n = 0;
        float func[IILE_MAX_LIGHTS], cdf[IILE_MAX_LIGHTS + 1], func_int = 0;
    };
    mutable std::unordered_map<uint64_t, LightDist> light_cache;
    void light_grid(V3 *pmin, V3 *pmax, int nv[3]) const {  // lightdistrib.cpp:91-110, maxVoxels = 64
        const iile_bvh_node &root = S.nodes[0];
        *pmin = V3(root.bmin[0], root.bmin[1], root.bmin[2]);
        *pmax = V3(root.bmax[0], root.bmax[1], root.bmax[2]);
        V3 diag = *pmax - *pmin;
        int me = (diag.x > diag.y && diag.x > diag.z) ? 0 : (diag.y > diag.z ? 1 : 2);  // MaximumExtent, geometry.h:771-779
        float bmax = diag[me];
        for (int i = 0; i < 3; ++i) nv[i] = std::max(1, int(std::round(diag[i] / bmax * 64)));
    }
    // Light::Sample_Li at an Interaction without normal or error bounds; returns Li, sets *pdf
    Rgb sample_li_plain(const iile_light &lt, V3 po, const float *u, float *pdf) const {
        const V3 pos(lt.pos[0], lt.pos[1], lt.pos[2]);
        const Rgb I(lt.lemit[0], lt.lemit[1], lt.lemit[2]);
        *pdf = 1;
        if (lt.type == IILE_LIGHT_INFINITE) {
            V3 wi, target;
            return inf_sample_li(lt, po, u, &wi, pdf, &target);
        }
        if (lt.type == IILE_LIGHT_DISTANT) return I;
        if (lt.type == IILE_LIGHT_POINT) return I / length_sq(pos - po);
        if (lt.type == IILE_LIGHT_SPOT) {
            const V3 w = -normalize(pos - po);
            V3 wl = normalize(V3(lt.w2l[0] * w.x + lt.w2l[1] * w.y + lt.w2l[2] * w.z,
                                 lt.w2l[3] * w.x + lt.w2l[4] * w.y + lt.w2l[5] * w.z,
                                 lt.w2l[6] * w.x + lt.w2l[7] * w.y + lt.w2l[8] * w.z));
            float cos_theta = wl.z, falloff;
            if (cos_theta < lt.cos_total_width)
                falloff = 0;
            else if (cos_theta >= lt.cos_falloff_start)
                falloff = 1;
            else {
                float delta = (cos_theta - lt.cos_total_width) / (lt.cos_falloff_start - lt.cos_total_width);
                falloff = (delta * delta) * (delta * delta);
            }
            return I * falloff / length_sq(pos - po);
        }
        // DiffuseAreaLight::Sample_Li, lights/diffuse.cpp:68-81
        Isect ref;
        ref.p = po;
        ref.perr = V3(0, 0, 0);
        ref.n = V3(0, 0, 0);
        LightSample ps = shape_sample(lt, ref, u, pdf);
        if (*pdf == 0 || length_sq(ps.p - po) == 0) {
            *pdf = 0;
            return Rgb(0.f);
        }
        V3 wi = normalize(ps.p - po);
        return light_L(lt, ps.n, -wi);
    }
    LightDist compute_light_distribution(const int pi[3]) const {  // lightdistrib.cpp:228-299
        V3 bmin, bmax;
        int nv[3];
        light_grid(&bmin, &bmax, nv);
        auto lerp = [](float t, float a, float b) { return (1 - t) * a + t * b; };  // pbrt.h:414
        V3 p0(float(pi[0]) / float(nv[0]), float(pi[1]) / float(nv[1]), float(pi[2]) / float(nv[2]));
        V3 p1(float(pi[0] + 1) / float(nv[0]), float(pi[1] + 1) / float(nv[1]), float(pi[2] + 1) / float(nv[2]));
        V3 vmin(lerp(p0.x, bmin.x, bmax.x), lerp(p0.y, bmin.y, bmax.y), lerp(p0.z, bmin.z, bmax.z));
        V3 vmax(lerp(p1.x, bmin.x, bmax.x), lerp(p1.y, bmin.y, bmax.y), lerp(p1.z, bmin.z, bmax.z));
        const int n_samples = 128, n = S.n_lights;
        float contrib[IILE_MAX_LIGHTS] = {0};
        for (int i = 0; i < n_samples; ++i) {
            V3 t(oracle_radical_inverse(0, i), oracle_radical_inverse(1, i), oracle_radical_inverse(2, i));
            V3 po(lerp(t.x, vmin.x, vmax.x), lerp(t.y, vmin.y, vmax.y), lerp(t.z, vmin.z, vmax.z));
            float u[2] = {oracle_radical_inverse(3, i), oracle_radical_inverse(4, i)};
            for (int j = 0; j < n; ++j) {
                float pdf;
                Rgb Li = sample_li_plain(S.lights[j], po, u, &pdf);
                if (pdf > 0) contrib[j] += Li.y() / pdf;
            }
        }
        float sum = 0;  // std::accumulate(..., Float(0))
        for (int j = 0; j < n; ++j) sum = sum + contrib[j];
        float avg = sum / (n_samples * n);
        float min_contrib = (avg > 0) ? float(.001 * avg) : 1.f;
        LightDist d;
        d.n = n;
        for (int j = 0; j < n; ++j) d.func[j] = std::max(contrib[j], min_contrib);
        finish_distribution(&d);
        return d;
    }
    // Distribution1D's constructor, sampling.h:57-69, over d->func[0 .. n)
    static void finish_distribution(LightDist *dp) {
        LightDist &d = *dp;
        const int n = d.n;
        d.cdf[0] = 0;
        for (int i = 1; i < n + 1; ++i) d.cdf[i] = d.cdf[i - 1] + d.func[i - 1] / n;
        d.func_int = d.cdf[n];
        if (d.func_int == 0)
            for (int i = 1; i < n + 1; ++i) d.cdf[i] = float(i) / float(n);
        else
            for (int i = 1; i < n + 1; ++i) d.cdf[i] /= d.func_int;
    }
    mutable LightDist fixed_dist;  // UniformLightDistribution / PowerLightDistribution: one Distribution1D for every point
    mutable bool fixed_dist_ready = false;
    const LightDist &light_distribution(V3 p) const {  // LightDistribution::Lookup
        if (S.integrator.light_strategy != IILE_LIGHTS_SPATIAL) {  // lightdistrib.cpp:65-82, integrator.cpp:217-225
            if (!fixed_dist_ready) {
                fixed_dist.n = S.n_lights;
                for (int i = 0; i < S.n_lights; ++i)
                    fixed_dist.func[i] = S.integrator.light_strategy == IILE_LIGHTS_UNIFORM ? 1.f : S.integrator.light_power[i];
                finish_distribution(&fixed_dist);
                fixed_dist_ready = true;
            }
            return fixed_dist;
        }
        // SpatialLightDistribution::Lookup, lightdistrib.cpp:134-226
        V3 bmin, bmax;
        int nv[3];
        light_grid(&bmin, &bmax, nv);
        V3 o = p - bmin;  // Bounds3::Offset, geometry.h:800-806
        if (bmax.x > bmin.x) o.x /= bmax.x - bmin.x;
        if (bmax.y > bmin.y) o.y /= bmax.y - bmin.y;
        if (bmax.z > bmin.z) o.z /= bmax.z - bmin.z;
        int pi[3];
        for (int i = 0; i < 3; ++i) pi[i] = std::min(std::max(int(o[i] * nv[i]), 0), nv[i] - 1);
        uint64_t key = (uint64_t(pi[0]) << 40) | (uint64_t(pi[1]) << 20) | uint64_t(pi[2]);
        auto it = light_cache.find(key);
        if (it == light_cache.end()) it = light_cache.emplace(key, compute_light_distribution(pi)).first;
        return it->second;
    }
    // Distribution1D::SampleDiscrete, sampling.h:90-100 with FindInterval, pbrt.h:399-412
    static int sample_discrete(const LightDist &d, float u, float *pdf) {
        int size = d.n + 1, first = 0, len = size;
        while (len > 0) {
            int half = len >> 1, middle = first + half;
            if (d.cdf[middle] <= u) {
                first = middle + 1;
                len -= half + 1;
            } else
                len = half;
        }
        int offset = std::min(std::max(first - 1, 0), size - 2);
        *pdf = (d.func_int > 0) ? d.func[offset] / (d.func_int * d.n) : 0;
        return offset;
    }

    // ------------------------------------------------------------------------
    // InfiniteAreaLight without an environment map (lights/infinite.cpp:42-174). Lmap holds one
    // texel; its MIPMap lookup is the triangle filter over that texel (mipmap.h:351-389), and the
    // sampling distribution a 2 x 2 Distribution2D built on the host (pbrt_loader.cpp).
    // Lmap->Lookup(st) = Lookup(st, width 0): level = Levels - 1 + Log2(1e-8) < 0 for any pyramid of fewer than
    // 27 levels -> triangle(0, st) (mipmap.h:233-262)
    Rgb inf_lookup(const iile_light &lt, float s_, float t_) const {
        const float st[2] = {s_, t_};
        return tex_triangle(S.textures[lt.env_tex], 0, st);
    }
    // Distribution1D::SampleContinuous over n entries {func[n], cdf[n + 1], funcInt}, sampling.h:71-89
    static float dist1d_sample(const float *d, int n, float u, float *pdf, int *off) {
        const float *func = d, *cdf = d + n;
        const float func_int = d[2 * n + 1];
        const int size = n + 1;
        int first = 0, len = size;
        while (len > 0) {
            int half = len >> 1, middle = first + half;
            if (cdf[middle] <= u) {
                first = middle + 1;
                len -= half + 1;
            } else
                len = half;
        }
        int offset = std::min(std::max(first - 1, 0), size - 2);
        if (off) *off = offset;
        float du = u - cdf[offset];
        if ((cdf[offset + 1] - cdf[offset]) > 0) du /= (cdf[offset + 1] - cdf[offset]);
        *pdf = (func_int > 0) ? func[offset] / func_int : 0;
        return (offset + du) / n;
    }
    const float *inf_cond(const iile_light &lt, int v) const { return S.env_dist + lt.dist_offset + size_t(2 * lt.dist_w + 2) * v; }
    const float *inf_marg(const iile_light &lt) const { return inf_cond(lt, lt.dist_h); }
    V3 inf_w2l(const iile_light &lt, V3 w) const {
        return V3(lt.w2l[0] * w.x + lt.w2l[1] * w.y + lt.w2l[2] * w.z, lt.w2l[3] * w.x + lt.w2l[4] * w.y + lt.w2l[5] * w.z,
                  lt.w2l[6] * w.x + lt.w2l[7] * w.y + lt.w2l[8] * w.z);
    }
    float spherical_theta(V3 v) const { return trig.acos_f(clampf(v.z, -1, 1)); }  // geometry.h:1474-1481
    float spherical_phi(V3 v) const {
        float p = trig.atan2_f(v.y, v.x);
        return (p < 0) ? (p + 2 * Pi) : p;
    }
    Rgb inf_le(const iile_light &lt, V3 d) const {  // InfiniteAreaLight::Le, infinite.cpp:99-104
        V3 w = normalize(inf_w2l(lt, d));
        return inf_lookup(lt, spherical_phi(w) * Inv2Pi, spherical_theta(w) * InvPi);
    }
    Rgb inf_sample_li(const iile_light &lt, V3 ref_p, const float *u, V3 *wi, float *pdf, V3 *target) const {  // :106-137
        float pdfs[2];
        int v;
        float d1 = dist1d_sample(inf_marg(lt), lt.dist_h, u[1], &pdfs[1], &v);
        float d0 = dist1d_sample(inf_cond(lt, v), lt.dist_w, u[0], &pdfs[0], nullptr);
        float map_pdf = pdfs[0] * pdfs[1];
        *pdf = 0;
        if (map_pdf == 0) return Rgb(0.f);
        float theta = d1 * Pi, phi = d0 * 2 * Pi;
        float cos_theta = trig.cos_f(theta), sin_theta = trig.sin_f(theta);
        float sin_phi = trig.sin_f(phi), cos_phi = trig.cos_f(phi);
        V3 wl(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
        *wi = V3(lt.l2w[0] * wl.x + lt.l2w[1] * wl.y + lt.l2w[2] * wl.z, lt.l2w[3] * wl.x + lt.l2w[4] * wl.y + lt.l2w[5] * wl.z,
                 lt.l2w[6] * wl.x + lt.l2w[7] * wl.y + lt.l2w[8] * wl.z);
        *pdf = map_pdf / (2 * Pi * Pi * sin_theta);
        if (sin_theta == 0) *pdf = 0;
        *target = ref_p + *wi * (2 * lt.world_radius);
        return inf_lookup(lt, d0, d1);
    }
    float inf_pdf_li(const iile_light &lt, V3 w) const {  // :139-148
        V3 wi = inf_w2l(lt, w);
        float theta = spherical_theta(wi), phi = spherical_phi(wi);
        float sin_theta = trig.sin_f(theta);
        if (sin_theta == 0) return 0;
        float p0 = phi * Inv2Pi, p1 = theta * InvPi;  // Distribution2D::Pdf, sampling.h:135-142
        int iu = std::min(std::max(int(p0 * lt.dist_w), 0), lt.dist_w - 1), iv = std::min(std::max(int(p1 * lt.dist_h), 0), lt.dist_h - 1);
        return (inf_cond(lt, iv)[iu] / inf_marg(lt)[2 * lt.dist_h + 1]) / (2 * Pi * Pi * sin_theta);
    }
    // EstimateDirect for the infinite light: both halves, with Le(ray) where the BSDF-sampled ray escapes
    Rgb estimate_direct_infinite(const Isect &it, const Bsdf &bsdf, const float *u_scatter, const iile_light &lt,
                                 const float *u_light) const {
        Rgb Ld(0.f);
        V3 wi, target;
        float light_pdf = 0, scattering_pdf = 0;
        Rgb Li = inf_sample_li(lt, it.p, u_light, &wi, &light_pdf, &target);
        if (light_pdf > 0 && !Li.is_black()) {
            Rgb f = bsdf_f(bsdf, it.wo, wi) * absdot(wi, it.sn);
            scattering_pdf = bsdf_pdf(bsdf, it.wo, wi);
            if (!f.is_black()) {
                V3 origin = offset_ray_origin(it.p, it.perr, it.n, target - it.p);
                V3 tgt = offset_ray_origin(target, V3(0, 0, 0), V3(0, 0, 0), origin - target);
                Ray sr{origin, tgt - origin, 1 - ShadowEpsilon};
                if (intersect_p(sr)) Li = Rgb(0.f);
                if (!Li.is_black()) {
                    float weight = power_heuristic(1, light_pdf, 1, scattering_pdf);
                    Ld = Ld + f * Li * weight / light_pdf;
                }
            }
        }
        {
            Rgb f = bsdf_sample_f(bsdf, it.wo, &wi, u_scatter, &scattering_pdf);
            f = f * absdot(wi, it.sn);
            if (!f.is_black() && scattering_pdf > 0) {
                light_pdf = inf_pdf_li(lt, wi);
                if (light_pdf == 0) return Ld;
                float weight = power_heuristic(1, scattering_pdf, 1, light_pdf);
                Isect li;
                Ray ray = spawn_ray(it, wi);
                bool found = intersect(ray, &li);
                Rgb Li2(0.f);
                if (!found) Li2 = inf_le(lt, ray.d);  // a surface hit never is this light
                if (!Li2.is_black()) Ld = Ld + f * Li2 * Rgb(1.f) * weight / scattering_pdf;
            }
        }
        return Ld;
    }

    // EstimateDirect for the delta lights (IsDeltaLight: no MIS weight, no BSDF-sampling half,
    // core/integrator.cpp:150-166). Sample_Li of PointLight (lights/point.cpp:43-52), SpotLight
    // (lights/spot.cpp:53-76) and DistantLight (lights/distant.cpp:50-61).
    Rgb estimate_direct_delta(const Isect &it, const Bsdf &bsdf, const iile_light &lt) const {
        Rgb Ld(0.f);
        const V3 pos(lt.pos[0], lt.pos[1], lt.pos[2]);
        const Rgb I(lt.lemit[0], lt.lemit[1], lt.lemit[2]);
        V3 wi, target;
        Rgb Li;
        if (lt.type == IILE_LIGHT_DISTANT) {
            wi = pos;  // wLight
            target = it.p + pos * (2 * lt.world_radius);  // pOutside
            Li = I;
        } else {
            wi = normalize(pos - it.p);
            target = pos;  // pLight
            if (lt.type == IILE_LIGHT_SPOT) {
                // Falloff(-wi), spot.cpp:66-76
                const V3 w = -wi;
                V3 wl = normalize(V3(lt.w2l[0] * w.x + lt.w2l[1] * w.y + lt.w2l[2] * w.z,
                                     lt.w2l[3] * w.x + lt.w2l[4] * w.y + lt.w2l[5] * w.z,
                                     lt.w2l[6] * w.x + lt.w2l[7] * w.y + lt.w2l[8] * w.z));
                float cos_theta = wl.z, falloff;
                if (cos_theta < lt.cos_total_width)
                    falloff = 0;
                else if (cos_theta >= lt.cos_falloff_start)
                    falloff = 1;
                else {
                    float delta = (cos_theta - lt.cos_total_width) / (lt.cos_falloff_start - lt.cos_total_width);
                    falloff = (delta * delta) * (delta * delta);
                }
                Li = I * falloff / length_sq(pos - it.p);
            } else {
                Li = I / length_sq(pos - it.p);  // I / DistanceSquared
            }
        }
        const float light_pdf = 1.f;
        if (light_pdf > 0 && !Li.is_black()) {
            Rgb f = bsdf_f(bsdf, it.wo, wi) * absdot(wi, it.sn);
            if (!f.is_black()) {
                // VisibilityTester(ref, Interaction(target)): the light-side interaction has no
                // normal and no error bounds, so its OffsetRayOrigin is the point itself
                V3 origin = offset_ray_origin(it.p, it.perr, it.n, target - it.p);
                V3 tgt = offset_ray_origin(target, V3(0, 0, 0), V3(0, 0, 0), origin - target);
                Ray sr{origin, tgt - origin, 1 - ShadowEpsilon};
                if (intersect_p(sr)) Li = Rgb(0.f);
                if (!Li.is_black()) Ld = Ld + f * Li / light_pdf;
            }
        }
        return Ld;
    }
    Rgb estimate_direct(const Isect &it, const Bsdf &bsdf, const float *u_scatter, int light_index,
                        const float *u_light) const {
        const iile_light &lt = S.lights[light_index];
        if (lt.type == IILE_LIGHT_INFINITE) return estimate_direct_infinite(it, bsdf, u_scatter, lt, u_light);
        if (lt.type != IILE_LIGHT_DIFFUSE_AREA && lt.type != IILE_LIGHT_AREA_TRIANGLE)
            return estimate_direct_delta(it, bsdf, lt);
        Rgb Ld(0.f);
        V3 wi;
        float light_pdf = 0, scattering_pdf = 0;
        // DiffuseAreaLight::Sample_Li, lights/diffuse.cpp:68-81
        Rgb Li(0.f);
        LightSample ps = shape_sample(lt, it, u_light, &light_pdf);
        if (light_pdf == 0 || length_sq(ps.p - it.p) == 0) {
            light_pdf = 0;
            Li = Rgb(0.f);
        } else {
            wi = normalize(ps.p - it.p);
            Li = light_L(lt, ps.n, -wi);
        }
        if (light_pdf > 0 && !Li.is_black()) {
            Rgb f = bsdf_f(bsdf, it.wo, wi) * absdot(wi, it.sn);
            scattering_pdf = bsdf_pdf(bsdf, it.wo, wi);
            if (!f.is_black()) {
                // VisibilityTester::Unoccluded -> SpawnRayTo(Interaction), interaction.h:73-78
                V3 origin = offset_ray_origin(it.p, it.perr, it.n, ps.p - it.p);
                V3 target = offset_ray_origin(ps.p, ps.perr, ps.n, origin - ps.p);
                Ray sr{origin, target - origin, 1 - ShadowEpsilon};
                if (intersect_p(sr)) Li = Rgb(0.f);
                if (!Li.is_black()) {
                    float weight = power_heuristic(1, light_pdf, 1, scattering_pdf);
                    Ld = Ld + f * Li * weight / light_pdf;
                }
            }
        }
        // BSDF sampling half
        {
            Rgb f = bsdf_sample_f(bsdf, it.wo, &wi, u_scatter, &scattering_pdf);
            f = f * absdot(wi, it.sn);
            if (!f.is_black() && scattering_pdf > 0) {
                light_pdf = shape_pdf(lt, it, wi);
                if (light_pdf == 0) return Ld;
                float weight = power_heuristic(1, scattering_pdf, 1, light_pdf);
                Isect li;
                Ray ray = spawn_ray(it, wi);
                bool found = intersect(ray, &li);
                Rgb Li2(0.f);
                if (found) {
                    if (S.prim_light[li.prim] == light_index) Li2 = isect_le(li, -wi);
                }
                if (!Li2.is_black()) Ld = Ld + f * Li2 * Rgb(1.f) * weight / scattering_pdf;
            }
        }
        return Ld;
    }

    // ------------------------------------------------------------------------
    // PathIntegrator::Li, integrators/path.cpp:64-194
    // aux != null: IISPTdIntegrator::Li (iispt_d.cpp:66-222) — no emitted light at the camera ray's own vertex
    // (hit or escaped), and the first hit's distance and camera-space normal recorded
    Rgb li(Ray ray, Sampler &smp, RayDiff rdiff = RayDiff(), float *aux = nullptr) const {
        Rgb L(0.f), beta(1.f);
        bool specular_bounce = false;
        int bounces;
        const int max_depth = S.integrator.max_depth;
        const float rr_threshold = S.integrator.rr_threshold;
        float eta_scale = 1;
        for (bounces = 0;; ++bounces) {
            Isect is;
            bool found = intersect(ray, &is);
            if (aux && bounces == 0) {  // iispt_d.cpp:96-113
                if (found) {
                    V3 cv = is.p - ray.o;
                    float d2 = dot(cv, cv);
                    aux[3] = std::sqrt(d2);
                    const float *mi = probe->w2c_minv;  // Transform::operator()(Normal3f): transpose of mInv
                    aux[0] = mi[0] * is.n.x + mi[4] * is.n.y + mi[8] * is.n.z;
                    aux[1] = mi[1] * is.n.x + mi[5] * is.n.y + mi[9] * is.n.z;
                    aux[2] = mi[2] * is.n.x + mi[6] * is.n.y + mi[10] * is.n.z;
                } else {
                    aux[0] = aux[1] = aux[2] = 0.f;
                    aux[3] = -1.f;  // NO_INTERSECTION_DISTANCE
                }
            }
            if ((bounces == 0 && !aux) || (bounces != 0 && specular_bounce)) {
                if (found)
                    L = L + beta * isect_le(is, -ray.d);
                else  // `for (const auto &light : scene.infiniteLights) L += beta * light->Le(ray)`, path.cpp:97-99
                    for (int l = 0; l < S.n_lights; ++l)
                        if (S.lights[l].type == IILE_LIGHT_INFINITE) L = L + beta * inf_le(S.lights[l], ray.d);
            }
            if (!found || bounces >= max_depth) break;
            // isect.ComputeScatteringFunctions(ray, ...): ComputeDifferentials first (interaction.cpp:95-101);
            // only the camera ray has differentials (spawned rays are plain Rays, path.cpp:159)
            if (S.n_textures > 0) compute_differentials(&is, rdiff);
            rdiff.has = false;
            {  // every material's ComputeScatteringFunctions starts with `if (bumpMap) Bump(bumpMap, si)`
                const iile_material &mb = S.materials[S.prim_material[is.prim]];
                if (S.n_textures > 0 && mb.bump_tex >= 0) bump(mb.bump_tex, &is);
            }
            Bsdf bsdf = make_bsdf(is);
            // UniformLightDistribution::Lookup ignores the point; SampleDiscrete
            // still consumes one 1D sample (integrator.cpp:95).
            if (bsdf.n_nonspec() > 0) {  // NumComponents(BSDF_ALL & ~BSDF_SPECULAR) > 0, path.cpp:118
                ++ctr->nee_evals;
                Rgb Ld_in(0.f);
                if (S.n_lights == 1) {
                    // UniformLightDistribution over one light (lightdistrib.cpp:50): SampleDiscrete
                    // returns light 0 with pdf 1 and still consumes a 1D sample (integrator.cpp:95)
                    smp.get1d();
                    float u_light[2], u_scatter[2];
                    smp.get2d(u_light);
                    smp.get2d(u_scatter);
                    Ld_in = estimate_direct(is, bsdf, u_scatter, 0, u_light) / 1.f;
                } else if (S.n_lights > 1) {
                    // UniformSampleOneLight with the spatial distribution (integrator.cpp:85-106)
                    const LightDist &dist = light_distribution(is.p);
                    float light_pdf;
                    int light_num = sample_discrete(dist, smp.get1d(), &light_pdf);
                    if (light_pdf != 0) {  // `if (lightPdf == 0) return Spectrum(0.f)` before any Get2D
                        float u_light[2], u_scatter[2];
                        smp.get2d(u_light);
                        smp.get2d(u_scatter);
                        Ld_in = estimate_direct(is, bsdf, u_scatter, light_num, u_light) / light_pdf;
                    }
                }
                Rgb Ld = beta * Ld_in;
                if (Ld.is_black()) ++ctr->zero_radiance;
                L = L + Ld;
            }
            V3 wo = -ray.d, wi;
            float pdf;
            float u[2];
            smp.get2d(u);
            pdf = 0;
            bool sampled_specular = false, sampled_transmission = false;
            Rgb f = bsdf_sample_f(bsdf, wo, &wi, u, &pdf, true, &sampled_specular, &sampled_transmission);
            if (f.is_black() || pdf == 0.f) break;
            beta = beta * (f * absdot(wi, is.sn) / pdf);
            if (beta.y() < 0.f || std::isnan(beta.y())) return L;
            specular_bounce = sampled_specular;
            if (sampled_specular && sampled_transmission) {  // path.cpp:151-157
                float eta = bsdf.eta;
                eta_scale *= (dot(wo, is.n) > 0) ? (eta * eta) : 1 / (eta * eta);
            }
            ray = spawn_ray(is, wi);
            Rgb rr_beta = beta * eta_scale;
            if (rr_beta.max_component() < rr_threshold && bounces > 3) {
                float q = std::max(.05f, 1 - rr_beta.max_component());
                if (smp.get1d() < q) break;
                beta = beta / (1 - q);
            }
        }
        ctr->path_length[std::min(bounces, 7)]++;
        return L;
    }

    // one camera sample -> guarded radiance (integrator.cpp:270-314)
    Rgb sample_radiance(int px, int py, int64_t k, float *pfilm) const {
        Sampler smp{this, sample_index(px, py, k), 0, px, py};
        float u[2];
        smp.get2d(u);
        pfilm[0] = float(px) + u[0];
        pfilm[1] = float(py) + u[1];
        smp.get1d();  // time
        float plens[2];
        smp.get2d(plens);
        RayDiff rdiff;
        Ray ray = probe ? probe_ray(pfilm[0], pfilm[1]) : camera_ray(pfilm[0], pfilm[1], plens, S.n_textures > 0 ? &rdiff : nullptr);
        if (probe && S.n_textures > 0) rdiff = probe_differentials(pfilm[0], pfilm[1], ray);
        ++ctr->camera_rays;
        Rgb L = li(ray, smp, rdiff, probe ? const_cast<float *>(probe_aux) : nullptr);
        if (L.has_nans())
            L = Rgb(0.f);
        else if (L.y() < -1e-5)
            L = Rgb(0.f);
        else if (std::isinf(L.y()))
            L = Rgb(0.f);
        return L;
    }

    // ========================================================================
    // The IISPT render runner's gather (second half): what IisptRenderRunner::run does with the predicted
    // hemispheres (integrators/iisptrenderrunner.cpp:414-596). The random numbers: the reference draws them from one
    // PCG32 per thread (IisptRng(thread_no)) in whatever order its tasks are scheduled; here every film pixel of a task
    // has its own stream RNG(rng_seed + pixel rank), and the camera samples come from the runner's "one sampler pixel per
    // call" counter (sampler_next_pixel, :941-953) in the single-thread order: hemi points row by row, then film pixels.
    // ========================================================================
    struct Pcg {  // core/rng.h:62-156
        uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
        explicit Pcg(uint64_t seq) {  // RNG(sequenceIndex) -> SetSequence
            state = 0u;
            inc = (seq << 1u) | 1u;
            uniform_u32();
            state += 0x853c49e6748fea9bULL;
            uniform_u32();
        }
        uint32_t uniform_u32() {
            uint64_t oldstate = state;
            state = oldstate * 0x5851f42d4c957f2dULL + inc;
            uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
            uint32_t rot = (uint32_t)(oldstate >> 59u);
            return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
        }
        uint32_t uniform_u32(uint32_t b) {
            uint32_t threshold = (~b + 1u) % b;
            while (true) {
                uint32_t r = uniform_u32();
                if (r >= threshold) return r % b;
            }
        }
        float uniform_float() { return std::min(OneMinusEpsilon, float(uniform_u32() * 0x1p-32f)); }
    };
    // ========================================================================
    // The IISPT integrator's DIRECT pass: DirectProgressiveIntegrator::Li / RenderOnePass
    // (integrators/directprogressiveintegrator.cpp:22-150) as IisptRenderRunner::run_direct drives it
    // (iisptrenderrunner.cpp:601-633), with the sampler CreateIISPTIntegrator makes (iispt.cpp:813-816): a RandomSampler of
    // PbrtOptions.iileDirectSamples = 16 samples per pixel (pbrt.h:178), cloned per runner thread with seed
    // 6284 + 17 * thread_no. preprocess() requests, for each of maxDepth = 5 levels and each light, two 2D arrays of
    // nLightSamples = RoundCount(light->nSamples) entries per pixel sample (killeroo-simple's area light: "nsamples" 8; round 3
    // assumed 1 everywhere, which its advisor caught); RenderOnePass calls StartPixel for every pixel of every pass and never
    // StartNextSample, so each pixel reads the first nLightSamples entries of every array and is a fresh draw per pass.
    //
    // Random numbers: the reference's RandomSampler is ONE PCG32 stream per thread, consumed pixel after pixel (arrays, camera
    // sample, Li), and which thread renders which pass is a race (getNextDirectPass) — its image is not a function of its
    // inputs. Restated here as a function: pass p is rendered with the seed a thread with thread_no = p would clone
    // (6284 + 17 p), and every pixel of it has its own stream RNG((seed << 32) + pixel rank in the sample bounds), consumed
    // in the reference's per-pixel order: RandomSampler::StartPixel fills ALL the arrays (16 entries each, x then y:
    // samplers/random.cpp:62-72), then GetCameraSample (pFilm, time, pLens: sampler.cpp:46-52), then Li's Get2D calls.
    // The functions are the reference's; the schedule of a thread pool is not restated (as for the gather above).
    struct DirectSampler {
        static constexpr int kSpp = 16, kMaxDepth = 5;
        Pcg rng;
        std::vector<float> entry0;   // per 2D array: its entries for pixel sample 0 (the first nSamples of its 16 x nSamples)
        std::vector<size_t> start;   // per 2D array: where they begin in entry0
        size_t array_offset = 0;     // Sampler::array2DOffset
        // n_samples[l] = nLightSamples of light l (Light::nSamples, RoundCount is the identity for a RandomSampler)
        DirectSampler(uint64_t seq, int n_lights, const int *n_samples) : rng(seq) {
            for (int d = 0; d < kMaxDepth; ++d)
                for (int l = 0; l < n_lights; ++l)
                    for (int rep = 0; rep < 2; ++rep) {  // Request2DArray(nLightSamples[j]) twice, preprocess()
                        const int n = n_samples[l];
                        start.push_back(entry0.size());
                        for (int j = 0; j < n * kSpp; ++j) {  // sampleArray2D[i][j] = {rng.UniformFloat(), rng.UniformFloat()}
                            const float x = rng.uniform_float(), y = rng.uniform_float();
                            if (j < n) entry0.push_back(x), entry0.push_back(y);
                        }
                    }
        }
        const float *get2d_array() {  // Sampler::Get2DArray(n), sampler.cpp:97-102, currentPixelSampleIndex = 0
            if (array_offset == start.size()) return nullptr;
            return entry0.data() + start[array_offset++];
        }
        float get1d() { return rng.uniform_float(); }
        void get2d(float *u) {
            u[0] = rng.uniform_float();
            u[1] = rng.uniform_float();
        }
    };
    // BSDF::Sample_f(wo, &wi, u, &pdf, BSDF_REFLECTION | BSDF_SPECULAR) (reflection.cpp:719-784): of the lobes built here only
    // SpecularReflection matches that type — the mirror / uber lobe, or glass's SpecularReflection(R, FresnelDielectric(1, eta)).
    // One matching lobe: the remapped u is not used.
    Rgb sample_specular_reflection(const Bsdf &b, V3 woW, V3 *wiW, float *pdf) const {
        *pdf = 0;
        if (!b.has_spec) return Rgb(0.f);  // matchingComps == 0
        if (b.spec_glass && b.kr.is_black()) return Rgb(0.f);  // glass.cpp:75: no reflection lobe for a black Kr
        V3 wo = b.to_local(woW);
        if (wo.z == 0) return Rgb(0.f);
        V3 wi = V3(-wo.x, -wo.y, wo.z);  // SpecularReflection::Sample_f, reflection.cpp:136-143
        *pdf = 1;
        const float fr = b.spec_noop ? 1.f : fr_dielectric(wi.z, 1.f, b.spec_eta);
        Rgb f = Rgb(fr) * b.kr / std::abs(wi.z);
        *wiW = b.to_world(wi);
        return f;
    }
    // BSDF::Sample_f(wo, &wi, u, &pdf, BSDF_TRANSMISSION | BSDF_SPECULAR) (reflection.cpp:719-784). The lobes of that type: the
    // SpecularTransmission(T, 1, eta, Radiance) glass carries beside its SpecularReflection when ComputeScatteringFunctions runs
    // with allowMultipleLobes = false (glass.cpp:62-90, interaction.h:130-133 — the direct integrator's Li does), and UberMaterial's
    // pass-through (first) and Kt lobe (last), uber.cpp:53-61, 94-99. With two of them u[0] picks one and the pdf is halved;
    // SpecularTransmission::Sample_f, reflection.cpp:154-170
    Rgb sample_specular_transmission(const Bsdf &b, V3 woW, const float u[2], V3 *wiW, float *pdf) const {
        *pdf = 0;
        Rgb lobe_t[3];
        float lobe_eta[3];
        int matching = 0;
        if (b.has_t0) lobe_t[matching] = b.t0, lobe_eta[matching++] = 1.f;
        if (b.has_spec && b.spec_glass && !b.kt.is_black()) lobe_t[matching] = b.kt, lobe_eta[matching++] = b.spec_eta;
        if (b.has_t1) lobe_t[matching] = b.t1, lobe_eta[matching++] = b.t1_eta;
        if (matching == 0) return Rgb(0.f);
        const int comp = std::min(int(std::floor(u[0] * matching)), matching - 1);
        V3 wo = b.to_local(woW);
        if (wo.z == 0) return Rgb(0.f);
        const float eta_a = 1.f, eta_b = lobe_eta[comp];
        const bool entering = wo.z > 0;
        const float eta_i = entering ? eta_a : eta_b, eta_t = entering ? eta_b : eta_a;
        // Refract(wo, Faceforward(Normal3f(0, 0, 1), wo), etaI / etaT, wi), reflection.h:96-108
        V3 n = (wo.z < 0.f) ? -V3(0, 0, 1) : V3(0, 0, 1);
        const float eta = eta_i / eta_t;
        const float cos_i = dot(n, wo);
        const float sin2_i = std::max(0.f, 1 - cos_i * cos_i);
        const float sin2_t = eta * eta * sin2_i;
        if (sin2_t >= 1) return Rgb(0.f);
        const float cos_t = std::sqrt(1 - sin2_t);
        V3 wi = eta * -wo + (eta * cos_i - cos_t) * n;
        *pdf = 1;
        Rgb ft = lobe_t[comp] * (1.f - fr_dielectric(wi.z, eta_a, eta_b));  // T * (Spectrum(1.) - fresnel.Evaluate(CosTheta(*wi)))
        ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
        *wiW = b.to_world(wi);
        if (matching > 1) *pdf /= matching;  // (a specular lobe: the other lobes' pdf and f are not added)
        return ft / std::abs(wi.z);
    }
    // UniformSampleAllLights, integrator.cpp:54-83
    Rgb uniform_sample_all_lights(const Isect &it, const Bsdf &bsdf, DirectSampler &smp) const {
        Rgb L(0.f);
        for (int j = 0; j < S.n_lights; ++j) {
            const int n_samples = std::max(1, int(S.lights[j].n_samples));
            const float *u_light_array = smp.get2d_array();
            const float *u_scattering_array = smp.get2d_array();
            if (!u_light_array || !u_scattering_array) {
                float u_light[2], u_scattering[2];
                smp.get2d(u_light);
                smp.get2d(u_scattering);
                L = L + estimate_direct(it, bsdf, u_scattering, j, u_light);
            } else {
                Rgb Ld(0.f);
                for (int k = 0; k < n_samples; ++k) Ld = Ld + estimate_direct(it, bsdf, u_scattering_array + 2 * k, j, u_light_array + 2 * k);
                L = L + Ld / float(n_samples);
            }
        }
        return L;
    }
    // DirectProgressiveIntegrator::Li, directprogressiveintegrator.cpp:22-58. Differentials: the camera ray's, and — they feed
    // the texture filtering of whatever a mirror shows — the reflected rays' (SpecularReflect, :165-184), from the hit's dpdx /
    // dpdy, du/dx .. and shading.dndu / dndv (triangle_interaction, sphere_interaction).
    Rgb direct_li(Ray ray, DirectSampler &smp, RayDiff rdiff, int depth) const {
        Rgb L(0.f);
        Isect is;
        if (!intersect(ray, &is)) {
            for (int l = 0; l < S.n_lights; ++l)  // `for (const auto &light : scene.lights) L += light->Le(ray)`
                if (S.lights[l].type == IILE_LIGHT_INFINITE) L = L + inf_le(S.lights[l], ray.d);
            return L;
        }
        if (S.n_textures > 0) compute_differentials(&is, rdiff);
        {
            const iile_material &mb = S.materials[S.prim_material[is.prim]];
            if (S.n_textures > 0 && mb.bump_tex >= 0) bump(mb.bump_tex, &is);
        }
        Bsdf bsdf = make_bsdf(is);  // (every primitive of a flattened scene has a material: isect.bsdf is never null)
        V3 wo = is.wo;
        L = L + isect_le(is, wo);
        if (S.n_lights > 0) L = L + uniform_sample_all_lights(is, bsdf, smp);
        if (depth + 1 < DirectSampler::kMaxDepth) {
            {  // SpecularReflect, :134-190
                float u[2];
                smp.get2d(u);
                V3 wi;
                float pdf;
                Rgb f = sample_specular_reflection(bsdf, wo, &wi, &pdf);
                Rgb R(0.f);
                if (pdf > 0.f && !f.is_black() && absdot(wi, is.sn) != 0.f) {
                    RayDiff rd;  // `RayDifferential rd = isect.SpawnRay(wi); if (ray.hasDifferentials) { ... }`
                    if (rdiff.has) {
                        const V3 ns = is.sn;
                        rd.has = true;
                        rd.rxo = is.p + is.dpdx;
                        rd.ryo = is.p + is.dpdy;
                        const V3 dndx = is.dndu * is.dudx + is.dndv * is.dvdx;
                        const V3 dndy = is.dndu * is.dudy + is.dndv * is.dvdy;
                        const V3 dwodx = -rdiff.rxd - wo, dwody = -rdiff.ryd - wo;
                        const float dDNdx = dot(dwodx, ns) + dot(wo, dndx);
                        const float dDNdy = dot(dwody, ns) + dot(wo, dndy);
                        rd.rxd = wi - dwodx + 2.f * V3(dot(wo, ns) * dndx + dDNdx * ns);
                        rd.ryd = wi - dwody + 2.f * V3(dot(wo, ns) * dndy + dDNdy * ns);
                    }
                    R = f * direct_li(spawn_ray(is, wi), smp, rd, depth + 1) * absdot(wi, is.sn) / pdf;
                }
                L = L + R;
            }
            {  // SpecularTransmit, :190-237: glass's and uber's SpecularTransmission lobes
                float u[2];
                smp.get2d(u);
                V3 wi;
                float pdf;
                Rgb f = sample_specular_transmission(bsdf, wo, u, &wi, &pdf);
                Rgb T(0.f);
                if (pdf > 0.f && !f.is_black() && absdot(wi, is.sn) != 0.f) {
                    RayDiff rd;
                    if (rdiff.has) {
                        const V3 ns = is.sn;
                        rd.has = true;
                        rd.rxo = is.p + is.dpdx;
                        rd.ryo = is.p + is.dpdy;
                        float eta = bsdf.eta;
                        const V3 w = -wo;
                        if (dot(wo, ns) < 0) eta = 1.f / eta;
                        const V3 dndx = is.dndu * is.dudx + is.dndv * is.dvdx;
                        const V3 dndy = is.dndu * is.dudy + is.dndv * is.dvdy;
                        const V3 dwodx = -rdiff.rxd - wo, dwody = -rdiff.ryd - wo;
                        const float dDNdx = dot(dwodx, ns) + dot(wo, dndx);
                        const float dDNdy = dot(dwody, ns) + dot(wo, dndy);
                        const float mu = eta * dot(w, ns) - dot(wi, ns);
                        const float dmudx = (eta - (eta * eta * dot(w, ns)) / dot(wi, ns)) * dDNdx;
                        const float dmudy = (eta - (eta * eta * dot(w, ns)) / dot(wi, ns)) * dDNdy;
                        rd.rxd = wi + eta * dwodx - V3(mu * dndx + dmudx * ns);
                        rd.ryd = wi + eta * dwody - V3(mu * dndy + dmudy * ns);
                    }
                    T = f * direct_li(spawn_ray(is, wi), smp, rd, depth + 1) * absdot(wi, is.sn) / pdf;
                }
                L = L + T;
            }
        }
        return L;
    }
    // one pixel of one pass: RenderOnePass's loop body (:84-140); returns false outside the pixel bounds
    Rgb direct_pixel(int px, int py, int pass, int rank) const {
        int n_samples[64];
        for (int l = 0; l < S.n_lights && l < 64; ++l) n_samples[l] = std::max(1, int(S.lights[l].n_samples));
        DirectSampler smp((uint64_t(6284 + 17 * pass) << 32) + uint64_t(rank), std::min(S.n_lights, 64), n_samples);
        float u[2], plens[2];
        smp.get2d(u);   // GetCameraSample: pFilm = pixel + Get2D(), time = Get1D(), pLens = Get2D()
        smp.get1d();
        smp.get2d(plens);
        RayDiff rdiff;
        // ray.ScaleDifferentials(1 / sqrt(samplesPerPixel)) with the RandomSampler's 16 samples per pixel
        Ray ray = camera_ray(float(px) + u[0], float(py) + u[1], plens, S.n_textures > 0 ? &rdiff : nullptr, false,
                             1 / std::sqrt(float(DirectSampler::kSpp)));
        Rgb L = direct_li(ray, smp, rdiff, 0);
        if (L.has_nans())
            L = Rgb(0.f);
        else if (L.y() < -1e-5)
            L = Rgb(0.f);
        else if (std::isinf(L.y()))
            L = Rgb(0.f);
        return L;
    }

    // a hemi point's camera as the gather sees it (HemisphericCamera: hemispheric.h:23-82, hemispheric.cpp:109-160)
    struct HemiCam {
        bool valid = false;
        float c2w[16], w2c[16];  // CameraToWorld's matrix; WorldToCamera's matrix (its numerical inverse)
        V3 origin, look;
        const float *nn;         // the predicted intensity image, hemi x hemi x 3, row 0 = top scanline (ImageFilm order)
    };
    static bool make_hemi_cam(const float pos[3], const float dir[3], const float *nn, HemiCam *hc) {
        ProbeCam pc;
        if (!make_probe_camera(pos, dir, 0, &pc)) return false;
        std::memcpy(hc->c2w, pc.c2w, sizeof(pc.c2w));
        if (!invert4(pc.c2w, hc->w2c)) return false;
        hc->origin = V3(pos[0], pos[1], pos[2]);
        hc->look = V3(dir[0], dir[1], dir[2]);
        hc->nn = nn;
        hc->valid = true;
        return true;
    }
    // sampler_next_pixel + GetCameraSample + GenerateRayDifferential for the counter-th call (counter >= 1): the
    // sampler sits on pixel (counter, 0), sample 0; the camera sample is `pixel` + its first two dimensions
    Ray iispt_camera_ray(int fx, int fy, uint32_t counter, Sampler *smp_out, RayDiff *rd) const {
        Sampler smp{this, sample_index(int(counter), 0, 0), 0, int(counter), 0};
        float u[2];
        smp.get2d(u);
        const float pfx = float(fx) + u[0], pfy = float(fy) + u[1];
        smp.get1d();  // time
        float plens[2];
        smp.get2d(plens);
        RayDiff rdiff;
        Ray ray = camera_ray(pfx, pfy, plens, S.n_textures > 0 ? &rdiff : nullptr, true);  // r.ScaleDifferentials(1.0)
        *smp_out = smp;
        *rd = rdiff;
        return ray;
    }
    // IisptRenderRunner::find_intersection, iisptrenderrunner.cpp:632-757 (without the emitted / background outputs,
    // which belong to the direct pass)
    bool iispt_find_intersection(Ray ray, RayDiff rdiff, Sampler &smp, Isect *is_out, Ray *ray_out, Rgb *beta_out) const {
        Rgb beta(1.f);
        for (int bounces = 0; bounces < 24; ++bounces) {
            Isect is;
            if (!intersect(ray, &is)) return false;
            if (S.prim_material[is.prim] < 0) {  // `if (!isect.bsdf)`: skip this intersection
                ray = spawn_ray(is, ray.d);
                rdiff.has = false;
                continue;
            }
            if (S.n_textures > 0) compute_differentials(&is, rdiff);
            rdiff.has = false;
            {
                const iile_material &mb = S.materials[S.prim_material[is.prim]];
                if (S.n_textures > 0 && mb.bump_tex >= 0) bump(mb.bump_tex, &is);
            }
            Bsdf bsdf = make_bsdf(is);
            const V3 wo = -ray.d;
            V3 wi;
            float pdf = 0, u[2];
            smp.get2d(u);
            bool spec = false, trans = false;
            Rgb f = bsdf_sample_f(bsdf, wo, &wi, u, &pdf, true, &spec, &trans);
            if (f.is_black() || pdf == 0.f) {
                *beta_out = Rgb(0.f);
                return true;
            }
            if (!spec) {  // the current bounce is not specular: IISPT proceeds from here
                *is_out = is;
                *ray_out = ray;
                *beta_out = beta;
                return true;
            }
            beta = beta * (f * absdot(wi, is.sn) / pdf);
            if (beta.y() < 0.f || std::isnan(beta.y())) {
                *beta_out = Rgb(0.f);
                return true;
            }
            ray = spawn_ray(is, wi);
        }
        *beta_out = Rgb(0.f);
        return true;
    }
    // the aux ray of a first hit: `isect.SpawnRay(Vector3f(surface_normal))` with the normal turned against the ray
    // (iisptrenderrunner.cpp:299-312 and :969-980)
    static Ray iispt_aux_ray(const Isect &is, const Ray &ray) {
        V3 n = is.n;
        if (dot(is.n, ray.d) > 0.0) n = -is.n;
        return spawn_ray(is, n);
    }
    // HemisphericCamera::get_light_sample_nn (hemispheric.cpp:89-105) / getLightSampleNn (:44-60) over
    // IntensityFilm::get_camera_coord_jacobian (film/intensityfilm.cpp:60-66); `jac` = sin(pi * y / hemi), y < hemi
    Rgb iispt_nn_pixel(const HemiCam &hc, int x, int y, int hemi, const float *jac) const {
        const float *px = hc.nn + 3 * (size_t(hemi - 1 - y) * hemi + x);  // film->get(x, height - 1 - y)
        return Rgb(px[0] * jac[y], px[1] * jac[y], px[2] * jac[y]);
    }
    Rgb iispt_light_sample_xy(const HemiCam &hc, int x, int y, int hemi, const float *jac, V3 *wi) const {
        float theta = Pi * y / hemi;
        float phi = Pi * x / hemi;
        V3 dir(trig.sin_f(theta) * trig.cos_f(phi), trig.cos_f(theta), trig.sin_f(theta) * trig.sin_f(phi));
        *wi = xf_vector(M4{hc.c2w}, dir);  // CameraToWorld(ray).d
        return iispt_nn_pixel(hc, x, y, hemi, jac);
    }
    Rgb iispt_light_sample_dir(const HemiCam &hc, V3 wi, int hemi, const float *jac) const {
        V3 wc = xf_vector(M4{hc.w2c}, wi);
        float theta = trig.acos_f(wc.y);
        float phi = trig.atan2_f(wc.z, wc.x);
        if (std::isnan(theta) || std::isnan(phi)) return Rgb(0.f);  // (int)NaN is INT_MIN on x86: outside the film
        int y = int(hemi * theta / Pi);
        int x = int(hemi * phi / Pi);
        if (x >= 0 && x < hemi && y >= 0 && y < hemi) return iispt_nn_pixel(hc, x, y, hemi, jac);
        return Rgb(0.f);
    }
    // estimate_direct, iisptrenderrunner.cpp:16-140
    Rgb iispt_estimate_direct(const Isect &it, const Bsdf &bsdf, int rx, int ry, const HemiCam &hc, int hemi, const float *jac, Pcg &rng) const {
        Rgb Ld(0.f);
        V3 wi;
        const float light_pdf = float(1.0 / 6.28);
        const float BSDF_RATIO = float(0.4394);
        const float EM_RATIO = float(1.098);
        float scattering_pdf = 0;
        Rgb Li = iispt_light_sample_xy(hc, rx, ry, hemi, jac, &wi);
        if (light_pdf > 0 && !Li.is_black()) {
            Rgb f = bsdf_f(bsdf, it.wo, wi) * absdot(wi, it.sn);
            scattering_pdf = bsdf_pdf(bsdf, it.wo, wi);
            if (!f.is_black()) {
                if (!Li.is_black()) {
                    float weight = power_heuristic(1, light_pdf, 1, scattering_pdf);
                    Ld = Ld + EM_RATIO * f * Li * weight / light_pdf;
                }
            }
        }
        {
            // `Point2f uScattering(rng->uniform_float(), rng->uniform_float())`: g++ evaluates the arguments right to
            // left, so the first draw is y
            float u[2];
            u[1] = rng.uniform_float();
            u[0] = rng.uniform_float();
            Rgb f = bsdf_sample_f(bsdf, it.wo, &wi, u, &scattering_pdf);
            f = f * absdot(wi, it.sn);
            if (!f.is_black() && scattering_pdf > 0) {
                float weight = power_heuristic(1, scattering_pdf, 1, light_pdf);  // (no specular lobe can be sampled)
                Rgb Li2 = iispt_light_sample_dir(hc, wi, hemi, jac);
                if (!Li2.is_black()) Ld = Ld + BSDF_RATIO * f * Li2 * weight / scattering_pdf;
            }
        }
        return Ld;
    }
    // IisptRenderRunner::sample_hemisphere, iisptrenderrunner.cpp:142-178 (HEMISPHERIC_IMPORTANCE_SAMPLES = 16)
    Rgb iispt_sample_hemisphere(const Isect &it, const Bsdf &bsdf, int len, const float *weights, const HemiCam *const *cams, int hemi,
                                const float *jac, Pcg &rng) const {
        Rgb L(0.f);
        int samples_taken = 0;
        for (int i = 0; i < len; i++) {
            for (int j = 0; j < 16; j++) {
                float rr = rng.uniform_float();
                if (rr < weights[i]) {
                    samples_taken++;
                    if (cams[i] != nullptr) {
                        int rx = int(rng.uniform_u32(uint32_t(hemi)));
                        int ry = int(rng.uniform_u32(uint32_t(hemi)));
                        L = L + iispt_estimate_direct(it, bsdf, rx, ry, *cams[i], hemi, jac, rng);
                    }
                }
            }
        }
        if (samples_taken > 0) return L / float(samples_taken);
        return Rgb(0.f);
    }
    // IisptRenderRunner::compute_fpixel_weights, iisptrenderrunner.cpp:961-1039 with tools/iisptmathutils.h:44-130, 179-197
    void iispt_fpixel_weights(int len, const int (*neigh)[2], const HemiCam *const *cams, int fx, int fy, const Isect &f_isect, int tilesize,
                              const Ray &f_ray, V3 main_cam_origin, float *out) const {
        const Ray aux = iispt_aux_ray(f_isect, f_ray);
        float wdpos[4], wdnor[4], wdd[4], wod[4];
        for (int i = 0; i < len; i++) {  // weighting_distance_positions
            float dx2 = float(fx - neigh[i][0]);
            dx2 = dx2 * dx2;
            float dy2 = float(fy - neigh[i][1]);
            dy2 = dy2 * dy2;
            const float pdist = std::sqrt(dx2 + dy2);
            const float tile_distance = float(tilesize);
            float res = tile_distance != 0.0 ? pdist / tile_distance : pdist;
            wdpos[i] = res < 0.0 ? 0.f : (res > 1.0 ? 1.f : res);
        }
        for (int i = 0; i < len; i++) {  // weighting_distance_normals
            if (!cams[i]) {
                wdnor[i] = 0.0f;
                continue;
            }
            V3 a = aux.d, b = cams[i]->look;
            const float al = length(a), bl = length(b);
            if (al <= 0.0 || bl <= 0.0) {
                wdnor[i] = 1.f;
                continue;
            }
            a = vdiv(a, al);
            b = vdiv(b, bl);
            const float dt = dot(a, b);
            wdnor[i] = dt < 0.0 ? 1.f : 1.f - dt;
        }
        for (int i = 0; i < len; i++) {  // weightingCameraDistance
            if (!cams[i]) {
                wdd[i] = 0.0f;
                continue;
            }
            const float i2c = length(main_cam_origin - f_isect.p);
            if (i2c < 1e-10) {
                wdd[i] = 0.f;
                continue;
            }
            const float s2c = length(main_cam_origin - cams[i]->origin);
            float rel = std::abs(i2c - s2c) / i2c;
            rel *= 1.f;
            const float w = 1.0f - rel;
            wdd[i] = w < 0.f ? 0.f : (w > 1.f ? 1.f : w);
        }
        for (int i = 0; i < len; i++) wod[i] = wdpos[i] * wdnor[i] + wdpos[i] * wdd[i] + wdpos[i];
        for (int i = 0; i < len; i++) out[i] = float(std::max(0.0, 2.0 - double(wod[i])) + 0.001);
        float tot = 0.0;
        for (int i = 0; i < len; i++) tot += out[i];
        if (tot > 0.0)
            for (int i = 0; i < len; i++) out[i] = out[i] / tot;
    }
};

// SampleDiscrete on a uniform distribution: FindInterval over cdf[i] = i/n picks
// the last i with cdf[i] <= u, i.e. min(int(u*n), n-1) for n == 1 (the only
// case this path supports on device); for n > 1 the product i/n is compared in
// float, which this helper does not reproduce bit-exactly — rejected at load.

// ----------------------------------------------------------------------------
// film tile (core/film.h:140-213, core/film.cpp:92-103, 135-148)
struct TilePixel {
    float rgb[3] = {0, 0, 0};
    float wsum = 0;
};
struct FilmTile {
    int x0, y0, x1, y1;  // pixel bounds
    std::vector<TilePixel> px;
};

}  // namespace

extern "C" {

}  // extern "C"

// SamplerIntegrator::Render's tile loop (integrator.cpp:227-339); with `probe`: IISPTdIntegrator::RenderView
// (iispt_d.cpp:388-470) — pixels outside the film's pixel bounds are skipped (:428-429) and the first hits' camera-space
// normals and distances go to aux_nd[(y * w + x) * 4]
static int render_impl(const iile_scene_desc *scene, int trig_mode, int n_threads, int k_begin, int k_end, int tile_rank,
                       int tile_nranks, float *film_xyzw, oracle_stats *stats, const ProbeCam *probe, float *aux_nd) {
    if (!scene || !film_xyzw) return 1;
    const iile_scene_desc &S = *scene;
    const iile_film_desc &F = S.film;
    if (k_end < 0) {
        k_begin = 0;
        k_end = S.halton.spp;
    }
    if (tile_nranks <= 0) {
        tile_rank = 0;
        tile_nranks = 1;
    }
    if (n_threads <= 0) n_threads = int(std::thread::hardware_concurrency());
    if (n_threads <= 0) n_threads = 1;
    const int tile_size = 16;
    const int sx = F.samp_x1 - F.samp_x0, sy = F.samp_y1 - F.samp_y0;
    const int ntx = (sx + tile_size - 1) / tile_size, nty = (sy + tile_size - 1) / tile_size;
    const int n_tiles = ntx * nty;
    std::vector<std::unique_ptr<FilmTile>> tiles(n_tiles);
    std::vector<Counters> counters(n_threads);
    std::atomic<int> next_tile(0);
    auto t_start = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        Oracle orc(S, trig_mode, &counters[tid]);
        orc.probe = probe;
        while (true) {
            int tile = next_tile.fetch_add(1);
            if (tile >= n_tiles) break;
            int tx = tile % ntx, ty = tile / ntx;
            if (iile_tile_owner(tx, ty, tile_nranks) != tile_rank) continue;
            int x0 = F.samp_x0 + tx * tile_size, x1 = std::min(x0 + tile_size, F.samp_x1);
            int y0 = F.samp_y0 + ty * tile_size, y1 = std::min(y0 + tile_size, F.samp_y1);
            // Film::GetFilmTile, film.cpp:92-103
            std::unique_ptr<FilmTile> ft(new FilmTile);
            ft->x0 = std::max(int(std::ceil(float(x0) - 0.5f - F.filter_rx)), F.crop_x0);
            ft->y0 = std::max(int(std::ceil(float(y0) - 0.5f - F.filter_ry)), F.crop_y0);
            ft->x1 = std::min(int(std::floor(float(x1) - 0.5f + F.filter_rx)) + 1, F.crop_x1);
            ft->y1 = std::min(int(std::floor(float(y1) - 0.5f + F.filter_ry)) + 1, F.crop_y1);
            const int tw = std::max(0, ft->x1 - ft->x0), th = std::max(0, ft->y1 - ft->y0);
            ft->px.resize(size_t(tw) * th);
            for (int py = y0; py < y1; ++py)
                for (int px = x0; px < x1; ++px)
                    for (int64_t k = k_begin; k < k_end; ++k) {
                        if (probe && (px < F.crop_x0 || px >= F.crop_x1 || py < F.crop_y0 || py >= F.crop_y1)) continue;
                        // `if (!InsideExclusive(pixel, pixelBounds)) continue;` (integrator.cpp:272; "pixelbounds", path.cpp:216-229)
                        const int32_t *pb = S.integrator.pixel_bounds;
                        if (!probe && !(px >= pb[0] && px < pb[2] && py >= pb[1] && py < pb[3])) continue;
                        float pf[2];
                        Rgb L = orc.sample_radiance(px, py, k, pf);
                        if (probe && aux_nd)
                            std::memcpy(aux_nd + 4 * (size_t(py - F.crop_y0) * (F.crop_x1 - F.crop_x0) + (px - F.crop_x0)),
                                        orc.probe_aux, 4 * sizeof(float));
                        if (L.y() > F.max_sample_luminance) L = L * (F.max_sample_luminance / L.y());
                        // FilmTile::AddSample, film.h:153-193
                        float dxf = pf[0] - 0.5f, dyf = pf[1] - 0.5f;
                        int ax0 = std::max(int(std::ceil(dxf - F.filter_rx)), ft->x0);
                        int ay0 = std::max(int(std::ceil(dyf - F.filter_ry)), ft->y0);
                        int ax1 = std::min(int(std::floor(dxf + F.filter_rx)) + 1, ft->x1);
                        int ay1 = std::min(int(std::floor(dyf + F.filter_ry)) + 1, ft->y1);
                        const float inv_rx = 1 / F.filter_rx, inv_ry = 1 / F.filter_ry;  // Filter::invRadius
                        for (int y = ay0; y < ay1; ++y) {
                            const float fy = std::abs((y - dyf) * inv_ry * 16);
                            const int ify = std::min(int(std::floor(fy)), 15);
                            for (int x = ax0; x < ax1; ++x) {
                                const float fx = std::abs((x - dxf) * inv_rx * 16);
                                const int ifx = std::min(int(std::floor(fx)), 15);
                                TilePixel &tp = ft->px[size_t(y - ft->y0) * tw + (x - ft->x0)];
                                const float fw = S.film_filter_table[ify * 16 + ifx];
                                for (int c = 0; c < 3; ++c) tp.rgb[c] += L.c[c] * 1.f * fw;
                                tp.wsum += fw;
                            }
                        }
                    }
            tiles[tile] = std::move(ft);
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n_threads; ++i) th.emplace_back(worker, i);
    worker(0);
    for (auto &t : th) t.join();
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();

    // Film::MergeFilmTile, film.cpp:135-148 — merged here in tile index order so
    // the oracle is deterministic (the reference merges in completion order).
    const int fw = F.crop_x1 - F.crop_x0, fh = F.crop_y1 - F.crop_y0;
    std::memset(film_xyzw, 0, sizeof(float) * 4 * size_t(fw) * fh);
    for (int t = 0; t < n_tiles; ++t) {
        if (!tiles[t]) continue;
        const FilmTile &ft = *tiles[t];
        const int tw = std::max(0, ft.x1 - ft.x0);
        for (int y = ft.y0; y < ft.y1; ++y)
            for (int x = ft.x0; x < ft.x1; ++x) {
                const TilePixel &tp = ft.px[size_t(y - ft.y0) * tw + (x - ft.x0)];
                float *out = film_xyzw + 4 * (size_t(y - F.crop_y0) * fw + (x - F.crop_x0));
                float xyz[3];  // RGBToXYZ, spectrum.h:62-66
                xyz[0] = 0.412453f * tp.rgb[0] + 0.357580f * tp.rgb[1] + 0.180423f * tp.rgb[2];
                xyz[1] = 0.212671f * tp.rgb[0] + 0.715160f * tp.rgb[1] + 0.072169f * tp.rgb[2];
                xyz[2] = 0.019334f * tp.rgb[0] + 0.119193f * tp.rgb[1] + 0.950227f * tp.rgb[2];
                for (int c = 0; c < 3; ++c) out[c] += xyz[c];
                out[3] += tp.wsum;
            }
    }
    if (stats) {
        Counters tot;
        for (const Counters &c : counters) tot.add(c);
        std::memset(stats, 0, sizeof(*stats));
        stats->camera_rays = tot.camera_rays;
        stats->regular_rays = tot.regular_rays;
        stats->shadow_rays = tot.shadow_rays;
        stats->tri_tests = tot.tri_tests;
        stats->tri_hits = tot.tri_hits;
        stats->sphere_tests = tot.sphere_tests;
        stats->nodes_closest = tot.nodes_closest;
        stats->nodes_any = tot.nodes_any;
        stats->nee_evals = tot.nee_evals;
        stats->zero_radiance = tot.zero_radiance;
        for (int i = 0; i < 8; ++i) stats->path_length[i] = tot.path_length[i];
        stats->max_stack_depth = tot.max_stack;
        stats->threads = n_threads;
        stats->seconds = secs;
    }
    return 0;
}

extern "C" {

int oracle_tile_owner(int tx, int ty, int nranks) { return iile_tile_owner(tx, ty, nranks); }

// ---- the IISPT runner's gather (iisptrenderrunner.cpp:248-596) -------------------------------------------------
// hemi points of a task: find_intersection for each, then the aux ray the probe camera is placed on
int oracle_iispt_hemi_points(const iile_scene_desc *scene, int trig_mode, const iile_iispt_task *task, uint8_t *valid, float *pos3, float *dir3) {
    Counters c;
    Oracle orc(*scene, trig_mode, &c);
    const int nx = iile_iispt_grid_count(task->x0, task->x1, task->tilesize), ny = iile_iispt_grid_count(task->y0, task->y1, task->tilesize);
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
            const int k = j * nx + i;
            const int tx = iile_iispt_grid_pos(task->x0, task->x1, task->tilesize, i), ty = iile_iispt_grid_pos(task->y0, task->y1, task->tilesize, j);
            Oracle::Sampler smp{&orc, 0, 0, 0, 0};
            RayDiff rd;
            Ray r = orc.iispt_camera_ray(tx, ty, task->counter_base + 1 + uint32_t(k), &smp, &rd);
            Isect is;
            Ray ray;
            Rgb beta;
            const bool found = orc.iispt_find_intersection(r, rd, smp, &is, &ray, &beta);
            valid[k] = 0;
            for (int a = 0; a < 3; ++a) pos3[3 * k + a] = 0, dir3[3 * k + a] = 0;
            if (!found || beta.y() <= 0.0) continue;  // "set a black hemi"
            const Ray aux = Oracle::iispt_aux_ray(is, ray);
            valid[k] = 1;
            pos3[3 * k] = aux.o.x, pos3[3 * k + 1] = aux.o.y, pos3[3 * k + 2] = aux.o.z;
            dir3[3 * k] = aux.d.x, dir3[3 * k + 1] = aux.d.y, dir3[3 * k + 2] = aux.d.z;
        }
    return nx * ny;
}
// the per-pixel loop: out_rgbw[4 * j ..] = {f_beta * L (RGB), weight 0.5} of film pixel j of the task (row-major), or
// zeros where the runner records nothing
int oracle_iispt_gather(const iile_scene_desc *scene, int trig_mode, const iile_iispt_task *task, const uint8_t *valid, const float *pos3,
                        const float *dir3, const float *nn_films, float *out_rgbw) {
    Counters c;
    Oracle orc(*scene, trig_mode, &c);
    const int hemi = scene->probe.hemi_size;
    const int nx = iile_iispt_grid_count(task->x0, task->x1, task->tilesize), ny = iile_iispt_grid_count(task->y0, task->y1, task->tilesize);
    std::vector<Oracle::HemiCam> cams(size_t(nx) * ny);
    for (int k = 0; k < nx * ny; ++k)
        if (valid[k] && !Oracle::make_hemi_cam(pos3 + 3 * k, dir3 + 3 * k, nn_films + size_t(k) * hemi * hemi * 3, &cams[size_t(k)])) return 1;
    std::vector<float> jac(static_cast<size_t>(hemi), 0.f);  // IntensityFilm::get_camera_coord_jacobian, intensityfilm.cpp:60-66
    for (int y = 0; y < hemi; ++y) {
        float abs_vertical_value = float(y) / hemi;
        float polar_vertical_value = float(M_PI * abs_vertical_value);
        // intensityfilm.cpp includes <math.h>: with libstdc++ `sin(Float)` resolves to the float overload (sinf)
        jac[size_t(y)] = std::sin(polar_vertical_value);
    }
    float zero_lens[2] = {0, 0};
    const V3 main_origin = orc.camera_ray(0.f, 0.f, zero_lens).o;  // Camera::getCameraWorldPosition, camera.cpp:115-124
    const int w = task->x1 - task->x0;
    for (int fy = task->y0; fy < task->y1; ++fy)
        for (int fx = task->x0; fx < task->x1; ++fx) {
            const int j = (fy - task->y0) * w + (fx - task->x0);
            float *out = out_rgbw + 4 * size_t(j);
            out[0] = out[1] = out[2] = out[3] = 0;
            const int ts = task->tilesize;
            auto pmod = [](int i, int n) { return (i % n + n) % n; };
            const int sx = fx - pmod(fx - task->x0, ts), sy = fy - pmod(fy - task->y0, ts);
            const int ex = std::min(sx + ts, task->x1 - 1), ey = std::min(sy + ts, task->y1 - 1);
            const int neigh[4][2] = {{sx, sy}, {ex, ey}, {ex, sy}, {sx, ey}};  // S, E, R, B
            const Oracle::HemiCam *hc[4];
            for (int i = 0; i < 4; ++i) {
                const int gi = iile_iispt_grid_index(task->x0, task->x1, ts, neigh[i][0]), gj = iile_iispt_grid_index(task->y0, task->y1, ts, neigh[i][1]);
                const Oracle::HemiCam &cam = cams[size_t(gj) * nx + gi];
                hc[i] = cam.valid ? &cam : nullptr;
            }
            Oracle::Sampler smp{&orc, 0, 0, 0, 0};
            RayDiff rd;
            Ray r = orc.iispt_camera_ray(fx, fy, task->counter_base + 1 + uint32_t(nx * ny) + uint32_t(j), &smp, &rd);
            Isect f_isect;
            Ray f_ray;
            Rgb f_beta;
            if (!orc.iispt_find_intersection(r, rd, smp, &f_isect, &f_ray, &f_beta)) continue;
            if (f_beta.y() <= 0.0) continue;
            float weights[4];
            orc.iispt_fpixel_weights(4, neigh, hc, fx, fy, f_isect, ts, f_ray, main_origin, weights);
            // f_isect.ComputeScatteringFunctions(f_ray, arena) again (iisptrenderrunner.cpp:548): f_isect came out of
            // find_intersection with its differentials and bump already applied; the BSDF is a function of that state
            const Oracle::Bsdf bsdf = orc.make_bsdf(f_isect);
            Oracle::Pcg rng(task->rng_seed + uint64_t(j));
            const Rgb L = orc.iispt_sample_hemisphere(f_isect, bsdf, 4, weights, hc, hemi, jac.data(), rng);
            const Rgb v = f_beta * L;
            out[0] = v.c[0], out[1] = v.c[1], out[2] = v.c[2], out[3] = 0.5f;
        }
    return 0;
}

int oracle_render(const iile_scene_desc *scene, int trig_mode, int n_threads, int k_begin, int k_end, int tile_rank,
                  int tile_nranks, float *film_xyzw, oracle_stats *stats) {
    return render_impl(scene, trig_mode, n_threads, k_begin, k_end, tile_rank, tile_nranks, film_xyzw, stats, nullptr, nullptr);
}

// One IISPT probe (iisptrenderrunner.cpp:316-346): CreateHemisphericCamera(hemi, hemi, pos, dir), IISPTdIntegrator::
// RenderView, then get_intensity_film / get_normal_film / get_distance_film. Outputs are indexed [y][x] in the camera's
// raster coordinates (the reference's ImageFilm stores row height - 1 - y, imagefilm.cpp:26-31, film.cpp:245-254).
int oracle_render_probe(const iile_scene_desc *scene, int trig_mode, const float *pos3, const float *dir3, float *intensity_rgb,
                        float *normals_xyz, float *distance) {
    if (!scene || !pos3 || !dir3) return 1;
    const iile_probe_setup &pr = scene->probe;
    ProbeCam cam;
    if (!make_probe_camera(pos3, dir3, pr.hemi_size, &cam)) return 3;
    iile_scene_desc sp = *scene;  // the probe's film, sampler and depth in place of the frame's
    sp.film = pr.film;
    sp.film_filter_wide = 1;
    std::memcpy(sp.film_filter_table, pr.filter_table, sizeof(pr.filter_table));
    sp.halton.spp = 1;
    sp.halton.sample_at_pixel_center = 0;  // HaltonSampler(1, sampleBounds): the default
    for (int i = 0; i < 2; ++i) {
        sp.halton.base_scales[i] = pr.base_scales[i];
        sp.halton.base_exponents[i] = pr.base_exponents[i];
        sp.halton.mult_inverse[i] = pr.mult_inverse[i];
    }
    sp.halton.sample_stride = pr.sample_stride;
    sp.integrator.max_depth = pr.max_depth;
    const int n = pr.hemi_size;
    std::vector<float> film(size_t(4) * n * n), aux(size_t(4) * n * n, 0.f);
    for (int i = 0; i < n * n; ++i) aux[4 * size_t(i) + 3] = 0.f;  // normal_film / distance_film->clear(): zeros
    int rc = render_impl(&sp, trig_mode, 1, 0, 1, 0, 1, film.data(), nullptr, &cam, aux.data());
    if (rc) return rc;
    for (int i = 0; i < n * n; ++i) {
        // Film::to_rgb_array(1.0), film.cpp:187-225
        const float *px = &film[4 * size_t(i)];
        float rgb[3];
        rgb[0] = 3.240479f * px[0] - 1.537150f * px[1] - 0.498535f * px[2];  // XYZToRGB, spectrum.h:56-60
        rgb[1] = -0.969256f * px[0] + 1.875991f * px[1] + 0.041556f * px[2];
        rgb[2] = 0.055648f * px[0] - 0.204043f * px[1] + 1.057311f * px[2];
        if (px[3] != 0) {
            float inv_wt = 1.f / px[3];
            for (int c = 0; c < 3; ++c) rgb[c] = std::max(0.f, rgb[c] * inv_wt);
        }
        for (int c = 0; c < 3; ++c) {
            rgb[c] += 1.f * 0.f;  // no splats
            rgb[c] *= sp.film.scale;
            if (intensity_rgb) intensity_rgb[3 * size_t(i) + c] = rgb[c];
            if (normals_xyz) normals_xyz[3 * size_t(i) + c] = aux[4 * size_t(i) + c];
        }
        if (distance) distance[i] = aux[4 * size_t(i) + 3];
    }
    return 0;
}

int64_t oracle_halton_index(const iile_scene_desc *scene, int px, int py, int64_t k) {
    Counters c;
    Oracle o(*scene, ORACLE_TRIG_LIBM, &c);
    return o.halton_index(px, py, k);
}
float oracle_halton_sample(const iile_scene_desc *scene, int64_t index, int dim) {
    Counters c;
    Oracle o(*scene, ORACLE_TRIG_LIBM, &c);
    return o.sample_dimension(index, dim);
}
// the scene's sampler (Halton or Sobol'): GetIndexForSample / SampleDimension for pixel (px, py)
int64_t oracle_sample_index(const iile_scene_desc *scene, int px, int py, int64_t k) {
    Counters c;
    Oracle o(*scene, ORACLE_TRIG_LIBM, &c);
    return o.sample_index(px, py, k);
}
float oracle_sample_dimension(const iile_scene_desc *scene, int64_t index, int dim, int px, int py) {
    Counters c;
    Oracle o(*scene, ORACLE_TRIG_LIBM, &c);
    return o.sample_dimension(index, dim, px, py);
}
// Generator-matrix helpers of core/lowdiscrepancy.h on caller-supplied matrices (the reference's own tests of them,
// src/tests/sampling.cpp:75-138, are re-run through these).
uint32_t oracle_reverse_bits32(uint32_t n) { return Oracle::reverse_bits32(n); }
uint32_t oracle_multiply_generator(const uint32_t *C, uint32_t a) {  // lowdiscrepancy.h:93-98
    uint32_t v = 0;
    for (int i = 0; a != 0; ++i, a >>= 1)
        if (a & 1) v ^= C[i];
    return v;
}
float oracle_sample_generator_matrix(const uint32_t *C, uint32_t a, uint32_t scramble) {  // lowdiscrepancy.h:100-109
    return std::min((oracle_multiply_generator(C, a) ^ scramble) * 0x1p-32f, OneMinusEpsilon);
}
void oracle_gray_code_sample(const uint32_t *C, uint32_t n, uint32_t scramble, float *p) {  // lowdiscrepancy.h:113-126
    uint32_t v = scramble;
    for (uint32_t i = 0; i < n; ++i) {
        p[i] = std::min(v * 0x1p-32f /* 1/2^32 */, OneMinusEpsilon);
        v ^= C[__builtin_ctz(i + 1)];
    }
}
// SobolSampleFloat / SobolSampleDouble (lowdiscrepancy.h:262-288) on full 52-column matrices
float oracle_sobol_sample_float(const uint32_t *m32, int64_t a, int dimension, uint32_t scramble) {
    uint32_t v = scramble;
    for (int i = dimension * 52; a != 0; a >>= 1, i++)
        if (a & 1) v ^= m32[i];
    return std::min(v * 0x1p-32f /* 1/2^32 */, OneMinusEpsilon);
}
double oracle_sobol_sample_double(const uint64_t *m64, int64_t a, int dimension, uint64_t scramble) {
    uint64_t result = scramble & ~-(1LL << 52);
    for (int i = dimension * 52; a != 0; a >>= 1, i++)
        if (a & 1) result ^= m64[i];
    return std::min(result * (1.0 / (1ULL << 52)), 0x1.fffffffffffffp-1);
}
// SobolIntervalToIndex (lowdiscrepancy.h:229-252) on caller-supplied VdC matrices
uint64_t oracle_sobol_interval_to_index(const uint64_t *vdc, const uint64_t *vdc_inv, uint32_t m, uint64_t frame, int px, int py) {
    if (m == 0) return 0;
    const uint32_t m2 = m << 1;
    uint64_t index = uint64_t(frame) << m2;
    uint64_t delta = 0;
    for (int c = 0; frame; frame >>= 1, ++c)
        if (frame & 1) delta ^= vdc[c];
    uint64_t b = (((uint64_t)((uint32_t)px) << m) | ((uint32_t)py)) ^ delta;
    for (int c = 0; b; b >>= 1, ++c)
        if (b & 1) index ^= vdc_inv[c];
    return index;
}
static const int kFirstPrimes[64] = {2,   3,   5,   7,   11,  13,  17,  19,  23,  29,  31,  37,  41,  43,  47,  53,
                                     59,  61,  67,  71,  73,  79,  83,  89,  97,  101, 103, 107, 109, 113, 127, 131,
                                     137, 139, 149, 151, 157, 163, 167, 173, 179, 181, 191, 193, 197, 199, 211, 223,
                                     227, 229, 233, 239, 241, 251, 257, 263, 269, 271, 277, 281, 283, 293, 307, 311};
float oracle_radical_inverse(int base_index, uint64_t a) {
    return Oracle::radical_inverse(base_index, kFirstPrimes[base_index & 63], a);
}
float oracle_scrambled_radical_inverse(const iile_scene_desc *scene, int base_index, uint64_t a) {
    const iile_halton &h = scene->halton;
    return Oracle::scrambled_radical_inverse(h.primes[base_index], h.perms + h.prime_sums[base_index], a);
}
float oracle_scrambled_radical_inverse_perm(int base, const uint16_t *perm, uint64_t a) {
    return Oracle::scrambled_radical_inverse(base, perm, a);
}
void oracle_camera_ray(const iile_scene_desc *scene, float pfx, float pfy, float plx, float ply, float *o3, float *d3) {
    Counters c;
    Oracle o(*scene, ORACLE_TRIG_PORTABLE, &c);
    float pl[2] = {plx, ply};
    Ray r = o.camera_ray(pfx, pfy, pl);
    for (int i = 0; i < 3; ++i) {
        o3[i] = r.o[i];
        d3[i] = r.d[i];
    }
}
void oracle_intersect(const iile_scene_desc *scene, int n, const float *o, const float *d, const float *tmax,
                      int32_t *prim, float *tb) {
    Counters c;
    Oracle orc(*scene, ORACLE_TRIG_PORTABLE, &c);
    for (int i = 0; i < n; ++i) {
        Ray r{V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]};
        Isect is;
        bool hit = orc.intersect(r, &is, false);
        prim[i] = hit ? is.prim : -1;
        tb[4 * i] = hit ? is.t : 0;
        tb[4 * i + 1] = hit ? is.b0 : 0;
        tb[4 * i + 2] = hit ? is.b1 : 0;
        tb[4 * i + 3] = hit ? is.b2 : 0;
    }
}
void oracle_intersect_p(const iile_scene_desc *scene, int n, const float *o, const float *d, const float *tmax,
                        int32_t *hit) {
    Counters c;
    Oracle orc(*scene, ORACLE_TRIG_PORTABLE, &c);
    for (int i = 0; i < n; ++i) {
        Ray r{V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]};
        hit[i] = orc.intersect_p(r) ? 1 : 0;
    }
}
void oracle_li(const iile_scene_desc *scene, int trig_mode, int n, const int32_t *px, const int32_t *py,
               const int32_t *k, float *L, int32_t *nrays) {
    Counters c;
    Oracle orc(*scene, trig_mode, &c);
    for (int i = 0; i < n; ++i) {
        uint64_t r0 = c.regular_rays, s0 = c.shadow_rays;
        float pf[2];
        Rgb v = orc.sample_radiance(px[i], py[i], k[i], pf);
        for (int j = 0; j < 3; ++j) L[3 * i + j] = v.c[j];
        if (nrays) {
            nrays[2 * i] = int32_t(c.regular_rays - r0);
            nrays[2 * i + 1] = int32_t(c.shadow_rays - s0);
        }
    }
}
// the material's BSDF in the canonical frame ns = ng = +z, ss = +x (constant parameters: image textures are not looked up)
static Oracle::Bsdf local_bsdf(const Oracle &orc, const iile_scene_desc *scene, int mat) {
    iile_material m = scene->materials[mat];
    m.kd_tex = m.ks_tex = m.kr_tex = m.kt_tex = m.bump_tex = m.rough_tex = m.sigma_tex = m.opacity_tex = -1;
    if (m.rough_tex_v >= 0) m.rough_tex_v = -1;
    Isect is;
    is.sn = is.n = V3(0, 0, 1);
    is.sdpdu = V3(1, 0, 0);
    return orc.make_bsdf_of(m, is);
}
void oracle_bsdf_eval(const iile_scene_desc *scene, int trig_mode, int mat, const float *wo3, const float *wi3,
                      float *f3, float *pdf) {
    Counters c;
    Oracle orc(*scene, trig_mode, &c);
    Oracle::Bsdf b = local_bsdf(orc, scene, mat);
    V3 wo(wo3[0], wo3[1], wo3[2]), wi(wi3[0], wi3[1], wi3[2]);
    Rgb f = Oracle::bsdf_f(b, wo, wi);
    for (int i = 0; i < 3; ++i) f3[i] = f.c[i];
    *pdf = Oracle::bsdf_pdf(b, wo, wi);
}
void oracle_bsdf_sample(const iile_scene_desc *scene, int trig_mode, int mat, const float *wo3, const float *u2,
                        float *wi3, float *f3, float *pdf) {
    Counters c;
    Oracle orc(*scene, trig_mode, &c);
    Oracle::Bsdf b = local_bsdf(orc, scene, mat);
    V3 wo(wo3[0], wo3[1], wo3[2]), wi;
    float p = 0;
    Rgb f = orc.bsdf_sample_f(b, wo, &wi, u2, &p);
    for (int i = 0; i < 3; ++i) {
        f3[i] = f.c[i];
        wi3[i] = wi[i];
    }
    *pdf = p;
}
void oracle_bsdf_sample_batch(const iile_scene_desc *scene, int trig_mode, int mat, const float *wo3, int n,
                              const float *u2n, float *wi3n, float *pdfn) {
    Counters c;
    Oracle orc(*scene, trig_mode, &c);
    Oracle::Bsdf b = local_bsdf(orc, scene, mat);
    V3 wo(wo3[0], wo3[1], wo3[2]);
    for (int i = 0; i < n; ++i) {
        V3 wi(0, 0, 0);
        float p = 0;
        Rgb f = orc.bsdf_sample_f(b, wo, &wi, u2n + 2 * i, &p);
        if (f.is_black()) p = 0;  // the reference's FrequencyTable skips black samples (bsdfs.cpp:78)
        for (int k = 0; k < 3; ++k) wi3n[3 * i + k] = wi[k];
        pdfn[i] = p;
    }
}
void oracle_bsdf_pdf_batch(const iile_scene_desc *scene, int trig_mode, int mat, const float *wo3, int n,
                           const float *wi3n, float *pdfn) {
    Counters c;
    Oracle orc(*scene, trig_mode, &c);
    Oracle::Bsdf b = local_bsdf(orc, scene, mat);
    V3 wo(wo3[0], wo3[1], wo3[2]);
    for (int i = 0; i < n; ++i)
        pdfn[i] = Oracle::bsdf_pdf(b, wo, V3(wi3n[3 * i], wi3n[3 * i + 1], wi3n[3 * i + 2]));
}
void oracle_texture_eval(const iile_scene_desc *scene, int trig_mode, int tex, int n, const float *uv2, const float *duv4,
                         float *rgb3) {
    Counters c;
    Oracle orc(*scene, trig_mode, &c);
    for (int i = 0; i < n; ++i) {
        Isect is;
        is.uv[0] = uv2[2 * i];
        is.uv[1] = uv2[2 * i + 1];
        is.dudx = duv4[4 * i];
        is.dvdx = duv4[4 * i + 1];
        is.dudy = duv4[4 * i + 2];
        is.dvdy = duv4[4 * i + 3];
        Rgb v = orc.tex_evaluate(tex, is);
        for (int k = 0; k < 3; ++k) rgb3[3 * i + k] = v.c[k];
    }
}
int oracle_camera_hit_differentials(const iile_scene_desc *scene, int trig_mode, float pfx, float pfy, float *out6) {
    Counters c;
    Oracle orc(*scene, trig_mode, &c);
    const float plens[2] = {0.5f, 0.5f};
    RayDiff rd;
    Ray ray = orc.camera_ray(pfx, pfy, plens, &rd);
    Isect is;
    if (!orc.intersect(ray, &is)) return 0;
    Oracle::compute_differentials(&is, rd);
    out6[0] = is.uv[0];
    out6[1] = is.uv[1];
    out6[2] = is.dudx;
    out6[3] = is.dvdx;
    out6[4] = is.dudy;
    out6[5] = is.dvdy;
    return 1;
}
// The closest hit's differential geometry as the direct pass reads it: out24 = {p, n, shading n, dpdu, dpdv, shading dndu, dndv, {prim, -, -}}
int oracle_hit_geometry(const iile_scene_desc *scene, int trig_mode, const float *o3, const float *d3, float *out24) {
    Counters c;
    Oracle orc(*scene, trig_mode, &c);
    Ray ray{V3(o3[0], o3[1], o3[2]), V3(d3[0], d3[1], d3[2]), std::numeric_limits<float>::infinity()};
    Isect is;
    if (!orc.intersect(ray, &is)) return 0;
    const V3 v[7] = {is.p, is.n, is.sn, is.dpdu, is.dpdv, is.dndu, is.dndv};
    for (int i = 0; i < 7; ++i) {
        out24[3 * i] = v[i].x;
        out24[3 * i + 1] = v[i].y;
        out24[3 * i + 2] = v[i].z;
    }
    out24[21] = float(is.prim);
    out24[22] = out24[23] = 0;
    return 1;
}
// Distribution1D over func[0 .. n), n <= IILE_MAX_LIGHTS (src/tests/sampling.cpp:231-304): mode 0 SampleDiscrete (the
// light selection of UniformSampleOneLight) -> returns the offset, *pdf = DiscretePDF-style pdf; mode 1
// SampleContinuous (the environment map's rows and columns) -> *value, *pdf, returns the offset
int oracle_distribution1d(const float *func, int n, int mode, float u, float *value, float *pdf) {
    if (n < 1 || n > IILE_MAX_LIGHTS) return -1;
    Oracle::LightDist d;
    d.n = n;
    for (int i = 0; i < n; ++i) d.func[i] = func[i];
    Oracle::finish_distribution(&d);
    if (mode == 0) return Oracle::sample_discrete(d, u, pdf);
    std::vector<float> flat(size_t(2 * n + 2));
    for (int i = 0; i < n; ++i) flat[size_t(i)] = d.func[i];
    for (int i = 0; i <= n; ++i) flat[size_t(n + i)] = d.cdf[i];
    flat[size_t(2 * n + 1)] = d.func_int;
    int off = 0;
    *value = Oracle::dist1d_sample(flat.data(), n, u, pdf, &off);
    return off;
}
float oracle_log(int trig_mode, float x) {
    Trig t{trig_mode};
    return t.log_f(x);
}
void oracle_sincos(int trig_mode, float x, float *s, float *c) {
    Trig t{trig_mode};
    *s = t.sin_f(x);
    *c = t.cos_f(x);
}
void oracle_sincos_d(int trig_mode, double x, double *s, double *c) {
    Trig t{trig_mode};
    *s = t.sin_d(x);
    *c = t.cos_d(x);
}
float oracle_atan2(int trig_mode, float y, float x) {
    Trig t{trig_mode};
    return t.atan2_f(y, x);
}
double oracle_atan2_d(double y, double x) { return portable_atan2(y, x); }
float oracle_acos(int trig_mode, float x) {
    Trig t{trig_mode};
    return t.acos_f(x);
}

}  // extern "C"

// ----------------------------------------------------------------------------
// property tests of the reference, see oracle.h
namespace {
struct SplitMix {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    float uniform() { return float(next() >> 40) * 0x1p-24f; }
    uint32_t below(uint32_t n) { return uint32_t(next() % n); }
};
// fp_tests.cpp:97-118 (getFloat): a random float with a wide range of exponents
float random_float(SplitMix &r, float min_exp = -6.f, float max_exp = 6.f) {
    float logu = min_exp + r.uniform() * (max_exp - min_exp);
    float sign = r.uniform() < .5f ? -1.f : 1.f;
    return sign * std::pow(10.f, logu);
}
EFloat random_efloat(SplitMix &r) {  // fp_tests.cpp:105-143
    float val = std::abs(random_float(r));
    float err = 0;
    switch (r.below(4)) {
    case 0: break;
    case 1: err = std::abs(b2f(f2b(val) + r.below(1024)) - val); break;
    case 2: err = std::abs(b2f(f2b(val) + r.below(1024 * 1024)) - val); break;
    default: err = (4 * r.uniform()) * std::abs(val);
    }
    float sign = r.uniform() < .5f ? -1.f : 1.f;
    return EFloat(sign * val, err);
}
double precise_in(const EFloat &e, SplitMix &r) {  // fp_tests.cpp:147-166
    switch (r.below(3)) {
    case 0: return e.low;
    case 1: return e.high;
    default: {
        float t = r.uniform();
        double p = (1 - t) * double(e.low) + t * double(e.high);
        return std::min(std::max(p, double(e.low)), double(e.high));
    }
    }
}
}  // namespace

void oracle_sphere_solid_angle(const iile_scene_desc *scene, int sphere, const float *p3, int n_samples,
                               double *by_sampling, double *by_uniform_directions) {
    Counters c;
    Oracle orc(*scene, ORACLE_TRIG_LIBM, &c);
    const iile_sphere &sp = scene->spheres[sphere];
    const V3 p(p3[0], p3[1], p3[2]);
    // the prim that is this sphere
    int prim = -1;
    for (int i = 0; i < scene->n_prims; ++i)
        if ((scene->prim_flags[i] & IILE_PRIM_SPHERE) && scene->prim_shape[i] == sphere) prim = i;
    Isect ref;  // Interaction(p, Normal3f(), Vector3f(), ...): no normal, no error bounds
    ref.p = p;
    ref.perr = V3(0, 0, 0);
    ref.n = V3(0, 0, 0);
    double sa = 0;
    int hits = 0;
    for (int i = 0; i < n_samples; ++i) {
        float u[2] = {oracle_radical_inverse(0, uint64_t(i)), oracle_radical_inverse(1, uint64_t(i))};
        float pdf = 0;
        Oracle::LightSample ps = orc.sphere_sample(sp, ref, u, &pdf);
        Ray r{p, ps.p - p, .999f};
        if (pdf > 0 && !orc.prim_intersects(r, prim)) sa += 1 / pdf;
        // UniformSampleSphere, sampling.cpp:98-103
        float z = 1 - 2 * u[0], rr = std::sqrt(std::max(0.f, 1.f - z * z)), phi = 2 * Pi * u[1];
        Ray w{p, V3(rr * std::cos(phi), rr * std::sin(phi), z), Infinity};
        if (orc.prim_intersects(w, prim)) ++hits;
    }
    *by_sampling = sa / n_samples;
    *by_uniform_directions = hits / ((1.0 / (4 * 3.14159265358979323846)) * n_samples);
}

void oracle_light_solid_angle(const iile_scene_desc *scene, int light, const float *p3, int n_samples,
                              double *by_sampling, double *by_uniform_directions) {
    Counters c;
    Oracle orc(*scene, ORACLE_TRIG_LIBM, &c);
    const iile_light &lt = scene->lights[light];
    const V3 p(p3[0], p3[1], p3[2]);
    int prim = lt.prim;
    if (lt.type == IILE_LIGHT_DIFFUSE_AREA)
        for (int i = 0; i < scene->n_prims; ++i)
            if ((scene->prim_flags[i] & IILE_PRIM_SPHERE) && scene->prim_shape[i] == lt.sphere) prim = i;
    Isect ref;  // Interaction(pc, Normal3f(), Vector3f(), ...)
    ref.p = p;
    ref.perr = V3(0, 0, 0);
    ref.n = V3(0, 0, 0);
    double sa = 0;
    int hits = 0;
    for (int i = 0; i < n_samples; ++i) {
        float u[2] = {oracle_radical_inverse(0, uint64_t(i)), oracle_radical_inverse(1, uint64_t(i))};
        float pdf = 0;
        (void)orc.shape_sample(lt, ref, u, &pdf);
        if (pdf > 0) sa += 1. / (double(n_samples) * pdf);
        float z = 1 - 2 * u[0], rr = std::sqrt(std::max(0.f, 1.f - z * z)), phi = 2 * Pi * u[1];
        Ray w{p, V3(rr * std::cos(phi), rr * std::sin(phi), z), Infinity};
        if (orc.prim_intersects(w, prim)) ++hits;
    }
    *by_sampling = sa;
    *by_uniform_directions = hits / (double(n_samples) * (1.0 / (4 * 3.14159265358979323846)));
}

int64_t oracle_check_next_float(int iters, uint64_t seed) {
    int64_t bad = 0;
    if (!(next_up(-0.f) > 0.f) || !(next_down(0.f) < 0.f)) ++bad;
    if (next_up(Infinity) != Infinity || !(next_down(Infinity) < Infinity)) ++bad;
    if (next_down(-Infinity) != -Infinity || !(next_up(-Infinity) > -Infinity)) ++bad;
    SplitMix r{seed};
    for (int i = 0; i < iters; ++i) {
        float f = b2f(uint32_t(r.next()));  // any bit pattern
        if (std::isinf(f) || std::isnan(f)) continue;
        if (std::nextafter(f, Infinity) != next_up(f)) ++bad;
        if (std::nextafter(f, -Infinity) != next_down(f)) ++bad;
    }
    return bad;
}

int64_t oracle_check_efloat(int iters, uint64_t seed) {
    int64_t bad = 0;
    for (int i = 0; i < iters; ++i) {
        SplitMix r{seed + uint64_t(i)};
        EFloat e[2] = {random_efloat(r), random_efloat(r)};
        double p[2] = {precise_in(e[0], r), precise_in(e[1], r)};
        auto check = [&](const EFloat &res, float precise) {  // the reference compares in float
            if (!(precise >= res.low && precise <= res.high)) ++bad;
        };
        check(e[0] + e[1], float(p[0] + p[1]));
        check(e[0] - e[1], float(p[0] - p[1]));
        check(e[0] * e[1], float(p[0] * p[1]));
        const float abs_err = (e[1].high - e[1].low) / 2;  // GetAbsoluteError, efloat.h:112
        if (!(double(e[1].low) * double(e[1].high) < 0. || abs_err > .25 * std::abs(e[1].low)))
            check(e[0] / e[1], float(p[0] / p[1]));
    }
    return bad;
}

int64_t oracle_check_reintersect(const iile_scene_desc *scene, int n, const float *o, const float *d, int n_out,
                                 uint64_t seed, int64_t *stats) {
    Counters c;
    Oracle orc(*scene, ORACLE_TRIG_LIBM, &c);
    SplitMix r{seed};
    int64_t bad = 0, hits = 0, tested = 0;
    for (int i = 0; i < n; ++i) {
        Ray ray{V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), Infinity};
        Isect is;
        if (!orc.intersect(ray, &is, true)) continue;
        ++hits;
        const bool convex_only = (scene->prim_flags[is.prim] & IILE_PRIM_SPHERE) != 0;
        for (int j = 0; j < n_out; ++j) {
            // UniformSampleSphere, sampling.cpp:98-103
            float u0 = r.uniform(), u1 = r.uniform();
            float z = 1 - 2 * u0, rr = std::sqrt(std::max(0.f, 1.f - z * z)), phi = 2 * Pi * u1;
            V3 w(rr * std::cos(phi), rr * std::sin(phi), z);
            // a sphere is only convex: stay on the side of the surface normal (shapes.cpp:402-404)
            if (convex_only && dot(w, is.n) < 0) w = -w;
            Ray out = Oracle::spawn_ray(is, w);
            ++tested;
            if (orc.prim_intersects(out, is.prim)) ++bad;
            // SpawnRayTo a random point (shapes.cpp:196-205, 411-423)
            V3 p2(random_float(r, -3.f, 3.f), random_float(r, -3.f, 3.f), random_float(r, -3.f, 3.f));
            if (convex_only) {
                V3 w2 = p2 - is.p;
                if (dot(w2, is.n) < 0) w2 = -w2;
                p2 = is.p + w2;
            }
            V3 origin = Oracle::offset_ray_origin(is.p, is.perr, is.n, p2 - is.p);
            Ray to{origin, p2 - is.p, 1 - ShadowEpsilon};  // Interaction::SpawnRayTo(Point3f), interaction.h:68-72
            ++tested;
            if (orc.prim_intersects(to, is.prim)) ++bad;
        }
    }
    if (stats) {
        stats[0] = hits;
        stats[1] = tested;
    }
    return bad;
}

extern "C" {

// The IISPT direct pass into a film monitor (IisptFilmMonitor::add_n_samples, iisptfilmmonitor.cpp:47-72: doubles): n_passes
// passes of DirectProgressiveIntegrator::RenderOnePass, pass p seeded as described at DirectSampler, added in pass order into
// film_rgbw[(y * w + x) * 4] = {sum r, sum g, sum b, sum of ray weights} over the film's cropped pixel bounds (zeroed first).
// 0 = ok, 1 = bad arguments.
// Glass: DirectProgressiveIntegrator::Li calls ComputeScatteringFunctions with allowMultipleLobes = false (interaction.h:130-133),
// so GlassMaterial adds SpecularReflection + SpecularTransmission (glass.cpp:62-90) and Li recurses through both: a tree, walked here
// by the recursion itself (round 3 assumed FresnelSpecular and rendered glass black; the device pass refuses glass).
int oracle_iispt_direct(const iile_scene_desc *scene, int trig_mode, int n_passes, int first_pass, int n_threads, double *film_rgbw) {
    if (!scene || !film_rgbw || n_passes < 0) return 1;
    const iile_scene_desc &S = *scene;
    const iile_film_desc &F = S.film;
    const int fw = F.crop_x1 - F.crop_x0, fh = F.crop_y1 - F.crop_y0;
    std::memset(film_rgbw, 0, sizeof(double) * 4 * size_t(fw) * fh);
    if (n_threads <= 0) n_threads = int(std::thread::hardware_concurrency());
    if (n_threads <= 0) n_threads = 1;
    const int sw = F.samp_x1 - F.samp_x0;
    std::vector<Counters> counters(static_cast<size_t>(n_threads));
    for (int p = 0; p < n_passes; ++p) {
        std::atomic<int> next_row(F.samp_y0);
        auto worker = [&](int tid) {
            Oracle orc(S, trig_mode, &counters[size_t(tid)]);
            for (;;) {
                const int y = next_row.fetch_add(1);
                if (y >= F.samp_y1) break;
                for (int x = F.samp_x0; x < F.samp_x1; ++x) {
                    // (StartPixel is called for every pixel of the sample bounds; one outside the pixel bounds is skipped)
                    if (x < F.crop_x0 || x >= F.crop_x1 || y < F.crop_y0 || y >= F.crop_y1) continue;
                    const Rgb L = orc.direct_pixel(x, y, first_pass + p, (y - F.samp_y0) * sw + (x - F.samp_x0));
                    double *out = film_rgbw + 4 * (size_t(y - F.crop_y0) * fw + (x - F.crop_x0));
                    out[0] += double(L.c[0]);  // pix.r += rgb[0] (float to double)
                    out[1] += double(L.c[1]);
                    out[2] += double(L.c[2]);
                    out[3] += 1.0;             // rayWeight of the perspective camera
                }
            }
        };
        std::vector<std::thread> th;
        for (int i = 1; i < n_threads; ++i) th.emplace_back(worker, i);
        worker(0);
        for (auto &t : th) t.join();
    }
    return 0;
}

// IisptFilmMonitor::merge_into (iisptfilmmonitor.cpp:231-275) followed by to_intensity_film (:158-196): both pixels
// normalised (sums over weight where the weight is positive), added, and the sum — weight 1 — converted to float.
void oracle_iispt_merge(int64_t n_pixels, const double *direct_rgbw, const double *indirect_rgbw, float *out_rgb) {
    for (int64_t i = 0; i < n_pixels; ++i) {
        double a[3] = {direct_rgbw[4 * i], direct_rgbw[4 * i + 1], direct_rgbw[4 * i + 2]};
        double b[3] = {indirect_rgbw[4 * i], indirect_rgbw[4 * i + 1], indirect_rgbw[4 * i + 2]};
        if (direct_rgbw[4 * i + 3] > 0.0)
            for (double &v : a) v /= direct_rgbw[4 * i + 3];
        if (indirect_rgbw[4 * i + 3] > 0.0)
            for (double &v : b) v /= indirect_rgbw[4 * i + 3];
        for (int c = 0; c < 3; ++c) out_rgb[3 * i + c] = float((a[c] + b[c]) / 1.0);
    }
}

}  // extern "C"

