// dmath.h — device-side scalar/vector math for the gfx950 path-tracing kernels.
//
// Everything here is compiled with -ffp-contract=off and without fast-math:
// each function evaluates the float expression of the reference line it cites
// (relative to /root/reference/src) with IEEE add/mul/div/sqrt, so that a ray
// takes the same branches on the GPU as on the CPU. min/max keep the
// std::min/std::max NaN behaviour (comparison + select), not fminf/fmaxf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEV __device__ __forceinline__

namespace iile {

// core/pbrt.h:196-208, core/rng.h:53
constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kInv2Pi = 0.15915494309189533577f;
constexpr float kPiOver2 = 1.57079632679489661923f;
constexpr float kPiOver4 = 0.78539816339744830961f;
constexpr float kMachineEpsilon = 5.9604644775390625e-08f;  // 2^-24
constexpr float kShadowEpsilon = 0.0001f;
constexpr float kOneMinusEpsilon = 0x1.fffffep-1f;
#define IILE_INF __builtin_huge_valf()
// gamma(n), core/pbrt.h:286-288 — folded at compile time in float
constexpr float gamma_c(int n) { return (n * kMachineEpsilon) / (1 - n * kMachineEpsilon); }
constexpr float kGamma2 = gamma_c(2), kGamma3 = gamma_c(3), kGamma5 = gamma_c(5), kGamma6 = gamma_c(6),
                kGamma7 = gamma_c(7);
constexpr float kSlabScale = 1 + 2 * gamma_c(3);  // geometry.h:1422

DEV float mn(float a, float b) { return b < a ? b : a; }  // std::min
DEV float mx(float a, float b) { return a < b ? b : a; }  // std::max
DEV float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
DEV uint32_t f2b(float f) { return __float_as_uint(f); }
// Keeps a loaded float4 whole: with its four lanes consumed at one point the compiler issues ONE 16-byte load instead of
// narrowing it into a 12-byte load where xyz is used and a 4-byte load where w is (every vector memory instruction of a
// wavefront costs address-unit cycles whatever its width: DESIGN.md §6 "What binds")
DEV void keep_whole(float4 &a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }
DEV void keep_whole(float4 &a, float4 &b, float4 &c) {
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w), "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w));
}
DEV float b2f(uint32_t u) { return __uint_as_float(u); }
DEV bool is_inf(float f) { return (f2b(f) & 0x7fffffffu) == 0x7f800000u; }
DEV bool is_nan(float f) { return (f2b(f) & 0x7fffffffu) > 0x7f800000u; }

// core/pbrt.h:238-262
DEV float next_up(float v) {
    if (is_inf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = f2b(v);
    if (v >= 0)
        ++ui;
    else
        --ui;
    return b2f(ui);
}
DEV float next_down(float v) {
    if (is_inf(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = f2b(v);
    if (v > 0)
        --ui;
    else
        ++ui;
    return b2f(ui);
}

struct F3 {
    float x, y, z;
};
DEV F3 mk3(float x, float y, float z) { return F3{x, y, z}; }
DEV F3 operator+(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV F3 operator-(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV F3 operator-(F3 a) { return F3{-a.x, -a.y, -a.z}; }
DEV F3 operator*(float s, F3 a) { return F3{s * a.x, s * a.y, s * a.z}; }
DEV F3 operator*(F3 a, float s) { return F3{s * a.x, s * a.y, s * a.z}; }
DEV F3 operator*(F3 a, F3 b) { return F3{a.x * b.x, a.y * b.y, a.z * b.z}; }  // spectrum product
DEV float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV float absdot(F3 a, F3 b) { return fabsf(dot(a, b)); }
DEV float length_sq(F3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
DEV float length(F3 a) { return sqrtf(length_sq(a)); }
// geometry.h:242-246 — vector / scalar multiplies by a float reciprocal
DEV F3 vdiv(F3 a, float f) {
    float inv = 1.f / f;
    return F3{a.x * inv, a.y * inv, a.z * inv};
}
DEV F3 normalize(F3 a) { return vdiv(a, length(a)); }
DEV F3 vabs(F3 a) { return F3{fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
// geometry.h:957-963 — cross product in double, rounded once per component
DEV F3 cross(F3 a, F3 b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return F3{float((ay * bz) - (az * by)), float((az * bx) - (ax * bz)), float((ax * by) - (ay * bx))};
}
DEV F3 faceforward(F3 n, F3 v) { return (dot(n, v) < 0.f) ? -n : n; }
DEV float max3(float a, float b, float c) { return mx(a, mx(b, c)); }
// geometry.h:1020-1027
DEV void coordinate_system(F3 v1, F3 *v2, F3 *v3) {
    if (fabsf(v1.x) > fabsf(v1.y))
        *v2 = vdiv(F3{-v1.z, 0, v1.x}, sqrtf(v1.x * v1.x + v1.z * v1.z));
    else
        *v2 = vdiv(F3{0, v1.z, -v1.y}, sqrtf(v1.y * v1.y + v1.z * v1.z));
    *v3 = cross(v1, *v2);
}
// spectrum helpers (core/spectrum.h: RGBSpectrum)
DEV bool is_black(F3 s) { return s.x == 0.f && s.y == 0.f && s.z == 0.f; }
DEV float lum_y(F3 s) { return 0.212671f * s.x + 0.715160f * s.y + 0.072169f * s.z; }
DEV F3 sdiv(F3 s, float a) { return F3{s.x / a, s.y / a, s.z / a}; }  // true division per channel

// ---------------------------------------------------------------------------
// Portable trigonometry. The CPU path calls glibc's sinf/cosf/acosf; those are
// not reproducible instruction-for-instruction on a GPU, so the device uses a
// fixed double-precision evaluation (Cody-Waite reduction by pi/2 + fdlibm
// minimax kernels), rounded once to float. The oracle's ORACLE_TRIG_PORTABLE
// mode evaluates the identical operation sequence; MI355X runs FP64 at full
// vector rate, so this costs a few dozen DP ops per call.
DEV void sincos_d(double x, double *s, double *c) {
    const double k = rint(x * 6.36619772367581382433e-01);
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
    const int q = int(k) & 3;
    *s = (q == 0) ? ps : (q == 1) ? pc : (q == 2) ? -ps : -pc;
    *c = (q == 0) ? pc : (q == 1) ? -ps : (q == 2) ? -pc : ps;
}
DEV void sincos_f(float x, float *s, float *c) {
    double sd, cd;
    sincos_d(double(x), &sd, &cd);
    *s = float(sd);
    *c = float(cd);
}
DEV double acos_d(double x) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                 pi = 3.14159265358979311600e+00;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const double ax = fabs(x);
    if (ax >= 1.0) {
        if (x == 1.0) return 0.0;
        if (x == -1.0) return pi + 2.0 * pio2_lo;
        return __builtin_nan("");
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
        const double s = sqrt(z);
        const double r = p / q;
        const double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    } else {
        const double z = (1.0 - x) * 0.5;
        const double s = sqrt(z);
        const double df = __longlong_as_double(__double_as_longlong(s) & 0xffffffff00000000LL);
        const double c = (z - df * df) / (s + df);
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        const double w = r * s + c;
        return 2.0 * (df + w);
    }
}
DEV float acos_f(float x) { return float(acos_d(double(x))); }
// atan / atan2 with the structure and coefficients of fdlibm's s_atan.c / e_atan2.c, operation for
// operation as oracle/oracle_path.cpp portable_atan / portable_atan2 (finite arguments only: the callers
// pass components of normalised directions)
DEV double atan_d(double x) {
    const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01,
                              1.57079632679489655800e+00};
    const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17,
                              6.12323399573676603587e-17};
    const bool neg = x < 0 || (x == 0 && __double_as_longlong(x) < 0);
    double ax = fabs(x);
    int id;
    if (!(ax < 7.378697629483821e19)) {
        if (x != x) return x + x;
        return neg ? -(atanhi[3] + atanlo[3]) : (atanhi[3] + atanlo[3]);
    }
    double hi = 0, lo = 0;
    if (ax < 0.4375) {
        if (ax < 1.862645149230957e-09) return x;
        id = -1;
        ax = x;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) {
            id = 0;
            hi = atanhi[0];
            lo = atanlo[0];
            ax = (2.0 * ax - 1.0) / (2.0 + ax);
        } else {
            id = 1;
            hi = atanhi[1];
            lo = atanlo[1];
            ax = (ax - 1.0) / (ax + 1.0);
        }
    } else if (ax < 2.4375) {
        id = 2;
        hi = atanhi[2];
        lo = atanlo[2];
        ax = (ax - 1.5) / (1.0 + 1.5 * ax);
    } else {
        id = 3;
        hi = atanhi[3];
        lo = atanlo[3];
        ax = -1.0 / ax;
    }
    const double z = ax * ax, w = z * z;
    const double s1 =
        z * (3.33333333333329318027e-01 +
             w * (1.42857142725034663711e-01 +
                  w * (9.09088713343650656196e-02 +
                       w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 +
                           w * (-1.11111104054623557880e-01 +
                                w * (-7.69187620504482999495e-02 +
                                     w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    if (id < 0) return ax - ax * (s1 + s2);
    const double r = hi - ((ax * (s1 + s2) - lo) - ax);
    return neg ? -r : r;
}
DEV double atan2_d(double y, double x) {
    const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16, pi_o_2 = 1.5707963267948965580E+00,
                 tiny = 1.0e-300;
    if (x != x || y != y) return x + y;
    if (x == 1.0) return atan_d(y);
    const bool sy = __double_as_longlong(y) < 0, sx = __double_as_longlong(x) < 0;
    const int m = (sy ? 1 : 0) | (sx ? 2 : 0);
    if (y == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (x == 0) return sy ? -pi_o_2 - tiny : pi_o_2 + tiny;
    // exponents as frexp reports them (the arguments are finite and non-zero here)
    int ey, ex;
    (void)frexp(y, &ey);
    (void)frexp(x, &ex);
    const int k = ey - ex;
    double z;
    if (k > 60)
        z = pi_o_2 + 0.5 * pi_lo;
    else if (sx && k < -60)
        z = 0.0;
    else
        z = atan_d(fabs(y / x));
    switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}
DEV float atan2_f(float y, float x) { return float(atan2_d(double(y), double(x))); }
// Natural logarithm, operation for operation as oracle/oracle_path.cpp portable_log (argument reduction to
// sqrt(1/2) < m <= sqrt(2), atanh series with the classic minimax coefficients). Callers pass positive floats.
DEV double log_d(double x) {
    if (x != x || x < 0) return __builtin_nan("");
    if (x == 0) return -__builtin_huge_val();
    if (x == __builtin_huge_val()) return x;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, Lg1 = 6.666666666666735130e-01,
                 Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    int k = int((bits >> 52) & 0x7ffull) - 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = __longlong_as_double((long long)bits);
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
DEV float log_f(float x) { return float(log_d(double(x))); }
DEV float log2_f(float x) { return log_f(x) * 1.442695040888963387004650940071f; }  // pbrt.h:325-328

// ---------------------------------------------------------------------------
// 4x4 row-major transforms (core/transform.h:217-410)
struct M44 {
    float m[16];
};
DEV F3 xf_point(const M44 &t, F3 p) {
    const float *m = t.m;
    float x = p.x, y = p.y, z = p.z;
    float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    if (wp == 1) return F3{xp, yp, zp};
    return vdiv(F3{xp, yp, zp}, wp);
}
DEV F3 xf_point_err(const M44 &t, F3 p, F3 *err) {  // transform.h:278-300
    const float *m = t.m;
    float x = p.x, y = p.y, z = p.z;
    float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    float xs = (fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3]));
    float ys = (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7]));
    float zs = (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11]));
    *err = kGamma3 * F3{xs, ys, zs};
    if (wp == 1) return F3{xp, yp, zp};
    return vdiv(F3{xp, yp, zp}, wp);
}
DEV F3 xf_point_err2(const M44 &t, F3 pt, F3 pe, F3 *err) {  // transform.h:302-331
    const float *m = t.m;
    float x = pt.x, y = pt.y, z = pt.z;
    float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    err->x = (kGamma3 + 1.f) * (fabsf(m[0]) * pe.x + fabsf(m[1]) * pe.y + fabsf(m[2]) * pe.z) +
             kGamma3 * (fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3]));
    err->y = (kGamma3 + 1.f) * (fabsf(m[4]) * pe.x + fabsf(m[5]) * pe.y + fabsf(m[6]) * pe.z) +
             kGamma3 * (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7]));
    err->z = (kGamma3 + 1.f) * (fabsf(m[8]) * pe.x + fabsf(m[9]) * pe.y + fabsf(m[10]) * pe.z) +
             kGamma3 * (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11]));
    if (wp == 1.f) return F3{xp, yp, zp};
    return vdiv(F3{xp, yp, zp}, wp);
}
DEV F3 xf_vector(const M44 &t, F3 v) {
    const float *m = t.m;
    float x = v.x, y = v.y, z = v.z;
    return F3{m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z, m[8] * x + m[9] * y + m[10] * z};
}
DEV F3 xf_vector_err(const M44 &t, F3 v, F3 *err) {  // transform.h:333-349
    const float *m = t.m;
    float x = v.x, y = v.y, z = v.z;
    err->x = kGamma3 * (fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z));
    err->y = kGamma3 * (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z));
    err->z = kGamma3 * (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z));
    return F3{m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z, m[8] * x + m[9] * y + m[10] * z};
}
// normals transform by the transpose of the inverse, transform.h:243-249
DEV F3 xf_normal(const M44 &tinv, F3 n) {
    const float *m = tinv.m;
    float x = n.x, y = n.y, z = n.z;
    return F3{m[0] * x + m[4] * y + m[8] * z, m[1] * x + m[5] * y + m[9] * z, m[2] * x + m[6] * y + m[10] * z};
}

// geometry.h:1440-1460
DEV F3 offset_ray_origin(F3 p, F3 perr, F3 n, F3 w) {
    float d = dot(vabs(n), perr);
    F3 off = d * n;
    if (dot(w, n) < 0) off = -off;
    F3 po = p + off;
    po.x = off.x > 0 ? next_up(po.x) : (off.x < 0 ? next_down(po.x) : po.x);
    po.y = off.y > 0 ? next_up(po.y) : (off.y < 0 ? next_down(po.y) : po.y);
    po.z = off.z > 0 ? next_up(po.z) : (off.z < 0 ? next_down(po.z) : po.z);
    return po;
}

// ---------------------------------------------------------------------------
// EFloat interval arithmetic (core/efloat.h, NDEBUG flavour)
struct EF {
    float v, lo, hi;
};
DEV EF ef(float v) { return EF{v, v, v}; }
DEV EF ef(float v, float err) {
    if (err == 0.f) return EF{v, v, v};
    return EF{v, next_down(v - err), next_up(v + err)};
}
DEV EF operator+(EF a, EF b) { return EF{a.v + b.v, next_down(a.lo + b.lo), next_up(a.hi + b.hi)}; }
DEV EF operator-(EF a, EF b) { return EF{a.v - b.v, next_down(a.lo - b.hi), next_up(a.hi - b.lo)}; }
DEV EF operator*(EF a, EF b) {
    float p0 = a.lo * b.lo, p1 = a.hi * b.lo, p2 = a.lo * b.hi, p3 = a.hi * b.hi;
    return EF{a.v * b.v, next_down(mn(mn(p0, p1), mn(p2, p3))), next_up(mx(mx(p0, p1), mx(p2, p3)))};
}
DEV EF operator/(EF a, EF b) {
    EF r;
    r.v = a.v / b.v;
    if (b.lo < 0 && b.hi > 0) {
        r.lo = -IILE_INF;
        r.hi = IILE_INF;
    } else {
        float d0 = a.lo / b.lo, d1 = a.hi / b.lo, d2 = a.lo / b.hi, d3 = a.hi / b.hi;
        r.lo = next_down(mn(mn(d0, d1), mn(d2, d3)));
        r.hi = next_up(mx(mx(d0, d1), mx(d2, d3)));
    }
    return r;
}
// efloat.h:267-285 — discriminant in double
DEV bool ef_quadratic(EF A, EF B, EF C, EF *t0, EF *t1) {
    double discrim = (double)B.v * (double)B.v - 4. * (double)A.v * (double)C.v;
    if (discrim < 0.) return false;
    double root = sqrt(discrim);
    EF froot = ef(float(root), float(double(kMachineEpsilon) * root));
    EF q = (B.v < 0) ? ef(-.5f) * (B - froot) : ef(-.5f) * (B + froot);
    *t0 = q / A;
    *t1 = C / q;
    if (t0->v > t1->v) {
        EF tmp = *t0;
        *t0 = *t1;
        *t1 = tmp;
    }
    return true;
}

}  // namespace iile
