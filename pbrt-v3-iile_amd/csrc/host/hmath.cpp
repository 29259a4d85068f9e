// hmath.cpp — matrix inverse and the transform constructors the killeroo-class
// scenes use. Follows /root/reference/src/core/transform.cpp (lines cited).
#include "hmath.h"

namespace iile {

// core/transform.cpp:82-141 — Gauss-Jordan elimination with full pivoting.
bool invert(const Mat4 &in, Mat4 *out) {
    int indxc[4], indxr[4];
    int ipiv[4] = {0, 0, 0, 0};
    float a[4][4];
    std::memcpy(a, in.m, sizeof(a));
    for (int i = 0; i < 4; i++) {
        int irow = 0, icol = 0;
        float big = 0.f;
        for (int j = 0; j < 4; j++) {
            if (ipiv[j] == 1) continue;
            for (int k = 0; k < 4; k++) {
                if (ipiv[k] == 0) {
                    if (std::abs(a[j][k]) >= big) {
                        big = std::abs(a[j][k]);
                        irow = j;
                        icol = k;
                    }
                } else if (ipiv[k] > 1)
                    return false;
            }
        }
        ++ipiv[icol];
        if (irow != icol)
            for (int k = 0; k < 4; ++k) std::swap(a[irow][k], a[icol][k]);
        indxr[i] = irow;
        indxc[i] = icol;
        if (a[icol][icol] == 0.f) return false;
        // `Float pivinv = 1. / minv[icol][icol];` — double divide, then rounded
        float pivinv = float(1. / double(a[icol][icol]));
        a[icol][icol] = 1.f;
        for (int j = 0; j < 4; j++) a[icol][j] *= pivinv;
        for (int j = 0; j < 4; j++) {
            if (j == icol) continue;
            float save = a[j][icol];
            a[j][icol] = 0;
            for (int k = 0; k < 4; k++) a[j][k] -= a[icol][k] * save;
        }
    }
    for (int j = 3; j >= 0; j--) {
        if (indxr[j] != indxc[j])
            for (int k = 0; k < 4; k++) std::swap(a[k][indxr[j]], a[k][indxc[j]]);
    }
    std::memcpy(out->m, a, sizeof(a));
    return true;
}

// core/transform.cpp:146-152
Xform xf_translate(V3 d) {
    Mat4 m(1, 0, 0, d.x, 0, 1, 0, d.y, 0, 0, 1, d.z, 0, 0, 0, 1);
    Mat4 mi(1, 0, 0, -d.x, 0, 1, 0, -d.y, 0, 0, 1, -d.z, 0, 0, 0, 1);
    return Xform(m, mi);
}

// core/transform.cpp:154-158
Xform xf_scale(float x, float y, float z) {
    Mat4 m(x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1);
    Mat4 mi(1 / x, 0, 0, 0, 0, 1 / y, 0, 0, 0, 0, 1 / z, 0, 0, 0, 0, 1);
    return Xform(m, mi);
}

// core/transform.cpp:184-205 — rotation about an arbitrary axis; sin/cos are
// the float overloads of the host libm, as in the reference.
Xform xf_rotate(float theta, V3 axis) {
    V3 a = normalize(axis);
    float s = std::sin(radians(theta));
    float c = std::cos(radians(theta));
    Mat4 m;
    m.m[0][0] = a.x * a.x + (1 - a.x * a.x) * c;
    m.m[0][1] = a.x * a.y * (1 - c) - a.z * s;
    m.m[0][2] = a.x * a.z * (1 - c) + a.y * s;
    m.m[0][3] = 0;
    m.m[1][0] = a.x * a.y * (1 - c) + a.z * s;
    m.m[1][1] = a.y * a.y + (1 - a.y * a.y) * c;
    m.m[1][2] = a.y * a.z * (1 - c) - a.x * s;
    m.m[1][3] = 0;
    m.m[2][0] = a.x * a.z * (1 - c) - a.y * s;
    m.m[2][1] = a.y * a.z * (1 - c) + a.x * s;
    m.m[2][2] = a.z * a.z + (1 - a.z * a.z) * c;
    m.m[2][3] = 0;
    return Xform(m, transpose(m));
}

// core/transform.cpp:207-242
bool xf_lookat(V3 pos, V3 look, V3 up, Xform *out) {
    Mat4 c2w;
    c2w.m[0][3] = pos.x;
    c2w.m[1][3] = pos.y;
    c2w.m[2][3] = pos.z;
    c2w.m[3][3] = 1;
    V3 dir = normalize(look - pos);
    if (length(cross(normalize(up), dir)) == 0) {
        *out = Xform();
        return false;
    }
    V3 right = normalize(cross(normalize(up), dir));
    V3 new_up = cross(dir, right);
    c2w.m[0][0] = right.x;
    c2w.m[1][0] = right.y;
    c2w.m[2][0] = right.z;
    c2w.m[3][0] = 0.;
    c2w.m[0][1] = new_up.x;
    c2w.m[1][1] = new_up.y;
    c2w.m[2][1] = new_up.z;
    c2w.m[3][1] = 0.;
    c2w.m[0][2] = dir.x;
    c2w.m[1][2] = dir.y;
    c2w.m[2][2] = dir.z;
    c2w.m[3][2] = 0.;
    Mat4 w2c;
    invert(c2w, &w2c);
    *out = Xform(w2c, c2w);
    return true;
}

// core/transform.cpp:303-311
Xform xf_perspective(float fov, float n, float f) {
    Mat4 persp(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, f / (f - n), -f * n / (f - n), 0, 0, 1, 0);
    float inv_tan = 1 / std::tan(radians(fov) / 2);
    return xf_scale(inv_tan, inv_tan, 1) * Xform(persp);
}

}  // namespace iile
