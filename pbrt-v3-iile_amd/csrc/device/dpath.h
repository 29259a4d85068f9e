// dpath.h — device functions of the wavefront path tracer: Halton sampling,
// camera rays, ray/primitive tests, BVH traversal with an LDS-resident stack,
// surface interactions, BSDFs and light sampling.
//
// Citations are relative to /root/reference/src. The traversal keeps the
// reference's node layout and near-first order (bvh.cpp:686-692), so equal-t
// ties resolve to the same primitive as on the CPU.
#pragma once
#include "dscene.h"

namespace iile {

// ===========================================================================
// Halton sampler (samplers/halton.cpp:96-127, core/lowdiscrepancy.cpp:389-427)
// ===========================================================================
// exact a / base for any 32-bit a (Granlund-Montgomery round-up form)
DEV uint32_t div_magic(uint32_t a, uint32_t magic, uint32_t shift) {
    uint32_t t = __umulhi(magic, a);
    return (t + ((a - t) >> 1)) >> shift;
}

// Global Halton index of sample k of pixel (px, py): offset(pixel mod 128) +
// k * stride. 32-bit arithmetic is exact here: the render entry point rejects
// (spp + 1) * stride >= 2^32.
DEV uint32_t halton_index(const DScene &S, int px, int py, uint32_t k) {
    int pmx = px - (px / 128) * 128, pmy = py - (py / 128) * 128;  // Mod(), pbrt.h:310-314
    if (pmx < 0) pmx += 128;
    if (pmy < 0) pmy += 128;
    // offsetForCurrentPixel (halton.cpp:96-122), tabulated at scene upload
    return S.pixel_offsets[pmy * 128 + pmx] + k * uint32_t(S.sample_stride);
}

DEV float radical_inverse_base2(uint32_t a) {
    // ReverseBits64(a) * 0x1p-64 with a < 2^32: the reversed bits land in the
    // upper word (lowdiscrepancy.cpp:430-434); double product, one rounding to float.
    unsigned long long rev = (unsigned long long)__brev(a) << 32;
    return float(double(rev) * 0x1p-64);
}
DEV float radical_inverse_base3(uint32_t a0) {
    // RadicalInverseSpecialized<3> (lowdiscrepancy.cpp:389-407); digits peeled in double
    // arithmetic, exact for every u32 (see scrambled_radical_inverse)
    const float inv_base = 1.f / 3.f;
    double a = double(a0), reversed = 0;
    float inv_base_n = 1;
    while (a != 0) {
        const double next = __builtin_trunc((a + 0.5) * (1.0 / 3.0));
        reversed = __builtin_fma(reversed, 3.0, __builtin_fma(-next, 3.0, a));
        inv_base_n *= inv_base;
        a = next;
    }
    return mn(float(reversed) * inv_base_n, kOneMinusEpsilon);
}
// `perms` is the concatenated permutation table, either in HBM (const uint16_t *) or
// staged in LDS by the shade kernel (lds_u16 *): a path needs one table lookup per
// digit per dimension, ~30 dependent lookups per bounce.
typedef __attribute__((address_space(3))) uint16_t lds_u16;
// ScrambledRadicalInverseSpecialized (lowdiscrepancy.cpp:409-424). The digits are peeled in
// double arithmetic: next = trunc((a + 0.5) / base) is exact for every u32 a (the fractional
// part of (a + 0.5) / base stays >= 0.5 / base away from an integer, far more than the 2^-22
// the rounded product can be off by), digit = a - next * base and reversed * base + perm are
// exact integers below 2^53, and the final uint64 -> float conversion of the reference rounds
// the same integer the same way as double -> float here.
template <typename PermPtr>
DEV float scrambled_radical_inverse(const DScene &S, PermPtr perms, int dim, uint32_t a0) {
    const DHaltonDim hd = S.hdims[dim];
    PermPtr perm = perms + hd.perm_offset;
    double a = double(a0), reversed = 0;
    float inv_base_n = 1;
    while (a != 0) {
        const double next = __builtin_trunc(__builtin_fma(a, hd.inv_base_d, 0.5 * hd.inv_base_d));  // (a + 0.5) / base, one rounding
        const uint32_t digit = uint32_t(__builtin_fma(-next, hd.base_d, a));
        reversed = __builtin_fma(reversed, hd.base_d, double(uint32_t(perm[digit])));
        inv_base_n *= hd.inv_base;
        a = next;
    }
    return mn(inv_base_n * (float(reversed) + hd.perm0_term), kOneMinusEpsilon);
}
// N consecutive dimensions of one sample index, `dim0` wave-uniform: the per-dimension
// constants come through scalar loads, and the N digit chains advance together so that their
// LDS lookups overlap (one chain alone waits out one LDS round trip per digit).
typedef const __attribute__((address_space(4))) DHaltonDim *HaltonDimConst;
// An entry of the sphere / light table whose index is the same in every lane (the scene's only sphere, the only light):
// read through the constant address space its fields arrive by scalar loads into SGPRs, once per wavefront, instead of
// one vector-memory instruction per field with 64 identical addresses (k_shade spent ~13 of its ~30 loads per hit so).
// (a copy, field by field out of that address space: a reference would have to pass through a generic pointer, and the
// constant address space is lost on the way whenever the index folds to a constant; loads of unused fields are dropped)
template <typename T>
DEV T uniform_entry(const T *table, int idx) {
    typedef const __attribute__((address_space(4))) T *ConstPtr;
    T out;
    __builtin_memcpy(&out, (ConstPtr)(table) + __builtin_amdgcn_readfirstlane(idx), sizeof(T));
    return out;
}
template <int N, typename PermPtr>
DEV void scrambled_radical_inverse_n(const DScene &S, PermPtr perms, int dim0, uint32_t index, float *out) {
    const HaltonDimConst hd = (HaltonDimConst)(S.hdims) + dim0;
    double a[N], reversed[N];
    float inv_base_n[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        a[i] = double(index);
        reversed[i] = 0;
        inv_base_n[i] = 1;
    }
    // all chains run unpredicated while every one of them still has digits left ...
    for (;;) {
        bool all = true;
#pragma unroll
        for (int i = 0; i < N; ++i) all = all && a[i] != 0;
        if (!all) break;
        uint32_t digit[N], p[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double next = __builtin_trunc(__builtin_fma(a[i], hd[i].inv_base_d, 0.5 * hd[i].inv_base_d));  // (a + 0.5) / base, one rounding
            digit[i] = uint32_t(__builtin_fma(-next, hd[i].base_d, a[i]));
            a[i] = next;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = uint32_t(perms[hd[i].perm_offset + digit[i]]);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            reversed[i] = __builtin_fma(reversed[i], hd[i].base_d, double(p[i]));
            inv_base_n[i] *= hd[i].inv_base;
        }
    }
    // ... then the larger bases' last digit or two, chain by chain
#pragma unroll
    for (int i = 0; i < N; ++i) {
        while (a[i] != 0) {
            const double next = __builtin_trunc(__builtin_fma(a[i], hd[i].inv_base_d, 0.5 * hd[i].inv_base_d));  // (a + 0.5) / base, one rounding
            const uint32_t digit = uint32_t(__builtin_fma(-next, hd[i].base_d, a[i]));
            reversed[i] = __builtin_fma(reversed[i], hd[i].base_d, double(uint32_t(perms[hd[i].perm_offset + digit])));
            inv_base_n[i] *= hd[i].inv_base;
            a[i] = next;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
        out[i] = mn(inv_base_n[i] * (float(reversed[i]) + hd[i].perm0_term), kOneMinusEpsilon);
}
// ===========================================================================
// Sobol' sampler (samplers/sobol.cpp:42-59, core/lowdiscrepancy.h:229-274): the sampler the fork's path integrator
// renders with under IILE_PATH_SAMPLES_OVERRIDE (integrators/path.cpp:202-212). Indices stay below 2^32 (checked on the
// host), so 32 columns per generator matrix are kept.
// ===========================================================================
// XOR of the columns a 32-bit word selects, through its four bytes (byte tables: DScene::sobol_bt)
DEV uint32_t sobol_xor4(const uint32_t *bt, uint32_t w) {
    return bt[w & 255u] ^ bt[256u + ((w >> 8) & 255u)] ^ bt[512u + ((w >> 16) & 255u)] ^ bt[768u + (w >> 24)];
}
DEV uint32_t sobol_index(const DScene &S, int px, int py, uint32_t k) {
    const int m = S.sobol_log2res, m2 = 2 * m;
    // `for (c ...) if ((k >> c) & 1) delta ^= vdc[c]` ("add flipped column m + c + 1"; k < 2^(32 - 2m))
    const uint32_t delta = sobol_xor4(S.sobol_vdc_bt, k);
    const uint32_t b = ((uint32_t(px - S.samp_x0) << m) | uint32_t(py - S.samp_y0)) ^ delta;  // flipped b, < 2^(2m)
    // `for (c ...) if ((b >> c) & 1) index ^= vdc_inv[c]` ("add column 2 m - c")
    return (k << m2) ^ sobol_xor4(S.sobol_vdc_bt + 1024, b);
}
// SobolSampleFloat (scramble 0) followed by SobolSampler::SampleDimension's remapping of the two pixel dimensions
template <typename PermPtr>
DEV float sobol_sample_dimension(const DScene &S, PermPtr mats, uint32_t index, int dim, int px, int py) {
    // SobolSample: XOR of the generator-matrix columns the index's bits select (lowdiscrepancy.h:229-241)
    (void)mats;
    const uint32_t v = sobol_xor4(S.sobol_bt + size_t(dim) * 1024, index);
    float s = mn(float(v) * 0x1p-32f /* 1/2^32 */, kOneMinusEpsilon);
    if (dim == 0 || dim == 1) {  // s * resolution + sampleBounds.pMin[dim], then Clamp(s - currentPixel[dim], 0, OneMinusEpsilon)
        const int pmin = dim == 0 ? S.samp_x0 : S.samp_y0, cur = dim == 0 ? px : py;
        s = s * float(S.sobol_res) + float(pmin);
        s = s - float(cur);
        s = s < 0.f ? 0.f : (s > kOneMinusEpsilon ? kOneMinusEpsilon : s);
    }
    return s;
}

// The scene's sampler: GetIndexForSample / SampleDimension of HaltonSampler or SobolSampler. (px, py) is the sample's
// pixel — only SobolSampler's dimensions 0 and 1 depend on it.
DEV uint32_t sample_index(const DScene &S, int px, int py, uint32_t k) {
    return S.sobol ? sobol_index(S, px, py, k) : halton_index(S, px, py, k);
}
template <typename PermPtr>
DEV float sample_dimension(const DScene &S, PermPtr perms, uint32_t index, int dim, int px = 0, int py = 0) {
    if (S.sobol) return sobol_sample_dimension(S, perms, index, dim, px, py);
    if (S.sample_center && dim < 2) return 0.5f;  // "samplepixelcenter", halton.cpp:119
    if (dim == 0) return radical_inverse_base2(index >> S.base_exp0);
    if (dim == 1) return radical_inverse_base3(index / uint32_t(S.base_scale1));
    return scrambled_radical_inverse(S, perms, dim, index);
}
// the same for a dimension the caller knows to be >= 2 (every dimension after the camera sample's film position): none of the
// code — and none of the hoisted constants — of the two pixel dimensions
template <typename PermPtr>
DEV float sample_dimension_hi(const DScene &S, PermPtr perms, uint32_t index, int dim) {
    if (dim < 2) __builtin_unreachable();
    return sample_dimension(S, perms, index, dim);
}
DEV float sample_dimension(const DScene &S, uint32_t index, int dim, int px = 0, int py = 0) {
    return sample_dimension(S, S.perms, index, dim, px, py);
}
// N consecutive dimensions >= 2 of one sample (the shade kernel's batches)
template <int N, typename PermPtr>
DEV void sample_dimensions_n(const DScene &S, PermPtr perms, int dim0, bool dim_uniform, int dim_lane, uint32_t index, float *out) {
    if (dim_lane < 2) __builtin_unreachable();
    if (S.sobol) {
#pragma unroll
        for (int i = 0; i < N; ++i) out[i] = sobol_sample_dimension(S, perms, index, dim_lane + i, 0, 0);
    } else if (dim_uniform) {
        scrambled_radical_inverse_n<N>(S, perms, dim0, index, out);
    } else {
        for (int i = 0; i < N; ++i) out[i] = sample_dimension_hi(S, perms, index, dim_lane + i);
    }
}

// ===========================================================================
// sampling warps (core/sampling.cpp:113-130, core/sampling.h:159-163)
// ===========================================================================
DEV void concentric_sample_disk(float u0, float u1, float *dx, float *dy) {
    float ox = 2.f * u0 - 1, oy = 2.f * u1 - 1;
    if (ox == 0 && oy == 0) {
        *dx = 0;
        *dy = 0;
        return;
    }
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) {
        r = ox;
        theta = kPiOver4 * (oy / ox);
    } else {
        r = oy;
        theta = kPiOver2 - kPiOver4 * (ox / oy);
    }
    float s, c;
    sincos_f(theta, &s, &c);
    *dx = r * c;
    *dy = r * s;
}
DEV F3 cosine_sample_hemisphere(float u0, float u1) {
    float dx, dy;
    concentric_sample_disk(u0, u1, &dx, &dy);
    float z = sqrtf(mx(0.f, 1 - dx * dx - dy * dy));
    return F3{dx, dy, z};
}

// ===========================================================================
// camera (cameras/perspective.cpp:100-149, core/transform.h:251-264)
// ===========================================================================
// `zero` is 0.f: a caller inside a persistent loop passes one the compiler cannot see through (opaque_zero), so that the
// products of matrix entries (SGPRs) with it are worked out where they are used instead of being hoisted out of the loop
// into VGPRs that then live — or spill — across the whole kernel.
DEV float opaque_zero() {
    float z = 0.f;
    asm volatile("" : "+v"(z));
    return z;
}
DEV void camera_ray(const DScene &S, float pfx, float pfy, float lu0, float lu1, F3 *o_out, F3 *d_out, float *tmax,
                    const float zero = 0.f) {
    F3 pcam = xf_point(S.raster_to_camera, F3{pfx, pfy, zero});
    F3 ro = F3{zero, zero, zero};
    F3 rd = normalize(pcam);
    if (S.lens_radius > 0) {
        float lx, ly;
        concentric_sample_disk(lu0, lu1, &lx, &ly);
        lx = S.lens_radius * lx;
        ly = S.lens_radius * ly;
        float ft = S.focal_distance / rd.z;
        F3 pfocus = ro + rd * ft;
        ro = F3{lx, ly, zero};
        rd = normalize(pfocus - ro);
    }
    F3 oerr;
    F3 o = xf_point_err(S.camera_to_world, ro, &oerr);
    F3 d = xf_vector(S.camera_to_world, rd);
    float len2 = length_sq(d);
    float tm = IILE_INF;
    if (len2 > 0) {
        float dt = dot(vabs(d), oerr) / len2;
        o = o + d * dt;
        tm -= dt;
    }
    *o_out = o;
    *d_out = d;
    *tmax = tm;
}

// HemisphericCamera::GenerateRay (hemispheric.cpp:15-41) + Transform::operator()(Ray) (transform.h:251-264)
DEV void probe_ray(const DScene &S, const DProbeCam &cam, float pfx, float pfy, F3 *o_out, F3 *d_out, float *tmax) {
    const float theta = kPi * pfy / float(S.yres);
    const float phi = kPi * pfx / float(S.xres);
    float st, ct, sp, cp;
    sincos_f(theta, &st, &ct);
    sincos_f(phi, &sp, &cp);
    const F3 dir = F3{st * cp, ct, st * sp};
    F3 oerr;
    F3 o = xf_point_err(cam.c2w, F3{0, 0, 0}, &oerr);
    const F3 d = xf_vector(cam.c2w, dir);
    const float len2 = length_sq(d);
    float tm = IILE_INF;
    if (len2 > 0) {
        const float dt = dot(vabs(d), oerr) / len2;
        o = o + d * dt;
        tm -= dt;
    }
    *o_out = o;
    *d_out = d;
    *tmax = tm;
}

// The auxiliary rays of the camera ray's RayDifferential: GenerateRayDifferential (perspective.cpp:124-148),
// Transform::operator()(RayDifferential) (transform.h:265-274) and the render loop's
// ScaleDifferentials(1 / sqrt(spp)) (geometry.h:908-913). (o, d) is the camera ray as camera_ray returns it.
struct RayDiff {
    F3 rxo, ryo, rxd, ryd;
};
DEV RayDiff camera_differentials(const DScene &S, float pfx, float pfy, float lu0, float lu1, F3 o, F3 d) {
    const F3 pcam = xf_point(S.raster_to_camera, F3{pfx, pfy, 0});
    const F3 dxc = F3{S.dx_camera[0], S.dx_camera[1], S.dx_camera[2]}, dyc = F3{S.dy_camera[0], S.dy_camera[1], S.dy_camera[2]};
    F3 rxo = F3{0, 0, 0}, ryo = F3{0, 0, 0}, rxd, ryd;
    if (S.lens_radius > 0) {
        float lx, ly;
        concentric_sample_disk(lu0, lu1, &lx, &ly);
        lx = S.lens_radius * lx;
        ly = S.lens_radius * ly;
        const F3 dx = normalize(pcam + dxc);
        float ft = S.focal_distance / dx.z;
        F3 pfocus = F3{0, 0, 0} + (ft * dx);
        rxo = F3{lx, ly, 0};
        rxd = normalize(pfocus - rxo);
        const F3 dy = normalize(pcam + dyc);
        ft = S.focal_distance / dy.z;
        pfocus = F3{0, 0, 0} + (ft * dy);
        ryo = F3{lx, ly, 0};
        ryd = normalize(pfocus - ryo);
    } else {
        rxd = normalize(pcam + dxc);
        ryd = normalize(pcam + dyc);
    }
    rxo = xf_point(S.camera_to_world, rxo);
    ryo = xf_point(S.camera_to_world, ryo);
    rxd = xf_vector(S.camera_to_world, rxd);
    ryd = xf_vector(S.camera_to_world, ryd);
    const float sc = S.diff_scale;
    RayDiff r;
    r.rxo = o + (rxo - o) * sc;
    r.ryo = o + (ryo - o) * sc;
    r.rxd = d + (rxd - d) * sc;
    r.ryd = d + (ryd - d) * sc;
    return r;
}
// Camera::GenerateRayDifferential (camera.cpp:60-96) for the hemispheric probe camera: the rays through the film
// points shifted by eps = 0.05 in x and in y, differenced; then ScaleDifferentials as above
DEV RayDiff probe_differentials(const DScene &S, const DProbeCam &cam, float pfx, float pfy, F3 o, F3 d) {
    const float eps = .05f;
    F3 xo, xd, yo, yd;
    float tm;
    probe_ray(S, cam, pfx + eps, pfy, &xo, &xd, &tm);
    probe_ray(S, cam, pfx, pfy + eps, &yo, &yd, &tm);
    const float inv = 1.f / eps;  // Vector3::operator/ multiplies by the reciprocal
    const F3 rxo = o + (xo - o) * inv, rxd = d + (xd - d) * inv;
    const F3 ryo = o + (yo - o) * inv, ryd = d + (yd - d) * inv;
    const float sc = S.diff_scale;
    RayDiff r;
    r.rxo = o + (rxo - o) * sc;
    r.ryo = o + (ryo - o) * sc;
    r.rxd = d + (rxd - d) * sc;
    r.ryd = d + (ryd - d) * sc;
    return r;
}

// ===========================================================================
// ray / primitive tests
// ===========================================================================
struct RayCtx {  // per-ray constants of the watertight test (triangle.cpp:206-226) and the slab test
    // origin as three scalars, not an F3: as a sub-struct it survived scalar replacement (the
    // vectorizer gave it overlapping float2 accesses) and lived in scratch / LDS, not registers
    float ox, oy, oz;
    DEV F3 o() const { return F3{ox, oy, oz}; }
    float Sx, Sy, Sz;
    F3 inv_dir;
    int neg_mask;  // bits 0..2: dirIsNeg[xyz] (bvh.cpp:667); bits 4..5: kz, the max-|d| axis;
                   // bit 7: some 1/d is infinite, so a slab product can be NaN (0 * inf);
                   // bytes 1..3: where the ray's ENTRY planes sit in a four-wide record (byte offsets of the x, y, z
                   // planes it meets first: min planes at 0 / 16 / 32, max planes 48 further on; trav_interior4)
};
DEV float comp(F3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
DEV RayCtx make_ray_ctx(F3 o, F3 d) {
    RayCtx c;
    c.ox = o.x;
    c.oy = o.y;
    c.oz = o.z;
    F3 ad = vabs(d);
    const int kz = (ad.x > ad.y) ? ((ad.x > ad.z) ? 0 : 2) : ((ad.y > ad.z) ? 1 : 2);  // MaxDimension
    // Permute(d, kx, ky, kz) with kx = kz+1, ky = kx+1 (mod 3)
    const float dx = kz == 0 ? d.y : (kz == 1 ? d.z : d.x);
    const float dy = kz == 0 ? d.z : (kz == 1 ? d.x : d.y);
    const float dz = kz == 0 ? d.x : (kz == 1 ? d.y : d.z);
    c.Sx = -dx / dz;
    c.Sy = -dy / dz;
    c.inv_dir = F3{1 / d.x, 1 / d.y, 1 / d.z};  // bvh.cpp:666
    // Sz = 1.f / dz with dz the kz-th component of d: the quotient invDir already holds (one IEEE division less per ray)
    c.Sz = kz == 0 ? c.inv_dir.x : (kz == 1 ? c.inv_dir.y : c.inv_dir.z);
    c.neg_mask = (c.inv_dir.x < 0 ? 1 : 0) | (c.inv_dir.y < 0 ? 2 : 0) | (c.inv_dir.z < 0 ? 4 : 0) | (kz << 4);
    c.neg_mask |= int(((c.inv_dir.x < 0 ? 48u : 0u) << 8) | ((c.inv_dir.y < 0 ? 64u : 16u) << 16) | ((c.inv_dir.z < 0 ? 80u : 32u) << 24));
    if (!(fabsf(c.inv_dir.x) < IILE_INF && fabsf(c.inv_dir.y) < IILE_INF && fabsf(c.inv_dir.z) < IILE_INF)) c.neg_mask |= 0x80;
    return c;
}

// Triangle::Intersect up to the conservative t test (triangle.cpp:196-275);
// the SurfaceInteraction part is deferred to the shade kernel.
DEV bool triangle_test(const RayCtx &rc, float tmax, F3 p0, F3 p1, F3 p2, float *t_out, float *b0o, float *b1o,
                       float *b2o) {
    const F3 ro = rc.o();
    F3 a = p0 - ro, b = p1 - ro, c = p2 - ro;
    const int kz = (rc.neg_mask >> 4) & 3;
    // Permute(p, kx, ky, kz): kz == 0 -> (y,z,x); kz == 1 -> (z,x,y); kz == 2 -> (x,y,z)
    float ax = kz == 0 ? a.y : (kz == 1 ? a.z : a.x), ay = kz == 0 ? a.z : (kz == 1 ? a.x : a.y),
          az = kz == 0 ? a.x : (kz == 1 ? a.y : a.z);
    float bx = kz == 0 ? b.y : (kz == 1 ? b.z : b.x), by = kz == 0 ? b.z : (kz == 1 ? b.x : b.y),
          bz = kz == 0 ? b.x : (kz == 1 ? b.y : b.z);
    float cx = kz == 0 ? c.y : (kz == 1 ? c.z : c.x), cy = kz == 0 ? c.z : (kz == 1 ? c.x : c.y),
          cz = kz == 0 ? c.x : (kz == 1 ? c.y : c.z);
    ax += rc.Sx * az;
    ay += rc.Sy * az;
    bx += rc.Sx * bz;
    by += rc.Sy * bz;
    cx += rc.Sx * cz;
    cy += rc.Sy * cz;
    float e0 = bx * cy - by * cx;
    float e1 = cx * ay - cy * ax;
    float e2 = ax * by - ay * bx;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {  // double-precision fallback on edges
        double p2txp1ty = (double)cx * (double)by;
        double p2typ1tx = (double)cy * (double)bx;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)ax * (double)cy;
        double p0typ2tx = (double)ay * (double)cx;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)bx * (double)ay;
        double p1typ0tx = (double)by * (double)ax;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    float det = e0 + e1 + e2;
    if (det == 0) return false;
    az *= rc.Sz;
    bz *= rc.Sz;
    cz *= rc.Sz;
    float t_scaled = e0 * az + e1 * bz + e2 * cz;
    if (det < 0 && (t_scaled >= 0 || t_scaled < tmax * det))
        return false;
    else if (det > 0 && (t_scaled <= 0 || t_scaled > tmax * det))
        return false;
    float inv_det = 1 / det;
    float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
    float t = t_scaled * inv_det;
    float max_zt = max3(fabsf(az), fabsf(bz), fabsf(cz));
    float delta_z = kGamma3 * max_zt;
    float max_xt = max3(fabsf(ax), fabsf(bx), fabsf(cx));
    float max_yt = max3(fabsf(ay), fabsf(by), fabsf(cy));
    float delta_x = kGamma5 * (max_xt + max_zt);
    float delta_y = kGamma5 * (max_yt + max_zt);
    float delta_e = 2 * (kGamma2 * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
    float max_e = max3(fabsf(e0), fabsf(e1), fabsf(e2));
    float delta_t = 3 * (kGamma3 * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * fabsf(inv_det);
    if (t <= delta_t) return false;
    *t_out = t;
    *b0o = b0;
    *b1o = b1;
    *b2o = b2;
    return true;
}

// Sphere::Intersect / IntersectP up to the hit decision (sphere.cpp:49-103), partial spheres included (zmin / zmax / phimax: the
// clipping branch of :89-104 — taken, and phi's atan2 evaluated, only for a sphere that is cut: a full sphere's test cannot fail).
// Outputs the object-space ray and refined hit point for sphere_interaction.
DEV bool sphere_test(const DSphere &sp, F3 ro, F3 rd, float tmax, float *t_hit, F3 *obj_d, F3 *phit) {
    F3 oerr, derr;
    F3 o = xf_point_err(sp.o2w_inv, ro, &oerr);
    F3 d = xf_vector_err(sp.o2w_inv, rd, &derr);
    float len2 = length_sq(d);
    if (len2 > 0) {  // transform.h:382-394 (tMax unchanged)
        float dt = dot(vabs(d), oerr) / len2;
        o = o + d * dt;
    }
    // Exact early-out on the value track of the interval arithmetic. Every EFloat
    // keeps lo <= v <= hi, and `v` never depends on lo/hi, so
    //   t0.v > tMax  =>  t0.hi > tMax   and   t1.v <= 0  =>  t1.lo <= 0,
    // i.e. the reference's rejection `t0.UpperBound() > tMax || t1.LowerBound() <= 0`
    // (sphere.cpp:72) is already decided. Shadow rays all end 1e-4 short of the
    // light they aim at, so nearly every sphere test on this path leaves here
    // without the ~40 next_up/next_down pairs of the interval track.
    {
        const float av = (d.x * d.x + d.y * d.y) + d.z * d.z;
        const float bv = 2.f * ((d.x * o.x + d.y * o.y) + d.z * o.z);
        const float cv = ((o.x * o.x + o.y * o.y) + o.z * o.z) - sp.radius * sp.radius;
        const double discrim = (double)bv * (double)bv - 4. * (double)av * (double)cv;
        if (discrim < 0.) return false;
        const float root = float(sqrt(discrim));
        const float qv = (bv < 0) ? -.5f * (bv - root) : -.5f * (bv + root);
        float t0v = qv / av, t1v = cv / qv;
        if (t0v > t1v) {
            const float tmp = t0v;
            t0v = t1v;
            t1v = tmp;
        }
        if (t0v > tmax || t1v <= 0) return false;
    }
    EF ox = ef(o.x, oerr.x), oy = ef(o.y, oerr.y), oz = ef(o.z, oerr.z);
    EF dx = ef(d.x, derr.x), dy = ef(d.y, derr.y), dz = ef(d.z, derr.z);
    EF a = dx * dx + dy * dy + dz * dz;
    EF b = ef(2.f) * (dx * ox + dy * oy + dz * oz);
    EF c = ox * ox + oy * oy + oz * oz - ef(sp.radius) * ef(sp.radius);
    EF t0, t1;
    if (!ef_quadratic(a, b, c, &t0, &t1)) return false;
    if (t0.hi > tmax || t1.lo <= 0) return false;
    EF ts = t0;
    if (ts.lo <= 0) {
        ts = t1;
        if (ts.hi > tmax) return false;
    }
    F3 ph = o + d * ts.v;
    float scale = sp.radius / length(ph);
    ph = F3{ph.x * scale, ph.y * scale, ph.z * scale};
    if (ph.x == 0 && ph.y == 0) ph.x = 1e-5f * sp.radius;
    const bool z_cut = sp.zmin > -sp.radius || sp.zmax < sp.radius, phi_cut = sp.phi_max < 6.2831853f;   // Radians(360) = 6.2831855f
    if (z_cut || phi_cut) {
        auto clipped = [&](F3 q) {
            bool out = (sp.zmin > -sp.radius && q.z < sp.zmin) || (sp.zmax < sp.radius && q.z > sp.zmax);
            if (phi_cut) {
                float phi = atan2_f(q.y, q.x);
                if (phi < 0) phi += 2 * kPi;
                out = out || phi > sp.phi_max;
            }
            return out;
        };
        if (clipped(ph)) {
            if (ts.v == t1.v) return false;
            if (t1.hi > tmax) return false;
            ts = t1;
            ph = o + d * ts.v;
            scale = sp.radius / length(ph);
            ph = F3{ph.x * scale, ph.y * scale, ph.z * scale};
            if (ph.x == 0 && ph.y == 0) ph.x = 1e-5f * sp.radius;
            if (clipped(ph)) return false;
        }
    }
    *t_hit = ts.v;
    *obj_d = d;
    *phit = ph;
    return true;
}

// What shading needs of a SurfaceInteraction (core/interaction.h)
struct Isect {
    F3 p, perr, n, wo, sn, sdpdu;
    // texture lookups (triangles): the hit's (u, v) and dp/du, dp/dv; dead code where no texture is read
    float u, v;
    F3 dpdu, dpdv;
    // bump mapping: shading.dpdv, shading.dndu / dndv, reverseOrientation ^ transformSwapsHandedness
    F3 sdpdv, dndu, dndv;
    bool flip;
};

// Sphere::Intersect's interaction + Transform::operator()(SurfaceInteraction)
// (sphere.cpp:104-155, interaction.cpp:44-70, transform.cpp:262-297)
template <bool DIFFS = false>
DEV void sphere_interaction(const DSphere &sp, F3 obj_d, F3 ph, Isect *is) {
    float theta = acos_f(clampf(ph.z / sp.radius, -1, 1));
    float z_radius = sqrtf(ph.x * ph.x + ph.y * ph.y);
    float inv_z_radius = 1 / z_radius;
    float cos_phi = ph.x * inv_z_radius;
    float sin_phi = ph.y * inv_z_radius;
    float st, ct;
    sincos_f(theta, &st, &ct);
    F3 dpdu = F3{-sp.phi_max * ph.y, sp.phi_max * ph.x, 0};
    F3 dpdv = (sp.theta_max - sp.theta_min) * F3{ph.z * cos_phi, ph.z * sin_phi, -sp.radius * st};
    F3 perr = kGamma5 * vabs(ph);
    F3 n = normalize(cross(dpdu, dpdv));
    F3 sn = n;
    if (sp.reverse_orientation ^ sp.swaps_handedness) {
        n = n * -1.f;
        sn = sn * -1.f;
    }
    F3 wo = normalize(-obj_d);
    is->p = xf_point_err2(sp.o2w, ph, perr, &is->perr);
    is->n = normalize(xf_normal(sp.o2w_inv, n));
    is->wo = normalize(xf_vector(sp.o2w, wo));
    F3 snw = normalize(xf_normal(sp.o2w_inv, sn));
    is->sdpdu = xf_vector(sp.o2w, dpdu);
    is->sn = faceforward(snw, is->n);
    if (DIFFS) {
        // what the direct pass's reflected-ray differentials need of a sphere hit (directprogressiveintegrator.cpp:165-184):
        // dpdu / dpdv for ComputeDifferentials and dndu / dndv from the fundamental forms (sphere.cpp:122-143), in world space
        // (transform.cpp:275-283: vectors by the matrix, Normal3f by the inverse transpose)
        const float dt = sp.theta_max - sp.theta_min;
        const F3 d2Pduu = (-sp.phi_max * sp.phi_max) * F3{ph.x, ph.y, 0};
        const F3 d2Pduv = (dt * ph.z * sp.phi_max) * F3{-sin_phi, cos_phi, 0.f};
        const F3 d2Pdvv = (-dt * dt) * F3{ph.x, ph.y, ph.z};
        const float E = dot(dpdu, dpdu), F = dot(dpdu, dpdv), G = dot(dpdv, dpdv);
        const F3 N = normalize(cross(dpdu, dpdv));
        const float e = dot(N, d2Pduu), f = dot(N, d2Pduv), g = dot(N, d2Pdvv);
        const float inv_egf2 = 1 / (E * G - F * F);
        const F3 dndu = ((f * F - e * G) * inv_egf2) * dpdu + ((e * F - f * E) * inv_egf2) * dpdv;
        const F3 dndv = ((g * F - f * G) * inv_egf2) * dpdu + ((f * F - g * E) * inv_egf2) * dpdv;
        is->dpdu = is->sdpdu;
        is->dpdv = is->sdpdv = xf_vector(sp.o2w, dpdv);
        is->dndu = xf_normal(sp.o2w_inv, dndu);
        is->dndv = xf_normal(sp.o2w_inv, dndv);
        // Point2f(u, v) of the hit (sphere.cpp:107-109); phi as Sphere::Intersect computes it from the refined hit point
        float phi = atan2_f(ph.y, ph.x);
        if (phi < 0) phi += 2 * kPi;
        is->u = phi / sp.phi_max;
        is->v = (theta - sp.theta_min) / (sp.theta_max - sp.theta_min);
        is->flip = sp.reverse_orientation ^ sp.swaps_handedness;
    }
}

// Triangle::Intersect's interaction (triangle.cpp:277-400) from the stored
// barycentrics of the closest hit.
DEV void triangle_interaction(const DScene &S, int prim, uint32_t flags, F3 p0, F3 p1, F3 p2, F3 ray_d, float b0,
                              float b1, float b2, Isect *is) {
    float uv00 = 0, uv01 = 0, uv10 = 1, uv11 = 0, uv20 = 1, uv21 = 1;  // triangle.h:98-108
    if (flags & 4u) {
        const float2 *u = S.tri_uv + 3 * size_t(prim);
        float2 a = u[0], b = u[1], c = u[2];
        uv00 = a.x;
        uv01 = a.y;
        uv10 = b.x;
        uv11 = b.y;
        uv20 = c.x;
        uv21 = c.y;
    }
    float duv02x = uv00 - uv20, duv02y = uv01 - uv21;
    float duv12x = uv10 - uv20, duv12y = uv11 - uv21;
    F3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = duv02x * duv12y - duv02y * duv12x;
    bool degenerate = double(fabsf(determinant)) < 1e-8;
    F3 dpdu = F3{0, 0, 0}, dpdv = F3{0, 0, 0};
    if (!degenerate) {
        float invdet = 1 / determinant;
        dpdu = (duv12y * dp02 - duv02y * dp12) * invdet;
        dpdv = (-duv12x * dp02 + duv02x * dp12) * invdet;
    }
    if (degenerate || length_sq(cross(dpdu, dpdv)) == 0)
        coordinate_system(normalize(cross(p2 - p0, p1 - p0)), &dpdu, &dpdv);
    float xs = (fabsf(b0 * p0.x) + fabsf(b1 * p1.x) + fabsf(b2 * p2.x));
    float ys = (fabsf(b0 * p0.y) + fabsf(b1 * p1.y) + fabsf(b2 * p2.y));
    float zs = (fabsf(b0 * p0.z) + fabsf(b1 * p1.z) + fabsf(b2 * p2.z));
    is->perr = kGamma7 * F3{xs, ys, zs};
    is->p = b0 * p0 + b1 * p1 + b2 * p2;
    is->u = b0 * uv00 + b1 * uv10 + b2 * uv20;  // uvHit, triangle.cpp:318
    is->v = b0 * uv01 + b1 * uv11 + b2 * uv21;
    is->dpdu = dpdu;
    is->dpdv = dpdv;
    is->wo = normalize(-ray_d);
    F3 n = normalize(cross(dp02, dp12));
    const bool flip = (flags & 8u) != 0;
    if (flags & 2u) {
        const float4 *nn = S.tri_norms + 3 * size_t(prim);
        float4 a = nn[0], b = nn[1], c = nn[2];
        F3 n0 = F3{a.x, a.y, a.z}, n1 = F3{b.x, b.y, b.z}, n2 = F3{c.x, c.y, c.z};
        F3 ns = (b0 * n0 + b1 * n1 + b2 * n2);
        if (length_sq(ns) > 0)
            ns = normalize(ns);
        else
            ns = n;
        F3 ss = normalize(dpdu);
        F3 ts = cross(ss, ns);
        if (length_sq(ts) > 0.f) {
            ts = normalize(ts);
            ss = cross(ts, ns);
        } else
            coordinate_system(ns, &ss, &ts);
        // dndu, dndv of the interpolated normal, triangle.cpp:374-392
        const F3 dn1 = n0 - n2, dn2 = n1 - n2;
        if (double(fabsf(determinant)) < 1e-8) {
            is->dndu = is->dndv = F3{0, 0, 0};
        } else {
            const float inv_det = 1 / determinant;
            is->dndu = (duv12y * dn1 - duv02y * dn2) * inv_det;
            is->dndv = (-duv12x * dn1 + duv02x * dn2) * inv_det;
        }
        F3 sn = normalize(cross(ss, ts));  // SetShadingGeometry, interaction.cpp:72-92
        if (flip) sn = -sn;
        n = faceforward(n, sn);
        is->sn = sn;
        is->sdpdu = ss;
        is->sdpdv = ts;
    } else {
        if (flip) n = -n;
        is->sn = n;
        is->sdpdu = dpdu;
        is->sdpdv = dpdv;
        is->dndu = is->dndv = F3{0, 0, 0};
    }
    is->flip = flip;
    is->n = n;
}

// ===========================================================================
// BVH traversal (accelerators/bvh.cpp:662-738, core/geometry.h:1411-1438)
// ===========================================================================
// Same tree, same visiting order and the same accept / reject decisions as the
// reference, restructured for 64-lane wavefronts:
//
//  * "Wide" 64-byte interior records hold the boxes of BOTH children
//    (children[0] = the node at i+1, children[1] = secondChildOffset), so one
//    fetch resolves two of the reference's node visits and leaves need no node
//    fetch at all (a leaf reference is ~firstPrimitive; the last primitive of a
//    leaf carries a flag bit in its vertex record).
//  * The near child is tested and entered at once. The far child's slab test
//    does not depend on ray.tMax except for its final `tMin < ray.tMax`
//    comparison (geometry.h:1437), so its tMin is cached on the stack and that
//    one comparison is repeated at pop time against the tMax of that moment —
//    the decision the reference takes when it visits the node later.
//  * while-while: lanes first walk interior nodes (cheap iterations), then all
//    lanes that reached a leaf run the triangle test together, instead of paying
//    the triangle code on every iteration because some lane is at a leaf.
//  * One ray per lane; per-lane stack of (ref, tMin) in LDS as stack[level][lane]
//    (64 dwords per level): lane l always hits bank l mod 32, so pushes and pops
//    are conflict-free whatever depth each lane is at. The LDS part is a ring holding the
//    newest kLdsStackDepth levels; older ones are evicted to an HBM column (see stack_push).
//
// LDS pointers carry their address space explicitly so that pushes and pops
// compile to ds_write_b32 / ds_read_b32 (a generic pointer would go through flat_*).
typedef __attribute__((address_space(3))) int lds_int;
#ifndef IILE_LDS_STACK
#define IILE_LDS_STACK 11  // 22 KB per block + the 3 KB copy of the tree's top: six blocks per 160 KB CU
#endif
constexpr int kLdsStackDepth = IILE_LDS_STACK;  // measured max depth on killeroo-simple: 19 (binary steps)
// The reference's stack holds 64 binary entries (bvh.cpp:670); a four-wide step defers up to
// three slots where the binary walk defers one child, so the same tree needs up to 1.5x that.
constexpr int kSpillStackDepth = 128 - IILE_LDS_STACK;
constexpr int kStackWordsPerWave = 2 * kLdsStackDepth * 64;  // ref plane + tMin plane

struct TraceStats {
    uint32_t nodes, tris, tri_hits, spheres;
};

struct HitRec {
    int prim;  // -1 = miss
    float t, b0, b1, b2;
};

// Bounds3::IntersectP(ray, invDir, dirIsNeg) without its final ray.tMax
// comparison: returns whether the slabs overlap with tMax_box > 0 and the entry
// distance tMin. The caller finishes with `tMin < ray.tMax`.
DEV bool slab_entry(const RayCtx &rc, float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz,
                    float *tmin_out) {
    const bool nx = rc.neg_mask & 1, ny = (rc.neg_mask & 2) != 0, nz = (rc.neg_mask & 4) != 0;
    float tmin = ((nx ? bmaxx : bminx) - rc.ox) * rc.inv_dir.x;
    float tmx = ((nx ? bminx : bmaxx) - rc.ox) * rc.inv_dir.x;
    float tymin = ((ny ? bmaxy : bminy) - rc.oy) * rc.inv_dir.y;
    float tymax = ((ny ? bminy : bmaxy) - rc.oy) * rc.inv_dir.y;
    tmx *= kSlabScale;
    tymax *= kSlabScale;
    bool ok = !(tmin > tymax || tymin > tmx);
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmx) tmx = tymax;
    float tzmin = ((nz ? bmaxz : bminz) - rc.oz) * rc.inv_dir.z;
    float tzmax = ((nz ? bminz : bmaxz) - rc.oz) * rc.inv_dir.z;
    tzmax *= kSlabScale;
    ok = ok && !(tmin > tzmax || tzmin > tmx);
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmx) tmx = tzmax;
    *tmin_out = tmin;
    return ok && (tmx > 0);
}

// The same test for a ray whose slab products cannot be NaN (all 1/d finite: neg_mask bit 7
// clear). Without NaNs the reference's compare-and-replace chain selects exactly
// max(tx0, ty0, tz0) and min(tx1, ty1, tz1), and its two overlap tests pass iff the three
// intervals share a point, i.e. iff that maximum <= that minimum. (The chain never tests an
// axis interval against itself; one can only be inverted by the 1+2*gamma(3) scaling of a
// negative far plane, and then tMax < 0 fails both formulations.) v_max3/v_min3 replace
// eight compare/select pairs; signed zeros can differ but tMin/tMax are only ever compared.
// Both child boxes at once, as float2 lanes (v_pk_add_f32 / v_pk_mul_f32 are full rate).
typedef float v2f __attribute__((ext_vector_type(2)));
DEV void slab_entry_finite2(const RayCtx &rc, const float4 q0, const float4 q1, const float4 q2, bool *ok_a,
                            bool *ok_b, float *tmin_a, float *tmin_b) {
    // children[0] = (q0.xyz, q0.w q1.xy), children[1] = (q1.zw q2.x, q2.yzw)
    const bool nx = rc.neg_mask & 1, ny = (rc.neg_mask & 2) != 0, nz = (rc.neg_mask & 4) != 0;
    const v2f x0 = v2f{nx ? q0.w : q0.x, nx ? q2.y : q1.z}, x1 = v2f{nx ? q0.x : q0.w, nx ? q1.z : q2.y};
    const v2f y0 = v2f{ny ? q1.x : q0.y, ny ? q2.z : q1.w}, y1 = v2f{ny ? q0.y : q1.x, ny ? q1.w : q2.z};
    const v2f z0 = v2f{nz ? q1.y : q0.z, nz ? q2.w : q2.x}, z1 = v2f{nz ? q0.z : q1.y, nz ? q2.x : q2.w};
    const float fox = rc.ox, foy = rc.oy, foz = rc.oz, fix = rc.inv_dir.x, fiy = rc.inv_dir.y, fiz = rc.inv_dir.z;
    const v2f ox = v2f{fox, fox}, oy = v2f{foy, foy}, oz = v2f{foz, foz}, ix = v2f{fix, fix}, iy = v2f{fiy, fiy},
              iz = v2f{fiz, fiz}, sc = v2f{kSlabScale, kSlabScale};
    const v2f tx0 = (x0 - ox) * ix, tx1 = (x1 - ox) * ix * sc;
    const v2f ty0 = (y0 - oy) * iy, ty1 = (y1 - oy) * iy * sc;
    const v2f tz0 = (z0 - oz) * iz, tz1 = (z1 - oz) * iz * sc;
    const float mna = __builtin_fmaxf(__builtin_fmaxf(tx0.x, ty0.x), tz0.x);
    const float mxa = __builtin_fminf(__builtin_fminf(tx1.x, ty1.x), tz1.x);
    const float mnb = __builtin_fmaxf(__builtin_fmaxf(tx0.y, ty0.y), tz0.y);
    const float mxb = __builtin_fminf(__builtin_fminf(tx1.y, ty1.y), tz1.y);
    *tmin_a = mna;
    *tmin_b = mnb;
    *ok_a = mna <= mxa && mxa > 0;
    *ok_b = mnb <= mxb && mxb > 0;
}

// Resumable traversal state of one lane. The persistent kernels keep a Trav per
// lane and run the interior / leaf phases for the whole wavefront, refilling
// lanes whose ray has finished; traverse() below is the single-ray wrapper.
struct Trav {
    RayCtx rc;
    float tmax;
    int cur;  // >= 0: wide interior record; < 0: leaf, ~cur = first primitive
    int sp;
    bool have;  // cur is a node that passed its box test and still has to be processed
    int hit_prim;      // -1: none; else primitive index | hit_tag(flags)
    float b0, b1, b2;  // barycentrics of the closest triangle hit (t itself is t.tmax)
};
// The vertex record's flag word carries the primitive's shading class (bits 5..7: material
// type, +4 for a sphere) and its (area light index + 1) (bits 8..11). Both ride along in
// hit_prim bits 24..30, so that extend can tag shade-queue entries with the class and the
// MIS kernel knows whether its ray ended on an emitter, without fetching the primitive again.
constexpr int kHitClassShift = 24;
constexpr int kHitLightShift = 27;
constexpr int kHitPrimMask = (1 << kHitClassShift) - 1;
DEV int hit_tag(uint32_t flags) { return int((flags >> 5) & 0x7fu) << kHitClassShift; }
DEV int hit_index(int hit_prim) { return hit_prim < 0 ? -1 : (hit_prim & kHitPrimMask); }
// The top of the tree in LDS. The traversal kernels are bound by the vector memory pipe, not by arithmetic (r03 counters:
// TA busy 0.74-0.86, TD busy 0.92-0.99 of the kernel's cycles; a divergent dwordx4 load costs ~26 address-unit cycles and an
// interior step issues seven of them against ~60 cycles of VALU per CU): the records every ray passes through first — the
// breadth-first top of the four-wide tree — are therefore read from a per-block LDS copy, which takes no part in that pipe.
// A reference to such a record is kTopFlag | slot (still > 0 = interior); the copies refer to each other that way and to
// everything below by the ordinary record index. Same records, same decisions.
constexpr int kTopStride = 144;  // bytes per LDS record: 128 + 16, so that neighbouring records start 4 banks apart
typedef __attribute__((address_space(3))) char lds_char;
struct StackRef {
    lds_int *lds;          // this lane's LDS column: ref plane [level*64], tMin plane [(kLdsStackDepth+level)*64]
    int *spill_base;       // HBM overflow: lane column = spill_base + spill_col, 2 ints per level
    uint32_t spill_col;
    uint32_t spill_stride;
    lds_char *top = nullptr;  // the block's copy of DScene::top4 (kTopStride bytes per record), or null
    int root = 0;             // where a traversal starts: DScene::root_ref_top with `top`, else DScene::root_ref
    DEV int *spill() const { return spill_base + spill_col; }
};
// a block's threads copy the top records into its LDS array (kMaxTop * kTopStride bytes); the caller synchronises
typedef float lds_v4f_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) lds_v4f_t lds_v4f;  // (HIP's float4 is a class: no assignment across address spaces)
DEV float4 lds_load4(lds_char *p) {
    const lds_v4f_t v = *reinterpret_cast<lds_v4f *>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
DEV void stage_top_records(const DScene &S, lds_char *top, int tid, int n_threads) {
    for (int i = tid; i < S.n_top * 8; i += n_threads) {
        const float4 v = S.top4[i];
        *reinterpret_cast<lds_v4f *>(top + (i >> 3) * kTopStride + (i & 7) * 16) = lds_v4f_t{v.x, v.y, v.z, v.w};
    }
}

template <bool COUNT>
DEV void trav_begin(const DScene &S, Trav &t, F3 ro, F3 rd, float tmax, TraceStats *st, int root) {
    t.rc = make_ray_ctx(ro, rd);
    t.tmax = tmax;
    t.sp = 0;
    t.hit_prim = -1;
    t.b0 = t.b1 = t.b2 = 0;
    t.cur = root;
    t.have = false;
    if (S.n_nodes == 0) return;
    // the root is visited like any node: its own box against ray.tMax
    float tmin;
    if (COUNT) ++st->nodes;
    const bool ok = slab_entry(t.rc, S.root_box[0], S.root_box[1], S.root_box[2], S.root_box[3], S.root_box[4],
                               S.root_box[5], &tmin);
    t.have = ok && (tmin < tmax);
}

// The per-lane stack keeps its *newest* kLdsStackDepth levels in LDS, as a ring
// (level l lives in LDS slot l mod kLdsStackDepth); when it grows beyond that, the oldest
// level is evicted to the lane's HBM column — a store nobody waits for — and comes back only
// if the traversal ever unwinds that far. (Spilling the newest levels instead, as a plain
// array would, puts an HBM round trip on the very next pop: 12 vs 14 LDS levels cost 5 ms
// per frame that way.) Trav::sp packs both cursors: bits 0..7 = number of levels on the
// stack, bits 8.. = number of levels that live in HBM (levels [0, lo)).
// level % kLdsStackDepth for level < 128 without an integer division (exact for depths 8..32)
static_assert(kLdsStackDepth >= 5 && kLdsStackDepth <= 32, "lds_slot's reciprocal is checked for depths 5..32 (levels < 256)");
DEV int lds_slot(int level) {
    constexpr int kRecip = (65536 + kLdsStackDepth - 1) / kLdsStackDepth;
    return level - kLdsStackDepth * ((level * kRecip) >> 16);
}
DEV int stack_size(const Trav &t) { return t.sp & 0xff; }
DEV void stack_push(Trav &t, const StackRef &sr, int ref, float tmin) {
    const int sp = t.sp & 0xff, lo = t.sp >> 8;
    if (sp - lo == kLdsStackDepth) {
        const int slot = lds_slot(lo);
        const size_t off = size_t(lo) * 2 * sr.spill_stride;
        sr.spill()[off] = sr.lds[slot * 64];
        sr.spill()[off + sr.spill_stride] = sr.lds[(kLdsStackDepth + slot) * 64];
        t.sp += 256;
    }
    const int slot = lds_slot(sp);
    sr.lds[slot * 64] = ref;
    sr.lds[(kLdsStackDepth + slot) * 64] = __float_as_int(tmin);
    t.sp += 1;
}
// resume at the most recent deferred (far) child that still passes `tMin < ray.tMax`
template <bool COUNT>
DEV void trav_pop(Trav &t, const StackRef &sr, TraceStats *st) {
    t.have = false;
    while ((t.sp & 0xff) > 0) {
        --t.sp;
        const int sp = t.sp & 0xff, lo = t.sp >> 8;
        int ref;
        float tmin;
        if (sp >= lo) {
            const int slot = lds_slot(sp);
            ref = sr.lds[slot * 64];
            tmin = __int_as_float(sr.lds[(kLdsStackDepth + slot) * 64]);
        } else {  // LDS ring empty: level sp is the newest one in HBM
            const size_t off = size_t(sp) * 2 * sr.spill_stride;
            ref = sr.spill()[off];
            tmin = __int_as_float(sr.spill()[off + sr.spill_stride]);
            t.sp = sp | (sp << 8);
        }
        if (COUNT) ++st->nodes;
        if (tmin < t.tmax) {
            t.cur = ref;
            t.have = true;
            break;
        }
    }
}

// one wide interior record (already fetched): two of the reference's node visits
template <bool COUNT>
DEV void trav_interior(Trav &t, const StackRef &sr, TraceStats *st, const float4 q0, const float4 q1, const float4 q2,
                       const float4 q3) {
    const int ref_a = __float_as_int(q3.x), ref_b = __float_as_int(q3.y);
    const int axis = __float_as_int(q3.z) & 3;
    // children[0] = (q0.xyz, q0.w q1.xy), children[1] = (q1.zw q2.x, q2.yzw)
    float tmin_a, tmin_b;
    bool ok_a, ok_b;
    if (__builtin_expect(__ballot(t.rc.neg_mask & 0x80) == 0, 1)) {
        slab_entry_finite2(t.rc, q0, q1, q2, &ok_a, &ok_b, &tmin_a, &tmin_b);
    } else {
        ok_a = slab_entry(t.rc, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, &tmin_a);
        ok_b = slab_entry(t.rc, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, &tmin_b);
    }
    // bvh.cpp:686-692: with a negative direction along the split axis the second
    // child is nearer; the other one is deferred
    const bool second_first = (t.rc.neg_mask >> axis) & 1;
    const int near_ref = second_first ? ref_b : ref_a, far_ref = second_first ? ref_a : ref_b;
    const bool near_ok = second_first ? ok_b : ok_a, far_ok = second_first ? ok_a : ok_b;
    const float near_tmin = second_first ? tmin_b : tmin_a, far_tmin = second_first ? tmin_a : tmin_b;
    if (COUNT || far_ok) {
        // a far child whose slabs can never pass is only kept for the visit count
        const float ft = far_ok ? far_tmin : IILE_INF;
        stack_push(t, sr, far_ref, ft);
    }
    if (COUNT) ++st->nodes;
    if (near_ok && near_tmin < t.tmax)
        t.cur = near_ref;
    else
        trav_pop<COUNT>(t, sr, st);
}
// Interior step of one lane: fetch its 64-byte record (4 x dwordx4) and process it.
// (A quad-cooperative fetch — four lanes loading one record per request and a DPP 4x4
// transpose — was measured: it removes the vector-L1 pending-miss stalls but its
// 64 extra VALU/DPP moves cost more than they save: 61 ms vs 49 ms per step.)
template <bool COUNT>
DEV void trav_interior_step(const DScene &S, Trav &t, const StackRef &sr, TraceStats *st) {
    // index clamped so that a load the compiler hoists above the loop test
    // (observed with hipcc 7.2 on this loop nest) can never leave the array
    const float4 *w = S.wide + 4 * size_t(t.cur < 0 ? 0 : t.cur);
    trav_interior<COUNT>(t, sr, st, w[0], w[1], w[2], w[3]);
}

// ---------------------------------------------------------------------------
// Four-wide step (uninstrumented kernels only). A wide4 record of binary node P holds the
// boxes and refs of its *grandchildren* in fixed slots — slots 0,1: children of P's first
// child L (or L itself in slot 0 when L is a leaf), slots 2,3: likewise for P's second child
// R — and the split axes of P, L and R. One step enters the first grandchild the reference
// would reach and defers the others in the reference's order (L's near, L's far, R's near,
// R's far, each pair and the pairs themselves ordered by dirIsNeg of the respective axis).
//
// The intermediate nodes L and R are never tested. That cannot change which leaves are tested,
// nor in which order: a child's box lies inside its parent's (Union is exact), every slab
// operation ((plane - o) * invDir, * (1 + 2 gamma3), max3/min3) is monotone, so for a ray
// without NaN slab products "child passes" implies "parent passes" — both the slab overlap
// with tMax > 0 and tMin < ray.tMax, whatever ray.tMax was when the parent was visited
// (it only shrinks). What the reference decides at L or R is therefore implied by what it
// decides at their children. Rays with an infinite 1/d (NaN-capable) take the binary step,
// which shares refs and stack entries with this one. The visit *counters* do need L and R,
// so the instrumented kernels keep the binary step.
// The six box planes of a record are loaded as "entry x / y / z" and "exit x / y / z": which of the min / max planes
// that is depends on the ray's direction signs only, so the choice is made in the load ADDRESS (per-lane anyway) from the
// three byte offsets RayCtx keeps, instead of 24 selects on loaded values per step. The exit plane of an axis is 48 bytes
// from its entry plane, in the direction an XOR gives (records are 128-byte aligned: offset bits 0..6 are the plane's).
struct Wide4Planes {
    float4 ex, ey, ez, lx, ly, lz, refs;
    uint32_t meta;
};
template <bool WITH_META>
DEV Wide4Planes load_wide4(const float4 *wide4, int cur, int neg_mask, lds_char *top = nullptr) {
    const char *base = reinterpret_cast<const char *>(wide4);
    const uint32_t nm = uint32_t(neg_mask);
    const uint32_t px = (nm >> 8) & 0xffu, py = (nm >> 16) & 0xffu, pz = nm >> 24;  // entry planes' offsets inside a record
    Wide4Planes w;
    const bool in_top = top != nullptr && cur >= kTopFlag;
    if (top != nullptr && __ballot(in_top) != 0) {
        if (in_top) {
            typedef __attribute__((address_space(3))) uint32_t lu32;
            lds_char *r = top + uint32_t(cur - kTopFlag) * uint32_t(kTopStride);
            w.ex = lds_load4(r + px);
            w.ey = lds_load4(r + py);
            w.ez = lds_load4(r + pz);
            w.lx = lds_load4(r + (px ^ 48u));
            w.ly = lds_load4(r + (py ^ 80u));
            w.lz = lds_load4(r + (pz ^ 112u));
            w.refs = lds_load4(r + 96u);
            w.meta = (WITH_META && !kRefShift) ? *reinterpret_cast<lu32 *>(r + 112u) : 0u;
        }
        if (__ballot(!in_top) == 0) return w;  // (wave-uniform: a wavefront fresh from a refill is all in the top levels)
    }
    if (!in_top) {
        const uint32_t rec = uint32_t(cur < 0 ? 0 : cur) * 128u;  // (32-bit offsets: checked at upload)
        const uint32_t ax = rec + px, ay = rec + py, az = rec + pz;
        w.ex = *reinterpret_cast<const float4 *>(base + ax);
        w.ey = *reinterpret_cast<const float4 *>(base + ay);
        w.ez = *reinterpret_cast<const float4 *>(base + az);
        w.lx = *reinterpret_cast<const float4 *>(base + (ax ^ 48u));
        w.ly = *reinterpret_cast<const float4 *>(base + (ay ^ 80u));
        w.lz = *reinterpret_cast<const float4 *>(base + (az ^ 112u));
        w.refs = *reinterpret_cast<const float4 *>(base + (rec + 96u));
        w.meta = (WITH_META && !kRefShift) ? *reinterpret_cast<const uint32_t *>(base + (rec + 112u)) : 0u;
    }
    return w;
}
// the four refs of a record as the step uses them, and its axes word (both unpacked from the refs)
DEV void unpack_refs(const Wide4Planes &w, int *r0, int *r1, int *r2, int *r3, uint32_t *meta) {
    const int p0 = __float_as_int(w.refs.x), p1 = __float_as_int(w.refs.y), p2 = __float_as_int(w.refs.z), p3 = __float_as_int(w.refs.w);
    if (kRefShift) {
        *meta = uint32_t(p0 & 3) | uint32_t(p1 & 3) << 2 | uint32_t(p2 & 3) << 4;
        *r0 = p0 >> kRefShift, *r1 = p1 >> kRefShift, *r2 = p2 >> kRefShift, *r3 = p3 >> kRefShift;  // (arithmetic: a leaf ref is negative)
    } else {
        *meta = w.meta;
        *r0 = p0, *r1 = p1, *r2 = p2, *r3 = p3;
    }
}
DEV void trav_interior4(Trav &t, const StackRef &sr, const Wide4Planes &w) {
    const RayCtx &rc = t.rc;
    // entry / exit planes of slots (0,1) and (2,3) as float2 lanes
    const v2f x0a = v2f{w.ex.x, w.ex.y}, x0b = v2f{w.ex.z, w.ex.w}, x1a = v2f{w.lx.x, w.lx.y}, x1b = v2f{w.lx.z, w.lx.w};
    const v2f y0a = v2f{w.ey.x, w.ey.y}, y0b = v2f{w.ey.z, w.ey.w}, y1a = v2f{w.ly.x, w.ly.y}, y1b = v2f{w.ly.z, w.ly.w};
    const v2f z0a = v2f{w.ez.x, w.ez.y}, z0b = v2f{w.ez.z, w.ez.w}, z1a = v2f{w.lz.x, w.lz.y}, z1b = v2f{w.lz.z, w.lz.w};
    const float fox = rc.ox, foy = rc.oy, foz = rc.oz, fix = rc.inv_dir.x, fiy = rc.inv_dir.y, fiz = rc.inv_dir.z;
    const v2f ox = v2f{fox, fox}, oy = v2f{foy, foy}, oz = v2f{foz, foz}, ix = v2f{fix, fix}, iy = v2f{fiy, fiy},
              iz = v2f{fiz, fiz}, sc = v2f{kSlabScale, kSlabScale};
    const v2f tx0a = (x0a - ox) * ix, tx0b = (x0b - ox) * ix;
    const v2f tx1a = (x1a - ox) * ix * sc, tx1b = (x1b - ox) * ix * sc;
    const v2f ty0a = (y0a - oy) * iy, ty0b = (y0b - oy) * iy;
    const v2f ty1a = (y1a - oy) * iy * sc, ty1b = (y1b - oy) * iy * sc;
    const v2f tz0a = (z0a - oz) * iz, tz0b = (z0b - oz) * iz;
    const v2f tz1a = (z1a - oz) * iz * sc, tz1b = (z1b - oz) * iz * sc;
    // per slot: tMin, and whether it is to be visited as things stand (key = tMin, else +inf;
    // a visitable tMin is < ray.tMax <= inf, so +inf is free to mean "no")
    auto slot_key = [&](float a0, float b0, float c0, float a1, float b1, float c1) {
        const float tmin = __builtin_fmaxf(__builtin_fmaxf(a0, b0), c0);
        const float tmx = __builtin_fminf(__builtin_fminf(a1, b1), c1);
        return (tmin <= tmx && tmx > 0 && tmin < t.tmax) ? tmin : IILE_INF;
    };
    const float k0 = slot_key(tx0a.x, ty0a.x, tz0a.x, tx1a.x, ty1a.x, tz1a.x);
    const float k1 = slot_key(tx0a.y, ty0a.y, tz0a.y, tx1a.y, ty1a.y, tz1a.y);
    const float k2 = slot_key(tx0b.x, ty0b.x, tz0b.x, tx1b.x, ty1b.x, tz1b.x);
    const float k3 = slot_key(tx0b.y, ty0b.y, tz0b.y, tx1b.y, ty1b.y, tz1b.y);
    int r0, r1, r2, r3;
    uint32_t meta;
    unpack_refs(w, &r0, &r1, &r2, &r3, &meta);
    // the reference's visiting order (bvh.cpp:686-692 applied at P, L and R)
    const bool swap_p = (rc.neg_mask >> (meta & 3u)) & 1, swap_l = (rc.neg_mask >> ((meta >> 2) & 3u)) & 1,
               swap_r = (rc.neg_mask >> ((meta >> 4) & 3u)) & 1;
    const int a0r = swap_l ? r1 : r0, a1r = swap_l ? r0 : r1, b0r = swap_r ? r3 : r2, b1r = swap_r ? r2 : r3;
    const float a0k = swap_l ? k1 : k0, a1k = swap_l ? k0 : k1, b0k = swap_r ? k3 : k2, b1k = swap_r ? k2 : k3;
    const int e0r = swap_p ? b0r : a0r, e1r = swap_p ? b1r : a1r, e2r = swap_p ? a0r : b0r, e3r = swap_p ? a1r : b1r;
    const float e0k = swap_p ? b0k : a0k, e1k = swap_p ? b1k : a1k, e2k = swap_p ? a0k : b0k, e3k = swap_p ? a1k : b1k;
    const bool v0 = e0k < IILE_INF, v1 = e1k < IILE_INF, v2 = e2k < IILE_INF, v3 = e3k < IILE_INF;
    // defer everything behind the first visitable slot, farthest first
    if (v3 && (v0 || v1 || v2)) stack_push(t, sr, e3r, e3k);
    if (v2 && (v0 || v1)) stack_push(t, sr, e2r, e2k);
    if (v1 && v0) stack_push(t, sr, e1r, e1k);
    if (v0 || v1 || v2 || v3)
        t.cur = v0 ? e0r : (v1 ? e1r : (v2 ? e2r : e3r));
    else
        trav_pop<false>(t, sr, nullptr);
}
// The same step for any-hit rays (BVHAccel::IntersectP, bvh.cpp:702-738) in the uninstrumented kernels: whether SOME
// primitive is hit does not depend on the order the tree is walked in, and ray.tMax never shrinks, so the reference's
// near / far ordering (three dirIsNeg decisions and the selects that apply them to four refs and keys) is dropped:
// enter the first visitable slot, defer the rest as they come. Only the visit counters depend on the order, and the
// instrumented kernels keep the ordered binary walk.
DEV void trav_interior4_any(Trav &t, const StackRef &sr, const Wide4Planes &w) {
    const RayCtx &rc = t.rc;
    const v2f x0a = v2f{w.ex.x, w.ex.y}, x0b = v2f{w.ex.z, w.ex.w}, x1a = v2f{w.lx.x, w.lx.y}, x1b = v2f{w.lx.z, w.lx.w};
    const v2f y0a = v2f{w.ey.x, w.ey.y}, y0b = v2f{w.ey.z, w.ey.w}, y1a = v2f{w.ly.x, w.ly.y}, y1b = v2f{w.ly.z, w.ly.w};
    const v2f z0a = v2f{w.ez.x, w.ez.y}, z0b = v2f{w.ez.z, w.ez.w}, z1a = v2f{w.lz.x, w.lz.y}, z1b = v2f{w.lz.z, w.lz.w};
    const float fox = rc.ox, foy = rc.oy, foz = rc.oz, fix = rc.inv_dir.x, fiy = rc.inv_dir.y, fiz = rc.inv_dir.z;
    const v2f ox = v2f{fox, fox}, oy = v2f{foy, foy}, oz = v2f{foz, foz}, ix = v2f{fix, fix}, iy = v2f{fiy, fiy},
              iz = v2f{fiz, fiz}, sc = v2f{kSlabScale, kSlabScale};
    const v2f tx0a = (x0a - ox) * ix, tx0b = (x0b - ox) * ix;
    const v2f tx1a = (x1a - ox) * ix * sc, tx1b = (x1b - ox) * ix * sc;
    const v2f ty0a = (y0a - oy) * iy, ty0b = (y0b - oy) * iy;
    const v2f ty1a = (y1a - oy) * iy * sc, ty1b = (y1b - oy) * iy * sc;
    const v2f tz0a = (z0a - oz) * iz, tz0b = (z0b - oz) * iz;
    const v2f tz1a = (z1a - oz) * iz * sc, tz1b = (z1b - oz) * iz * sc;
    auto visit = [&](float a0, float b0, float c0, float a1, float b1, float c1) {
        const float tmin = __builtin_fmaxf(__builtin_fmaxf(a0, b0), c0);
        const float tmx = __builtin_fminf(__builtin_fminf(a1, b1), c1);
        return tmin <= tmx && tmx > 0 && tmin < t.tmax;
    };
    const bool v0 = visit(tx0a.x, ty0a.x, tz0a.x, tx1a.x, ty1a.x, tz1a.x);
    const bool v1 = visit(tx0a.y, ty0a.y, tz0a.y, tx1a.y, ty1a.y, tz1a.y);
    const bool v2 = visit(tx0b.x, ty0b.x, tz0b.x, tx1b.x, ty1b.x, tz1b.x);
    const bool v3 = visit(tx0b.y, ty0b.y, tz0b.y, tx1b.y, ty1b.y, tz1b.y);
    int r0, r1, r2, r3;
    uint32_t meta_unused;
    unpack_refs(w, &r0, &r1, &r2, &r3, &meta_unused);
    // (the cached tMin of a deferred slot only feeds trav_pop's `tMin < ray.tMax`, already decided: any value below tMax)
    if (v3 && (v0 || v1 || v2)) stack_push(t, sr, r3, 0.f);
    if (v2 && (v0 || v1)) stack_push(t, sr, r2, 0.f);
    if (v1 && v0) stack_push(t, sr, r1, 0.f);
    if (v0 || v1 || v2 || v3)
        t.cur = v0 ? r0 : (v1 ? r1 : (v2 ? r2 : r3));
    else
        trav_pop<false>(t, sr, nullptr);
}
// One interior step of the uninstrumented kernels: the four-wide record, unless a lane of the
// wavefront carries a NaN-capable ray (or the scene's boxes are not nested, which a BVH built as
// bvh.cpp:236-402 builds it cannot produce; iile_scene_create checks).
template <bool ANY = false>
DEV void trav_interior_step_fast(const DScene &S, Trav &t, const StackRef &sr) {
    if (__builtin_expect(S.boxes_nested && __ballot(t.rc.neg_mask & 0x80) == 0, 1)) {
        if (ANY)   // an any-hit ray takes its children in record order (no near-first sort: any hit ends it)
            trav_interior4_any(t, sr, load_wide4<false>(S.wide4, t.cur, t.rc.neg_mask, sr.top));
        else
            trav_interior4(t, sr, load_wide4<true>(S.wide4, t.cur, t.rc.neg_mask, sr.top));
    } else {
        int g = t.cur < 0 ? 0 : t.cur;
        // a record of the LDS top carries its own index among the binary records behind its axes word
        if (sr.top != nullptr && g >= kTopFlag)
            g = int(*reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(sr.top + uint32_t(g - kTopFlag) * uint32_t(kTopStride) + 116u));
        const float4 *w = S.wide + 4 * size_t(g);
        trav_interior<false>(t, sr, nullptr, w[0], w[1], w[2], w[3]);
    }
}
template <bool COUNT, bool ANY = false>
DEV void trav_step(const DScene &S, Trav &t, const StackRef &sr, TraceStats *st) {
    if (COUNT)
        trav_interior_step<true>(S, t, sr, st);
    else
        trav_interior_step_fast<ANY>(S, t, sr);
}

// screen-space derivatives of a hit's (u, v): SurfaceInteraction::dudx ... (interaction.h:127-128)
struct TexDiff {
    float dudx, dvdx, dudy, dvdy;
};
DEV F3 tex_evaluate(const DScene &S, int tex, float u, float v, const TexDiff &td);  // defined with the textures below

// The alpha test of Triangle::Intersect / IntersectP (triangle.cpp:325-331, 509-541) on a hit that passed the
// geometric test: isectLocal carries uvHit and zero differentials, so an ImageTexture<Float, Float> filters
// bilinearly at level 0. any_hit (IntersectP) also asks the shadow alpha mask.
DEV bool alpha_rejects(const DScene &S, int prim, uint32_t flags, float b0, float b1, float b2, bool any_hit) {
    float uv00 = 0, uv01 = 0, uv10 = 1, uv11 = 0, uv20 = 1, uv21 = 1;  // triangle.h:98-108
    if (flags & 4u) {
        const float2 *u = S.tri_uv + 3 * size_t(prim);
        const float2 a = u[0], b = u[1], c = u[2];
        uv00 = a.x, uv01 = a.y, uv10 = b.x, uv11 = b.y, uv20 = c.x, uv21 = c.y;
    }
    const float u = b0 * uv00 + b1 * uv10 + b2 * uv20, v = b0 * uv01 + b1 * uv11 + b2 * uv21;
    const int2 masks = S.prim_alpha[prim];
    const TexDiff zero = TexDiff{0, 0, 0, 0};
    if (masks.x == -2) return true;  // IILE_ALPHA_ZERO
    if (masks.x >= 0 && tex_evaluate(S, masks.x, u, v, zero).x == 0) return true;
    if (any_hit) {
        if (masks.y == -2) return true;
        if (masks.y >= 0 && tex_evaluate(S, masks.y, u, v, zero).x == 0) return true;
    }
    return false;
}

// ray_d: the float4 record holding the ray direction — only the (rare) sphere
// test needs it, so it is re-read there instead of living in registers.
// ALPHA: some mesh of the scene has an alpha mask (vertex-record flag bit 12 marks its triangles)
template <bool COUNT, bool ALPHA = false>
DEV bool trav_leaf(const DScene &S, Trav &t, const StackRef &sr, TraceStats *st, const bool any_hit,
                   const float4 *ray_d) {
    int prim = t.cur < 0 ? ~t.cur : 0;  // clamped like trav_interior's index; tri_verts has one pad record
    bool last;
    do {
        // the primitive's three records in one round trip: the flag word (sphere? last of its leaf?) rides in the first one, and
        // waiting for it before asking for the other two made every leaf step two dependent trips to memory
        // (the room 474 -> 455 ms, killeroo 47.3 -> 46.7: profiles/r04_ab_traversal_scheduling.txt)
        float4 v0 = S.tri_verts[3 * size_t(prim)];
        float4 v1_ = S.tri_verts[3 * size_t(prim) + 1];
        float4 v2_ = S.tri_verts[3 * size_t(prim) + 2];
        asm volatile("" : "+v"(v0.w), "+v"(v1_.x), "+v"(v2_.x));  // (keeps the compiler from sinking the two loads behind the flag test)
        const uint32_t flags = f2b(v0.w);
        last = (flags & 16u) != 0;
        if (flags & 1u) {
            if (COUNT) ++st->spheres;
            float th;
            F3 od, ph;
            const float4 d4 = *ray_d;
            // the lanes that test the same sphere go together, the sphere's fields in SGPRs (uniform_entry): one turn of the
            // loop when the scene has one sphere, which is the common case
            const int sphere = S.n_spheres == 1 ? 0 : S.prim_shape[prim];
            bool sphere_hit = false;
            for (bool pending = true; pending;) {
                const int s_now = __builtin_amdgcn_readfirstlane(sphere);
                if (sphere == s_now) {
                    sphere_hit = sphere_test(uniform_entry(S.spheres, s_now), t.rc.o(), F3{d4.x, d4.y, d4.z}, t.tmax, &th, &od, &ph);
                    pending = false;
                }
            }
            if (sphere_hit) {
                if (any_hit) {
                    t.have = false;
                    return true;
                }
                t.tmax = th;
                t.hit_prim = prim | hit_tag(flags);
                t.b0 = t.b1 = t.b2 = 0;
            }
        } else {
            const float4 v1 = v1_, v2 = v2_;
            if (COUNT) ++st->tris;
            float th, b0, b1, b2;
            if (triangle_test(t.rc, t.tmax, F3{v0.x, v0.y, v0.z}, F3{v1.x, v1.y, v1.z}, F3{v2.x, v2.y, v2.z}, &th, &b0,
                              &b1, &b2) &&
                !(ALPHA && (flags & 4096u) && alpha_rejects(S, prim, flags, b0, b1, b2, any_hit))) {
                if (COUNT) ++st->tri_hits;
                if (any_hit) {
                    t.have = false;
                    return true;
                }
                t.tmax = th;
                t.hit_prim = prim | hit_tag(flags);
                t.b0 = b0;
                t.b1 = b1;
                t.b2 = b2;
            }
        }
        ++prim;
        // one primitive per step: the wavefront's next vote sees the lanes whose leaf goes on,
        // instead of every lane waiting for the longest leaf (most leaves hold one primitive)
        if (!last) {
            t.cur = ~prim;
            return false;
        }
    } while (!last);
    trav_pop<COUNT>(t, sr, st);
    return false;
}

// single-ray wrapper (kernel-level probes)
template <bool ANY_HIT, bool COUNT, bool ALPHA = true>
DEV bool traverse(const DScene &S, F3 ro, F3 rd, float tmax, lds_int *lds_stack, int *spill, uint32_t spill_stride,
                  HitRec *hit, TraceStats *st) {
    Trav t;
    StackRef sr{lds_stack, spill, 0u, spill_stride};
    sr.root = S.root_ref;
    const float4 d4 = make_float4(rd.x, rd.y, rd.z, 0.f);
    trav_begin<COUNT>(S, t, ro, rd, tmax, st, sr.root);
    while (t.have) {
        while (t.have && t.cur >= 0) trav_step<COUNT, ANY_HIT>(S, t, sr, st);
        if (t.have && trav_leaf<COUNT, ALPHA>(S, t, sr, st, ANY_HIT, &d4)) return true;
    }
    hit->prim = hit_index(t.hit_prim);
    hit->t = t.tmax;
    hit->b0 = t.b0;
    hit->b1 = t.b1;
    hit->b2 = t.b2;
    return t.hit_prim >= 0;
}

// ===========================================================================
// BSDF (core/reflection.{h,cpp}, core/microfacet.cpp)
// ===========================================================================
struct Bsdf {
    F3 ns, ng, ss, ts;
    F3 kd, ks, kr, kt;
    float alpha, eta;
    float alpha_y;  // TrowbridgeReitzDistribution(alphax = alpha, alphay): uber's / glass's "vroughness"; the same value as alpha otherwise
    // UberMaterial's SpecularTransmission lobes (uber.cpp:53-61, 94-99): the pass-through of a surface that is not opaque —
    // SpecularTransmission(t0 = 1 - opacity, 1, 1), the FIRST lobe — and SpecularTransmission(kt = opacity Kt, 1, eta), the LAST;
    // path_eta = BSDF::eta (1 with the pass-through, else the material's: path.cpp:151-157 reads it)
    F3 t0;
    bool has_t0, has_t1;
    float path_eta;
    // rough glass (glass.cpp:66-90): MicrofacetReflection(kr -> ks, FresnelDielectric(1, eta)) is the microfacet lobe;
    // MicrofacetTransmission(kt, distrib, 1, eta, Radiance) — glossy, not specular
    bool has_mtrans;
    int n_lobes;  // nBxDFs; BxDF order: [pass-through], Lambertian, microfacet, specular reflection, [uber's Kt lobe]
    float on_a, on_b;  // Oren-Nayar constants of the diffuse lobe (oren_nayar set)
    bool oren_nayar;
    int mtype;    // kMat*: selects the Fresnel terms (plastic 1.5/1; uber 1/eta; mirror none) and, for
                  // glass, makes the specular lobe a FresnelSpecular(kr, kt, 1, eta)
    bool has_lambert, has_micro, has_spec;
};
DEV int n_nonspec(const Bsdf &b) { return (b.has_lambert ? 1 : 0) + (b.has_micro ? 1 : 0) + (b.has_mtrans ? 1 : 0); }
DEV F3 to_local(const Bsdf &b, F3 v) { return F3{dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)}; }
DEV F3 to_world(const Bsdf &b, F3 v) {
    return F3{b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z,
              b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z};
}
// ===========================================================================
// image textures: SurfaceInteraction::ComputeDifferentials (interaction.cpp:103-149),
// UVMapping2D::Map (texture.cpp:93-99), MIPMap<RGBSpectrum>::Lookup / triangle / EWA / Texel
// (mipmap.h:210-355) over the host-built pyramid — operation for operation as the oracle's tex_* functions
// ===========================================================================
DEV bool solve_2x2(float a00, float a01, float a10, float a11, float b0, float b1, float *x0, float *x1) {  // transform.cpp:41-49
    const float det = a00 * a11 - a01 * a10;
    if (fabsf(det) < 1e-10f) return false;
    *x0 = (a11 * b0 - a01 * b1) / det;
    *x1 = (a00 * b1 - a10 * b0) / det;
    if (*x0 != *x0 || *x1 != *x1) return false;
    return true;
}
DEV bool is_inf_or_nan(float v) { return !(fabsf(v) < IILE_INF); }
// (dpdx / dpdy, interaction.cpp:117-118 — zero when the auxiliary rays miss the tangent plane —, are what the direct pass's
//  reflected-ray differentials start from; every other caller leaves them out)
DEV TexDiff compute_differentials(const Isect &is, const RayDiff &rd, F3 *dpdx = nullptr, F3 *dpdy = nullptr) {
    TexDiff t = TexDiff{0, 0, 0, 0};
    if (dpdx) *dpdx = *dpdy = F3{0, 0, 0};
    const F3 n = is.n, p = is.p;
    const float d = dot(n, p);
    const float tx = -(dot(n, rd.rxo) - d) / dot(n, rd.rxd);
    if (is_inf_or_nan(tx)) return t;
    const F3 px = rd.rxo + tx * rd.rxd;
    const float ty = -(dot(n, rd.ryo) - d) / dot(n, rd.ryd);
    if (is_inf_or_nan(ty)) return t;
    const F3 py = rd.ryo + ty * rd.ryd;
    if (dpdx) *dpdx = px - p, *dpdy = py - p;
    int d0, d1;
    if (fabsf(n.x) > fabsf(n.y) && fabsf(n.x) > fabsf(n.z)) {
        d0 = 1;
        d1 = 2;
    } else if (fabsf(n.y) > fabsf(n.z)) {
        d0 = 0;
        d1 = 2;
    } else {
        d0 = 0;
        d1 = 1;
    }
    const float a00 = comp(is.dpdu, d0), a01 = comp(is.dpdv, d0), a10 = comp(is.dpdu, d1), a11 = comp(is.dpdv, d1);
    const float bx0 = comp(px, d0) - comp(p, d0), bx1 = comp(px, d1) - comp(p, d1);
    const float by0 = comp(py, d0) - comp(p, d0), by1 = comp(py, d1) - comp(p, d1);
    if (!solve_2x2(a00, a01, a10, a11, bx0, bx1, &t.dudx, &t.dvdx)) t.dudx = t.dvdx = 0;
    if (!solve_2x2(a00, a01, a10, a11, by0, by1, &t.dudy, &t.dvdy)) t.dudy = t.dvdy = 0;
    return t;
}
DEV int mod_i(int a, int b) {  // pbrt.h:310-314
    const int r = a - (a / b) * b;
    return r < 0 ? r + b : r;
}
DEV F3 tex_texel(const DScene &S, const DTexture &t, int level, int s, int tt) {
    const int w = t.level_w[level], h = t.level_h[level];
    if (t.wrap == kWrapRepeat) {
        // level sizes are powers of two: Mod is a mask (two's complement handles negative s)
        s = s & (w - 1);
        tt = tt & (h - 1);
    } else if (t.wrap == kWrapClamp) {
        s = s < 0 ? 0 : (s > w - 1 ? w - 1 : s);
        tt = tt < 0 ? 0 : (tt > h - 1 ? h - 1 : tt);
    } else if (s < 0 || s >= w || tt < 0 || tt >= h) {
        return F3{0, 0, 0};
    }
    const float4 c = S.texels[t.level_offset[level] + (long long)tt * w + s];
    return F3{c.x, c.y, c.z};
}
DEV F3 tex_triangle(const DScene &S, const DTexture &t, int level, float st0, float st1) {
    level = level < 0 ? 0 : (level > t.n_levels - 1 ? t.n_levels - 1 : level);
    const float s = st0 * float(t.level_w[level]) - 0.5f;
    const float tt = st1 * float(t.level_h[level]) - 0.5f;
    const float fs = floorf(s), ft = floorf(tt);
    const int s0 = int(fs), t0 = int(ft);
    const float ds = s - float(s0), dt = tt - float(t0);
    return tex_texel(S, t, level, s0, t0) * ((1 - ds) * (1 - dt)) + tex_texel(S, t, level, s0, t0 + 1) * ((1 - ds) * dt) +
           tex_texel(S, t, level, s0 + 1, t0) * (ds * (1 - dt)) + tex_texel(S, t, level, s0 + 1, t0 + 1) * (ds * dt);
}
DEV F3 lerp_f3(float t, F3 a, F3 b) { return a * (1 - t) + b * t; }
DEV F3 tex_lookup_width(const DScene &S, const DTexture &t, float st0, float st1, float width) {  // mipmap.h:233-250
    const float level = float(t.n_levels - 1) + log2_f(mx(width, 1e-8f));
    if (level < 0) return tex_triangle(S, t, 0, st0, st1);
    if (level >= float(t.n_levels - 1)) return tex_texel(S, t, t.n_levels - 1, 0, 0);
    const int il = int(floorf(level));
    const float delta = level - float(il);
    return lerp_f3(delta, tex_triangle(S, t, il, st0, st1), tex_triangle(S, t, il + 1, st0, st1));
}
DEV F3 tex_ewa(const DScene &S, const DTexture &t, int level, float st0, float st1, float d00, float d01, float d10, float d11) {
    if (level >= t.n_levels) return tex_texel(S, t, t.n_levels - 1, 0, 0);
    const float w = float(t.level_w[level]), h = float(t.level_h[level]);
    st0 = st0 * w - 0.5f;
    st1 = st1 * h - 0.5f;
    d00 *= w;
    d01 *= h;
    d10 *= w;
    d11 *= h;
    float A = d01 * d01 + d11 * d11 + 1;
    float B = -2 * (d00 * d01 + d10 * d11);
    float C = d00 * d00 + d10 * d10 + 1;
    const float invF = 1 / (A * C - B * B * 0.25f);
    A *= invF;
    B *= invF;
    C *= invF;
    const float det = -B * B + 4 * A * C;
    const float inv_det = 1 / det;
    const float u_sqrt = sqrtf(det * C), v_sqrt = sqrtf(A * det);
    const int s0 = int(ceilf(st0 - 2 * inv_det * u_sqrt));
    const int s1 = int(floorf(st0 + 2 * inv_det * u_sqrt));
    const int t0 = int(ceilf(st1 - 2 * inv_det * v_sqrt));
    const int t1 = int(floorf(st1 + 2 * inv_det * v_sqrt));
    F3 sum = F3{0, 0, 0};
    float sum_wts = 0;
    for (int it = t0; it <= t1; ++it) {
        const float tt = float(it) - st1;
        for (int is = s0; is <= s1; ++is) {
            const float ss = float(is) - st0;
            const float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
            if (r2 < 1) {
                int index = int(r2 * 128.f);
                index = index < 127 ? index : 127;
                const float weight = S.ewa_lut[index];
                sum = sum + tex_texel(S, t, level, is, it) * weight;
                sum_wts += weight;
            }
        }
    }
    return F3{sum.x / sum_wts, sum.y / sum_wts, sum.z / sum_wts};
}
DEV F3 tex_evaluate(const DScene &S, int tex, float u, float v, const TexDiff &td) {
    const DTexture &t = S.textures[tex];
    float d00 = t.su * td.dudx, d01 = t.sv * td.dvdx, d10 = t.su * td.dudy, d11 = t.sv * td.dvdy;
    const float st0 = t.su * u + t.du, st1 = t.sv * v + t.dv;
    if (t.trilinear) {
        const float width = mx(mx(fabsf(d00), fabsf(d01)), mx(fabsf(d10), fabsf(d11)));
        return tex_lookup_width(S, t, st0, st1, 2 * width);
    }
    if (d00 * d00 + d01 * d01 < d10 * d10 + d11 * d11) {
        float tmp = d00;
        d00 = d10;
        d10 = tmp;
        tmp = d01;
        d01 = d11;
        d11 = tmp;
    }
    const float major = sqrtf(d00 * d00 + d01 * d01);
    float minor = sqrtf(d10 * d10 + d11 * d11);
    if (minor * t.max_aniso < major && minor > 0) {
        const float scale = major / (minor * t.max_aniso);
        d10 *= scale;
        d11 *= scale;
        minor *= scale;
    }
    if (minor == 0) return tex_triangle(S, t, 0, st0, st1);
    const float lod = mx(0.f, float(t.n_levels) - 1.f + log2_f(minor));
    const int ilod = int(floorf(lod));
    return lerp_f3(lod - float(ilod), tex_ewa(S, t, ilod, st0, st1, d00, d01, d10, d11),
                   tex_ewa(S, t, ilod + 1, st0, st1, d00, d01, d10, d11));
}
// Material::Bump (material.cpp:45-86) with an ImageTexture<Float, Float> displacement, then
// SetShadingGeometry(dpdu, dpdv, dndu, dndv, false) (interaction.cpp:72-92)
DEV void bump(const DScene &S, int tex, const TexDiff &td, Isect *is) {
    float du = .5f * (fabsf(td.dudx) + fabsf(td.dudy));
    if (du == 0) du = .0005f;
    const float u_displace = tex_evaluate(S, tex, is->u + du, is->v + 0.f, td).x;
    float dv = .5f * (fabsf(td.dvdx) + fabsf(td.dvdy));
    if (dv == 0) dv = .0005f;
    const float v_displace = tex_evaluate(S, tex, is->u + 0.f, is->v + dv, td).x;
    const float displace = tex_evaluate(S, tex, is->u, is->v, td).x;
    const F3 dpdu = is->sdpdu + (u_displace - displace) / du * is->sn + displace * is->dndu;
    const F3 dpdv = is->sdpdv + (v_displace - displace) / dv * is->sn + displace * is->dndv;
    F3 sn = normalize(cross(dpdu, dpdv));
    if (is->flip) sn = -sn;
    sn = faceforward(sn, is->n);
    is->sn = sn;
    is->sdpdu = dpdu;
    is->sdpdv = dpdv;
}

// the material with its textured parameters looked up at the hit (Texture::Evaluate(*si))
DEV DMaterial textured_material(const DScene &S, const DMaterial &m, const Isect &is, const TexDiff &td) {
    DMaterial r = m;
    if (m.kd_tex >= 0) {
        const F3 c = tex_evaluate(S, m.kd_tex, is.u, is.v, td);  // times the constant: 1, or a "scale" texture's factor
        r.kd[0] = c.x * m.kd[0], r.kd[1] = c.y * m.kd[1], r.kd[2] = c.z * m.kd[2];
    }
    if (m.ks_tex >= 0) {
        const F3 c = tex_evaluate(S, m.ks_tex, is.u, is.v, td);  // times the constant: 1, or a "scale" texture's factor
        r.ks[0] = c.x * m.ks[0], r.ks[1] = c.y * m.ks[1], r.ks[2] = c.z * m.ks[2];
    }
    if (m.kr_tex >= 0) {
        const F3 c = tex_evaluate(S, m.kr_tex, is.u, is.v, td);  // times the constant: 1, or a "scale" texture's factor
        r.kr[0] = c.x * m.kr[0], r.kr[1] = c.y * m.kr[1], r.kr[2] = c.z * m.kr[2];
    }
    if (m.sigma_tex >= 0) {  // sigma->Evaluate(*si), matte.cpp:56-61; OrenNayar's constants, reflection.h:414-420
        const float sig = clampf(tex_evaluate(S, m.sigma_tex, is.u, is.v, td).x, 0.f, 90.f);
        r.on_a = 1.f;
        r.on_b = 0.f;
        if (sig != 0) {
            const float sg = (kPi / 180) * sig;
            const float sigma2 = sg * sg;
            r.on_a = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
            r.on_b = 0.45f * sigma2 / (sigma2 + 0.09f);
        }
    }
    if (m.rough_tex >= 0) {  // roughness->Evaluate(*si), then RoughnessToAlpha (microfacet.h:123-128)
        float rough = tex_evaluate(S, m.rough_tex, is.u, is.v, td).x;
        if (m.remap_roughness) {
            rough = mx(rough, 1e-3f);
            const float x = log_f(rough);
            rough = 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
        }
        r.alpha = rough;
        if (m.rough_tex_v == -2) r.alpha_y = rough;   // roughv = roughu, uber.cpp:83-84 (plastic: one roughness)
    }
    if (m.rough_tex_v >= 0) {  // "vroughness" as a float image (uber.cpp:76, 83)
        float rough = tex_evaluate(S, m.rough_tex_v, is.u, is.v, td).x;
        if (m.remap_roughness) {
            rough = mx(rough, 1e-3f);
            const float x = log_f(rough);
            rough = 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
        }
        r.alpha_y = rough;
    }
    if (m.opacity_tex >= 0) {  // opacity->Evaluate(*si), uber.cpp:53
        const F3 c = tex_evaluate(S, m.opacity_tex, is.u, is.v, td);
        r.opacity[0] = c.x * m.opacity[0], r.opacity[1] = c.y * m.opacity[1], r.opacity[2] = c.z * m.opacity[2];
    }
    if (m.kt_tex >= 0) {
        const F3 c = tex_evaluate(S, m.kt_tex, is.u, is.v, td);  // times the constant: 1, or a "scale" texture's factor
        r.kt[0] = c.x * m.kt[0], r.kt[1] = c.y * m.kt[1], r.kt[2] = c.z * m.kt[2];
    }
    return r;
}

// Matte / Plastic / Uber / Mirror ComputeScatteringFunctions (matte.cpp:45-62, plastic.cpp:45-70,
// uber.cpp:45-100 with opacity 1 and Kt 0, mirror.cpp:44-55)
// EXT = false: the scene has matte and plastic only (checked at upload); the specular lobes then fold away
template <bool EXT = true>
DEV Bsdf make_bsdf(const DMaterial &m, const Isect &is) {
    Bsdf b;
    b.ns = is.sn;
    b.ng = is.n;
    b.ss = normalize(is.sdpdu);
    b.ts = cross(b.ns, b.ss);
    b.n_lobes = 0;
    // the material's first 32 bytes as two 16-byte loads (field by field they come as four)
    float4 m_kd, m_ks;  // (bitcast type, kd), (ks, alpha)
    __builtin_memcpy(&m_kd, &m.type, 16);
    __builtin_memcpy(&m_ks, &m.ks[0], 16);
    keep_whole(m_kd);
    keep_whole(m_ks);
    const int m_type = int(f2b(m_kd.x));
    // UberMaterial (uber.cpp:53-61): op = opacity.Clamp(), t = (-op + Spectrum(1.f)).Clamp(); every other coefficient is op * K.Clamp()
    const bool uber = EXT && m_type == kMatUber;
    F3 op = F3{1.f, 1.f, 1.f};
    b.t0 = F3{0, 0, 0};
    b.has_t0 = b.has_t1 = false;
    if (uber) {
        op = F3{clampf(m.opacity[0], 0, IILE_INF), clampf(m.opacity[1], 0, IILE_INF), clampf(m.opacity[2], 0, IILE_INF)};
        b.t0 = F3{clampf(-op.x + 1.f, 0, IILE_INF), clampf(-op.y + 1.f, 0, IILE_INF), clampf(-op.z + 1.f, 0, IILE_INF)};
        b.has_t0 = !is_black(b.t0);
        if (b.has_t0) ++b.n_lobes;
    }
    b.kd = F3{clampf(m_kd.y, 0, IILE_INF), clampf(m_kd.z, 0, IILE_INF), clampf(m_kd.w, 0, IILE_INF)};
    if (uber) b.kd = op * b.kd;
    b.has_lambert = !is_black(b.kd);
    if (b.has_lambert) ++b.n_lobes;
    b.ks = F3{0, 0, 0};
    b.has_micro = false;
    b.alpha = m_ks.w;
    b.alpha_y = (EXT && (m_type == kMatUber || m_type == kMatGlass)) ? m.alpha_y : b.alpha;   // (uber with a roughness image: textured_material sets both)
    b.oren_nayar = EXT && m_type == kMatMatte && m.on_b != 0.f;  // matte.cpp:56-61 (B == 0 iff sigma == 0)
    b.on_a = m.on_a;
    b.on_b = m.on_b;
    b.mtype = EXT ? m_type : kMatPlastic;
    b.eta = EXT ? m.eta : 1.f;  // (only uber, mirror and glass read it)
    b.path_eta = b.has_t0 ? 1.f : b.eta;   // BSDF(*si, 1.f) / BSDF(*si, e), uber.cpp:56-61
    if (m_type == kMatPlastic || (EXT && m_type == kMatUber)) {
        b.ks = F3{clampf(m_ks.x, 0, IILE_INF), clampf(m_ks.y, 0, IILE_INF), clampf(m_ks.z, 0, IILE_INF)};
        if (uber) b.ks = op * b.ks;
        b.has_micro = !is_black(b.ks);
        if (b.has_micro) ++b.n_lobes;
    }
    b.kr = F3{0, 0, 0};
    b.kt = F3{0, 0, 0};
    b.has_spec = false;
    if (EXT && (m_type == kMatUber || m_type == kMatMirror)) {
        b.kr = F3{clampf(m.kr[0], 0, IILE_INF), clampf(m.kr[1], 0, IILE_INF), clampf(m.kr[2], 0, IILE_INF)};
        if (uber) b.kr = op * b.kr;
        b.has_spec = !is_black(b.kr);
        if (b.has_spec) ++b.n_lobes;
    }
    if (uber) {   // SpecularTransmission(op * Kt.Clamp(), 1, e), uber.cpp:94-99
        b.kt = op * F3{clampf(m.kt[0], 0, IILE_INF), clampf(m.kt[1], 0, IILE_INF), clampf(m.kt[2], 0, IILE_INF)};
        b.has_t1 = !is_black(b.kt);
        if (b.has_t1) ++b.n_lobes;
    }
    b.has_mtrans = false;
    if (EXT && m_type == kMatGlass && (m_ks.w != 0.f || m.alpha_y != 0.f)) {  // glass.cpp:63-90: a rough dielectric (an alpha != 0)
        b.ks = F3{clampf(m.kr[0], 0, IILE_INF), clampf(m.kr[1], 0, IILE_INF), clampf(m.kr[2], 0, IILE_INF)};   // R: MicrofacetReflection
        b.kt = F3{clampf(m.kt[0], 0, IILE_INF), clampf(m.kt[1], 0, IILE_INF), clampf(m.kt[2], 0, IILE_INF)};   // T: MicrofacetTransmission
        b.has_micro = !is_black(b.ks);
        b.has_mtrans = !is_black(b.kt);
        if (b.has_micro) ++b.n_lobes;
        if (b.has_mtrans) ++b.n_lobes;
    } else if (EXT && m_type == kMatGlass) {  // glass.cpp:45-66 with isSpecular && allowMultipleLobes
        b.kr = F3{clampf(m.kr[0], 0, IILE_INF), clampf(m.kr[1], 0, IILE_INF), clampf(m.kr[2], 0, IILE_INF)};
        b.kt = F3{clampf(m.kt[0], 0, IILE_INF), clampf(m.kt[1], 0, IILE_INF), clampf(m.kt[2], 0, IILE_INF)};
        b.has_spec = !(is_black(b.kr) && is_black(b.kt));
        if (b.has_spec) ++b.n_lobes;
    }
    return b;
}
// reflection.h:56-84
DEV float cos2_theta(F3 w) { return w.z * w.z; }
DEV float sin2_theta(F3 w) { return mx(0.f, 1.f - cos2_theta(w)); }
DEV float sin_theta(F3 w) { return sqrtf(sin2_theta(w)); }
DEV float tan_theta(F3 w) { return sin_theta(w) / w.z; }
DEV float tan2_theta(F3 w) { return sin2_theta(w) / cos2_theta(w); }
DEV float cos_phi(F3 w) {
    float st = sin_theta(w);
    return (st == 0) ? 1 : clampf(w.x / st, -1, 1);
}
DEV float sin_phi(F3 w) {
    float st = sin_theta(w);
    return (st == 0) ? 0 : clampf(w.y / st, -1, 1);
}
DEV float cos2_phi(F3 w) { return cos_phi(w) * cos_phi(w); }
DEV float sin2_phi(F3 w) { return sin_phi(w) * sin_phi(w); }
DEV bool same_hemisphere(F3 a, F3 b) { return a.z * b.z > 0; }
// FrDielectric, reflection.cpp:47-68
DEV float fr_dielectric(float cos_i, float eta_i, float eta_t) {
    cos_i = clampf(cos_i, -1, 1);
    bool entering = cos_i > 0.f;
    if (!entering) {
        float tmp = eta_i;
        eta_i = eta_t;
        eta_t = tmp;
        cos_i = fabsf(cos_i);
    }
    float sin_i = sqrtf(mx(0.f, 1 - cos_i * cos_i));
    float sin_t = eta_i / eta_t * sin_i;
    if (sin_t >= 1) return 1;
    float cos_t = sqrtf(mx(0.f, 1 - sin_t * sin_t));
    float r_parl = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
    float r_perp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return (r_parl * r_parl + r_perp * r_perp) / 2;
}
// TrowbridgeReitzDistribution::D / Lambda, microfacet.cpp:155-163, 176-184
// (ax, ay: alphax, alphay. Where the scene has no anisotropic material — the plain build always — the two are one value and the
//  expressions below compile to what they were with one alpha)
DEV float tr_d(F3 wh, float ax, float ay) {
    float t2 = tan2_theta(wh);
    if (is_inf(t2)) return 0.f;
    const float cos4 = cos2_theta(wh) * cos2_theta(wh);
    float e = (cos2_phi(wh) / (ax * ax) + sin2_phi(wh) / (ay * ay)) * t2;
    return 1 / (kPi * ax * ay * cos4 * (1 + e) * (1 + e));
}
DEV float tr_lambda(F3 w, float ax, float ay) {
    float abs_tan = fabsf(tan_theta(w));
    if (is_inf(abs_tan)) return 0.f;
    float alpha = sqrtf(cos2_phi(w) * ax * ax + sin2_phi(w) * ay * ay);
    float a2t2 = (alpha * abs_tan) * (alpha * abs_tan);
    return (-1 + sqrtf(1.f + a2t2)) / 2;
}
DEV float tr_g1(F3 w, float ax, float ay) { return 1 / (1 + tr_lambda(w, ax, ay)); }
DEV float tr_g(F3 wo, F3 wi, float ax, float ay) { return 1 / (1 + tr_lambda(wo, ax, ay) + tr_lambda(wi, ax, ay)); }
DEV float tr_pdf(F3 wo, F3 wh, float ax, float ay) { return tr_d(wh, ax, ay) * tr_g1(wo, ax, ay) * absdot(wo, wh) / fabsf(wo.z); }
// TrowbridgeReitzSample11, microfacet.cpp:238-283. The normal-incidence branch
// evaluates sqrt/cos/sin through the C (double) overloads in the reference.
DEV void tr_sample11(float cos_theta, float U1, float U2, float *slope_x, float *slope_y) {
    if (double(cos_theta) > .9999) {
        float r = float(sqrt(double(U1 / (1 - U1))));
        float phi = float(6.28318530718 * double(U2));
        double s, c;
        sincos_d(double(phi), &s, &c);
        *slope_x = float(double(r) * c);
        *slope_y = float(double(r) * s);
        return;
    }
    float sin_t = sqrtf(mx(0.f, 1.f - cos_theta * cos_theta));
    float tan_t = sin_t / cos_theta;
    float a = 1 / tan_t;
    float G1 = 2 / (1 + sqrtf(1.f + 1.f / (a * a)));
    float A = 2 * U1 / G1 - 1;
    float tmp = 1.f / (A * A - 1.f);
    if (double(tmp) > 1e10) tmp = 1e10f;
    float B = tan_t;
    float D = sqrtf(mx(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.f));
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
    *slope_y = Sg * z * sqrtf(1.f + *slope_x * *slope_x);
}
// TrowbridgeReitzSample + Sample_wh (visible-area), microfacet.cpp:285-336
DEV F3 tr_sample_wh(F3 wo, float u0, float u1, float ax, float ay) {
    bool flip = wo.z < 0;
    F3 wi = flip ? -wo : wo;
    F3 ws = normalize(F3{ax * wi.x, ay * wi.y, wi.z});
    float sx, sy;
    tr_sample11(ws.z, u0, u1, &sx, &sy);
    float tmp = cos_phi(ws) * sx - sin_phi(ws) * sy;
    sy = sin_phi(ws) * sx + cos_phi(ws) * sy;
    sx = tmp;
    sx = ax * sx;
    sy = ay * sy;
    F3 wh = normalize(F3{-sx, -sy, 1.f});
    if (flip) wh = -wh;
    return wh;
}
// MicrofacetReflection::f, reflection.cpp:226-236, with FresnelDielectric(1.5, 1) (plastic) or (1, e) (uber)
DEV F3 micro_f(const Bsdf &b, F3 wo, F3 wi) {
    float cos_o = fabsf(wo.z), cos_i = fabsf(wi.z);
    F3 wh = wi + wo;
    if (cos_i == 0 || cos_o == 0) return F3{0, 0, 0};
    if (wh.x == 0 && wh.y == 0 && wh.z == 0) return F3{0, 0, 0};
    wh = normalize(wh);
    float Fr = (b.mtype == kMatUber || b.mtype == kMatGlass) ? fr_dielectric(dot(wi, wh), 1.f, b.eta) : fr_dielectric(dot(wi, wh), 1.5f, 1.f);
    F3 F = F3{Fr, Fr, Fr};
    return sdiv(b.ks * tr_d(wh, b.alpha, b.alpha_y) * tr_g(wo, wi, b.alpha, b.alpha_y) * F, 4 * cos_i * cos_o);
}
DEV float micro_pdf(const Bsdf &b, F3 wo, F3 wi) {
    if (!same_hemisphere(wo, wi)) return 0;
    F3 wh = normalize(wo + wi);
    return tr_pdf(wo, wh, b.alpha, b.alpha_y) / (4 * dot(wo, wh));
}
// LambertianReflection::f (reflection.cpp:178-180) or OrenNayar::f (reflection.cpp:197-219)
DEV F3 diffuse_f(const Bsdf &b, F3 wo, F3 wi) {
    if (!b.oren_nayar) return b.kd * kInvPi;
    const float sin_i = sin_theta(wi), sin_o = sin_theta(wo);
    float max_cos = 0;
    if (double(sin_i) > 1e-4 && double(sin_o) > 1e-4) {
        const float sin_phi_i = sin_phi(wi), cos_phi_i = cos_phi(wi);
        const float sin_phi_o = sin_phi(wo), cos_phi_o = cos_phi(wo);
        const float d_cos = cos_phi_i * cos_phi_o + sin_phi_i * sin_phi_o;
        max_cos = mx(0.f, d_cos);
    }
    float sin_alpha, tan_beta;
    if (fabsf(wi.z) > fabsf(wo.z)) {
        sin_alpha = sin_o;
        tan_beta = sin_i / fabsf(wi.z);
    } else {
        sin_alpha = sin_i;
        tan_beta = sin_o / fabsf(wo.z);
    }
    return b.kd * kInvPi * (b.on_a + b.on_b * max_cos * sin_alpha * tan_beta);
}
DEV float lambert_pdf(F3 wo, F3 wi) { return same_hemisphere(wo, wi) ? fabsf(wi.z) * kInvPi : 0; }
// Refract, reflection.h:96-108
DEV bool refract_dir(F3 wi, F3 n, float eta, F3 *wt) {
    const float cos_i = dot(n, wi);
    const float sin2_i = mx(0.f, 1 - cos_i * cos_i);
    const float sin2_t = eta * eta * sin2_i;
    if (sin2_t >= 1) return false;
    const float cos_t = sqrtf(1 - sin2_t);
    *wt = eta * -wi + (eta * cos_i - cos_t) * n;
    return true;
}
// MicrofacetTransmission::f, reflection.cpp:244-266 (etaA = 1, etaB = b.eta, mode == Radiance)
DEV F3 mtrans_f(const Bsdf &b, F3 wo, F3 wi) {
    if (same_hemisphere(wo, wi)) return F3{0, 0, 0};
    const float cos_o = wo.z, cos_i = wi.z;
    if (cos_i == 0 || cos_o == 0) return F3{0, 0, 0};
    const float eta_a = 1.f, eta_b = b.eta;
    const float eta = wo.z > 0 ? (eta_b / eta_a) : (eta_a / eta_b);
    F3 wh = normalize(wo + wi * eta);
    if (wh.z < 0) wh = -wh;
    const float F = fr_dielectric(dot(wo, wh), eta_a, eta_b);
    const float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
    const float factor = 1 / eta;
    const float omf = 1.f - F;
    return F3{omf, omf, omf} * b.kt *
           fabsf(tr_d(wh, b.alpha, b.alpha_y) * tr_g(wo, wi, b.alpha, b.alpha_y) * eta * eta * absdot(wi, wh) * absdot(wo, wh) * factor * factor /
                 (cos_i * cos_o * sqrt_denom * sqrt_denom));
}
// MicrofacetTransmission::Pdf, reflection.cpp:435-447
DEV float mtrans_pdf(const Bsdf &b, F3 wo, F3 wi) {
    if (same_hemisphere(wo, wi)) return 0;
    const float eta_a = 1.f, eta_b = b.eta;
    const float eta = wo.z > 0 ? (eta_b / eta_a) : (eta_a / eta_b);
    const F3 wh = normalize(wo + wi * eta);
    const float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
    const float dwh_dwi = fabsf((eta * eta * dot(wi, wh)) / (sqrt_denom * sqrt_denom));
    return tr_pdf(wo, wh, b.alpha, b.alpha_y) * dwh_dwi;
}
DEV F3 lobes_f(const Bsdf &b, F3 wo, F3 wi) {
    F3 f = F3{0, 0, 0};
    if (b.has_lambert) f = f + diffuse_f(b, wo, wi);
    if (b.has_micro) f = f + micro_f(b, wo, wi);
    return f;
}
// BSDF::f, reflection.cpp:686-699
DEV F3 bsdf_f(const Bsdf &b, F3 woW, F3 wiW) {
    F3 wi = to_local(b, wiW), wo = to_local(b, woW);
    if (wo.z == 0) return F3{0, 0, 0};
    bool reflect = dot(wiW, b.ng) * dot(woW, b.ng) > 0;
    if (reflect) return lobes_f(b, wo, wi);
    return b.has_mtrans ? F3{0, 0, 0} + mtrans_f(b, wo, wi) : F3{0, 0, 0};   // `(!reflect && (bxdfs[i]->type & BSDF_TRANSMISSION))`
}
// BSDF::Pdf, reflection.cpp:786-801
DEV float bsdf_pdf(const Bsdf &b, F3 woW, F3 wiW) {
    if (b.n_lobes == 0) return 0.f;
    F3 wo = to_local(b, woW), wi = to_local(b, wiW);
    if (wo.z == 0) return 0.f;
    float pdf = 0.f;
    if (b.has_lambert) pdf += lambert_pdf(wo, wi);
    if (b.has_micro) pdf += micro_pdf(b, wo, wi);
    if (b.has_mtrans) pdf += mtrans_pdf(b, wo, wi);
    const int matching = n_nonspec(b);  // flags = BSDF_ALL & ~BSDF_SPECULAR
    return matching > 0 ? pdf / matching : 0.f;
}
// BSDF::Sample_f, reflection.cpp:719-784. *pdf is untouched on the early
// `wo.z == 0` return, as in the reference.
DEV F3 bsdf_sample_f(const Bsdf &b, F3 woW, F3 *wiW, float u0, float u1, float *pdf, const bool allow_specular = false,
                     bool *sampled_specular = nullptr, bool *sampled_transmission = nullptr) {
    // `type` is BSDF_ALL (allow_specular) or BSDF_ALL & ~BSDF_SPECULAR
    if (sampled_specular) *sampled_specular = false;
    if (sampled_transmission) *sampled_transmission = false;
    const int matching = allow_specular ? b.n_lobes : n_nonspec(b);
    if (matching == 0) {
        *pdf = 0;
        return F3{0, 0, 0};
    }
    int comp = int(floorf(u0 * matching));
    if (comp > matching - 1) comp = matching - 1;
    // the comp-th present lobe in BxDF order: [3 uber's pass-through], 0 Lambertian, 1 microfacet, 2 specular reflection, [4 uber's Kt lobe]
    int pick, count = comp;
    if (allow_specular && b.has_t0 && count-- == 0)
        pick = 3;
    else if (b.has_lambert && count-- == 0)
        pick = 0;
    else if (b.has_micro && count-- == 0)
        pick = 1;
    else if (b.has_mtrans && count-- == 0)
        pick = 5;   // rough glass: MicrofacetTransmission behind MicrofacetReflection (glass.cpp:74-90)
    else if (!(allow_specular && b.has_t1) || (b.has_spec && count-- == 0))
        pick = 2;
    else
        pick = 4;
    // (comp < matching: the specular lobe is only ever picked when there is one — said aloud so that the builds whose
    // materials have none, where has_spec is a constant, drop that branch and the loads that feed it)
    if (pick == 2 && !b.has_spec) __builtin_unreachable();
    const float ur0 = mn(u0 * matching - comp, kOneMinusEpsilon);
    F3 wo = to_local(b, woW);
    if (wo.z == 0) return F3{0, 0, 0};
    *pdf = 0;
    F3 wi = F3{0, 0, 0}, f;
    if (pick == 0) {  // BxDF::Sample_f, reflection.cpp:378-385
        wi = cosine_sample_hemisphere(ur0, u1);
        if (wo.z < 0) wi.z *= -1;
        *pdf = lambert_pdf(wo, wi);
        f = diffuse_f(b, wo, wi);
    } else if (pick == 1) {  // MicrofacetReflection::Sample_f, reflection.cpp:405-417
        F3 wh = tr_sample_wh(wo, ur0, u1, b.alpha, b.alpha_y);
        wi = -wo + 2 * dot(wo, wh) * wh;
        if (!same_hemisphere(wo, wi))
            f = F3{0, 0, 0};
        else {
            *pdf = tr_pdf(wo, wh, b.alpha, b.alpha_y) / (4 * dot(wo, wh));
            f = micro_f(b, wo, wi);
        }
    } else if (pick == 5) {  // MicrofacetTransmission::Sample_f, reflection.cpp:425-433
        const F3 wh = tr_sample_wh(wo, ur0, u1, b.alpha, b.alpha_y);
        const float eta_a = 1.f, eta_b = b.eta;
        const float eta = wo.z > 0 ? (eta_a / eta_b) : (eta_b / eta_a);
        if (!refract_dir(wo, wh, eta, &wi)) return F3{0, 0, 0};  // `return 0`, pdf stays 0
        *pdf = mtrans_pdf(b, wo, wi);
        f = mtrans_f(b, wo, wi);
    } else if (pick >= 3) {  // SpecularTransmission::Sample_f, reflection.cpp:154-170 (mode == Radiance)
        const float eta_a = 1.f, eta_b = pick == 3 ? 1.f : b.eta;
        const bool entering = wo.z > 0;
        const float eta_i = entering ? eta_a : eta_b, eta_t = entering ? eta_b : eta_a;
        // Refract(wo, Faceforward(Normal3f(0, 0, 1), wo), etaI / etaT, wi), reflection.h:96-108; -n carries negative zeros, as there
        const F3 n = (wo.z < 0.f) ? -F3{0, 0, 1} : F3{0, 0, 1};
        const float eta = eta_i / eta_t;
        const float cos_i = dot(n, wo);
        const float sin2_i = mx(0.f, 1 - cos_i * cos_i);
        const float sin2_t = eta * eta * sin2_i;
        if (sin2_t >= 1) return F3{0, 0, 0};  // `return 0`, pdf stays 0
        const float cos_t = sqrtf(1 - sin2_t);
        wi = eta * -wo + (eta * cos_i - cos_t) * n;
        *pdf = 1;
        F3 ft = (pick == 3 ? b.t0 : b.kt) * (1.f - fr_dielectric(wi.z, eta_a, eta_b));
        ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
        f = sdiv(ft, fabsf(wi.z));
        if (sampled_specular) *sampled_specular = true;
        if (sampled_transmission) *sampled_transmission = true;
    } else if (b.mtype == kMatGlass) {  // FresnelSpecular::Sample_f, reflection.cpp:477-511 (mode == Radiance)
        const float eta_a = 1.f, eta_b = b.eta;
        const float F = fr_dielectric(wo.z, eta_a, eta_b);
        if (ur0 < F) {
            wi = F3{-wo.x, -wo.y, wo.z};
            *pdf = F;
            f = sdiv(F * b.kr, fabsf(wi.z));
        } else {
            const bool entering = wo.z > 0;
            const float eta_i = entering ? eta_a : eta_b, eta_t = entering ? eta_b : eta_a;
            // Refract(wo, Faceforward(Normal3f(0, 0, 1), wo), etaI / etaT, wi), reflection.h:96-108;
            // -n carries negative zeros, as there
            const F3 n = (wo.z < 0.f) ? -F3{0, 0, 1} : F3{0, 0, 1};
            const float eta = eta_i / eta_t;
            const float cos_i = dot(n, wo);
            const float sin2_i = mx(0.f, 1 - cos_i * cos_i);
            const float sin2_t = eta * eta * sin2_i;
            if (sin2_t >= 1) return F3{0, 0, 0};  // total internal reflection: `return 0`, pdf stays 0
            const float cos_t = sqrtf(1 - sin2_t);
            wi = eta * -wo + (eta * cos_i - cos_t) * n;
            F3 ft = b.kt * (1 - F);
            ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
            *pdf = 1 - F;
            f = sdiv(ft, fabsf(wi.z));
            if (sampled_transmission) *sampled_transmission = true;
        }
        if (sampled_specular) *sampled_specular = true;
    } else {  // SpecularReflection::Sample_f, reflection.cpp:136-143
        wi = F3{-wo.x, -wo.y, wo.z};
        *pdf = 1;
        const float fr = b.mtype == kMatMirror ? 1.f : fr_dielectric(wi.z, 1.f, b.eta);
        f = sdiv(F3{fr, fr, fr} * b.kr, fabsf(wi.z));
        if (sampled_specular) *sampled_specular = true;
    }
    if (*pdf == 0) {
        if (sampled_specular) *sampled_specular = false;
        if (sampled_transmission) *sampled_transmission = false;
        return F3{0, 0, 0};
    }
    *wiW = to_world(b, wi);
    const bool glossy = pick < 2 || pick == 5;
    if (glossy && matching > 1) {  // a specular lobe's Pdf() and f() are 0
        if (pick != 0 && b.has_lambert) *pdf += lambert_pdf(wo, wi);
        if (pick != 1 && b.has_micro) *pdf += micro_pdf(b, wo, wi);
        if (pick != 5 && b.has_mtrans) *pdf += mtrans_pdf(b, wo, wi);
    }
    if (matching > 1) *pdf /= matching;
    if (glossy && matching > 1) {
        bool reflect = dot(*wiW, b.ng) * dot(woW, b.ng) > 0;
        f = reflect ? lobes_f(b, wo, wi) : (b.has_mtrans ? F3{0, 0, 0} + mtrans_f(b, wo, wi) : F3{0, 0, 0});
    }
    return f;
}

// ===========================================================================
// sphere emitter (shapes/sphere.cpp:219-306, core/shape.cpp:72-87)
// ===========================================================================
struct LightSample {
    F3 p, perr, n;
};
DEV float sphere_area(const DSphere &sp) { return sp.phi_max * sp.radius * (sp.zmax - sp.zmin); }
DEV LightSample sphere_sample_area(const DSphere &sp, float u0, float u1, float *pdf) {
    float z = 1 - 2 * u0;  // UniformSampleSphere, sampling.cpp:98-103
    float r = sqrtf(mx(0.f, 1.f - z * z));
    float phi = 2 * kPi * u1;
    float s, c;
    sincos_f(phi, &s, &c);
    F3 us = F3{r * c, r * s, z};
    F3 pobj = F3{0, 0, 0} + sp.radius * us;
    LightSample it;
    it.n = normalize(xf_normal(sp.o2w_inv, pobj));
    if (sp.reverse_orientation) it.n = it.n * -1.f;
    float scale = sp.radius / length(pobj);
    pobj = F3{pobj.x * scale, pobj.y * scale, pobj.z * scale};
    F3 pobj_err = kGamma5 * vabs(pobj);
    it.p = xf_point_err2(sp.o2w, pobj, pobj_err, &it.perr);
    *pdf = 1 / sphere_area(sp);
    return it;
}
DEV LightSample sphere_sample(const DSphere &sp, const Isect &ref, float u0, float u1, float *pdf) {
    const F3 pc = F3{sp.center[0], sp.center[1], sp.center[2]};  // (*ObjectToWorld)(Point3f(0, 0, 0)), see DSphere
    F3 porigin = offset_ray_origin(ref.p, ref.perr, ref.n, pc - ref.p);
    if (length_sq(porigin - pc) <= sp.radius * sp.radius) {
        LightSample intr = sphere_sample_area(sp, u0, u1, pdf);
        F3 wi = intr.p - ref.p;
        if (length_sq(wi) == 0)
            *pdf = 0;
        else {
            wi = normalize(wi);
            *pdf *= length_sq(ref.p - intr.p) / absdot(intr.n, -wi);
        }
        if (is_inf(*pdf)) *pdf = 0.f;
        return intr;
    }
    F3 wc = normalize(pc - ref.p);
    F3 wcx, wcy;
    coordinate_system(wc, &wcx, &wcy);
    float sin_tmax2 = sp.radius * sp.radius / length_sq(ref.p - pc);
    float cos_tmax = sqrtf(mx(0.f, 1 - sin_tmax2));
    float cos_t = (1 - u0) + u0 * cos_tmax;
    float sin_t = sqrtf(mx(0.f, 1 - cos_t * cos_t));
    float phi = u1 * 2 * kPi;
    float dc = length(ref.p - pc);
    float ds = dc * cos_t - sqrtf(mx(0.f, sp.radius * sp.radius - dc * dc * sin_t * sin_t));
    float cos_a = (dc * dc + sp.radius * sp.radius - ds * ds) / (2 * dc * sp.radius);
    float sin_a = sqrtf(mx(0.f, 1 - cos_a * cos_a));
    float sphi, cphi;
    sincos_f(phi, &sphi, &cphi);
    // SphericalDirection(sinAlpha, cosAlpha, phi, -wcX, -wcY, -wc), geometry.h:1467-1472
    F3 nw = sin_a * cphi * (-wcx) + sin_a * sphi * (-wcy) + cos_a * (-wc);
    F3 pw = pc + sp.radius * nw;
    LightSample it;
    it.p = pw;
    it.perr = kGamma5 * vabs(pw);
    it.n = nw;
    if (sp.reverse_orientation) it.n = it.n * -1.f;
    *pdf = 1 / (2 * kPi * (1 - cos_tmax));
    return it;
}
DEV float sphere_pdf(const DSphere &sp, const Isect &ref, F3 wi) {
    const F3 pc = F3{sp.center[0], sp.center[1], sp.center[2]};  // (*ObjectToWorld)(Point3f(0, 0, 0)), see DSphere
    F3 porigin = offset_ray_origin(ref.p, ref.perr, ref.n, pc - ref.p);
    if (length_sq(porigin - pc) <= sp.radius * sp.radius) {
        // Shape::Pdf, shape.cpp:72-87 — the shape alone, not a scene ray
        F3 ro = offset_ray_origin(ref.p, ref.perr, ref.n, wi);
        float t;
        F3 od, ph;
        if (!sphere_test(sp, ro, wi, IILE_INF, &t, &od, &ph)) return 0;
        Isect li;
        sphere_interaction(sp, od, ph, &li);
        float pdf = length_sq(ref.p - li.p) / (absdot(li.n, -wi) * sphere_area(sp));
        if (is_inf(pdf)) pdf = 0.f;
        return pdf;
    }
    float sin_tmax2 = sp.radius * sp.radius / length_sq(ref.p - pc);
    float cos_tmax = sqrtf(mx(0.f, 1 - sin_tmax2));
    return 1 / (2 * kPi * (1 - cos_tmax));
}
// InfiniteAreaLight (lights/infinite.cpp:42-174), operation for operation as the oracle's inf_* functions: Lmap is
// a host-built pyramid among the textures (one texel without an environment map), the Distribution2D a table
// in HBM: per row {func[w], cdf[w + 1], funcInt}, then the marginal {func[h], cdf[h + 1], funcInt}.
DEV F3 inf_lookup(const DScene &S, const DLight &lt, float s_, float t_) {  // Lmap->Lookup(st) -> triangle(0, st), mipmap.h:233-262
    return tex_triangle(S, S.textures[lt.env_tex], 0, s_, t_);
}
DEV float dist1d_sample(const float *d, int n, float u, float *pdf, int *off) {  // Distribution1D::SampleContinuous, sampling.h:71-89
    const float *cdf = d + n;
    // FindInterval(n + 1, cdf[i] <= u), pbrt.h:399-412
    int first = 0, len = n + 1;
    while (len > 0) {
        const int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) {
            first = middle + 1;
            len -= half + 1;
        } else
            len = half;
    }
    int offset = first - 1;
    offset = offset < 0 ? 0 : (offset > n - 1 ? n - 1 : offset);
    if (off) *off = offset;
    const float lo = cdf[offset], hi = cdf[offset + 1];
    float du = u - lo;
    if ((hi - lo) > 0) du /= (hi - lo);
    const float func_int = d[2 * n + 1];
    *pdf = (func_int > 0) ? d[offset] / func_int : 0.f;
    return (float(offset) + du) / float(n);
}
DEV const float *inf_cond(const DScene &S, const DLight &lt, int v) { return S.env_dist + lt.dist_offset + (long long)(2 * lt.dist_w + 2) * v; }
DEV F3 inf_w2l(const DLight &lt, F3 w) {
    return F3{lt.w2l[0] * w.x + lt.w2l[1] * w.y + lt.w2l[2] * w.z, lt.w2l[3] * w.x + lt.w2l[4] * w.y + lt.w2l[5] * w.z,
              lt.w2l[6] * w.x + lt.w2l[7] * w.y + lt.w2l[8] * w.z};
}
DEV float spherical_theta(F3 v) { return acos_f(clampf(v.z, -1, 1)); }  // geometry.h:1474-1481
DEV float spherical_phi(F3 v) {
    const float p = atan2_f(v.y, v.x);
    return (p < 0) ? (p + 2 * kPi) : p;
}
DEV F3 inf_le(const DScene &S, const DLight &lt, F3 d) {  // InfiniteAreaLight::Le, infinite.cpp:99-104
    const F3 w = normalize(inf_w2l(lt, d));
    return inf_lookup(S, lt, spherical_phi(w) * kInv2Pi, spherical_theta(w) * kInvPi);
}
DEV F3 inf_sample_li(const DScene &S, const DLight &lt, F3 ref_p, float u0, float u1, F3 *wi, float *pdf, F3 *target) {  // :106-137
    float pdf0, pdf1;
    int v;
    const float d1 = dist1d_sample(inf_cond(S, lt, lt.dist_h), lt.dist_h, u1, &pdf1, &v);
    const float d0 = dist1d_sample(inf_cond(S, lt, v), lt.dist_w, u0, &pdf0, nullptr);
    const float map_pdf = pdf0 * pdf1;
    *pdf = 0;
    if (map_pdf == 0) return F3{0, 0, 0};
    const float theta = d1 * kPi, phi = d0 * 2 * kPi;
    float sin_theta, cos_theta, sin_phi, cos_phi;
    sincos_f(theta, &sin_theta, &cos_theta);
    sincos_f(phi, &sin_phi, &cos_phi);
    const F3 wl = F3{sin_theta * cos_phi, sin_theta * sin_phi, cos_theta};
    *wi = F3{lt.l2w[0] * wl.x + lt.l2w[1] * wl.y + lt.l2w[2] * wl.z, lt.l2w[3] * wl.x + lt.l2w[4] * wl.y + lt.l2w[5] * wl.z,
             lt.l2w[6] * wl.x + lt.l2w[7] * wl.y + lt.l2w[8] * wl.z};
    *pdf = map_pdf / (2 * kPi * kPi * sin_theta);
    if (sin_theta == 0) *pdf = 0;
    *target = ref_p + *wi * (2 * lt.world_radius);
    return inf_lookup(S, lt, d0, d1);
}
DEV float inf_pdf_li(const DScene &S, const DLight &lt, F3 w) {  // :139-148 with Distribution2D::Pdf, sampling.h:135-142
    const F3 wi = inf_w2l(lt, w);
    const float theta = spherical_theta(wi), phi = spherical_phi(wi);
    float sin_theta, cos_theta;
    sincos_f(theta, &sin_theta, &cos_theta);
    if (sin_theta == 0) return 0;
    const float p0 = phi * kInv2Pi, p1 = theta * kInvPi;
    int iu = int(p0 * float(lt.dist_w)), iv = int(p1 * float(lt.dist_h));
    iu = iu < 0 ? 0 : (iu > lt.dist_w - 1 ? lt.dist_w - 1 : iu);
    iv = iv < 0 ? 0 : (iv > lt.dist_h - 1 ? lt.dist_h - 1 : iv);
    const float func = inf_cond(S, lt, iv)[iu];
    return (func / inf_cond(S, lt, lt.dist_h)[2 * lt.dist_h + 1]) / (2 * kPi * kPi * sin_theta);
}

// Triangle emitter (shapes/triangle.cpp:546-579) through the generic Shape::Sample(ref, u) /
// Shape::Pdf(ref, wi) (core/shape.cpp:56-87), and the sphere / triangle dispatch of an area light
DEV float triangle_area(const DScene &S, int prim) {
    const float4 v0 = S.tri_verts[3 * size_t(prim)], v1 = S.tri_verts[3 * size_t(prim) + 1],
                 v2 = S.tri_verts[3 * size_t(prim) + 2];
    const F3 p0 = F3{v0.x, v0.y, v0.z}, p1 = F3{v1.x, v1.y, v1.z}, p2 = F3{v2.x, v2.y, v2.z};
    return float(0.5 * double(length(cross(p1 - p0, p2 - p0))));
}
DEV LightSample triangle_sample_area(const DScene &S, int prim, float u0, float u1, float *pdf) {
    const float su0 = sqrtf(u0);  // UniformSampleTriangle, sampling.cpp:154-157
    const float b0 = 1 - su0, b1 = u1 * su0;
    const float4 v0 = S.tri_verts[3 * size_t(prim)], v1 = S.tri_verts[3 * size_t(prim) + 1],
                 v2 = S.tri_verts[3 * size_t(prim) + 2];
    const F3 p0 = F3{v0.x, v0.y, v0.z}, p1 = F3{v1.x, v1.y, v1.z}, p2 = F3{v2.x, v2.y, v2.z};
    const uint32_t flags = f2b(v0.w);
    LightSample it;
    it.p = b0 * p0 + b1 * p1 + (1 - b0 - b1) * p2;
    it.n = normalize(cross(p1 - p0, p2 - p0));
    if (flags & 2u) {  // the mesh has normals
        const float4 a = S.tri_norms[3 * size_t(prim)], b = S.tri_norms[3 * size_t(prim) + 1],
                     c = S.tri_norms[3 * size_t(prim) + 2];
        const F3 ns = b0 * F3{a.x, a.y, a.z} + b1 * F3{b.x, b.y, b.z} + (1 - b0 - b1) * F3{c.x, c.y, c.z};
        it.n = faceforward(it.n, ns);
    } else if (flags & 8u)  // reverseOrientation ^ transformSwapsHandedness
        it.n = it.n * -1.f;
    const F3 abs_sum = vabs(b0 * p0) + vabs(b1 * p1) + vabs((1 - b0 - b1) * p2);
    it.perr = kGamma6 * abs_sum;
    *pdf = 1 / triangle_area(S, prim);
    return it;
}
DEV LightSample shape_sample(const DScene &S, const DLight &lt, const Isect &ref, float u0, float u1, float *pdf) {
    if (lt.type == kLightDiffuseArea) return sphere_sample(S.spheres[lt.sphere], ref, u0, u1, pdf);
    LightSample intr = triangle_sample_area(S, lt.prim, u0, u1, pdf);  // Shape::Sample(ref, u, pdf), shape.cpp:56-70
    F3 wi = intr.p - ref.p;
    if (length_sq(wi) == 0)
        *pdf = 0;
    else {
        wi = normalize(wi);
        *pdf *= length_sq(ref.p - intr.p) / absdot(intr.n, -wi);
        if (is_inf(*pdf)) *pdf = 0.f;
    }
    return intr;
}
// n_tests / n_hits: Triangle::Intersect counts its calls wherever they come from (stats of the
// instrumented kernels)
DEV float shape_pdf(const DScene &S, const DLight &lt, const Isect &ref, F3 wi, unsigned long long *n_tests,
                    unsigned long long *n_hits) {
    if (lt.type == kLightDiffuseArea) return sphere_pdf(S.spheres[lt.sphere], ref, wi);
    // Shape::Pdf(ref, wi), shape.cpp:72-87: intersect the shape alone
    const F3 o = offset_ray_origin(ref.p, ref.perr, ref.n, wi);
    const RayCtx rc = make_ray_ctx(o, wi);
    const int prim = lt.prim;
    const float4 v0 = S.tri_verts[3 * size_t(prim)], v1 = S.tri_verts[3 * size_t(prim) + 1],
                 v2 = S.tri_verts[3 * size_t(prim) + 2];
    const F3 p0 = F3{v0.x, v0.y, v0.z}, p1 = F3{v1.x, v1.y, v1.z}, p2 = F3{v2.x, v2.y, v2.z};
    float t, b0, b1, b2;
    ++*n_tests;
    if (!triangle_test(rc, IILE_INF, p0, p1, p2, &t, &b0, &b1, &b2)) return 0;
    ++*n_hits;
    Isect li;
    triangle_interaction(S, prim, f2b(v0.w), p0, p1, p2, wi, b0, b1, b2, &li);
    float pdf = length_sq(ref.p - li.p) / (absdot(li.n, -wi) * triangle_area(S, prim));
    if (is_inf(pdf)) pdf = 0.f;
    return pdf;
}
DEV float power_heuristic(float fpdf, float gpdf) {  // sampling.h:169-172 with nf = ng = 1
    float f = 1 * fpdf, g = 1 * gpdf;
    return (f * f) / (f * f + g * g);
}

}  // namespace iile
