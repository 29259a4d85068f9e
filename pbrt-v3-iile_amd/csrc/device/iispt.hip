// iispt.hip — the IISPT render runner's gather on the device (second half).
//
// What IisptRenderRunner::run does around the probe pass and the network (integrators/iisptrenderrunner.cpp:216-596):
//   k_iispt_hemi_points   the hemi points of a task: camera sample, find_intersection (:632-757), the aux ray the probe
//                         camera is placed on (:299-312) — the inputs of iile_render_probes
//   k_iispt_gather        the per-pixel loop (:414-596): camera sample, find_intersection, compute_fpixel_weights
//                         (:961-1039), sample_hemisphere (:142-178) / estimate_direct (:16-140) over the four neighbouring
//                         hemispheres the network predicted (read where the network left them in HBM), f_beta * L
// One thread per hemi point / film pixel: a task has 10^4..10^6 pixels with ~16 BSDF evaluations and one traversal
// each, so this pass is two orders of magnitude below the probe pass and the network in cost; it is written for
// exactness (bit for bit the oracle's restatement), not yet tuned.
// Random numbers: every film pixel draws from its own PCG32 stream RNG(rng_seed + pixel rank) (iile_iispt_task).
#include "dpath.h"
#include "kernels.h"

namespace iile {

namespace {
constexpr int kIisptBlock = 256;

// core/rng.h:62-156
struct Pcg {
    unsigned long long state, inc;
    DEV explicit Pcg(unsigned long long seq) {  // RNG(sequenceIndex) -> SetSequence
        state = 0u;
        inc = (seq << 1u) | 1u;
        uniform_u32();
        state += 0x853c49e6748fea9bULL;
        uniform_u32();
    }
    DEV uint32_t uniform_u32() {
        const unsigned long long oldstate = state;
        state = oldstate * 0x5851f42d4c957f2dULL + inc;
        const uint32_t xorshifted = uint32_t(((oldstate >> 18u) ^ oldstate) >> 27u);
        const uint32_t rot = uint32_t(oldstate >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    DEV uint32_t uniform_u32(uint32_t b) {
        const uint32_t threshold = (~b + 1u) % b;
        while (true) {
            const uint32_t r = uniform_u32();
            if (r >= threshold) return r % b;
        }
    }
    DEV float uniform_float() { return mn(kOneMinusEpsilon, float(uniform_u32()) * 0x1p-32f); }
};

struct FirstHit {
    bool found;  // find_intersection's return value
    Isect is;
    Bsdf bsdf;   // of the returned intersection (valid when found && beta != 0)
    F3 ray_d;    // direction of the ray that found it
    F3 beta;
};

// sampler_next_pixel + GetCameraSample + GenerateRayDifferential (iisptrenderrunner.cpp:262-272, 941-953) for the
// counter-th call, then find_intersection (:632-757): camera ray, specular chain, the first non-specular vertex
DEV FirstHit iispt_first_hit(const DScene &S, int fx, int fy, uint32_t counter, lds_int *my_stack, int *my_spill, uint32_t spill_stride) {
    FirstHit out;
    out.found = false;
    out.beta = F3{0, 0, 0};
    out.ray_d = F3{0, 0, 1};
    const int cpx = int(counter), cpy = 0;  // the sampler's pixel
    const uint32_t idx = sample_index(S, cpx, cpy, 0u);
    int dim = 0;
    const float u0 = sample_dimension(S, idx, 0, cpx, cpy), u1 = sample_dimension(S, idx, 1, cpx, cpy);
    float l0 = 0, l1 = 0;
    if (S.lens_radius > 0) {
        l0 = sample_dimension(S, idx, 3, cpx, cpy);
        l1 = sample_dimension(S, idx, 4, cpx, cpy);
    }
    dim = 5;
    const float pfx = float(fx) + u0, pfy = float(fy) + u1;
    F3 ro, rd;
    float tmax;
    camera_ray(S, pfx, pfy, l0, l1, &ro, &rd, &tmax);
    bool have_diff = S.textured_materials != 0;
    F3 beta = F3{1, 1, 1};
    TraceStats st = {0, 0, 0, 0};
    for (int bounces = 0; bounces < 24; ++bounces) {
        HitRec h;
        h.t = h.b0 = h.b1 = h.b2 = 0;
        h.prim = -1;
        if (!traverse<false, false, true>(S, ro, rd, tmax, my_stack, my_spill, spill_stride, &h, &st)) return out;  // no intersection
        const int prim = h.prim;
        const float4 v0 = S.tri_verts[3 * size_t(prim)], v1 = S.tri_verts[3 * size_t(prim) + 1], v2 = S.tri_verts[3 * size_t(prim) + 2];
        const uint32_t flags = f2b(v0.w);
        const int material = int(f2b(v1.w));
        Isect is;
        if (flags & 1u) {
            float t;
            F3 od, ph;
            const DSphere &sp = S.spheres[S.prim_shape[prim]];
            sphere_test(sp, ro, rd, IILE_INF, &t, &od, &ph);
            sphere_interaction(sp, od, ph, &is);
        } else {
            triangle_interaction(S, prim, flags, F3{v0.x, v0.y, v0.z}, F3{v1.x, v1.y, v1.z}, F3{v2.x, v2.y, v2.z}, rd, h.b0, h.b1, h.b2, &is);
        }
        if (material < 0) {  // `if (!isect.bsdf)`: skip this intersection
            ro = offset_ray_origin(is.p, is.perr, is.n, rd);
            tmax = IILE_INF;
            have_diff = false;
            continue;
        }
        Bsdf bsdf;
        const DMaterial &m0 = S.materials[material];
        if (S.textured_materials &&
            (m0.kd_tex >= 0 || m0.ks_tex >= 0 || m0.kr_tex >= 0 || m0.kt_tex >= 0 || m0.bump_tex >= 0 || m0.rough_tex >= 0 || m0.sigma_tex >= 0)) {
            TexDiff td = TexDiff{0, 0, 0, 0};
            if (have_diff) td = compute_differentials(is, camera_differentials(S, pfx, pfy, l0, l1, ro, rd));  // (S.diff_scale is 1 here)
            if (m0.bump_tex >= 0) bump(S, m0.bump_tex, td, &is);
            const DMaterial mm = textured_material(S, m0, is, td);
            bsdf = make_bsdf<true>(mm, is);
        } else {
            bsdf = make_bsdf<true>(m0, is);
        }
        have_diff = false;
        const float us0 = sample_dimension(S, idx, dim, cpx, cpy), us1 = sample_dimension(S, idx, dim + 1, cpx, cpy);
        dim += 2;
        F3 wi = F3{0, 0, 0};
        float pdf = 0;
        bool spec = false, trans = false;
        const F3 f = bsdf_sample_f(bsdf, -rd, &wi, us0, us1, &pdf, true, &spec, &trans);
        out.found = true;
        if (is_black(f) || pdf == 0.f) return out;  // beta 0
        if (!spec) {
            out.is = is;
            out.bsdf = bsdf;
            out.ray_d = rd;
            out.beta = beta;
            return out;
        }
        beta = beta * sdiv(f * absdot(wi, is.sn), pdf);
        const float by = lum_y(beta);
        if (by < 0.f || is_nan(by)) return out;
        ro = offset_ray_origin(is.p, is.perr, is.n, wi);
        rd = wi;
        tmax = IILE_INF;
    }
    out.found = true;  // max depth reached: 0 beta
    return out;
}
// the aux ray: isect.SpawnRay(surface normal turned against the ray)
DEV void iispt_aux_ray(const Isect &is, F3 ray_d, F3 *o, F3 *d) {
    F3 n = is.n;
    if (double(dot(is.n, ray_d)) > 0.0) n = -is.n;
    *o = offset_ray_origin(is.p, is.perr, is.n, n);
    *d = n;
}
}  // namespace

__global__ __launch_bounds__(kIisptBlock) void k_iispt_hemi_points(DScene S, iile_iispt_task T, int nx, int ny, uint8_t *valid, float *pos3, float *dir3,
                                                                  int *SPILL) {
    __shared__ int lds_stack[kIisptBlock / 64][2 * kLdsStackDepth][64];
    lds_int *my_stack = (lds_int *)&lds_stack[threadIdx.x >> 6][0][threadIdx.x & 63];
    const uint32_t spill_stride = gridDim.x * kIisptBlock;
    int *my_spill = SPILL + blockIdx.x * kIisptBlock + threadIdx.x;
    const int n = nx * ny;
    for (int k = blockIdx.x * kIisptBlock + threadIdx.x; k < n; k += gridDim.x * kIisptBlock) {
        const int i = k % nx, j = k / nx;
        const int tx = iile_iispt_grid_pos(T.x0, T.x1, T.tilesize, i), ty = iile_iispt_grid_pos(T.y0, T.y1, T.tilesize, j);
        const FirstHit fh = iispt_first_hit(S, tx, ty, T.counter_base + 1u + uint32_t(k), my_stack, my_spill, spill_stride);
        F3 o = F3{0, 0, 0}, d = F3{0, 0, 0};
        const bool ok = fh.found && double(lum_y(fh.beta)) > 0.0;  // else "set a black hemi"
        if (ok) iispt_aux_ray(fh.is, fh.ray_d, &o, &d);
        valid[k] = ok ? 1 : 0;
        pos3[3 * k] = o.x, pos3[3 * k + 1] = o.y, pos3[3 * k + 2] = o.z;
        dir3[3 * k] = d.x, dir3[3 * k + 1] = d.y, dir3[3 * k + 2] = d.z;
    }
}

namespace {
// IntensityFilm::get_camera_coord_jacobian (film/intensityfilm.cpp:60-66) on the predicted image of a hemi point:
// [y][x][3] with row 0 the top scanline (film->get(x, height - 1 - y)); jac[y] = sin(pi * y / hemi) (host table)
DEV F3 nn_pixel(const float *nn, int x, int y, int hemi, const float *jac) {
    const float *px = nn + 3 * (size_t(hemi - 1 - y) * hemi + x);
    const float j = jac[y];
    return F3{px[0] * j, px[1] * j, px[2] * j};
}
DEV F3 xf3(const float m[9], F3 v) { return F3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z}; }
// HemisphericCamera::get_light_sample_nn, hemispheric.cpp:89-105
DEV F3 light_sample_xy(const DHemiCam &hc, const float *nn, int x, int y, int hemi, const float *jac, F3 *wi) {
    const float theta = kPi * y / hemi;
    const float phi = kPi * x / hemi;
    float st, ct, sp, cp;
    sincos_f(theta, &st, &ct);
    sincos_f(phi, &sp, &cp);
    *wi = xf_vector(hc.c2w, F3{st * cp, ct, st * sp});
    return nn_pixel(nn, x, y, hemi, jac);
}
// HemisphericCamera::getLightSampleNn, hemispheric.cpp:44-60
DEV F3 light_sample_dir(const DHemiCam &hc, const float *nn, F3 wi, int hemi, const float *jac) {
    const F3 wc = xf3(hc.w2c, wi);
    const float theta = acos_f(wc.y);
    const float phi = atan2_f(wc.z, wc.x);
    if (is_nan(theta) || is_nan(phi)) return F3{0, 0, 0};  // (int)NaN is INT_MIN on x86: outside the film
    const int y = int(hemi * theta / kPi);
    const int x = int(hemi * phi / kPi);
    if (x >= 0 && x < hemi && y >= 0 && y < hemi) return nn_pixel(nn, x, y, hemi, jac);
    return F3{0, 0, 0};
}
// estimate_direct, iisptrenderrunner.cpp:16-140
DEV F3 estimate_direct_nn(const Isect &it, const Bsdf &bsdf, int rx, int ry, const DHemiCam &hc, const float *nn, int hemi, const float *jac, Pcg &rng) {
    F3 Ld = F3{0, 0, 0};
    F3 wi = F3{0, 0, 0};
    const float light_pdf = float(1.0 / 6.28);
    const float BSDF_RATIO = float(0.4394);
    const float EM_RATIO = float(1.098);
    float scattering_pdf = 0;
    const F3 Li = light_sample_xy(hc, nn, rx, ry, hemi, jac, &wi);
    if (light_pdf > 0 && !is_black(Li)) {
        const F3 f = bsdf_f(bsdf, it.wo, wi) * absdot(wi, it.sn);
        scattering_pdf = bsdf_pdf(bsdf, it.wo, wi);
        if (!is_black(f)) {
            const float weight = power_heuristic(light_pdf, scattering_pdf);
            Ld = Ld + sdiv(f * EM_RATIO * Li * weight, light_pdf);
        }
    }
    {
        float u[2];
        u[1] = rng.uniform_float();  // the constructor's arguments are evaluated right to left by g++
        u[0] = rng.uniform_float();
        F3 f = bsdf_sample_f(bsdf, it.wo, &wi, u[0], u[1], &scattering_pdf);
        f = f * absdot(wi, it.sn);
        if (!is_black(f) && scattering_pdf > 0) {
            const float weight = power_heuristic(scattering_pdf, light_pdf);
            const F3 Li2 = light_sample_dir(hc, nn, wi, hemi, jac);
            if (!is_black(Li2)) Ld = Ld + sdiv(f * BSDF_RATIO * Li2 * weight, scattering_pdf);
        }
    }
    return Ld;
}
}  // namespace

__global__ __launch_bounds__(kIisptBlock) void k_iispt_gather(DScene S, iile_iispt_task T, int nx, int ny, const DHemiCam *cams, const float *nn_films,
                                                             const float *jac, float4 *out, int *SPILL) {
    __shared__ int lds_stack[kIisptBlock / 64][2 * kLdsStackDepth][64];
    lds_int *my_stack = (lds_int *)&lds_stack[threadIdx.x >> 6][0][threadIdx.x & 63];
    const uint32_t spill_stride = gridDim.x * kIisptBlock;
    int *my_spill = SPILL + blockIdx.x * kIisptBlock + threadIdx.x;
    const int hemi = 32;  // PbrtOptions.iisptHemiSize (checked on the host)
    const int w = T.x1 - T.x0, n = w * (T.y1 - T.y0), ts = T.tilesize;
    // Camera::getCameraWorldPosition (camera.cpp:115-124): the origin of the ray through film point (0, 0), lens (0, 0)
    F3 main_o, main_d;
    float main_t;
    camera_ray(S, 0.f, 0.f, 0.f, 0.f, &main_o, &main_d, &main_t);
    for (int j = blockIdx.x * kIisptBlock + threadIdx.x; j < n; j += gridDim.x * kIisptBlock) {
        const int fx = T.x0 + j % w, fy = T.y0 + j / w;
        float4 res = make_float4(0, 0, 0, 0);
        const FirstHit fh = iispt_first_hit(S, fx, fy, T.counter_base + 1u + uint32_t(nx * ny) + uint32_t(j), my_stack, my_spill, spill_stride);
        if (fh.found && double(lum_y(fh.beta)) > 0.0) {
            // the four neighbouring hemi points: S top left, E bottom right, R top right, B bottom left (:424-468)
            const int mx = (fx - T.x0) % ts, my = (fy - T.y0) % ts;
            const int sx = fx - mx, sy = fy - my;
            const int ex = min(sx + ts, T.x1 - 1), ey = min(sy + ts, T.y1 - 1);
            const int neigh[4][2] = {{sx, sy}, {ex, ey}, {ex, sy}, {sx, ey}};
            int cam_index[4];
            for (int i = 0; i < 4; ++i) {
                const int gi = iile_iispt_grid_index(T.x0, T.x1, ts, neigh[i][0]), gj = iile_iispt_grid_index(T.y0, T.y1, ts, neigh[i][1]);
                const int c = gj * nx + gi;
                cam_index[i] = cams[c].valid ? c : -1;
            }
            // compute_fpixel_weights (:961-1039) with tools/iisptmathutils.h:44-130, 179-197
            F3 aux_o, aux_d;
            iispt_aux_ray(fh.is, fh.ray_d, &aux_o, &aux_d);
            float weights[4];
            float tot = 0.f;
            for (int i = 0; i < 4; ++i) {
                float dx2 = float(fx - neigh[i][0]);
                dx2 = dx2 * dx2;
                float dy2 = float(fy - neigh[i][1]);
                dy2 = dy2 * dy2;
                const float pdist = sqrtf(dx2 + dy2);
                const float res_p = pdist / float(ts);  // tilesize >= 1
                const float wdpos = double(res_p) < 0.0 ? 0.f : (double(res_p) > 1.0 ? 1.f : res_p);
                float wdnor = 0.f, wdd = 0.f;
                if (cam_index[i] >= 0) {
                    const DHemiCam &hc = cams[cam_index[i]];
                    F3 a = aux_d, b = F3{hc.look[0], hc.look[1], hc.look[2]};
                    const float al = length(a), bl = length(b);
                    if (double(al) <= 0.0 || double(bl) <= 0.0)
                        wdnor = 1.f;
                    else {
                        a = vdiv(a, al);
                        b = vdiv(b, bl);
                        const float dt = dot(a, b);
                        wdnor = double(dt) < 0.0 ? 1.f : 1.f - dt;
                    }
                    const float i2c = length(main_o - fh.is.p);
                    if (double(i2c) < 1e-10)
                        wdd = 0.f;
                    else {
                        const float s2c = length(main_o - F3{hc.origin[0], hc.origin[1], hc.origin[2]});
                        float rel = fabsf(i2c - s2c) / i2c;
                        rel *= 1.f;
                        const float wgt = 1.0f - rel;
                        wdd = wgt < 0.f ? 0.f : (wgt > 1.f ? 1.f : wgt);
                    }
                }
                const float wod = wdpos * wdnor + wdpos * wdd + wdpos;
                const double d2 = 2.0 - double(wod);
                weights[i] = float((d2 > 0.0 ? d2 : 0.0) + 0.001);
            }
            for (int i = 0; i < 4; ++i) tot += weights[i];
            if (double(tot) > 0.0)
                for (int i = 0; i < 4; ++i) weights[i] = weights[i] / tot;
            // sample_hemisphere (:142-178), HEMISPHERIC_IMPORTANCE_SAMPLES = 16
            Pcg rng(T.rng_seed + (unsigned long long)(j));
            F3 L = F3{0, 0, 0};
            int samples_taken = 0;
            for (int i = 0; i < 4; ++i)
                for (int s = 0; s < 16; ++s) {
                    const float rr = rng.uniform_float();
                    if (rr < weights[i]) {
                        samples_taken++;
                        if (cam_index[i] >= 0) {
                            const int rx = int(rng.uniform_u32(uint32_t(hemi)));
                            const int ry = int(rng.uniform_u32(uint32_t(hemi)));
                            L = L + estimate_direct_nn(fh.is, fh.bsdf, rx, ry, cams[cam_index[i]], nn_films + size_t(cam_index[i]) * hemi * hemi * 3, hemi,
                                                       jac, rng);
                        }
                    }
                }
            if (samples_taken > 0) {
                L = sdiv(L, float(samples_taken));
                const F3 v = fh.beta * L;
                res = make_float4(v.x, v.y, v.z, 0.5f);
            } else {
                res = make_float4(0.f, 0.f, 0.f, 0.5f);
            }
        }
        out[j] = res;
    }
}

void launch_iispt_hemi_points(const DScene &S, const iile_iispt_task &T, int nx, int ny, uint8_t *valid, float *pos3, float *dir3, int *spill,
                              const LaunchCfg &cfg) {
    const int n = nx * ny;
    const int blocks = std::max(1, std::min((n + kIisptBlock - 1) / kIisptBlock, cfg.n_cus * 4));
    hipLaunchKernelGGL(k_iispt_hemi_points, dim3(blocks), dim3(kIisptBlock), 0, cfg.stream, S, T, nx, ny, valid, pos3, dir3, spill);
}
void launch_iispt_gather(const DScene &S, const iile_iispt_task &T, int nx, int ny, const DHemiCam *cams, const float *nn_films, const float *jac,
                         float4 *out, int *spill, const LaunchCfg &cfg) {
    const int n = (T.x1 - T.x0) * (T.y1 - T.y0);
    const int blocks = std::max(1, std::min((n + kIisptBlock - 1) / kIisptBlock, cfg.n_cus * 4));
    hipLaunchKernelGGL(k_iispt_gather, dim3(blocks), dim3(kIisptBlock), 0, cfg.stream, S, T, nx, ny, cams, nn_films, jac, out,
                       spill);
}

}  // namespace iile
