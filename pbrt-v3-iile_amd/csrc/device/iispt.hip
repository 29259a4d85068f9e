// iispt.hip — the IISPT render runner's gather on the device (second half).
//
// What IisptRenderRunner::run does around the probe pass and the network (integrators/iisptrenderrunner.cpp:216-596),
// as a small wavefront pipeline over the items of a task (its hemi points, then its film pixels):
//   k_iispt_begin    sampler_next_pixel + GetCameraSample + GenerateRayDifferential (:262-272, 941-953) per item
//   k_iispt_trace    BVHAccel::Intersect for the items still looking for their vertex
//   k_iispt_vertex   one iteration of find_intersection's loop (:632-757): interaction, BSDF, Sample_f; the item ends
//                    (no surface / black / first non-specular vertex) or follows the specular bounce
//   k_iispt_hemi_out the aux ray a hemi point's probe camera is placed on (:299-312) — the inputs of iile_render_probes
//   k_iispt_gather   the per-pixel loop (:414-596): compute_fpixel_weights (:961-1039), sample_hemisphere (:142-178) /
//                    estimate_direct (:16-140) over the four neighbouring hemispheres the network predicted (read where
//                    the network left them in HBM), f_beta * L
// The host (api.hip) repeats trace + vertex until no item is left on a specular chain (at most 24 times, as the
// reference's loop). Each kernel is one thread per item and about the size of the kernel-level probes of kernels.hip:
// a first version that ran the whole loop, traversal included, inside one kernel (245 VGPRs, 170-200 spilled SGPRs)
// produced kernels that read outside their buffers or returned garbage for glossy hits depending on the optimisation
// level and on the heap layout of the process. This pass costs two orders of magnitude less than the probe pass and
// the network it sits between; it is written for exactness (bit for bit the oracle's restatement).
// Random numbers: every film pixel draws from its own PCG32 stream RNG(rng_seed + pixel rank) (iile_iispt_task).
#include "dpath.h"
#include "kernels.h"

namespace iile {

namespace {
constexpr int kIisptBlock = 256;
enum { kItemActive = 0, kItemNoSurface = 1, kItemBlack = 2, kItemVertex = 3 };

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


// item -> film pixel it looks through and its camera-sample counter (hemi points first, row by row, then film pixels)
DEV void item_pixel(const iile_iispt_task &T, int nx, int n_hemi, int item, int *fx, int *fy) {
    if (item < n_hemi) {
        *fx = iile_iispt_grid_pos(T.x0, T.x1, T.tilesize, item % nx);
        *fy = iile_iispt_grid_pos(T.y0, T.y1, T.tilesize, item / nx);
    } else {
        const int j = item - n_hemi, w = T.x1 - T.x0;
        *fx = T.x0 + j % w;
        *fy = T.y0 + j / w;
    }
}
// the vertex of a finished item again: SurfaceInteraction + BSDF from its hit record, exactly as k_iispt_vertex built
// them (ComputeScatteringFunctions is a pure function of the hit, the ray and — for textures — the camera sample)
DEV void vertex_state(const DScene &S, const float4 o4, const float4 d4, const float4 h4, const float4 pf4, bool camera_ray_hit, Isect *is,
                      Bsdf *bsdf, int *material_out) {
    const int prim = int(f2b(h4.x));
    const F3 ro = F3{o4.x, o4.y, o4.z}, rd = F3{d4.x, d4.y, d4.z};
    const float4 v0 = S.tri_verts[3 * size_t(prim)], v1 = S.tri_verts[3 * size_t(prim) + 1], v2 = S.tri_verts[3 * size_t(prim) + 2];
    const uint32_t flags = f2b(v0.w);
    const int material = int(f2b(v1.w));
    *material_out = material;
    if (flags & 1u) {
        float t;
        F3 od, ph;
        const DSphere &sp = S.spheres[S.prim_shape[prim]];
        sphere_test(sp, ro, rd, IILE_INF, &t, &od, &ph);
        sphere_interaction<true>(sp, od, ph, is);   // (with (u, v) and the derivatives: a textured or bump-mapped sphere looks them up below)
    } else {
        triangle_interaction(S, prim, flags, F3{v0.x, v0.y, v0.z}, F3{v1.x, v1.y, v1.z}, F3{v2.x, v2.y, v2.z}, rd, h4.y, h4.z, h4.w, is);
    }
    if (material < 0) return;
    const DMaterial &m0 = S.materials[material];
    if (S.textured_materials &&
        (m0.kd_tex >= 0 || m0.ks_tex >= 0 || m0.kr_tex >= 0 || m0.kt_tex >= 0 || m0.bump_tex >= 0 || m0.rough_tex >= 0 || m0.sigma_tex >= 0 || m0.opacity_tex >= 0 || m0.rough_tex_v >= 0)) {
        TexDiff td = TexDiff{0, 0, 0, 0};
        // only the camera ray carries differentials (r.ScaleDifferentials(1.0): S.diff_scale is 1 in these kernels)
        if (camera_ray_hit) td = compute_differentials(*is, camera_differentials(S, pf4.x, pf4.y, pf4.z, pf4.w, ro, rd));
        if (m0.bump_tex >= 0) bump(S, m0.bump_tex, td, is);
        const DMaterial mm = textured_material(S, m0, *is, td);
        *bsdf = make_bsdf<true>(mm, *is);
    } else {
        *bsdf = make_bsdf<true>(m0, *is);
    }
}
}  // namespace

// Per-item records (IisptItems, kernels.h; float4 planes of n_items each, api.hip allocates them):
//   ro = (ray o, tMax)   rd = (ray d, bitcast {state | bounce << 8 | sampler dimension << 16})   beta = (rgb, -)
//   hit = (bitcast prim, b0, b1, b2)   pf = (pFilm.xy, lens u)   idx[] = Halton index of the item's camera sample
// Every kernel runs all tasks of a batch at once: blockIdx.y is the task (its IisptJob is read from HBM, uniform loads),
// blockIdx.x strides over that task's items. A frame's tasks are 100 x 100 pixels: alone, one fills a sixth of the chip.
__global__ __launch_bounds__(kIisptBlock) void k_iispt_begin(DScene S, const IisptJob *jobs) {
    const iile_iispt_task T = jobs[blockIdx.y].T;
    const IisptItems I = jobs[blockIdx.y].I;
    for (int item = blockIdx.x * kIisptBlock + threadIdx.x; item < I.n_items; item += gridDim.x * kIisptBlock) {
        int fx, fy;
        item_pixel(T, I.nx, I.n_hemi, item, &fx, &fy);
        const int cpx = int(T.counter_base + 1u + uint32_t(item)), cpy = 0;  // the sampler's pixel for this call
        const uint32_t idx = sample_index(S, cpx, cpy, 0u);
        const float u0 = sample_dimension(S, idx, 0, cpx, cpy), u1 = sample_dimension(S, idx, 1, cpx, cpy);
        float l0 = 0, l1 = 0;
        if (S.lens_radius > 0) {
            l0 = sample_dimension(S, idx, 3, cpx, cpy);
            l1 = sample_dimension(S, idx, 4, cpx, cpy);
        }
        const float pfx = float(fx) + u0, pfy = float(fy) + u1;
        F3 ro, rd;
        float tmax;
        camera_ray(S, pfx, pfy, l0, l1, &ro, &rd, &tmax);
        I.ro[item] = make_float4(ro.x, ro.y, ro.z, tmax);
        I.rd[item] = make_float4(rd.x, rd.y, rd.z, b2f(uint32_t(kItemActive) | (0u << 8) | (5u << 16)));
        I.beta[item] = make_float4(1, 1, 1, 0);
        I.pf[item] = make_float4(pfx, pfy, l0, l1);
        I.hit[item] = make_float4(b2f(0xffffffffu), 0, 0, 0);
        I.idx[item] = idx;
    }
}

__global__ __launch_bounds__(kIisptBlock) void k_iispt_trace(DScene S, const IisptJob *jobs, int *SPILL) {
    __shared__ int lds_stack[kIisptBlock / 64][2 * kLdsStackDepth][64];
    const IisptItems I = jobs[blockIdx.y].I;
    lds_int *my_stack = (lds_int *)&lds_stack[threadIdx.x >> 6][0][threadIdx.x & 63];
    const uint32_t spill_stride = gridDim.x * gridDim.y * kIisptBlock;
    int *my_spill = SPILL + (blockIdx.y * gridDim.x + blockIdx.x) * kIisptBlock + threadIdx.x;
    TraceStats st = {0, 0, 0, 0};
    for (int item = blockIdx.x * kIisptBlock + threadIdx.x; item < I.n_items; item += gridDim.x * kIisptBlock) {
        const float4 o4 = I.ro[item], d4 = I.rd[item];
        if ((f2b(d4.w) & 0xffu) != uint32_t(kItemActive)) continue;
        HitRec h;
        h.t = h.b0 = h.b1 = h.b2 = 0;
        h.prim = -1;
        const bool found = traverse<false, false, true>(S, F3{o4.x, o4.y, o4.z}, F3{d4.x, d4.y, d4.z}, o4.w, my_stack, my_spill, spill_stride, &h, &st);
        I.hit[item] = make_float4(b2f(uint32_t(found ? h.prim : -1)), h.b0, h.b1, h.b2);
    }
}

// one iteration of find_intersection's loop for every active item
__global__ __launch_bounds__(kIisptBlock) void k_iispt_vertex(DScene S, const IisptJob *jobs) {
    const iile_iispt_task T = jobs[blockIdx.y].T;
    const IisptItems I = jobs[blockIdx.y].I;
    for (int item = blockIdx.x * kIisptBlock + threadIdx.x; item < I.n_items; item += gridDim.x * kIisptBlock) {
        const float4 o4 = I.ro[item], d4 = I.rd[item];
        const uint32_t word = f2b(d4.w);
        if ((word & 0xffu) != uint32_t(kItemActive)) continue;
        const int bounce = int((word >> 8) & 0xffu);
        int dim = int(word >> 16);
        const float4 h4 = I.hit[item];
        const F3 rd = F3{d4.x, d4.y, d4.z};
        uint32_t state = kItemActive;
        F3 new_o = F3{o4.x, o4.y, o4.z}, new_d = rd;
        float4 beta4 = I.beta[item];
        if (int(f2b(h4.x)) < 0) {
            state = kItemNoSurface;  // no intersection: find_intersection returns false
        } else {
            Isect is;
            Bsdf bsdf;
            int material;
            vertex_state(S, o4, d4, h4, I.pf[item], bounce == 0, &is, &bsdf, &material);
            if (material < 0) {  // `if (!isect.bsdf)`: skip this intersection
                new_o = offset_ray_origin(is.p, is.perr, is.n, rd);
            } else {
                const int cpx = int(T.counter_base + 1u + uint32_t(item)), cpy = 0;
                const uint32_t idx = I.idx[item];
                const float us0 = sample_dimension(S, idx, dim, cpx, cpy), us1 = sample_dimension(S, idx, dim + 1, cpx, cpy);
                dim += 2;
                F3 wi = F3{0, 0, 0};
                float pdf = 0;
                bool spec = false, trans = false;
                const F3 f = bsdf_sample_f(bsdf, -rd, &wi, us0, us1, &pdf, true, &spec, &trans);
                if (is_black(f) || pdf == 0.f) {
                    state = kItemBlack;
                } else if (!spec) {
                    state = kItemVertex;  // the first non-specular vertex: IISPT proceeds from here (ray, hit, beta stay)
                } else {
                    F3 beta = F3{beta4.x, beta4.y, beta4.z};
                    beta = beta * sdiv(f * absdot(wi, is.sn), pdf);
                    const float by = lum_y(beta);
                    beta4 = make_float4(beta.x, beta.y, beta.z, 0);
                    if (by < 0.f || is_nan(by)) {
                        state = kItemBlack;
                    } else {
                        new_o = offset_ray_origin(is.p, is.perr, is.n, wi);
                        new_d = wi;
                    }
                }
            }
        }
        int next_bounce = bounce;
        if (state == uint32_t(kItemActive)) {
            next_bounce = bounce + 1;
            if (next_bounce >= 24) state = kItemBlack;  // "max depth reached, return 0 beta"
        }
        if (state == uint32_t(kItemActive)) {
            I.ro[item] = make_float4(new_o.x, new_o.y, new_o.z, IILE_INF);
            I.beta[item] = beta4;
            atomicAdd(I.n_active, 1u);
        }
        if (state == uint32_t(kItemActive))
            I.rd[item] = make_float4(new_d.x, new_d.y, new_d.z, b2f(state | (uint32_t(next_bounce) << 8) | (uint32_t(dim) << 16)));
        else  // finished: the ray that found the vertex stays, with the bounce it was found at
            I.rd[item] = make_float4(d4.x, d4.y, d4.z, b2f(state | (uint32_t(bounce) << 8) | (uint32_t(dim) << 16)));
    }
}

namespace {
// a finished item has a vertex the runner works with: find_intersection returned true and beta.y() > 0
DEV bool item_has_vertex(const float4 d4, const float4 beta4) {
    return (f2b(d4.w) & 0xffu) == uint32_t(kItemVertex) && double(lum_y(F3{beta4.x, beta4.y, beta4.z})) > 0.0;
}
// the aux ray: isect.SpawnRay(surface normal turned against the ray)
DEV void iispt_aux_ray(const Isect &is, F3 ray_d, F3 *o, F3 *d) {
    F3 n = is.n;
    if (double(dot(is.n, ray_d)) > 0.0) n = -is.n;
    *o = offset_ray_origin(is.p, is.perr, is.n, n);
    *d = n;
}
}  // namespace

__global__ __launch_bounds__(kIisptBlock) void k_iispt_hemi_out(DScene S, const IisptJob *jobs) {
    const IisptItems I = jobs[blockIdx.y].I;
    uint8_t *valid = jobs[blockIdx.y].valid;
    float *pos3 = jobs[blockIdx.y].pos3, *dir3 = jobs[blockIdx.y].dir3;
    for (int k = blockIdx.x * kIisptBlock + threadIdx.x; k < I.n_hemi; k += gridDim.x * kIisptBlock) {
        const float4 o4 = I.ro[k], d4 = I.rd[k], beta4 = I.beta[k];
        F3 o = F3{0, 0, 0}, d = F3{0, 0, 0};
        const bool ok = item_has_vertex(d4, beta4);  // else "set a black hemi"
        if (ok) {
            Isect is;
            Bsdf bsdf;
            int material;
            vertex_state(S, o4, d4, I.hit[k], I.pf[k], ((f2b(d4.w) >> 8) & 0xffu) == 0u, &is, &bsdf, &material);
            iispt_aux_ray(is, F3{d4.x, d4.y, d4.z}, &o, &d);
        }
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

__global__ __launch_bounds__(kIisptBlock) void k_iispt_gather(DScene S, const IisptJob *jobs, const float *jac) {
    const iile_iispt_task T = jobs[blockIdx.y].T;
    const IisptItems I = jobs[blockIdx.y].I;
    const DHemiCam *cams = jobs[blockIdx.y].cams;
    const float *nn_films = jobs[blockIdx.y].nn_films;
    float4 *out = jobs[blockIdx.y].out;
    const int nx = I.nx;
    const int hemi = 32;  // PbrtOptions.iisptHemiSize (checked on the host)
    const int w = T.x1 - T.x0, n = w * (T.y1 - T.y0), ts = T.tilesize;
    // Camera::getCameraWorldPosition (camera.cpp:115-124): the origin of the ray through film point (0, 0), lens (0, 0)
    F3 main_o, main_d;
    float main_t;
    camera_ray(S, 0.f, 0.f, 0.f, 0.f, &main_o, &main_d, &main_t);
    for (int j = blockIdx.x * kIisptBlock + threadIdx.x; j < n; j += gridDim.x * kIisptBlock) {
        const int fx = T.x0 + j % w, fy = T.y0 + j / w;
        const int item = I.n_hemi + j;
        float4 res = make_float4(0, 0, 0, 0);
        const float4 o4 = I.ro[item], d4 = I.rd[item], beta4 = I.beta[item];
        if (item_has_vertex(d4, beta4)) {
            Isect f_is;
            Bsdf f_bsdf;
            int material;
            vertex_state(S, o4, d4, I.hit[item], I.pf[item], ((f2b(d4.w) >> 8) & 0xffu) == 0u, &f_is, &f_bsdf, &material);
            const F3 f_ray_d = F3{d4.x, d4.y, d4.z}, f_beta = F3{beta4.x, beta4.y, beta4.z};
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
            iispt_aux_ray(f_is, f_ray_d, &aux_o, &aux_d);
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
                    const float i2c = length(main_o - f_is.p);
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
            // `for (i < 4) for (s < 16) if (uniform_float() < weights[i]) { taken; if (camera i) L += estimate_direct(...) }`, the same
            // draws in the same order per pixel, arranged for 64 lanes: a lane runs ahead through the draws it does not take (a PCG
            // step and a comparison each) to its next estimate, and the wavefront evaluates estimates with every lane that still has
            // one — taking the estimate inside the double loop leaves three lanes in four idle (a quarter of the draws are taken).
            int c = 0;   // draws done: neighbour c >> 4, its sample c & 15
            while (true) {
                int ci = -1;
                while (c < 64) {
                    const int i = c >> 4;
                    ++c;
                    if (rng.uniform_float() < weights[i]) {
                        samples_taken++;
                        if (cam_index[i] >= 0) {
                            ci = cam_index[i];
                            break;
                        }
                    }
                }
                if (ci < 0) break;
                const int rx = int(rng.uniform_u32(uint32_t(hemi)));
                const int ry = int(rng.uniform_u32(uint32_t(hemi)));
                L = L + estimate_direct_nn(f_is, f_bsdf, rx, ry, cams[ci], nn_films + size_t(ci) * hemi * hemi * 3, hemi, jac, rng);
            }
            if (samples_taken > 0) {
                L = sdiv(L, float(samples_taken));
                const F3 v = f_beta * L;
                res = make_float4(v.x, v.y, v.z, 0.5f);
            } else {
                res = make_float4(0.f, 0.f, 0.f, 0.5f);
            }
        }
        out[j] = res;
    }
}

namespace {
// blocks per task: enough for its items, and no more blocks in all than the traversal stacks' spill columns were sized for
dim3 job_grid(int max_items, int n_jobs, const LaunchCfg &cfg) {
    const int budget = std::max(1, cfg.n_cus * 6 / std::max(n_jobs, 1));
    return dim3(unsigned(std::max(1, std::min((max_items + kIisptBlock - 1) / kIisptBlock, budget))), unsigned(n_jobs));
}
}  // namespace
void launch_iispt_first_hits(const DScene &S, const IisptJob *jobs, int n_jobs, int max_items, uint32_t *n_active, int *spill, const LaunchCfg &cfg) {
    const dim3 grid = job_grid(max_items, n_jobs, cfg);
    hipLaunchKernelGGL(k_iispt_begin, grid, dim3(kIisptBlock), 0, cfg.stream, S, jobs);
    // find_intersection's loop: trace, then one vertex step; again while some item (of any task) follows a specular bounce
    for (int bounce = 0; bounce < 24; ++bounce) {
        (void)hipMemsetAsync(n_active, 0, sizeof(uint32_t), cfg.stream);
        hipLaunchKernelGGL(k_iispt_trace, grid, dim3(kIisptBlock), 0, cfg.stream, S, jobs, spill);
        hipLaunchKernelGGL(k_iispt_vertex, grid, dim3(kIisptBlock), 0, cfg.stream, S, jobs);
        uint32_t host_active = 0;
        if (hipMemcpyAsync(&host_active, n_active, sizeof(uint32_t), hipMemcpyDeviceToHost, cfg.stream) != hipSuccess) return;
        if (hipStreamSynchronize(cfg.stream) != hipSuccess) return;
        if (host_active == 0) break;
    }
}
void launch_iispt_hemi_out(const DScene &S, const IisptJob *jobs, int n_jobs, int max_hemi, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_iispt_hemi_out, job_grid(max_hemi, n_jobs, cfg), dim3(kIisptBlock), 0, cfg.stream, S, jobs);
}
void launch_iispt_gather(const DScene &S, const IisptJob *jobs, int n_jobs, int max_pixels, const float *jac, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_iispt_gather, job_grid(max_pixels, n_jobs, cfg), dim3(kIisptBlock), 0, cfg.stream, S, jobs, jac);
}

// IisptFilmMonitor::add_n_samples (src/integrators/iisptfilmmonitor.cpp:47-72) for every pixel of every task of a batch: the
// gather's {f_beta * L, weight} per task pixel (task after task, row-major inside a task) added to the monitor's double sums.
// rects[t] = {x0, y0, x1, y1} in film pixels, first[t] = the task's first pixel in `out`. The tasks must not overlap (one sweep).
namespace {
__global__ __launch_bounds__(256) void k_iispt_film_add(const int4 *rects, const uint32_t *first, const float4 *out, double *film, int film_w) {
    const int4 r = rects[blockIdx.y];
    const int w = r.z - r.x, n = w * (r.w - r.y);
    const float4 *src = out + first[blockIdx.y];
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        const float4 v = src[p];
        double *d = film + 4 * (size_t(r.y + p / w) * film_w + size_t(r.x + p % w));
        d[0] += double(v.x);
        d[1] += double(v.y);
        d[2] += double(v.z);
        d[3] += double(v.w);
    }
}
// IisptFilmMonitor::merge_into + to_intensity_film (iisptfilmmonitor.cpp:231-275, iisptpixel.h): both monitors normalised
// (sums over the weight where it is positive), added, as float RGB
__global__ __launch_bounds__(256) void k_iispt_film_merge(const double *a, const double *b, float *rgb, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double pa[3] = {a[4 * i], a[4 * i + 1], a[4 * i + 2]}, pb[3] = {b[4 * i], b[4 * i + 1], b[4 * i + 2]};
    const double wa = a[4 * i + 3], wb = b[4 * i + 3];
    if (wa > 0.0) {
        pa[0] /= wa;
        pa[1] /= wa;
        pa[2] /= wa;
    }
    if (wb > 0.0) {
        pb[0] /= wb;
        pb[1] /= wb;
        pb[2] /= wb;
    }
    for (int c = 0; c < 3; ++c) rgb[3 * i + c] = float(pa[c] + pb[c]);   // (the merged pixel's weight is 1)
}
}  // namespace
void launch_iispt_film_add(const int4 *rects, const uint32_t *first, int n_tasks, int max_pixels, const float4 *out, double *film, int film_w, hipStream_t stream) {
    const unsigned gx = unsigned(std::max(1, std::min((max_pixels + 255) / 256, 64)));
    hipLaunchKernelGGL(k_iispt_film_add, dim3(gx, unsigned(n_tasks)), dim3(256), 0, stream, rects, first, out, film, film_w);
}
void launch_iispt_film_merge(const double *a, const double *b, float *rgb, long long n, hipStream_t stream) {
    hipLaunchKernelGGL(k_iispt_film_merge, dim3(unsigned((n + 255) / 256)), dim3(256), 0, stream, a, b, rgb, n);
}

}  // namespace iile
