// kernels.hip — hand-written gfx950 kernels of the wavefront path tracer.
//
// Pipeline per pass (one pass = owned tiles x a chunk of sample indices):
//   generate -> [ extend -> shade -> mis -> mis_lit -> shadow ] x (maxDepth + 1) -> film_accumulate
// and, after the last pass, film_resolve.
//
//   generate  HaltonSampler + PerspectiveCamera::GenerateRayDifferential; fills ray queue 0
//   extend    BVHAccel::Intersect over a ray queue (closest hit), LDS traversal stacks
//   shade     the body of PathIntegrator::Li for one bounce: interaction, Le, BSDF, light
//             sampling + BSDF sampling of EstimateDirect (emits one NEE record holding a
//             shadow ray and an MIS ray), next direction, Russian roulette; compacts the
//             surviving paths into the next ray queue with ballot + one atomic per wavefront
//   shadow    BVHAccel::IntersectP for the shadow ray of each NEE record
//   mis       BVHAccel::Intersect for the MIS ray of each NEE record
//   shadow    BVHAccel::IntersectP for its shadow ray; L += beta * Ld
//
// All kernels are persistent grid-stride loops that read their queue length
// from device memory, so a whole pass is enqueued without host synchronisation.
// 64-lane wavefronts throughout: ballots are 64-bit, lane = threadIdx.x & 63.
#include "kcommon.h"

namespace iile {

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_generate(DScene S, PassDesc P, PassBuffers B, int count_stats) {
    unsigned long long n_cam = 0;
    for (uint32_t base = blockIdx.x * kBlock; base < P.n_paths; base += gridDim.x * kBlock) {
        const uint32_t pid = base + threadIdx.x;
        bool valid = pid < P.n_paths;
        int px = 0, py = 0;
        uint32_t k = 0;
        if (valid) valid = path_pixel(S, P, pid, &px, &py, &k);
        F3 o = F3{0, 0, 0}, d = F3{0, 0, 1};
        float tmax = 0;
        if (valid) {
            // Sampler::GetCameraSample (sampler.cpp:46-52): dims 0,1 film, 2 time, 3,4 lens
            const uint32_t idx = sample_index(S, px, py, k);
            const float u0 = sample_dimension(S, idx, 0, px, py), u1 = sample_dimension(S, idx, 1, px, py);
            float l0 = 0, l1 = 0;
            if (S.lens_radius > 0) {
                l0 = sample_dimension(S, idx, 3);
                l1 = sample_dimension(S, idx, 4);
            }
            if (P.probe_mode) {
                const uint32_t probe = (pid / uint32_t(P.kc)) / (256u * uint32_t(P.probe_tiles));
                probe_ray(S, P.probe_cams[probe], float(px) + u0, float(py) + u1, &o, &d, &tmax);
                B.aux[pid] = make_float4(0, 0, 0, -1.f);  // no intersection: normal 0, NO_INTERSECTION_DISTANCE
            } else
                camera_ray(S, float(px) + u0, float(py) + u1, l0, l1, &o, &d, &tmax);
            if (!P.list_px && !P.probe_mode) flag_whole_film_position(B, pid, px, py, k, float(px) + u0, float(py) + u1, u0, u1);
            B.hindex[pid] = idx;
            B.L[pid] = make_float4(0, 0, 0, 0);
            if (B.nray_out) {
                B.nray_out[2 * pid] = 0;
                B.nray_out[2 * pid + 1] = 0;
            }
            ++n_cam;
        } else if (pid < P.n_paths) {
            B.L[pid] = make_float4(0, 0, 0, 0);
        }
        // queue 0 is dense (slot == pid); pixel slots outside the sample bounds are INVALID records
        if (pid < P.n_paths) {
            B.ray_o[0][pid] = make_float4(o.x, o.y, o.z, b2f(valid ? pid : kInvalid));
            B.ray_d[0][pid] = make_float4(d.x, d.y, d.z, tmax);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) B.counts[kCntRay] = P.n_paths;
    if (count_stats) flush_counter(&B.counters->camera_rays, n_cam);
}

// miss: a ray that left the scene at the first vertex or after a specular bounce picks up the
// infinite lights' radiance (path.cpp:91-99). Only launched for scenes that have one: a pass over
// the hit records of the bounce, so the traversal kernels stay as they are.
__global__ __launch_bounds__(kBlock) void k_miss(DScene S, PassBuffers B, int bounce) {
    const uint32_t count = B.counts[kCntRay + bounce];
    const float4 *ro = B.ray_o[bounce & 1], *rd = B.ray_d[bounce & 1];
    for (uint32_t slot = blockIdx.x * kBlock + threadIdx.x; slot < count; slot += gridDim.x * kBlock) {
        const float4 h4 = B.hits[slot];
        if (int(f2b(h4.x)) >= 0) continue;
        const uint32_t pid = f2b(ro[slot].w);
        if (pid == kInvalid) continue;
        const float4 d4 = rd[slot];
        // (path state rides with the ray from bounce 1 on: throughput in ray_s, specularBounce in the direction record's .w)
        const float4 beta4 = bounce == 0 ? make_float4(1, 1, 1, b2f(5u)) : B.ray_s[bounce & 1][slot];
        const uint32_t state_w = bounce == 0 ? 5u : f2b(d4.w);
        if (!((bounce == 0 && !S.probe_mode) || (bounce != 0 && (state_w >> 16) != 0))) continue;  // iispt_d.cpp:124-131
        const F3 beta = F3{beta4.x, beta4.y, beta4.z}, d = F3{d4.x, d4.y, d4.z};
        const float4 L4 = B.L[pid];
        F3 L = F3{L4.x, L4.y, L4.z};
        for (int l = 0; l < S.n_lights; ++l)
            if (S.lights[l].type == kLightInfinite) L = L + beta * inf_le(S, S.lights[l], d);
        B.L[pid] = make_float4(L.x, L.y, L.z, 0);
    }
}

// mis_lit: k_mis leaves, per record, the (area light index + 1) of the primitive its MIS ray
// ended on. This pass over that byte plane turns it into "the ray reached the *sampled* light,
// on its emitting side" (`lightIsect.primitive->GetAreaLight() == &light`, then
// SurfaceInteraction::Le -> DiffuseAreaLight::L; integrator.cpp:205-209). Only the rare records
// whose ray did end on an emitter are looked at any further.
__global__ __launch_bounds__(kBlock) void k_mis_lit(DScene S, PassBuffers B, int bounce, uint32_t plane) {
    const uint32_t count = B.counts[kCntMis + bounce];
    for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < count; q += gridDim.x * kBlock) {
        const float4 n2 = B.nee[2 * size_t(plane) + q];
        const uint32_t e = f2b(n2.w);  // the NEE record of this MIS ray
        if (e == kInvalid) continue;
        const uint32_t mis = B.nee_mis[e];
        if (mis == 0) continue;
        const float4 n3 = B.nee[3 * size_t(plane) + q];
        bool lit = false;
        {
            const int li = int(f2b(n3.w));
            if (mis == 255u) {
                lit = S.lights[li].type == kLightInfinite;  // `else Li = light.Le(ray)`, integrator.cpp:209-210
            } else if (int(mis) == li + 1) {
                const DLight &lt = S.lights[li];
                const F3 mo = F3{n2.x, n2.y, n2.z}, md = F3{n3.x, n3.y, n3.z};
                Isect lis;
                if (lt.type == kLightAreaTriangle) {
                    const float4 h4 = B.mis_hit[e];  // left by k_mis
                    const int prim = int(f2b(h4.x));
                    const float4 v0 = S.tri_verts[3 * size_t(prim)], v1 = S.tri_verts[3 * size_t(prim) + 1],
                                 v2 = S.tri_verts[3 * size_t(prim) + 2];
                    triangle_interaction(S, prim, f2b(v0.w), F3{v0.x, v0.y, v0.z}, F3{v1.x, v1.y, v1.z},
                                         F3{v2.x, v2.y, v2.z}, md, h4.y, h4.z, h4.w, &lis);
                } else {
                    const DSphere &sp = S.spheres[lt.sphere];
                    float th;
                    F3 od, ph;
                    // the closest hit was this sphere: redo its root selection for the hit point
                    sphere_test(sp, mo, md, IILE_INF, &th, &od, &ph);
                    sphere_interaction(sp, od, ph, &lis);
                }
                lit = lt.two_sided || dot(lis.n, -md) > 0;
            }
        }
        B.nee_mis[e] = lit ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------
// radiance guards of SamplerIntegrator::Render (integrator.cpp:293-314) and the
// luminance clamp of FilmTile::AddSample (film.h:157-158)
DEV F3 guard_radiance(const DScene &S, F3 L) {
    const float y = lum_y(L);
    if (is_nan(L.x) || is_nan(L.y) || is_nan(L.z))
        L = F3{0, 0, 0};
    else if (double(y) < -1e-5)
        L = F3{0, 0, 0};
    else if (is_inf(y))
        L = F3{0, 0, 0};
    const float y2 = lum_y(L);
    if (y2 > S.max_sample_luminance) L = L * (S.max_sample_luminance / y2);
    return L;
}

// film_accumulate: one thread per owned pixel adds this pass's samples, in
// sample order, to the pixel's RGB contribSum (FilmTile::AddSample, film.h:153-193
// with the box filter: weight 1 for the pixel containing pFilm).
__global__ __launch_bounds__(kBlock) void k_film_accumulate(DScene S, PassDesc P, PassBuffers B, FilmBuffers F) {
    const uint32_t n_pix = uint32_t(P.n_pass_tiles) * 256u;
    for (uint32_t lpt = blockIdx.x * kBlock + threadIdx.x; lpt < n_pix; lpt += gridDim.x * kBlock) {
        int px, py;
        uint32_t k;
        const uint32_t pid0 = lpt * uint32_t(P.kc);
        if (!path_pixel(S, P, pid0, &px, &py, &k)) continue;
        const uint32_t pt = uint32_t(P.slot0) * 256u + lpt;  // the pixel's slot among all owned tiles
        float4 acc = F.tile_rgbw[pt];
        // A lane's samples are consecutive in memory (1 KB apart from its neighbour's at 64 spp):
        // eight loads = one whole 128-byte line are issued together, then summed in sample order.
        for (int k8 = 0; k8 < P.kc; k8 += 8) {
            float4 Lb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) Lb[j] = B.L[pid0 + uint32_t(k8 + j < P.kc ? k8 + j : k8)];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = k8 + j;
                if (kk >= P.kc) break;
                const F3 L = guard_radiance(S, F3{Lb[j].x, Lb[j].y, Lb[j].z});
                acc.x += L.x;
                acc.y += L.y;
                acc.z += L.z;
                acc.w += 1.f;
                if (P.k0 + kk == 0) {
                    // a sample whose fractional film offset is exactly 0 also lands in the
                    // left / upper neighbour (support ceil(pd-.5) .. floor(pd+.5))
                    const uint32_t idx = sample_index(S, px, py, 0u);  // (= the index its camera ray was made with)
                    uint32_t mask = 0;
                    if (sample_dimension(S, idx, 0, px, py) == 0.f) mask |= 1u;
                    if (sample_dimension(S, idx, 1, px, py) == 0.f) mask |= 2u;
                    F.k0_rgbv[pt] = make_float4(L.x, L.y, L.z, b2f(mask));
                }
            }
        }
        F.tile_rgbw[pt] = acc;
    }
}

// Filters wider than a pixel (gaussian, mitchell, sinc, triangle, larger boxes). A pixel then sums samples of
// neighbouring pixels and tiles, in the order FilmTile::AddSample / MergeFilmTile give: inside a tile in pixel
// (row-major) then sample order, tiles in index order. The neighbouring pixels may belong to tiles of another pass, so
// the frame's samples are kept — k_film_store, 24 B each — and summed once at the end by a gather per film pixel
// (k_film_gather).
__global__ __launch_bounds__(kBlock) void k_film_store(DScene S, PassDesc P, PassBuffers B, FilmBuffers F, int k_begin, int n_samples) {
    for (uint32_t pid = blockIdx.x * kBlock + threadIdx.x; pid < P.n_paths; pid += gridDim.x * kBlock) {
        int px, py;
        uint32_t k;
        if (!path_pixel(S, P, pid, &px, &py, &k)) continue;
        const float4 L4 = B.L[pid];
        const F3 L = guard_radiance(S, F3{L4.x, L4.y, L4.z});
        const uint32_t idx = sample_index(S, px, py, k);  // (= the index its camera ray was made with)
        // [tile slot][k][pixel of the tile]: the gather's lanes (neighbouring film pixels) read neighbouring records
        const uint32_t pt = uint32_t(P.slot0) * 256u + pid / uint32_t(P.kc);
        const size_t at = (size_t(pt >> 8) * size_t(n_samples) + size_t(int(k) - k_begin)) * 256u + size_t(pt & 255u);
        F.wide_L[at] = make_float4(L.x, L.y, L.z, 0.f);
        F.wide_pf[at] = make_float2(float(px) + sample_dimension(S, idx, 0, px, py), float(py) + sample_dimension(S, idx, 1, px, py));
    }
}

DEV bool tile_owned(const PassDesc &P, int tx, int ty, uint32_t *slot) {
    if (tx < 0 || ty < 0 || tx >= P.n_tiles_x || ty >= P.n_tiles_y) return false;
    const int t = ty * P.n_tiles_x + tx;
    const int s = P.slot_of_tile ? P.slot_of_tile[t] : t;
    if (s < 0) return false;
    *slot = uint32_t(s);
    return true;
}
DEV void add_xyz(float4 *out, float r, float g, float b, float w) {  // RGBToXYZ, spectrum.h:62-66
    out->x += 0.412453f * r + 0.357580f * g + 0.180423f * b;
    out->y += 0.212671f * r + 0.715160f * g + 0.072169f * b;
    out->z += 0.019334f * r + 0.119193f * g + 0.950227f * b;
    out->w += w;
}

// Probe pass: film + aux images of every probe. Film::to_rgb_array (film.cpp:187-225) on the gathered {X, Y, Z, w},
// and the first hits' normals / distances picked from the paths (1 spp): outputs [probe][y][x].
__global__ __launch_bounds__(kBlock) void k_probe_finish(DScene S, PassDesc P, PassBuffers B, FilmBuffers F, int n_probes, float *intensity,
                                                         float *normals, float *distance) {
    const int fw = S.crop_x1 - S.crop_x0, fh = S.crop_y1 - S.crop_y0;
    const uint32_t per = uint32_t(fw) * uint32_t(fh), n = per * uint32_t(n_probes);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 px = F.film_xyzw[i];
        float rgb[3];
        rgb[0] = 3.240479f * px.x - 1.537150f * px.y - 0.498535f * px.z;  // XYZToRGB, spectrum.h:56-60
        rgb[1] = -0.969256f * px.x + 1.875991f * px.y + 0.041556f * px.z;
        rgb[2] = 0.055648f * px.x - 0.204043f * px.y + 1.057311f * px.z;
        if (px.w != 0) {
            const float inv_wt = 1.f / px.w;
            for (int c = 0; c < 3; ++c) rgb[c] = mx(0.f, rgb[c] * inv_wt);
        }
        for (int c = 0; c < 3; ++c) intensity[3 * size_t(i) + c] = (rgb[c] + 0.f) * 1.f;
        const uint32_t probe = i / per, loc = i % per;
        const int x = S.crop_x0 + int(loc % uint32_t(fw)), y = S.crop_y0 + int(loc / uint32_t(fw));
        const float4 a = B.aux[probe_record(S, P, probe, x, y) * size_t(P.kc)];
        normals[3 * size_t(i)] = a.x;
        normals[3 * size_t(i) + 1] = a.y;
        normals[3 * size_t(i) + 2] = a.z;
        distance[i] = a.w;
    }
}

// One film pixel of the gather: the sum over the tiles whose FilmTile holds (x, y), in tile index order, of the samples
// FilmTile::AddSample would have added to it, in pixel then sample order. rec1(qx, qy, row): the record {film position, radiance} of
// sample pixel (qx, qy) when there is one sample per pixel; the several-samples form reads F directly.
template <typename Rec1>
DEV float4 film_gather_pixel(const DScene &S, const PassDesc &P, const FilmBuffers &F, const float *s_table, int x, int y, int n_samples, Rec1 rec1) {
    const float rx = S.filter_rx, ry = S.filter_ry;
    const float inv_rx = 1 / rx, inv_ry = 1 / ry;  // Filter::invRadius
    // sample pixels that can reach (x, y): |q + u - 0.5 - x| <= r with u in [0, 1), i.e. x - r - 0.5 < q <= x + r + 0.5
    // (floor / ceil keep a pixel of slack on either side against the rounding of the sums in AddSample)
    // (a probe's RenderView takes no samples outside the film's pixel bounds: those records do not exist)
    // (the path pass takes none outside the integrator's pixel bounds — the sample bounds unless "pixelbounds" was given)
    const int lo_x = P.probe_mode ? S.crop_x0 : S.pb_x0, hi_x = P.probe_mode ? S.crop_x1 : S.pb_x1;
    const int lo_y = P.probe_mode ? S.crop_y0 : S.pb_y0, hi_y = P.probe_mode ? S.crop_y1 : S.pb_y1;
    const int qx0 = max(int(floorf(float(x) - rx - 0.5f)), lo_x), qx1 = min(int(ceilf(float(x) + rx + 0.5f)), hi_x - 1);
    const int qy0 = max(int(floorf(float(y) - ry - 0.5f)), lo_y), qy1 = min(int(ceilf(float(y) + ry + 0.5f)), hi_y - 1);
    float4 out = make_float4(0, 0, 0, 0);
    if (qx0 <= qx1 && qy0 <= qy1) {
        const int tx0 = (qx0 - S.samp_x0) / kTile, tx1 = (qx1 - S.samp_x0) / kTile;
        const int ty0 = (qy0 - S.samp_y0) / kTile, ty1 = (qy1 - S.samp_y0) / kTile;
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) {  // tile index order
                uint32_t slot = 0;  // (probe pass: the tiles here are the reference's, the records are found per pixel)
                if (!P.probe_mode && !tile_owned(P, tx, ty, &slot)) continue;
                // the tile's FilmTile (Film::GetFilmTile, film.cpp:92-103) must hold (x, y)
                const int sx0 = S.samp_x0 + tx * kTile, sy0 = S.samp_y0 + ty * kTile;
                const int sx1 = min(sx0 + kTile, S.samp_x1), sy1 = min(sy0 + kTile, S.samp_y1);
                const int fx0 = max(int(ceilf(float(sx0) - 0.5f - rx)), S.crop_x0), fx1 = min(int(floorf(float(sx1) - 0.5f + rx)) + 1, S.crop_x1);
                const int fy0 = max(int(ceilf(float(sy0) - 0.5f - ry)), S.crop_y0), fy1 = min(int(floorf(float(sy1) - 0.5f + ry)) + 1, S.crop_y1);
                if (x < fx0 || x >= fx1 || y < fy0 || y >= fy1) continue;
                float r = 0, g = 0, b = 0, w = 0;
                // FilmTile::AddSample's support test and table lookup for this pixel (film.h:159-188)
                auto add_sample = [&](float2 pf, float4 L) {
                    const float dxf = pf.x - 0.5f, dyf = pf.y - 0.5f;
                    const bool in = x >= max(int(ceilf(dxf - rx)), fx0) && x < min(int(floorf(dxf + rx)) + 1, fx1) &&
                                    y >= max(int(ceilf(dyf - ry)), fy0) && y < min(int(floorf(dyf + ry)) + 1, fy1);
                    if (in) {
                        const float ffx = fabsf((float(x) - dxf) * inv_rx * 16.f), ffy = fabsf((float(y) - dyf) * inv_ry * 16.f);
                        const int ifx = min(int(floorf(ffx)), 15), ify = min(int(floorf(ffy)), 15);
                        const float fwt = s_table[ify * 16 + ifx];
                        r += L.x * 1.f * fwt;
                        g += L.y * 1.f * fwt;
                        b += L.z * 1.f * fwt;
                        w += fwt;
                    }
                };
                const int xa = max(qx0, sx0), xb = min(qx1, sx1 - 1);
                for (int qy = max(qy0, sy0); qy <= min(qy1, sy1 - 1); ++qy) {
                    const size_t row = size_t(slot) * size_t(n_samples) * 256u + size_t((qy - sy0) * kTile - sx0);
                    if (n_samples == 1) {
                        // one sample per pixel (the probe pass): four neighbouring pixels' records in flight
                        for (int q4 = xa; q4 <= xb; q4 += 4) {
                            float2 pfb[4];
                            float4 Lb[4];
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) rec1(q4 + jj <= xb ? q4 + jj : q4, qy, row, &pfb[jj], &Lb[jj]);
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
                                if (q4 + jj <= xb) add_sample(pfb[jj], Lb[jj]);
                        }
                        continue;
                    }
                    for (int qx = xa; qx <= xb; ++qx) {
                        const size_t base = row + size_t(qx);
                        for (int k4 = 0; k4 < n_samples; k4 += 4) {
                            // four records in flight, then summed in sample order
                            float2 pfb[4];
                            float4 Lb[4];
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const size_t at = base + size_t(k4 + jj < n_samples ? k4 + jj : k4) * 256u;
                                pfb[jj] = F.wide_pf[at];
                                Lb[jj] = F.wide_L[at];
                            }
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
                                if (k4 + jj < n_samples) add_sample(pfb[jj], Lb[jj]);
                        }
                    }
                }
                add_xyz(&out, r, g, b, w);
            }
    }
    return out;
}

__global__ __launch_bounds__(kBlock) void k_film_gather(DScene S, PassDesc P, FilmBuffers F, int n_samples) {
    __shared__ float s_table[256];
    for (int j = threadIdx.x; j < 256; j += kBlock) s_table[j] = S.filter_table[j];
    __syncthreads();
    const int fw = S.crop_x1 - S.crop_x0, fh = S.crop_y1 - S.crop_y0;
    const uint32_t per = uint32_t(fw) * uint32_t(fh);
    const uint32_t n = per * (P.probe_mode ? uint32_t(P.n_owned_tiles / P.probe_tiles) : 1u);  // one film per probe
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t probe = i / per, loc = i % per;
        const int x = S.crop_x0 + int(loc % uint32_t(fw)), y = S.crop_y0 + int(loc / uint32_t(fw));
        F.film_xyzw[i] = film_gather_pixel(S, P, F, s_table, x, y, n_samples, [&](int qx, int qy, size_t row, float2 *pf, float4 *L) {
            const size_t at = P.probe_mode ? probe_record(S, P, probe, qx, qy) : row + size_t(qx);
            *pf = F.wide_pf[at];
            *L = F.wide_L[at];
        });
    }
}

// The probe pass's film in one kernel (k_film_store + k_film_gather + k_probe_finish for films of at most kProbeFilmMax pixels, the
// IISPT network's 32 x 32): a block takes a probe, keeps the film's samples — guarded radiance, film position — in LDS, and
// every thread gathers its pixels from there (each sample is read by up to 36 pixels) and writes the three images.
constexpr int kProbeFilmMax = 1024;
__global__ __launch_bounds__(kBlock) void k_probe_film(DScene S, PassDesc P, PassBuffers B, int n_probes, float *intensity, float *normals, float *distance) {
    __shared__ float s_table[256];
    __shared__ float2 s_pf[kProbeFilmMax];
    __shared__ float4 s_L[kProbeFilmMax];
    for (int j = threadIdx.x; j < 256; j += kBlock) s_table[j] = S.filter_table[j];
    const int fw = S.crop_x1 - S.crop_x0, fh = S.crop_y1 - S.crop_y0;
    const int per = fw * fh;
    FilmBuffers none = {};
    for (uint32_t probe = blockIdx.x; probe < uint32_t(n_probes); probe += gridDim.x) {
        __syncthreads();   // (the table; the previous probe's gathers)
        for (int j = threadIdx.x; j < per; j += kBlock) {   // k_film_store
            const int x = S.crop_x0 + j % fw, y = S.crop_y0 + j / fw;
            const float4 L4 = B.L[probe_record(S, P, probe, x, y)];
            const F3 L = guard_radiance(S, F3{L4.x, L4.y, L4.z});
            const uint32_t idx = sample_index(S, x, y, 0);
            s_L[j] = make_float4(L.x, L.y, L.z, 0.f);
            s_pf[j] = make_float2(float(x) + sample_dimension(S, idx, 0, x, y), float(y) + sample_dimension(S, idx, 1, x, y));
        }
        __syncthreads();
        for (int j = threadIdx.x; j < per; j += kBlock) {
            const int x = S.crop_x0 + j % fw, y = S.crop_y0 + j / fw;
            const float4 px = film_gather_pixel(S, P, none, s_table, x, y, 1, [&](int qx, int qy, size_t, float2 *pf, float4 *L) {
                const int at = (qy - S.crop_y0) * fw + (qx - S.crop_x0);
                *pf = s_pf[at];
                *L = s_L[at];
            });
            // k_probe_finish
            float rgb[3];
            rgb[0] = 3.240479f * px.x - 1.537150f * px.y - 0.498535f * px.z;  // XYZToRGB, spectrum.h:56-60
            rgb[1] = -0.969256f * px.x + 1.875991f * px.y + 0.041556f * px.z;
            rgb[2] = 0.055648f * px.x - 0.204043f * px.y + 1.057311f * px.z;
            if (px.w != 0) {
                const float inv_wt = 1.f / px.w;
                for (int c = 0; c < 3; ++c) rgb[c] = mx(0.f, rgb[c] * inv_wt);
            }
            const size_t i = size_t(probe) * size_t(per) + size_t(j);
            for (int c = 0; c < 3; ++c) intensity[3 * i + c] = (rgb[c] + 0.f) * 1.f;
            const float4 a = B.aux[probe_record(S, P, probe, x, y) * size_t(P.kc)];
            normals[3 * i] = a.x;
            normals[3 * i + 1] = a.y;
            normals[3 * i + 2] = a.z;
            distance[i] = a.w;
        }
    }
}

// film_resolve: Film::MergeFilmTile (film.cpp:135-148) as a gather. For film
// pixel Q the contributions are grouped by the tile whose FilmTile holds them
// (Q's own tile, then the tiles right / below / diagonal whose k=0 samples
// splat onto Q), each group summed in RGB in pixel order, converted to XYZ and
// added in tile-index order.
__global__ __launch_bounds__(kBlock) void k_film_resolve(DScene S, PassDesc P, FilmBuffers F) {
    const int fw = S.crop_x1 - S.crop_x0, fh = S.crop_y1 - S.crop_y0;
    const uint32_t n = uint32_t(fw) * uint32_t(fh);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int qx = S.crop_x0 + int(i % uint32_t(fw)), qy = S.crop_y0 + int(i / uint32_t(fw));
        float4 out = make_float4(0, 0, 0, 0);
        const int lx = qx - S.samp_x0, ly = qy - S.samp_y0;
        const bool q_in = lx >= 0 && ly >= 0 && qx < S.samp_x1 && qy < S.samp_y1;
        const int tx0 = lx >= 0 ? lx / kTile : -1, ty0 = ly >= 0 ? ly / kTile : -1;
        const int tx1 = (lx + 1) >= 0 ? (lx + 1) / kTile : -1, ty1 = (ly + 1) >= 0 ? (ly + 1) / kTile : -1;
        // sources: S1 = (qx+1, qy) needs mask bit0; S2 = (qx, qy+1) bit1; S3 = (qx+1, qy+1) both
        auto source = [&](int sx, int sy, int tx, int ty, uint32_t need, float *r, float *g, float *b, float *w) {
            if (sx >= S.samp_x1 || sy >= S.samp_y1 || sx < S.samp_x0 || sy < S.samp_y0) return;
            uint32_t slot;
            if (!tile_owned(P, tx, ty, &slot)) return;
            const uint32_t pix = uint32_t((sy - S.samp_y0 - ty * kTile) * kTile + (sx - S.samp_x0 - tx * kTile));
            const float4 v = F.k0_rgbv[slot * 256u + pix];
            if ((f2b(v.w) & need) != need) return;
            *r += v.x;
            *g += v.y;
            *b += v.z;
            *w += 1.f;
        };
        // group 0: Q's own tile
        {
            uint32_t slot;
            float r = 0, g = 0, b = 0, w = 0;
            bool any = false;
            if (q_in && tile_owned(P, tx0, ty0, &slot)) {
                const float4 own = F.tile_rgbw[slot * 256u + uint32_t((ly - ty0 * kTile) * kTile + (lx - tx0 * kTile))];
                r = own.x;
                g = own.y;
                b = own.z;
                w = own.w;
                any = true;
                if (tx1 == tx0) source(qx + 1, qy, tx0, ty0, 1u, &r, &g, &b, &w);
                if (ty1 == ty0) source(qx, qy + 1, tx0, ty0, 2u, &r, &g, &b, &w);
                if (tx1 == tx0 && ty1 == ty0) source(qx + 1, qy + 1, tx0, ty0, 3u, &r, &g, &b, &w);
            }
            if (any) add_xyz(&out, r, g, b, w);
        }
        // group 1: tile to the right in the same tile row
        if (tx1 != tx0 && ty0 >= 0) {
            float r = 0, g = 0, b = 0, w = 0;
            source(qx + 1, qy, tx1, ty0, 1u, &r, &g, &b, &w);
            if (ty1 == ty0) source(qx + 1, qy + 1, tx1, ty0, 3u, &r, &g, &b, &w);
            if (w > 0) add_xyz(&out, r, g, b, w);
        }
        // group 2: tile below in the same tile column
        if (ty1 != ty0 && tx0 >= 0) {
            float r = 0, g = 0, b = 0, w = 0;
            source(qx, qy + 1, tx0, ty1, 2u, &r, &g, &b, &w);
            if (tx1 == tx0) source(qx + 1, qy + 1, tx0, ty1, 3u, &r, &g, &b, &w);
            if (w > 0) add_xyz(&out, r, g, b, w);
        }
        // group 3: diagonal tile
        if (tx1 != tx0 && ty1 != ty0) {
            float r = 0, g = 0, b = 0, w = 0;
            source(qx + 1, qy + 1, tx1, ty1, 3u, &r, &g, &b, &w);
            if (w > 0) add_xyz(&out, r, g, b, w);
        }
        F.film_xyzw[i] = out;
    }
}

// ---------------------------------------------------------------------------
// kernel-level probes for parity tests
__global__ void k_halton(DScene S, int n, const int *px, const int *py, const int *k, int dim0, int ndims, float *out,
                         uint32_t *index_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t idx = sample_index(S, px[i], py[i], uint32_t(k[i]));
    if (index_out) index_out[i] = idx;
    for (int d = 0; d < ndims; ++d) out[size_t(i) * ndims + d] = sample_dimension(S, idx, dim0 + d, px[i], py[i]);
}
__global__ void k_camera(DScene S, int n, const float *pfilm, const float *plens, float *o, float *d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 ro, rd;
    float tm;
    camera_ray(S, pfilm[2 * i], pfilm[2 * i + 1], plens ? plens[2 * i] : 0.f, plens ? plens[2 * i + 1] : 0.f, &ro, &rd,
               &tm);
    o[3 * i] = ro.x;
    o[3 * i + 1] = ro.y;
    o[3 * i + 2] = ro.z;
    d[3 * i] = rd.x;
    d[3 * i + 1] = rd.y;
    d[3 * i + 2] = rd.z;
}
// BSDF in a canonical frame (ns = ng = +z, ss = +x). sample == 0: out = {f.xyz, pdf}
// for (wo, wi); sample == 1: out = {wi.xyz, f.xyz, pdf} for (wo, u).
__global__ void k_bsdf_probe(DScene S, int n, int mat, const float *wo, const float *wi_or_u, int sample, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Isect is;
    is.sn = F3{0, 0, 1};
    is.n = F3{0, 0, 1};
    is.sdpdu = F3{1, 0, 0};
    is.p = is.perr = is.wo = F3{0, 0, 0};
    Bsdf b = make_bsdf(S.materials[mat], is);
    b.ss = F3{1, 0, 0};
    b.ts = cross(b.ns, b.ss);
    const F3 w = F3{wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]};
    if (!sample) {
        const F3 wi = F3{wi_or_u[3 * i], wi_or_u[3 * i + 1], wi_or_u[3 * i + 2]};
        const F3 f = bsdf_f(b, w, wi);
        out[4 * i] = f.x;
        out[4 * i + 1] = f.y;
        out[4 * i + 2] = f.z;
        out[4 * i + 3] = bsdf_pdf(b, w, wi);
    } else {
        F3 wi = F3{0, 0, 0};
        float pdf = 0;
        const F3 f = bsdf_sample_f(b, w, &wi, wi_or_u[2 * i], wi_or_u[2 * i + 1], &pdf);
        out[7 * i] = wi.x;
        out[7 * i + 1] = wi.y;
        out[7 * i + 2] = wi.z;
        out[7 * i + 3] = f.x;
        out[7 * i + 4] = f.y;
        out[7 * i + 5] = f.z;
        out[7 * i + 6] = pdf;
    }
}
__global__ void k_trig_probe(int n, const float *x, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    sincos_f(x[i], &s, &c);
    out[3 * i] = s;
    out[3 * i + 1] = c;
    out[3 * i + 2] = acos_f(clampf(x[i], -1.f, 1.f));
}

// ---------------------------------------------------------------------------
// launchers
void launch_generate(const DScene &S, const PassDesc &P, const PassBuffers &B, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_generate, dim3(grid_blocks(P.n_paths, cfg.n_cus, 8)), dim3(kBlock), 0, cfg.stream, S, P, B,
                       cfg.count_stats ? 1 : 0);
}
void launch_miss(const DScene &S, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg) {
    const dim3 grid(grid_blocks(max_rays, cfg.n_cus, 8));
    hipLaunchKernelGGL(k_miss, grid, dim3(kBlock), 0, cfg.stream, S, B, bounce);
}
void launch_mis_lit(const DScene &S, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg) {
    const dim3 grid(grid_blocks(max_rays, cfg.n_cus, 8));
    hipLaunchKernelGGL(k_mis_lit, grid, dim3(kBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
}
void launch_film_accumulate(const DScene &S, const PassDesc &P, const PassBuffers &B, const FilmBuffers &F,
                            const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_film_accumulate, dim3(grid_blocks(uint32_t(P.n_pass_tiles) * 256u, cfg.n_cus, 8)),
                       dim3(kBlock), 0, cfg.stream, S, P, B, F);
}
void launch_film_store(const DScene &S, const PassDesc &P, const PassBuffers &B, const FilmBuffers &F, int k_begin, int n_samples,
                       const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_film_store, dim3(grid_blocks(P.n_paths, cfg.n_cus, 8)), dim3(kBlock), 0, cfg.stream, S, P, B, F, k_begin, n_samples);
}
void launch_probe_finish(const DScene &S, const PassDesc &P, const PassBuffers &B, const FilmBuffers &F, int n_probes,
                         float *intensity, float *normals, float *distance, const LaunchCfg &cfg) {
    const uint32_t n = uint32_t(S.crop_x1 - S.crop_x0) * uint32_t(S.crop_y1 - S.crop_y0) * uint32_t(n_probes);
    hipLaunchKernelGGL(k_probe_finish, dim3(grid_blocks(n, cfg.n_cus, 8)), dim3(kBlock), 0, cfg.stream, S, P, B, F, n_probes, intensity,
                       normals, distance);
}
bool launch_probe_film(const DScene &S, const PassDesc &P, const PassBuffers &B, int n_probes, float *intensity, float *normals, float *distance,
                       const LaunchCfg &cfg) {
    const uint32_t per = uint32_t(S.crop_x1 - S.crop_x0) * uint32_t(S.crop_y1 - S.crop_y0);
    if (per > uint32_t(kProbeFilmMax) || P.kc != 1) return false;   // (larger probe films: the three general kernels)
    hipLaunchKernelGGL(k_probe_film, dim3(unsigned(std::min(n_probes, cfg.n_cus * 8))), dim3(kBlock), 0, cfg.stream, S, P, B, n_probes, intensity, normals,
                       distance);
    return true;
}
void launch_film_gather(const DScene &S, const PassDesc &P, const FilmBuffers &F, int n_samples, const LaunchCfg &cfg) {
    const uint32_t n = uint32_t(S.crop_x1 - S.crop_x0) * uint32_t(S.crop_y1 - S.crop_y0) *
                       (P.probe_mode ? uint32_t(P.n_owned_tiles / P.probe_tiles) : 1u);
    hipLaunchKernelGGL(k_film_gather, dim3(grid_blocks(n, cfg.n_cus, 8)), dim3(kBlock), 0, cfg.stream, S, P, F, n_samples);
}
void launch_film_resolve(const DScene &S, const PassDesc &P, const FilmBuffers &F, const LaunchCfg &cfg) {
    const uint32_t n = uint32_t(S.crop_x1 - S.crop_x0) * uint32_t(S.crop_y1 - S.crop_y0);
    hipLaunchKernelGGL(k_film_resolve, dim3(grid_blocks(n, cfg.n_cus, 8)), dim3(kBlock), 0, cfg.stream, S, P, F);
}
void launch_halton(const DScene &S, int n, const int *px, const int *py, const int *k, int dim0, int ndims,
                   float *out, uint32_t *index_out, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_halton, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, S, n, px, py, k, dim0, ndims, out,
                       index_out);
}
void launch_camera(const DScene &S, int n, const float *pfilm, const float *plens, float *o, float *d,
                   const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_camera, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, S, n, pfilm, plens, o, d);
}
void launch_bsdf_probe(const DScene &S, int n, int mat, const float *wo, const float *wi_or_u, int sample,
                       float *out, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_bsdf_probe, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, S, n, mat, wo, wi_or_u, sample,
                       out);
}
// ImageTexture::Evaluate for given (u, v) and differentials (test probe)
__global__ void k_texture_probe(DScene S, int n, int tex, const float *uv, const float *duv, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TexDiff td = TexDiff{duv[4 * i], duv[4 * i + 1], duv[4 * i + 2], duv[4 * i + 3]};
    const F3 c = tex_evaluate(S, tex, uv[2 * i], uv[2 * i + 1], td);
    out[3 * i] = c.x;
    out[3 * i + 1] = c.y;
    out[3 * i + 2] = c.z;
}
void launch_texture_probe(const DScene &S, int n, int tex, const float *uv, const float *duv, float *out, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_texture_probe, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, S, n, tex, uv, duv, out);
}
__global__ void k_gather4(const float4 *src, const uint32_t *idx, int n, float4 *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
__global__ void k_scatter4(float4 *dst, const uint32_t *idx, int n, const float4 *in) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = in[i];
}
// One exact FilmTile sum per thread for a pixel that receives flagged samples from pixels generated before it in its
// own tile: those samples first, then its own kc samples, then the flagged samples of later pixels — all in generation
// order, guarded like k_film_accumulate (see patch_pass_finish in api.hip).
__global__ void k_patch_own(DScene S, const float4 *L, int n, const uint32_t *local_slot, const uint32_t *range3, const uint32_t *flag_pid,
                            int kc, float4 *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0, g = 0, b = 0, w = 0;
    auto add = [&](float4 v) {
        const F3 c = guard_radiance(S, F3{v.x, v.y, v.z});
        r += c.x * 1.f * 1.f;
        g += c.y * 1.f * 1.f;
        b += c.z * 1.f * 1.f;
        w += 1.f;
    };
    const uint32_t b0 = range3[3 * i], b1 = range3[3 * i + 1], b2 = range3[3 * i + 2];
    for (uint32_t h = b0; h < b1; ++h) add(L[flag_pid[h]]);
    const uint32_t first = local_slot[i] * uint32_t(kc);
    for (int k = 0; k < kc; ++k) add(L[first + uint32_t(k)]);
    for (uint32_t h = b1; h < b2; ++h) add(L[flag_pid[h]]);
    out[i] = make_float4(r, g, b, w);
}
// ---------------------------------------------------------------------------
// Exact film finish on the device. What patch_prepare / patch_pass_finish / patch_merge of api.hip do with the host in the
// loop (kept behind IILE_DEBUG_HOST_FILM_FINISH as the A/B witness), as four small kernels that never leave the stream:
//   k_patch_hits    every flagged sample (kcommon.h flag_whole_film_position) -> one hit record per OTHER pixel it lands in
//                   (FilmTile::AddSample's support, film.h:159-166, clipped to its tile's FilmTile bounds, film.cpp:92-103),
//                   linked into the list of its destination through a hash table over film indices
//   k_patch_dests   the thread that finds its hit at the head of a destination's list walks that list in GENERATION order
//                   (tile, pixel of the tile, sample: the order in which one thread of the reference would have added them)
//                   and forms, per contributing tile, the exact FilmTile sum for that pixel -> an entry
//   k_patch_index / k_patch_merge   after k_film_resolve: per film pixel with an entry that the resolve kernel cannot have
//                   placed, the tiles' sums (and the pixel's own tile sum if no entry covers it) converted to XYZ and added
//                   in tile index order (Film::MergeFilmTile, film.cpp:135-148)
// Lists are short (a sample lands in at most three other pixels, a pixel is reached by a handful): they are walked by repeated
// selection of the next key instead of being sorted, which needs no bound on their length.
namespace {
struct FlagRec {
    int px, py, k, tile, pix;
    float pfx, pfy;
    uint32_t pid;
    bool plain_k0;
};
DEV FlagRec flag_decode(const DScene &S, int ntx, const float *flag_rec, uint32_t i) {
    const float *r = flag_rec + 6 * size_t(i);
    const uint32_t u2 = f2b(r[2]);
    FlagRec f;
    f.px = int(f2b(r[0])), f.py = int(f2b(r[1])), f.k = int(u2 & 0x3fffffffu);
    f.pfx = r[3], f.pfy = r[4];
    f.pid = f2b(r[5]);
    const int tx = (f.px - S.samp_x0) / kTile, ty = (f.py - S.samp_y0) / kTile;
    f.pix = (f.py - S.samp_y0 - ty * kTile) * kTile + (f.px - S.samp_x0 - tx * kTile);
    f.tile = ty * ntx + tx;
    const bool zero_x = (u2 >> 30) & 1u, zero_y = (u2 >> 31) & 1u;
    const bool whole_x = f.pfx == float(f.px) || f.pfx == float(f.px + 1), whole_y = f.pfy == float(f.py) || f.pfy == float(f.py + 1);
    f.plain_k0 = f.k == 0 && (!whole_x || zero_x) && (!whole_y || zero_y);  // k_film_resolve places these by itself
    return f;
}
DEV unsigned long long gen_key(const FlagRec &f) {
    return (static_cast<unsigned long long>(uint32_t(f.tile)) << 40) | (static_cast<unsigned long long>(uint32_t(f.pix)) << 32) |
           static_cast<unsigned long long>(uint32_t(f.k));
}
DEV uint32_t patch_hash(uint32_t key) { return key * 2654435761u; }
// claim (or find) the table slot of `key`; kPatchNil when the table is full
DEV uint32_t patch_slot(const PatchDev &D, uint32_t key, bool insert) {
    uint32_t s = patch_hash(key) & D.table_mask;
    for (uint32_t probe = 0; probe <= D.table_mask; ++probe, s = (s + 1) & D.table_mask) {
        const uint32_t seen = insert ? atomicCAS(&D.keys[s], kPatchNil, key) : D.keys[s];
        if (seen == key || (insert && seen == kPatchNil)) return s;
        if (!insert && seen == kPatchNil) return kPatchNil;
    }
    return kPatchNil;
}
}  // namespace

__global__ __launch_bounds__(kBlock) void k_patch_hits(DScene S, int ntx, PatchDev D, const uint32_t *flag_count, const float *flag_rec) {
    uint32_t n_flag = *flag_count;
    if (n_flag > kMaxFlagged) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&D.counters[2], 1u);
        n_flag = kMaxFlagged;
    }
    const int fw = S.crop_x1 - S.crop_x0;
    const float r = 0.5f;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_flag; i += gridDim.x * kBlock) {
        const FlagRec f = flag_decode(S, ntx, flag_rec, i);
        const int tx = f.tile % ntx, ty = f.tile / ntx;
        const int sx0 = S.samp_x0 + tx * kTile, sy0 = S.samp_y0 + ty * kTile;
        const int sx1 = min(sx0 + kTile, S.samp_x1), sy1 = min(sy0 + kTile, S.samp_y1);
        // Film::GetFilmTile bounds of the sample's tile, film.cpp:92-103
        const int fx0 = max(int(ceilf(float(sx0) - 0.5f - r)), S.crop_x0), fx1 = min(int(floorf(float(sx1) - 0.5f + r)) + 1, S.crop_x1);
        const int fy0 = max(int(ceilf(float(sy0) - 0.5f - r)), S.crop_y0), fy1 = min(int(floorf(float(sy1) - 0.5f + r)) + 1, S.crop_y1);
        const float dxf = f.pfx - 0.5f, dyf = f.pfy - 0.5f;
        const int ax0 = max(int(ceilf(dxf - r)), fx0), ax1 = min(int(floorf(dxf + r)) + 1, fx1);
        const int ay0 = max(int(ceilf(dyf - r)), fy0), ay1 = min(int(floorf(dyf + r)) + 1, fy1);
        for (int y = ay0; y < ay1; ++y)
            for (int x = ax0; x < ax1; ++x) {
                if (x == f.px && y == f.py) continue;
                const uint32_t dest = uint32_t(y - S.crop_y0) * uint32_t(fw) + uint32_t(x - S.crop_x0);
                const uint32_t h = atomicAdd(&D.counters[0], 1u);
                if (h >= D.cap_hits) {
                    atomicOr(&D.counters[2], 1u);
                    continue;
                }
                const uint32_t s = patch_slot(D, dest, true);
                if (s == kPatchNil) {
                    atomicOr(&D.counters[2], 2u);
                    D.hits[h] = make_uint4(dest, i, kPatchNil, 1u);  // (never a list head: no thread processes it)
                    continue;
                }
                const uint32_t before = atomicExch(&D.heads[s], h);
                D.hits[h] = make_uint4(dest, i, before, 0u);
            }
    }
}

__global__ __launch_bounds__(kBlock) void k_patch_dests(DScene S, PassDesc P, PassBuffers B, FilmBuffers F, PatchDev D, int whole_frame) {
    const uint32_t n_hits = min(D.counters[0], D.cap_hits);
    const int ntx = P.n_tiles_x, fw = S.crop_x1 - S.crop_x0;
    for (uint32_t h0 = blockIdx.x * kBlock + threadIdx.x; h0 < n_hits; h0 += gridDim.x * kBlock) {
        const uint4 me = D.hits[h0];
        if (me.w != 0u) continue;
        const uint32_t slot_t = patch_slot(D, me.x, false);
        if (slot_t == kPatchNil || D.heads[slot_t] != h0) continue;  // one thread per destination: the head of its list
        // the destination pixel
        const uint32_t film_index = me.x;
        const int qx = S.crop_x0 + int(film_index % uint32_t(fw)), qy = S.crop_y0 + int(film_index / uint32_t(fw));
        const bool in_bounds = qx >= S.samp_x0 && qx < S.samp_x1 && qy >= S.samp_y0 && qy < S.samp_y1;
        int d_tile = -1, d_pix = 0;
        bool own_in_pass = false;
        uint32_t own_slot = 0;
        if (in_bounds) {
            const int tx = (qx - S.samp_x0) / kTile, ty = (qy - S.samp_y0) / kTile;
            d_pix = (qy - S.samp_y0 - ty * kTile) * kTile + (qx - S.samp_x0 - tx * kTile);
            d_tile = ty * ntx + tx;
            const int slot = P.slot_of_tile ? P.slot_of_tile[d_tile] : d_tile;
            if (slot >= 0) {
                own_in_pass = slot >= P.slot0 && slot < P.slot0 + P.n_pass_tiles;
                own_slot = uint32_t(slot) * 256u + uint32_t(d_pix);
            }
        }
        bool all_plain = true, need_own = false;
        for (uint32_t h = h0; h != kPatchNil; h = D.hits[h].z) {
            const FlagRec f = flag_decode(S, ntx, B.flag_rec, D.hits[h].y);
            all_plain = all_plain && f.plain_k0;
            if (own_in_pass && f.tile == d_tile && f.pix < d_pix) need_own = true;
        }
        // a destination outside the integrator's pixel bounds took no samples of its own ("pixelbounds"): nothing to put in order, and
        // its slot of tile_rgbw holds the zeros it was cleared to
        if (!(qx >= S.pb_x0 && qx < S.pb_x1 && qy >= S.pb_y0 && qy < S.pb_y1)) need_own = false;
        // the pass is the whole frame: a pixel reached only by samples k_film_resolve places itself needs nothing
        if (whole_frame && all_plain) continue;
        float rr = 0, gg = 0, bb = 0, ww = 0;
        auto add = [&](uint32_t pid) {
            const float4 v = B.L[pid];
            const F3 c = guard_radiance(S, F3{v.x, v.y, v.z});
            rr += c.x * 1.f * 1.f;
            gg += c.y * 1.f * 1.f;
            bb += c.z * 1.f * 1.f;
            ww += 1.f;
        };
        auto add_own_samples = [&]() {  // (need_own: the pixel's own samples between the earlier and the later pixels' ones)
            const uint32_t first = (own_slot - uint32_t(P.slot0) * 256u) * uint32_t(P.kc);
            for (int k = 0; k < P.kc; ++k) add(first + uint32_t(k));
        };
        int cur_tile = -1;
        bool nonplain = false, own_run = false, own_done = false;
        auto emit = [&]() {
            if (own_run && need_own && !own_done) add_own_samples();
            const uint32_t e = atomicAdd(&D.counters[1], 1u);
            if (e >= D.cap_entries) {
                atomicOr(&D.counters[2], 1u);
                return;
            }
            D.ent_a[e] = make_uint4(film_index, uint32_t(cur_tile), nonplain ? 1u : 0u, kPatchNil);
            D.ent_b[e] = make_float4(rr, gg, bb, ww);
        };
        unsigned long long prev = 0;
        bool have_prev = false;
        for (;;) {
            // the next hit in generation order
            uint32_t best = kPatchNil;
            unsigned long long best_key = ~0ull;
            for (uint32_t h = h0; h != kPatchNil; h = D.hits[h].z) {
                const unsigned long long key = gen_key(flag_decode(S, ntx, B.flag_rec, D.hits[h].y));
                if ((!have_prev || key > prev) && key < best_key) best = h, best_key = key;
            }
            if (best == kPatchNil) break;
            prev = best_key, have_prev = true;
            const FlagRec f = flag_decode(S, ntx, B.flag_rec, D.hits[best].y);
            if (f.tile != cur_tile) {
                if (cur_tile >= 0) emit();
                cur_tile = f.tile;
                rr = gg = bb = ww = 0.f;
                nonplain = false;
                own_run = own_in_pass && f.tile == d_tile;
                own_done = false;
                if (own_run && !need_own) {  // the finished sum of its own samples (k_film_accumulate), the later pixels' ones follow
                    const float4 own = F.tile_rgbw[own_slot];
                    rr = own.x, gg = own.y, bb = own.z, ww = own.w;
                }
            }
            if (own_run && need_own && !own_done && f.pix > d_pix) {
                add_own_samples();
                own_done = true;
            }
            add(f.pid);
            nonplain = nonplain || !f.plain_k0;
        }
        if (cur_tile >= 0) emit();
    }
}

__global__ __launch_bounds__(kBlock) void k_patch_index(PatchDev D) {
    const uint32_t n = min(D.counters[1], D.cap_entries);
    for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < n; e += gridDim.x * kBlock) {
        const uint32_t s = patch_slot(D, D.ent_a[e].x, true);
        if (s == kPatchNil) {
            atomicOr(&D.counters[2], 2u);
            continue;
        }
        D.ent_a[e].w = atomicExch(&D.heads[s], e);
    }
}

__global__ __launch_bounds__(kBlock) void k_patch_merge(DScene S, PassDesc P, FilmBuffers F, PatchDev D) {
    const uint32_t n = min(D.counters[1], D.cap_entries);
    const int ntx = P.n_tiles_x, fw = S.crop_x1 - S.crop_x0;
    for (uint32_t e0 = blockIdx.x * kBlock + threadIdx.x; e0 < n; e0 += gridDim.x * kBlock) {
        const uint32_t film_index = D.ent_a[e0].x;
        const uint32_t slot_t = patch_slot(D, film_index, false);
        if (slot_t == kPatchNil || D.heads[slot_t] != e0) continue;  // one thread per film pixel
        bool nonplain = false;
        for (uint32_t e = e0; e != kPatchNil; e = D.ent_a[e].w) nonplain = nonplain || D.ent_a[e].z != 0u;
        if (!nonplain) continue;  // k_film_resolve has placed everything that reaches this pixel
        // the pixel's own tile, if it is owned and no entry covers it: its finished sum takes its place in tile order
        const int qx = S.crop_x0 + int(film_index % uint32_t(fw)), qy = S.crop_y0 + int(film_index / uint32_t(fw));
        int own_tile = -1;
        float4 own = make_float4(0, 0, 0, 0);
        if (qx >= S.samp_x0 && qx < S.samp_x1 && qy >= S.samp_y0 && qy < S.samp_y1) {
            const int tx = (qx - S.samp_x0) / kTile, ty = (qy - S.samp_y0) / kTile, t = ty * ntx + tx;
            bool covered = false;
            for (uint32_t e = e0; e != kPatchNil; e = D.ent_a[e].w) covered = covered || int(D.ent_a[e].y) == t;
            const int slot = P.slot_of_tile ? P.slot_of_tile[t] : t;
            if (!covered && slot >= 0) {
                own_tile = t;
                own = F.tile_rgbw[uint32_t(slot) * 256u + uint32_t((qy - S.samp_y0 - ty * kTile) * kTile + (qx - S.samp_x0 - tx * kTile))];
            }
        }
        float4 o = make_float4(0, 0, 0, 0);
        bool own_added = own_tile < 0;
        int prev_tile = -1;
        for (;;) {
            uint32_t best = kPatchNil;
            int best_tile = 0x7fffffff;
            for (uint32_t e = e0; e != kPatchNil; e = D.ent_a[e].w) {
                const int t = int(D.ent_a[e].y);
                if (t > prev_tile && t < best_tile) best = e, best_tile = t;
            }
            if (best == kPatchNil) break;
            prev_tile = best_tile;
            if (!own_added && own_tile < best_tile) {
                add_xyz(&o, own.x, own.y, own.z, own.w);
                own_added = true;
            }
            const float4 v = D.ent_b[best];
            add_xyz(&o, v.x, v.y, v.z, v.w);
        }
        if (!own_added) add_xyz(&o, own.x, own.y, own.z, own.w);
        F.film_xyzw[film_index] = o;
    }
}

void launch_patch_pass(const DScene &S, const PassDesc &P, const PassBuffers &B, const FilmBuffers &F, const PatchDev &D, const LaunchCfg &cfg) {
    // (the table is cleared by the caller: two hipMemsetAsync of 0xff)
    const int whole_frame = P.n_pass_tiles == P.n_owned_tiles ? 1 : 0;
    hipLaunchKernelGGL(k_patch_hits, dim3(64), dim3(kBlock), 0, cfg.stream, S, P.n_tiles_x, D, B.flag_count, B.flag_rec);
    hipLaunchKernelGGL(k_patch_dests, dim3(128), dim3(kBlock), 0, cfg.stream, S, P, B, F, D, whole_frame);
}
void launch_patch_merge(const DScene &S, const PassDesc &P, const FilmBuffers &F, const PatchDev &D, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_patch_index, dim3(64), dim3(kBlock), 0, cfg.stream, D);
    hipLaunchKernelGGL(k_patch_merge, dim3(128), dim3(kBlock), 0, cfg.stream, S, P, F, D);
}

void launch_patch_own(const DScene &S, const float4 *L, int n, const uint32_t *local_slot, const uint32_t *range3, const uint32_t *flag_pid,
                      int kc, float4 *out, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_patch_own, dim3((n + 63) / 64), dim3(64), 0, cfg.stream, S, L, n, local_slot, range3, flag_pid, kc, out);
}
void launch_gather4(const float4 *src, const uint32_t *idx, int n, float4 *out, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_gather4, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, src, idx, n, out);
}
void launch_scatter4(float4 *dst, const uint32_t *idx, int n, const float4 *in, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_scatter4, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, dst, idx, n, in);
}
void launch_trig_probe(int n, const float *x, float *out, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_trig_probe, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, n, x, out);
}


}  // namespace iile
