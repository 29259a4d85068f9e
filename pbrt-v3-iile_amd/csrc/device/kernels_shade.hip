// kernels_shade.hip — k_shade: one bounce of PathIntegrator::Li for every hit of a bounce (pipeline overview in
// kernels.hip), and the tabulation of the spatial light distribution it samples from.
#include <cstdlib>

#include "kcommon.h"

namespace iile {

// ---------------------------------------------------------------------------
// DiffuseAreaLight::L (lights/diffuse.h:56-58)
DEV F3 light_L(const DLight &lt, F3 n, F3 w) {
    return (lt.two_sided || dot(n, w) > 0) ? F3{lt.lemit[0], lt.lemit[1], lt.lemit[2]} : F3{0, 0, 0};
}

// ---------------------------------------------------------------------------
// SpatialLightDistribution (core/lightdistrib.cpp:91-299): with more than one light the path
// integrator picks the light to sample from a per-voxel distribution. The reference fills a
// hash table lazily; a voxel's distribution is a pure function of its index, so all of them are
// tabulated once at scene creation (k_light_distributions) and looked up densely.
// Light::Sample_Li at an Interaction without normal or error bounds (lightdistrib.cpp:258-262)
DEV F3 sample_li_plain(const DScene &S, const DLight &lt, F3 po, float u0, float u1, float *pdf) {
    const F3 pos = F3{lt.pos[0], lt.pos[1], lt.pos[2]};
    const F3 I = F3{lt.lemit[0], lt.lemit[1], lt.lemit[2]};
    *pdf = 1;
    if (lt.type == kLightInfinite) {
        F3 wi, target;
        return inf_sample_li(S, lt, po, u0, u1, &wi, pdf, &target);
    }
    if (lt.type == kLightDistant) return I;
    if (lt.type == kLightPoint) return sdiv(I, length_sq(pos - po));
    if (lt.type == kLightSpot) {
        const F3 w = -normalize(pos - po);
        const F3 wl = normalize(F3{lt.w2l[0] * w.x + lt.w2l[1] * w.y + lt.w2l[2] * w.z,
                                   lt.w2l[3] * w.x + lt.w2l[4] * w.y + lt.w2l[5] * w.z,
                                   lt.w2l[6] * w.x + lt.w2l[7] * w.y + lt.w2l[8] * w.z});
        const float cos_theta = wl.z;
        float falloff;
        if (cos_theta < lt.cos_total_width)
            falloff = 0;
        else if (cos_theta >= lt.cos_falloff_start)
            falloff = 1;
        else {
            const float delta = (cos_theta - lt.cos_total_width) / (lt.cos_falloff_start - lt.cos_total_width);
            falloff = (delta * delta) * (delta * delta);
        }
        return sdiv(I * falloff, length_sq(pos - po));
    }
    Isect ref;  // DiffuseAreaLight::Sample_Li, lights/diffuse.cpp:68-81
    ref.p = po;
    ref.perr = F3{0, 0, 0};
    ref.n = F3{0, 0, 0};
    const LightSample ps = shape_sample(S, lt, ref, u0, u1, pdf);
    if (*pdf == 0 || length_sq(ps.p - po) == 0) {
        *pdf = 0;
        return F3{0, 0, 0};
    }
    const F3 wi = normalize(ps.p - po);
    return light_L(lt, ps.n, -wi);
}
DEV float lerp_f(float t, float a, float b) { return (1 - t) * a + t * b; }  // pbrt.h:414
// SpatialLightDistribution::ComputeDistribution (lightdistrib.cpp:228-299), one thread per voxel.
// samples: RadicalInverse(0..4, i) for i < 128 (host table)
__global__ void k_light_distributions(DScene S, const float *samples, float *out) {
    const int nv0 = S.light_nv[0], nv1 = S.light_nv[1], nv2 = S.light_nv[2];
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv0 * nv1 * nv2) return;
    const int pi2 = v % nv2, pi1 = (v / nv2) % nv1, pi0 = v / (nv2 * nv1);
    const F3 bmin = F3{S.root_box[0], S.root_box[1], S.root_box[2]}, bmax = F3{S.root_box[3], S.root_box[4], S.root_box[5]};
    const F3 p0 = F3{float(pi0) / float(nv0), float(pi1) / float(nv1), float(pi2) / float(nv2)};
    const F3 p1 = F3{float(pi0 + 1) / float(nv0), float(pi1 + 1) / float(nv1), float(pi2 + 1) / float(nv2)};
    const F3 vmin = F3{lerp_f(p0.x, bmin.x, bmax.x), lerp_f(p0.y, bmin.y, bmax.y), lerp_f(p0.z, bmin.z, bmax.z)};
    const F3 vmax = F3{lerp_f(p1.x, bmin.x, bmax.x), lerp_f(p1.y, bmin.y, bmax.y), lerp_f(p1.z, bmin.z, bmax.z)};
    const int n = S.n_lights;
    float contrib[kMaxLights];
#pragma unroll
    for (int j = 0; j < kMaxLights; ++j) contrib[j] = 0;
    for (int i = 0; i < 128; ++i) {
        const float *t = samples + 5 * i;
        const F3 po = F3{lerp_f(t[0], vmin.x, vmax.x), lerp_f(t[1], vmin.y, vmax.y), lerp_f(t[2], vmin.z, vmax.z)};
#pragma unroll
        for (int j = 0; j < kMaxLights; ++j) {
            if (j < n) {
                float pdf;
                const F3 Li = sample_li_plain(S, S.lights[j], po, t[3], t[4], &pdf);
                if (pdf > 0) contrib[j] += lum_y(Li) / pdf;
            }
        }
    }
    float sum = 0;
#pragma unroll
    for (int j = 0; j < kMaxLights; ++j)
        if (j < n) sum = sum + contrib[j];
    const float avg = sum / float(128 * n);
    const float min_contrib = (avg > 0) ? float(.001 * double(avg)) : 1.f;
    float *d = out + size_t(v) * kLightDistStride;
    float cdf = 0;
    d[kMaxLights] = 0;
#pragma unroll
    for (int j = 0; j < kMaxLights; ++j) {
        if (j < n) {
            const float f = mx(contrib[j], min_contrib);
            d[j] = f;
            cdf = cdf + f / float(n);
            d[kMaxLights + 1 + j] = cdf;
        }
    }
    const float func_int = cdf;
    d[2 * kMaxLights + 1] = func_int;
    for (int i = 1; i < n + 1; ++i) {
        if (func_int == 0)
            d[kMaxLights + i] = float(i) / float(n);
        else
            d[kMaxLights + i] = d[kMaxLights + i] / func_int;
    }
}
// SpatialLightDistribution::Lookup + Distribution1D::SampleDiscrete (sampling.h:90-100, FindInterval pbrt.h:399-412)
DEV int sample_light(const DScene &S, F3 p, float u, float *pdf) {
    const F3 bmin = F3{S.root_box[0], S.root_box[1], S.root_box[2]}, bmax = F3{S.root_box[3], S.root_box[4], S.root_box[5]};
    F3 o = p - bmin;  // Bounds3::Offset, geometry.h:800-806
    if (bmax.x > bmin.x) o.x = o.x / (bmax.x - bmin.x);
    if (bmax.y > bmin.y) o.y = o.y / (bmax.y - bmin.y);
    if (bmax.z > bmin.z) o.z = o.z / (bmax.z - bmin.z);
    int pi0 = int(o.x * float(S.light_nv[0])), pi1 = int(o.y * float(S.light_nv[1])), pi2 = int(o.z * float(S.light_nv[2]));
    pi0 = pi0 < 0 ? 0 : (pi0 > S.light_nv[0] - 1 ? S.light_nv[0] - 1 : pi0);
    pi1 = pi1 < 0 ? 0 : (pi1 > S.light_nv[1] - 1 ? S.light_nv[1] - 1 : pi1);
    pi2 = pi2 < 0 ? 0 : (pi2 > S.light_nv[2] - 1 ? S.light_nv[2] - 1 : pi2);
    const float *d = S.light_dist + size_t((pi0 * S.light_nv[1] + pi1) * S.light_nv[2] + pi2) * kLightDistStride;
    const int n = S.n_lights, size = n + 1;
    int first = 0, len = size;
    while (len > 0) {
        const int half = len >> 1, middle = first + half;
        if (d[kMaxLights + middle] <= u) {
            first = middle + 1;
            len -= half + 1;
        } else
            len = half;
    }
    int offset = first - 1;
    offset = offset < 0 ? 0 : (offset > size - 2 ? size - 2 : offset);
    const float func_int = d[2 * kMaxLights + 1];
    *pdf = (func_int > 0) ? d[offset] / (func_int * float(n)) : 0.f;
    return offset;
}

// shade: one bounce of PathIntegrator::Li (path.cpp:81-191) for every hit of the
// queue that extend just resolved.
// Waves per SIMD = resident blocks per CU. The plain and the extended builds run at 4 (<= 128 VGPRs, ~10 spilled to scratch
// outside the hot sections: killeroo 18.3 -> 17.2 ms, the closed room 53.5 -> 50.1 ms against 3 waves, profiles/r03_ab_shade_4_waves.txt
// — at the 168 VGPRs the kernel wanted before its uniform table reads became scalar loads, 4 waves lost: 23.6 vs 22.0 ms);
// the textured build is better off at 3 (<= 168 VGPRs; 50.2 vs 52.8 ms at 4).
// TEX: some material takes a parameter from an image texture (implies EXT)
#ifndef IILE_SHADE_WAVES
#define IILE_SHADE_WAVES 4
#endif
#ifndef IILE_SHADE_WAVES_TEX
#define IILE_SHADE_WAVES_TEX 3
#endif
constexpr int shade_waves(bool tex) { return tex ? IILE_SHADE_WAVES_TEX : IILE_SHADE_WAVES; }
template <bool COUNT, bool EXT, bool TEX>
__global__ __launch_bounds__(kBlock, shade_waves(TEX)) void k_shade(DScene S, PassDesc P, PassBuffers B, int bounce, uint32_t plane) {
    // digit permutations of the Halton sampler staged in LDS (dynamic shared memory)
    extern __shared__ __attribute__((aligned(16))) uint16_t s_perms_raw[];
    if (S.sobol) {
        // SobolSampler: four byte-table lookups per dimension, read through the caches (DScene::sobol_bt) — nothing staged
    } else {
        for (int i = threadIdx.x; i < S.n_perms; i += kBlock) s_perms_raw[i] = S.perms[i];
    }
    __syncthreads();
    lds_u16 *const s_perms = (lds_u16 *)s_perms_raw;
    const uint32_t count = B.counts[kCntShade + bounce];
    const float4 *ro = B.ray_o[bounce & 1], *rd = B.ray_d[bounce & 1];
    float4 *no = B.ray_o[(bounce + 1) & 1], *nd = B.ray_d[(bounce + 1) & 1];
    const float4 *rs = B.ray_s[bounce & 1];
    float4 *ns = B.ray_s[(bounce + 1) & 1];
    unsigned long long n_nee = 0, n_term = 0, n_pdf_tests = 0, n_pdf_hits = 0;
#ifdef IILE_SHADE_STAMPS
    // diagnostic build only: wave cycles (s_memtime) per section of a round, summed per wavefront, added to
    // DCounters::path_length[0..7] at the end: 0 regroup, 1 loads + Halton, 2 interaction + BSDF, 3 light half, 4 BSDF half,
    // 5 record stores, 6 continuation, 7 next-ray store
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#define SHADE_STAMP(i)                                                  \
    do {                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        stamp_sum[i] += now_ - stamp_t;                                 \
        stamp_t = now_;                                                 \
    } while (0)
#else
#define SHADE_STAMP(i) \
    do {               \
    } while (0)
#endif
    WaveOut ray_out{0, 0}, nee_out{0, 0}, mis_out{0, 0};
    auto pad_ray = [&](uint32_t sl) { no[sl] = make_float4(0, 0, 0, b2f(kInvalid)); };
    auto pad_nee = [&](uint32_t sl) { B.nee[plane + sl] = B.nee[4 * plane + sl] = make_float4(0, 0, 0, b2f(kInvalid)); };
    auto pad_mis = [&](uint32_t sl) { B.nee[2 * plane + sl] = make_float4(0, 0, 0, b2f(kInvalid)); };
    __shared__ uint32_t s_entry[kWavesPerBlock][kShadeChunk];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *const head = &B.counts[kCntShdHead + bounce];
    for (;;) {
        // chunks are drawn dynamically: a chunk of glossy hits costs several matte ones
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(head, uint32_t(kShadeChunk));
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= count) break;
        // Each wavefront takes kShadeChunk consecutive hits and regroups them by shading class
        // (material type, sphere) so that its rounds below run one code path each: past the
        // first bounce neighbouring queue entries hit unrelated materials (VALU lane
        // utilisation 39% -> 60% at bounce 1). The hits stay inside their chunk, so the queues
        // written here keep their locality. Wave-local counting sort: no block barrier.
        uint32_t ent[kShadeChunk / 64];
#pragma unroll
        for (int j = 0; j < kShadeChunk / 64; ++j) {
            const uint32_t qi = base + uint32_t(j) * 64u + uint32_t(lane);
            ent[j] = qi < count ? B.shade_q[qi] : kInvalid;
        }
        uint32_t run = 0;
        for (uint32_t c = 0; c < 8; ++c) {
#pragma unroll
            for (int j = 0; j < kShadeChunk / 64; ++j) {
                const bool is_c = (ent[j] == kInvalid ? 7u : ent[j] >> kSlotBits) == c;
                const uint64_t m = __ballot(is_c);
                if (is_c)
                    s_entry[wave][run + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u))] = ent[j];
                run += uint32_t(__popcll(m));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        SHADE_STAMP(0);
      for (int round = 0; round < kShadeChunk / 64; ++round) {
        const uint32_t mine = s_entry[wave][round * 64 + lane];
        const bool valid = mine != kInvalid;
        if (__ballot(valid) == 0) break;  // padding sorts last
        const uint32_t slot = mine & ((1u << kSlotBits) - 1u);
        // The loop body is two converged sections, each ending in a queue append, so that the
        // NEE record's ~20 registers are dead before the continuation is sampled:
        //   A: interaction, Le, BSDF, both halves of EstimateDirect  -> NEE record
        //   B: next direction, throughput, Russian roulette          -> next ray
        bool surface = false;  // a hit that still scatters (bounces < maxDepth)
        bool alive = false, returned_early = false;
        uint32_t pid = 0, hidx = 0;
        int dim = 0;
        Isect is;
        Bsdf bsdf;
        F3 beta = F3{0, 0, 0}, ray_d = F3{0, 0, 1};
        uint32_t state_next = 0;
        {
            bool emit_nee = false;
            F3 so = F3{0, 0, 0}, sd = F3{0, 0, 1}, mo = F3{0, 0, 0}, md = F3{0, 0, 1};
            F3 A = F3{0, 0, 0}, Bc = F3{0, 0, 0};
            uint32_t nee_flags = 0, nee_light = 0;
            float light_sel_pdf = 1.f;  // lightPdf of UniformSampleOneLight: Ld is divided by it
            if (valid) {
                float4 o4, d4;
                uint32_t fused_hidx = 0;
                const float4 h4 = B.hits[slot];
                if (bounce == 0 && P.gen_fused) {
                    // the camera ray again, as the first k_extend made it (queue 0 is dense: slot == path id)
                    const float4 cpf = B.beta[slot];  // pFilm and the Halton index, left by k_extend
                    fused_hidx = f2b(cpf.z);
                    float cl0 = 0, cl1 = 0;
                    if (S.lens_radius > 0) {
                        cl0 = sample_dimension(S, s_perms, fused_hidx, 3);
                        cl1 = sample_dimension(S, s_perms, fused_hidx, 4);
                    }
                    F3 co, cd;
                    float ctm;
                    camera_ray(S, cpf.x, cpf.y, cl0, cl1, &co, &cd, &ctm, opaque_zero());
                    o4 = make_float4(co.x, co.y, co.z, b2f(slot));
                    d4 = make_float4(cd.x, cd.y, cd.z, ctm);
                } else {
                    o4 = ro[slot];
                    d4 = rd[slot];
                }
                pid = f2b(o4.w);
                // a path arrives at its first vertex with beta = 1 at sampler dimension 5 (after
                // the camera sample): k_generate does not spend 16 B per path on saying so
                // from bounce 1 on the path's state arrived with its ray (PassBuffers::ray_s; dimension | specularBounce << 16 in
                // the direction record's .w)
                const float4 beta4 = bounce == 0 ? make_float4(1, 1, 1, 0) : rs[slot];
                const uint32_t state_w = bounce == 0 ? 5u : f2b(d4.w);
                beta = F3{beta4.x, beta4.y, beta4.z};
                dim = int(state_w & 0xffffu);
                const bool prev_specular = EXT && (state_w >> 16) != 0;  // specularBounce of path.cpp:150
                hidx = bounce == 0 ? ((P.gen_fused) ? fused_hidx : B.hindex[pid]) : f2b(beta4.w);
                // The four samples of EstimateDirect (dims dim+1 .. dim+4; dim itself is the
                // 1D sample SampleDiscrete consumes), drawn here while few registers are live.
                // Every path of a bounce normally sits at the same dimension.
                float u_nee[4] = {0, 0, 0, 0};
                if (bounce < S.max_depth) {
                    const int dim_u = __builtin_amdgcn_readfirstlane(dim);
                    sample_dimensions_n<4>(S, s_perms, dim_u + 1, __ballot(dim != dim_u) == 0, dim + 1, hidx, u_nee);
                }
                SHADE_STAMP(1);
                const int prim = int(f2b(h4.x));
                const F3 ray_o = F3{o4.x, o4.y, o4.z};
                ray_d = F3{d4.x, d4.y, d4.z};
                float4 v0 = S.tri_verts[3 * size_t(prim)];
                float4 v1 = S.tri_verts[3 * size_t(prim) + 1];
                float4 v2 = S.tri_verts[3 * size_t(prim) + 2];
                keep_whole(v0, v1, v2);  // three 16-byte loads (not three of 12 bytes and three of 4)
                const uint32_t flags = f2b(v0.w);
                const int material = int(f2b(v1.w)), light = int(f2b(v2.w));
                if (flags & 1u) {
                    // the closest hit was the sphere: redo its (deterministic) root
                    // selection to recover the object-space ray and refined hit point
                    float t;
                    F3 od, ph;
                    const DSphere &sp = S.spheres[S.prim_shape[prim]];
                    sphere_test(sp, ray_o, ray_d, IILE_INF, &t, &od, &ph);
                    sphere_interaction<TEX>(sp, od, ph, &is);   // (TEX: with (u, v), dp/du, dp/dv and dn/du, dn/dv for the texture lookups and Material::Bump)
                } else {
                    triangle_interaction(S, prim, flags, F3{v0.x, v0.y, v0.z}, F3{v1.x, v1.y, v1.z},
                                         F3{v2.x, v2.y, v2.z}, ray_d, h4.y, h4.z, h4.w, &is);
                }
                if (EXT && S.probe_mode && bounce == 0) {  // IISPTdIntegrator::Li, iispt_d.cpp:96-108
                    const F3 cv = is.p - ray_o;
                    const DProbeCam &cam = P.probe_cams[(pid / uint32_t(P.kc)) / (256u * uint32_t(P.probe_tiles))];
                    B.aux[pid] = make_float4(cam.nrm[0] * is.n.x + cam.nrm[1] * is.n.y + cam.nrm[2] * is.n.z,
                                             cam.nrm[3] * is.n.x + cam.nrm[4] * is.n.y + cam.nrm[5] * is.n.z,
                                             cam.nrm[6] * is.n.x + cam.nrm[7] * is.n.y + cam.nrm[8] * is.n.z, sqrtf(dot(cv, cv)));
                }
                // emitted light at the first vertex and after a specular bounce (path.cpp:91-101); the probe
                // integrator leaves out the camera ray's own vertex (iispt_d.cpp:116-123)
                if (((bounce == 0 && !(EXT && S.probe_mode)) || prev_specular) && light >= 0) {
                    const float4 L4 = B.L[pid];
                    const F3 L = F3{L4.x, L4.y, L4.z} + beta * light_L(S.lights[light], is.n, -ray_d);
                    B.L[pid] = make_float4(L.x, L.y, L.z, 0);
                }
                if (bounce < S.max_depth) {
                    surface = true;
                    if (TEX && S.textured_materials) {
                        // isect.ComputeScatteringFunctions(ray, ...): ComputeDifferentials (interaction.cpp:95-149)
                        // then the material's Texture::Evaluate calls. Only the camera ray carries differentials
                        // (path.cpp:159 spawns plain Rays); its auxiliary rays are a function of the camera
                        // sample, rebuilt here from the path's pixel instead of travelling with the ray.
                        const DMaterial &m0 = S.materials[material];
                        if (m0.kd_tex >= 0 || m0.ks_tex >= 0 || m0.kr_tex >= 0 || m0.kt_tex >= 0 || m0.bump_tex >= 0 || m0.rough_tex >= 0 || m0.sigma_tex >= 0 || m0.opacity_tex >= 0 || m0.rough_tex_v >= 0) {
                            TexDiff td = TexDiff{0, 0, 0, 0};
                            if (bounce == 0) {
                                int px = 0, py = 0;
                                uint32_t kk = 0;
                                path_pixel(S, P, pid, &px, &py, &kk);
                                const float u0 = sample_dimension(S, s_perms, hidx, 0, px, py), u1 = sample_dimension(S, s_perms, hidx, 1, px, py);
                                float l0 = 0, l1 = 0;
                                if (S.lens_radius > 0) {
                                    l0 = sample_dimension(S, s_perms, hidx, 3);
                                    l1 = sample_dimension(S, s_perms, hidx, 4);
                                }
                                const RayDiff rdiff =
                                    S.probe_mode ? probe_differentials(S, P.probe_cams[(pid / uint32_t(P.kc)) / (256u * uint32_t(P.probe_tiles))],
                                                                       float(px) + u0, float(py) + u1, ray_o, ray_d)
                                                 : camera_differentials(S, float(px) + u0, float(py) + u1, l0, l1, ray_o, ray_d);
                                td = compute_differentials(is, rdiff);
                            }
                            if (m0.bump_tex >= 0) bump(S, m0.bump_tex, td, &is);  // `if (bumpMap) Bump(bumpMap, si)` comes first
                            const DMaterial mm = textured_material(S, m0, is, td);
                            bsdf = make_bsdf<EXT>(mm, is);
                        } else {
                            bsdf = make_bsdf<EXT>(m0, is);
                        }
                    } else {
                        bsdf = make_bsdf<EXT>(S.materials[material], is);
                    }
                    SHADE_STAMP(2);
                    if (n_nonspec(bsdf) > 0) {  // NumComponents(BSDF_ALL & ~BSDF_SPECULAR) > 0, path.cpp:118
                        ++n_nee;
                        // UniformSampleOneLight (integrator.cpp:85-106). One light: it is chosen with pdf 1
                        // (SampleDiscrete still consumes a 1D sample). Several: through the voxel's
                        // distribution of the spatial light distribution (lightdistrib.cpp:134-226,
                        // tabulated at scene creation); a zero pdf returns before any further sample.
                        int li = 0;
                        if (EXT && S.n_lights > 1) {
                            const float ul = sample_dimension_hi(S, s_perms, hidx, dim);
                            li = sample_light(S, is.p, ul, &light_sel_pdf);
                        }
                        if (S.n_lights > 0) ++dim;
                        if (S.n_lights > 0 && light_sel_pdf != 0) {
                            // (the plain build has one light, an emitting sphere: both wave-uniform, see uniform_sphere)
                            DLight lt_u;
                            DSphere lsp_u;
                            if (!EXT) {
                                lt_u = uniform_entry(S.lights, 0);
                                lsp_u = uniform_entry(S.spheres, lt_u.sphere);
                            }
                            const DLight &lt = EXT ? S.lights[li] : lt_u;
                            const DSphere &lsp = EXT ? S.spheres[lt.sphere] : lsp_u;
                            if (EXT && lt.type == kLightInfinite) {
                                // EstimateDirect for the infinite light (integrator.cpp:108-215): both halves; the
                                // BSDF-sampled ray contributes Le(ray) when it escapes (:209-210)
                                const float ul0 = u_nee[0], ul1 = u_nee[1], us0 = u_nee[2], us1 = u_nee[3];
                                dim += 4;
                                float light_pdf = 0, scattering_pdf = 0;
                                F3 wi = F3{0, 0, 0}, target = F3{0, 0, 0};
                                const F3 Li = inf_sample_li(S, lt, is.p, ul0, ul1, &wi, &light_pdf, &target);
                                if (light_pdf > 0 && !is_black(Li)) {
                                    const F3 f = bsdf_f(bsdf, is.wo, wi) * absdot(wi, is.sn);
                                    scattering_pdf = bsdf_pdf(bsdf, is.wo, wi);
                                    if (!is_black(f)) {
                                        so = offset_ray_origin(is.p, is.perr, is.n, target - is.p);
                                        sd = target - so;
                                        const float weight = power_heuristic(light_pdf, scattering_pdf);
                                        A = sdiv(f * Li * weight, light_pdf);
                                        nee_flags |= NEE_HAS_SHADOW;
                                    }
                                }
                                F3 f2 = bsdf_sample_f(bsdf, is.wo, &wi, us0, us1, &scattering_pdf);
                                f2 = f2 * absdot(wi, is.sn);
                                if (!is_black(f2) && scattering_pdf > 0) {
                                    const float lp = inf_pdf_li(S, lt, wi);
                                    if (lp != 0) {
                                        const float weight = power_heuristic(scattering_pdf, lp);
                                        mo = offset_ray_origin(is.p, is.perr, is.n, wi);
                                        md = wi;
                                        // Li is Le(ray) when the MIS ray escapes the scene
                                        Bc = sdiv(f2 * inf_le(S, lt, wi) * weight, scattering_pdf);
                                        nee_flags |= NEE_HAS_MIS;
                                    }
                                }
                            } else if (EXT && lt.type != kLightDiffuseArea && lt.type != kLightAreaTriangle) {
                                // EstimateDirect for a delta light (integrator.cpp:150-166): light sample
                                // only, weight 1. Sample_Li of PointLight (lights/point.cpp:43-52),
                                // SpotLight (spot.cpp:53-76), DistantLight (distant.cpp:50-61).
                                dim += 4;  // uLight and uScattering are drawn all the same
                                const F3 pos = F3{lt.pos[0], lt.pos[1], lt.pos[2]};
                                const F3 I = F3{lt.lemit[0], lt.lemit[1], lt.lemit[2]};
                                F3 wi, target, Li;
                                if (lt.type == kLightDistant) {
                                    wi = pos;                                     // wLight
                                    target = is.p + pos * (2 * lt.world_radius);  // pOutside
                                    Li = I;
                                } else {
                                    wi = normalize(pos - is.p);
                                    target = pos;  // pLight
                                    if (lt.type == kLightSpot) {  // Falloff(-wi)
                                        const F3 w = -wi;
                                        const F3 wl = normalize(F3{lt.w2l[0] * w.x + lt.w2l[1] * w.y + lt.w2l[2] * w.z,
                                                                   lt.w2l[3] * w.x + lt.w2l[4] * w.y + lt.w2l[5] * w.z,
                                                                   lt.w2l[6] * w.x + lt.w2l[7] * w.y + lt.w2l[8] * w.z});
                                        const float cos_theta = wl.z;
                                        float falloff;
                                        if (cos_theta < lt.cos_total_width)
                                            falloff = 0;
                                        else if (cos_theta >= lt.cos_falloff_start)
                                            falloff = 1;
                                        else {
                                            const float delta =
                                                (cos_theta - lt.cos_total_width) / (lt.cos_falloff_start - lt.cos_total_width);
                                            falloff = (delta * delta) * (delta * delta);
                                        }
                                        Li = sdiv(I * falloff, length_sq(pos - is.p));
                                    } else {
                                        Li = sdiv(I, length_sq(pos - is.p));
                                    }
                                }
                                if (!is_black(Li)) {
                                    const F3 f = bsdf_f(bsdf, is.wo, wi) * absdot(wi, is.sn);
                                    if (!is_black(f)) {
                                        // the light-side Interaction has neither normal nor error bounds:
                                        // its OffsetRayOrigin is the point itself (interaction.h:73-78)
                                        so = offset_ray_origin(is.p, is.perr, is.n, target - is.p);
                                        sd = target - so;
                                        A = sdiv(f * Li, 1.f);
                                        nee_flags |= NEE_HAS_SHADOW;
                                    }
                                }
                            } else {
                                const float ul0 = u_nee[0], ul1 = u_nee[1], us0 = u_nee[2], us1 = u_nee[3];
                                dim += 4;
                                // EstimateDirect, light-sampling half (integrator.cpp:117-163)
                                float light_pdf = 0, scattering_pdf = 0;
                                F3 wi = F3{0, 0, 0}, Li = F3{0, 0, 0};
                                LightSample ps = EXT ? shape_sample(S, lt, is, ul0, ul1, &light_pdf)
                                                     : sphere_sample(lsp, is, ul0, ul1, &light_pdf);
                                if (light_pdf == 0 || length_sq(ps.p - is.p) == 0) {
                                    light_pdf = 0;
                                } else {
                                    wi = normalize(ps.p - is.p);
                                    Li = light_L(lt, ps.n, -wi);
                                }
                                if (light_pdf > 0 && !is_black(Li)) {
                                    F3 f = bsdf_f(bsdf, is.wo, wi) * absdot(wi, is.sn);
                                    scattering_pdf = bsdf_pdf(bsdf, is.wo, wi);
                                    if (!is_black(f)) {
                                        // VisibilityTester -> SpawnRayTo(Interaction), interaction.h:73-78
                                        so = offset_ray_origin(is.p, is.perr, is.n, ps.p - is.p);
                                        F3 target = offset_ray_origin(ps.p, ps.perr, ps.n, so - ps.p);
                                        sd = target - so;
                                        const float weight = power_heuristic(light_pdf, scattering_pdf);
                                        A = sdiv(f * Li * weight, light_pdf);
                                        nee_flags |= NEE_HAS_SHADOW;
                                    }
                                }
                                SHADE_STAMP(3);
                                // BSDF-sampling half (integrator.cpp:165-213)
                                F3 f2 = bsdf_sample_f(bsdf, is.wo, &wi, us0, us1, &scattering_pdf);
                                f2 = f2 * absdot(wi, is.sn);
                                if (!is_black(f2) && scattering_pdf > 0) {
                                    mo = offset_ray_origin(is.p, is.perr, is.n, wi);
                                    md = wi;
                                    // The ray only matters if its closest hit is the sampled light (integrator.cpp:205-209),
                                    // and Sphere::Pdf is the cone's pdf for ANY direction (sphere.cpp:294-306): most of these
                                    // rays point away from the light. The traversal would run Sphere::Intersect on this very
                                    // ray with some tMax <= inf, and every rejection of that test that depends on tMax only
                                    // gets stricter as tMax shrinks (t0.hi > tMax, ts.hi > tMax): a ray the sphere test
                                    // rejects at tMax = inf can never end on the light, whatever else it hits. Those rays are
                                    // not traced by the uninstrumented kernels (the instrumented build traces them all: the
                                    // reference's ray counters are part of parity), and nothing else of this half is worked
                                    // out for them — the test comes first, so a wavefront whose rays all miss skips the light's
                                    // pdf, the weight and the contribution. Triangle emitters: Shape::Pdf intersects the
                                    // triangle with this ray anyway (lp == 0 on a miss).
                                    bool can_reach = true;
                                    if (!COUNT && lt.type == kLightDiffuseArea) {
                                        float t_l;
                                        F3 od_l, ph_l;
                                        can_reach = sphere_test(lsp, mo, md, IILE_INF, &t_l, &od_l, &ph_l);
                                    }
                                    if (can_reach) {
                                        const float lp = EXT ? shape_pdf(S, lt, is, wi, &n_pdf_tests, &n_pdf_hits)
                                                             : sphere_pdf(lsp, is, wi);
                                        if (lp != 0) {
                                            const float weight = power_heuristic(scattering_pdf, lp);
                                            // Li is Lemit when the MIS ray finds this light facing it
                                            Bc = sdiv(f2 * F3{lt.lemit[0], lt.lemit[1], lt.lemit[2]} * weight, scattering_pdf);
                                            nee_flags |= NEE_HAS_MIS;
                                        }
                                    }
                                }
                            }
                            nee_light = uint32_t(li);
                            // a record with neither ray adds nothing to L; only the instrumented build needs it (zero_radiance)
                            emit_nee = COUNT || nee_flags != 0;
                        }
                    }
                }
            }
            SHADE_STAMP(4);
            const uint32_t eslot = out_take(nee_out, &B.counts[kCntNee + bounce], emit_nee, pad_nee);
            // the MIS rays go to a dense queue of their own (planes 2 and 3): most records have none
            const bool emit_mis = emit_nee && (nee_flags & NEE_HAS_MIS) != 0;
            const uint32_t mslot = out_take(mis_out, &B.counts[kCntMis + bounce], emit_mis, pad_mis);
            if (emit_nee) {
                B.nee[eslot] = make_float4(so.x, so.y, so.z, light_sel_pdf);
                B.nee[plane + eslot] = make_float4(sd.x, sd.y, sd.z, b2f(nee_flags));
                if (nee_flags & NEE_HAS_MIS) {
                    // the general record: k_shadow forms beta * (([unoccluded] A + [MIS ray lit] B) / lightPdf) once it knows both
                    B.nee[4 * plane + eslot] = make_float4(A.x, A.y, A.z, b2f(pid));
                    B.nee[5 * plane + eslot] = make_float4(Bc.x, Bc.y, Bc.z, b2f(nee_light));
                    B.nee[6 * plane + eslot] = make_float4(beta.x, beta.y, beta.z, b2f(pid));  // beta before this bounce
                } else {
                    // 98 % of the records have no MIS ray: what k_shadow would compute for "unoccluded" is known here — the same
                    // operations in the same order, beta * ((0 + A) / lightPdf) (x / 1 is x: skipped where every lane's is 1) —
                    // so the record carries that product and no throughput plane (16 B less written and read per record)
                    const F3 Ld = F3{0, 0, 0} + A;
                    const F3 pre = (__ballot(light_sel_pdf != 1.f) == 0) ? beta * Ld : beta * sdiv(Ld, light_sel_pdf);
                    B.nee[4 * plane + eslot] = make_float4(pre.x, pre.y, pre.z, b2f(pid));
                }
            }
            if (emit_mis) {
                B.nee[2 * plane + mslot] = make_float4(mo.x, mo.y, mo.z, b2f(eslot));  // + the record it belongs to
                B.nee[3 * plane + mslot] = make_float4(md.x, md.y, md.z, b2f(nee_light));
            }
        }
        SHADE_STAMP(5);
        F3 next_o = F3{0, 0, 0}, next_d = F3{0, 0, 1};
        if (P.skip_last_bounce == 2 && bounce + 1 >= S.max_depth) surface = false;  // the next vertex could add nothing: see PassDesc
        if (surface) {
            // next direction (path.cpp:133-156)
            float u_bsdf[2];
            {
                const int dim_u = __builtin_amdgcn_readfirstlane(dim);
                sample_dimensions_n<2>(S, s_perms, dim_u, __ballot(dim != dim_u) == 0, dim, hidx, u_bsdf);
            }
            const float u0 = u_bsdf[0], u1 = u_bsdf[1];
            dim += 2;
            float pdf = 0;
            F3 wi = F3{0, 0, 0};
            bool sampled_specular = false, sampled_transmission = false;
            const F3 f = bsdf_sample_f(bsdf, -ray_d, &wi, u0, u1, &pdf, EXT, &sampled_specular, &sampled_transmission);
            // etaScale (path.cpp:81, 151-157): a path state of its own, touched only in scenes with glass
            float eta_scale = 1.f;
            if (EXT && S.has_glass && bounce > 0) eta_scale = B.eta_scale[pid];
            if (!(is_black(f) || pdf == 0.f)) {
                beta = beta * sdiv(f * absdot(wi, is.sn), pdf);
                const float by = lum_y(beta);
                if (by < 0.f || is_nan(by)) {
                    returned_early = true;  // `return L` (path.cpp:143-145)
                } else {
                    next_o = offset_ray_origin(is.p, is.perr, is.n, wi);
                    next_d = wi;
                    alive = true;
                    if (sampled_specular && sampled_transmission) {
                        const float eta = bsdf.path_eta;   // BSDF::eta
                        eta_scale *= (dot(-ray_d, is.n) > 0) ? (eta * eta) : 1 / (eta * eta);
                    }
                    // Russian roulette on rrBeta = beta * etaScale (path.cpp:182-190)
                    const F3 rr_beta = beta * eta_scale;
                    const float mc = max3(rr_beta.x, rr_beta.y, rr_beta.z);
                    if (mc < S.rr_threshold && bounce > 3) {
                        const float q = mx(.05f, 1 - mc);
                        const float ur = sample_dimension_hi(S, s_perms, hidx, dim);
                        ++dim;
                        if (ur < q)
                            alive = false;
                        else
                            beta = sdiv(beta, 1 - q);
                    }
                }
            }
            // last shaded vertex: only a specular continuation can still pick up emitted light (PassDesc::skip_last_bounce)
            if (EXT && P.skip_last_bounce == 1 && bounce + 1 >= S.max_depth && !sampled_specular) alive = false;
            if (EXT && alive && S.has_glass) B.eta_scale[pid] = eta_scale;
            // sampler dimension | specularBounce << 16
            state_next = uint32_t(dim) | (sampled_specular ? 0x10000u : 0u);  // sampler dimension | specularBounce << 16
        }
        // ReportValue(pathLength, bounces): a path that ends in this iteration leaves the
        // loop with bounces == bounce (not counted on the early `return L`)
        if (COUNT && valid && !alive && !returned_early) ++n_term;
        SHADE_STAMP(6);
        const uint32_t nslot = out_take(ray_out, &B.counts[kCntRay + bounce + 1], alive, pad_ray);
        if (alive) {
            no[nslot] = make_float4(next_o.x, next_o.y, next_o.z, b2f(pid));
            nd[nslot] = make_float4(next_d.x, next_d.y, next_d.z, b2f(state_next));
            ns[nslot] = make_float4(beta.x, beta.y, beta.z, b2f(hidx));
        }
        SHADE_STAMP(7);
      }
        __builtin_amdgcn_wave_barrier();  // the next chunk overwrites s_entry
    }
    out_flush(ray_out, pad_ray);
    out_flush(nee_out, pad_nee);
    out_flush(mis_out, pad_mis);
#ifdef IILE_SHADE_STAMPS
    if (!COUNT && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&B.counters->path_length[i], stamp_sum[i]);
#endif
    if (COUNT) {
        flush_counter(&B.counters->nee_evals, n_nee);
        flush_counter(&B.counters->path_length[bounce < 7 ? bounce : 7], n_term);
        flush_counter(&B.counters->tri_tests, n_pdf_tests);  // Triangle::Intersect calls of Shape::Pdf
        flush_counter(&B.counters->tri_hits, n_pdf_hits);
    }
}


// ---------------------------------------------------------------------------
// launchers
void launch_shade(const DScene &S, const PassDesc &P, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg) {
    // as many blocks as are resident at the build's waves per SIMD: the static split has no tail
    const bool tex_build = cfg.count_stats || S.textured_materials || S.probe_mode;
    const dim3 grid(grid_blocks(max_rays, cfg.n_cus, shade_waves(tex_build)));
    const size_t perm_bytes = S.sobol ? size_t(16) : (size_t(S.n_perms) * sizeof(uint16_t) + 15) & ~size_t(15);
    if (cfg.count_stats)
        hipLaunchKernelGGL((k_shade<true, true, true>), grid, dim3(kBlock), perm_bytes, cfg.stream, S, P, B, bounce, B.queue_cap);
    else
        // Scenes of killeroo-simple's kind (one emitting sphere, matte / plastic only) run a build of the
        // kernel without the code for the wider feature set: it costs them registers otherwise (+0.7 ms);
        // likewise image textures have their own build
        if (S.textured_materials || S.probe_mode)
            hipLaunchKernelGGL((k_shade<false, true, true>), grid, dim3(kBlock), perm_bytes, cfg.stream, S, P, B, bounce, B.queue_cap);
        else if (S.extended_features)
            hipLaunchKernelGGL((k_shade<false, true, false>), grid, dim3(kBlock), perm_bytes, cfg.stream, S, P, B, bounce, B.queue_cap);
        else
            hipLaunchKernelGGL((k_shade<false, false, false>), grid, dim3(kBlock), perm_bytes, cfg.stream, S, P, B, bounce, B.queue_cap);
}
void launch_light_distributions(const DScene &S, const float *samples, float *out, const LaunchCfg &cfg) {
    const int n = S.light_nv[0] * S.light_nv[1] * S.light_nv[2];
    hipLaunchKernelGGL(k_light_distributions, dim3((n + 127) / 128), dim3(128), 0, cfg.stream, S, samples, out);
}

}  // namespace iile
