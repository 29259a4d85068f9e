// kernels_direct.hip — the IISPT integrator's DIRECT pass on the device:
// DirectProgressiveIntegrator::Li / RenderOnePass (src/integrators/directprogressiveintegrator.cpp:22-150) as
// IisptRenderRunner::run_direct drives it (src/integrators/iisptrenderrunner.cpp:601-633), with the RandomSampler
// CreateIISPTIntegrator makes (src/integrators/iispt.cpp:813-816). Operation for operation the oracle's "DIRECT pass"
// (oracle/oracle_path.cpp: DirectSampler, direct_li, direct_pixel), which also says how the reference's per-thread random
// stream is restated as a function of (pass, pixel).
//
// One pass = one camera sample per pixel through the wavefront pipeline of the path integrator, with three kernels of its own:
//   k_direct_generate   camera sample from the pixel's PCG32 stream (behind the 2D arrays StartPixel fills first)
//   [k_extend]          closest hit, as ever
//   k_direct_shade      one vertex of Li: Le -> E[depth], nSamples EstimateDirect calls per light -> NEE records whose results the
//                       ordinary k_mis / k_mis_lit / k_shadow leave in D[depth * (sum of nSamples) + sample], the mirror direction
//                       (SpecularReflect) -> F[depth] and the next ray
//   k_direct_fold       L = ((Le + sum of the lights' Ld) + f * L(next vertex) * |cos| / pdf) + 0 from the deepest vertex
//                       back to the camera — the recursion's own order of float operations —, the render loop's radiance
//                       guards, IisptFilmMonitor::add_n_samples in double precision
// Li's recursion is a chain here, never a tree: of the lobes of matte / plastic / uber (Kr) / mirror only SpecularReflection
// matches BSDF_REFLECTION | BSDF_SPECULAR and none matches BSDF_TRANSMISSION | BSDF_SPECULAR, so SpecularTransmit returns 0.
// Glass branches — Li builds its BSDF with allowMultipleLobes = false (interaction.h:130-133), GlassMaterial then adds a
// SpecularReflection and a SpecularTransmission lobe (glass.cpp:62-90) and both recursions fire —: scenes with glass do not take
// this wavefront but k_direct_tree below, one thread per pixel walking the tree depth first (round 3 rendered glass black beyond
// its direct light, which the round-3 advisor showed to be wrong).
#include "kcommon.h"

namespace iile {

namespace {

// PCG32 (core/rng.h:62-156)
struct DPcg {
    unsigned long long state, inc;
};
DEV uint32_t pcg_u32(DPcg &r) {
    const unsigned long long old = r.state;
    r.state = old * 0x5851f42d4c957f2dULL + r.inc;
    const uint32_t xorshifted = uint32_t(((old >> 18u) ^ old) >> 27u);
    const uint32_t rot = uint32_t(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
DEV float pcg_float(DPcg &r) { return mn(kOneMinusEpsilon, float(pcg_u32(r)) * 0x1p-32f); }
DEV DPcg pcg_seed(unsigned long long seq) {  // RNG(sequenceIndex) -> SetSequence
    DPcg r;
    r.state = 0u;
    r.inc = (seq << 1u) | 1u;
    (void)pcg_u32(r);
    r.state += 0x853c49e6748fea9bULL;
    (void)pcg_u32(r);
    return r;
}
// the stream n draws further on: state_n = A_n state + G_n inc (A_n = a^n, G_n = 1 + a + .. + a^(n-1), host-made table)
DEV DPcg pcg_at(const DPcg &base, const unsigned long long *jump, int i) {
    DPcg r;
    r.state = jump[2 * i] * base.state + jump[2 * i + 1] * base.inc;
    r.inc = base.inc;
    return r;
}
// pass: the pass inside this launch (a launch renders PassDesc::kc passes of the frame at once: path id = (pixel slot, pass);
// P.direct_seed is the first one's seed, 6284 + 17 p for pass p)
DEV DPcg pixel_stream(const DScene &S, const PassDesc &P, int px, int py, uint32_t pass) {
    const uint32_t rank = uint32_t(py - S.samp_y0) * uint32_t(S.samp_x1 - S.samp_x0) + uint32_t(px - S.samp_x0);
    return pcg_seed((static_cast<unsigned long long>(P.direct_seed + 17u * pass) << 32) + rank);
}

DEV F3 area_light_L(const DLight &lt, F3 n, F3 w) {  // DiffuseAreaLight::L, lights/diffuse.h:56-58
    return (lt.two_sided || dot(n, w) > 0) ? F3{lt.lemit[0], lt.lemit[1], lt.lemit[2]} : F3{0, 0, 0};
}

}  // namespace

// One EstimateDirect call of UniformSampleAllLights (integrator.cpp:108-215) as a REQUEST: the shadow ray of the light-sampling
// half with what it adds if unoccluded (A), the closest-hit ray of the BSDF-sampling half with what it adds if it ends on the
// sampled light (Bc). The wavefront pass turns the request into an NEE record (k_mis / k_mis_lit / k_shadow resolve it), the
// per-pixel pass for glass scenes (k_direct_tree) traces the two rays on the spot. Returns NEE_HAS_SHADOW | NEE_HAS_MIS.
DEV uint32_t direct_light_request(const DScene &S, const DLight &lt, const Isect &is, const Bsdf &bsdf, float ul0, float ul1, float us0, float us1,
                                  F3 &so, F3 &sd, F3 &A, F3 &mo, F3 &md, F3 &Bc) {
    uint32_t nee_flags = 0;
    if (lt.type == kLightInfinite) {
        // EstimateDirect for the infinite light (integrator.cpp:108-215), as k_shade has it: the light-sampling half
        // through the environment map's Distribution2D, the BSDF-sampling half whose ray contributes Le(ray) when it
        // escapes (:209-210; k_mis marks escaped rays, k_mis_lit accepts them for an infinite light)
        float light_pdf = 0, scattering_pdf = 0;
        F3 wi = F3{0, 0, 0}, target = F3{0, 0, 0};
        const F3 Li = inf_sample_li(S, lt, is.p, ul0, ul1, &wi, &light_pdf, &target);
        if (light_pdf > 0 && !is_black(Li)) {
            const F3 f = bsdf_f(bsdf, is.wo, wi) * absdot(wi, is.sn);
            scattering_pdf = bsdf_pdf(bsdf, is.wo, wi);
            if (!is_black(f)) {
                so = offset_ray_origin(is.p, is.perr, is.n, target - is.p);
                sd = target - so;
                A = sdiv(f * Li * power_heuristic(light_pdf, scattering_pdf), light_pdf);
                nee_flags |= NEE_HAS_SHADOW;
            }
        }
        F3 f2 = bsdf_sample_f(bsdf, is.wo, &wi, us0, us1, &scattering_pdf);
        f2 = f2 * absdot(wi, is.sn);
        if (!is_black(f2) && scattering_pdf > 0) {
            const float lp = inf_pdf_li(S, lt, wi);
            if (lp != 0) {
                mo = offset_ray_origin(is.p, is.perr, is.n, wi);
                md = wi;
                Bc = sdiv(f2 * inf_le(S, lt, wi) * power_heuristic(scattering_pdf, lp), scattering_pdf);
                nee_flags |= NEE_HAS_MIS;
            }
        }
    } else if (lt.type != kLightDiffuseArea && lt.type != kLightAreaTriangle) {
        // EstimateDirect for a delta light (integrator.cpp:150-166): light sample only. PointLight (lights/point.cpp:
        // 43-52), SpotLight (spot.cpp:53-76), DistantLight (distant.cpp:50-61).
        const F3 pos = F3{lt.pos[0], lt.pos[1], lt.pos[2]};
        const F3 I = F3{lt.lemit[0], lt.lemit[1], lt.lemit[2]};
        F3 wi, target, Li;
        if (lt.type == kLightDistant) {
            wi = pos;
            target = is.p + pos * (2 * lt.world_radius);
            Li = I;
        } else {
            wi = normalize(pos - is.p);
            target = pos;
            if (lt.type == kLightSpot) {
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
                    const float delta = (cos_theta - lt.cos_total_width) / (lt.cos_falloff_start - lt.cos_total_width);
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
                so = offset_ray_origin(is.p, is.perr, is.n, target - is.p);
                sd = target - so;
                A = sdiv(f * Li, 1.f);
                nee_flags |= NEE_HAS_SHADOW;
            }
        }
    } else {
        // EstimateDirect, light-sampling half (integrator.cpp:117-163)
        float light_pdf = 0, scattering_pdf = 0;
        F3 wi = F3{0, 0, 0}, Li = F3{0, 0, 0};
        const LightSample ps = shape_sample(S, lt, is, ul0, ul1, &light_pdf);
        if (light_pdf == 0 || length_sq(ps.p - is.p) == 0) {
            light_pdf = 0;
        } else {
            wi = normalize(ps.p - is.p);
            Li = area_light_L(lt, ps.n, -wi);
        }
        if (light_pdf > 0 && !is_black(Li)) {
            const F3 f = bsdf_f(bsdf, is.wo, wi) * absdot(wi, is.sn);
            scattering_pdf = bsdf_pdf(bsdf, is.wo, wi);
            if (!is_black(f)) {
                so = offset_ray_origin(is.p, is.perr, is.n, ps.p - is.p);
                const F3 target = offset_ray_origin(ps.p, ps.perr, ps.n, so - ps.p);
                sd = target - so;
                A = sdiv(f * Li * power_heuristic(light_pdf, scattering_pdf), light_pdf);
                nee_flags |= NEE_HAS_SHADOW;
            }
        }
        // BSDF-sampling half (integrator.cpp:165-213). A ray that the light's sphere rejects at tMax = inf can never end on the
        // light whatever else it hits (DESIGN.md "MIS rays that cannot score": every tMax-dependent branch of Sphere::Intersect is
        // a rejection), so its term is exactly zero and it is not queued — with nSamples = 8 that is most of the pass's rays.
        F3 f2 = bsdf_sample_f(bsdf, is.wo, &wi, us0, us1, &scattering_pdf);
        f2 = f2 * absdot(wi, is.sn);
        if (!is_black(f2) && scattering_pdf > 0) {
            const F3 m_o = offset_ray_origin(is.p, is.perr, is.n, wi);
            bool can_reach = true;
            if (lt.type == kLightDiffuseArea) {
                float t_l;
                F3 od_l, ph_l;
                can_reach = sphere_test(S.spheres[lt.sphere], m_o, wi, IILE_INF, &t_l, &od_l, &ph_l);
            }
            if (can_reach) {
                unsigned long long nt = 0, nh = 0;
                const float lp = shape_pdf(S, lt, is, wi, &nt, &nh);
                if (lp != 0) {
                    mo = m_o;
                    md = wi;
                    Bc = sdiv(f2 * F3{lt.lemit[0], lt.lemit[1], lt.lemit[2]} * power_heuristic(scattering_pdf, lp), scattering_pdf);
                    nee_flags |= NEE_HAS_MIS;
                }
            }
        }
    }
    return nee_flags;
}

// jump[2 i], jump[2 i + 1]: the stream at array i's first entry (the arrays before it hold 16 x nSamples entries of two draws
// each: RandomSampler::StartPixel, random.cpp:62-72); entry n_arrays: the camera sample
__global__ __launch_bounds__(kBlock) void k_direct_generate(DScene S, PassDesc P, PassBuffers B) {
    for (uint32_t pid = blockIdx.x * kBlock + threadIdx.x; pid < P.n_paths; pid += gridDim.x * kBlock) {
        int px = 0, py = 0;
        uint32_t k = 0;
        const bool valid = path_pixel(S, P, pid, &px, &py, &k) && px >= S.crop_x0 && px < S.crop_x1 && py >= S.crop_y0 && py < S.crop_y1;
        F3 o = F3{0, 0, 0}, d = F3{0, 0, 1};
        float tmax = 0;
        if (valid) {
            DPcg r = pcg_at(pixel_stream(S, P, px, py, k), P.direct_jump, P.direct_arrays);
            // GetCameraSample: pFilm = pixel + Get2D(), time = Get1D(), pLens = Get2D()
            const float u0 = pcg_float(r), u1 = pcg_float(r);
            (void)pcg_float(r);
            const float l0 = pcg_float(r), l1 = pcg_float(r);
            camera_ray(S, float(px) + u0, float(py) + u1, l0, l1, &o, &d, &tmax);
            reinterpret_cast<float2 *>(&B.beta[pid])[0] = make_float2(float(px) + u0, float(py) + u1);  // for the differentials
            reinterpret_cast<float2 *>(&B.beta[pid])[1] = make_float2(l0, l1);
        }
        B.ray_o[0][pid] = make_float4(o.x, o.y, o.z, b2f(valid ? pid : kInvalid));
        B.ray_d[0][pid] = make_float4(d.x, d.y, d.z, tmax);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) B.counts[kCntRay] = P.n_paths;
}

// One vertex of DirectProgressiveIntegrator::Li for every hit of the bounce's shade queue.
#ifndef IILE_DIRECT_SHADE_WAVES
#define IILE_DIRECT_SHADE_WAVES 2  // waves per SIMD the register allocation aims at
#endif
template <bool TEX>
__global__ __launch_bounds__(kBlock, IILE_DIRECT_SHADE_WAVES) void k_direct_shade(DScene S, PassDesc P, PassBuffers B, int depth, uint32_t plane) {
    const uint32_t count = B.counts[kCntShade + depth];
    const float4 *ro = B.ray_o[depth & 1], *rd = B.ray_d[depth & 1];
    float4 *no = B.ray_o[(depth + 1) & 1], *nd = B.ray_d[(depth + 1) & 1];
    WaveOut ray_out{0, 0}, nee_out{0, 0}, mis_out{0, 0};
    auto pad_ray = [&](uint32_t sl) { no[sl] = make_float4(0, 0, 0, b2f(kInvalid)); };
    auto pad_nee = [&](uint32_t sl) { B.nee[plane + sl] = B.nee[4 * plane + sl] = make_float4(0, 0, 0, b2f(kInvalid)); };
    auto pad_mis = [&](uint32_t sl) { B.nee[2 * plane + sl] = make_float4(0, 0, 0, b2f(kInvalid)); };
    const uint32_t rounds = (count + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
    for (uint32_t it = 0; it < rounds; ++it) {  // (every lane walks every round: the queue appends are wavefront-wide)
        const uint32_t qi = (it * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
        const uint32_t ent = qi < count ? B.shade_q[qi] : kInvalid;
        const bool valid = ent != kInvalid;
        const uint32_t slot = ent & ((1u << kSlotBits) - 1u);
        uint32_t pid = 0;
        Isect is;
        Bsdf bsdf;
        F3 ray_d = F3{0, 0, 1};
        bool lit_surface = false;  // a surface with a non-specular lobe: EstimateDirect can return something
        DPcg stream{0, 1};
        RayDiff rdiff = RayDiff{F3{0, 0, 0}, F3{0, 0, 0}, F3{0, 0, 1}, F3{0, 0, 1}};
        bool has_diff = false;
        TexDiff td = TexDiff{0, 0, 0, 0};
        F3 dpdx = F3{0, 0, 0}, dpdy = F3{0, 0, 0};
        if (valid) {
            const float4 h4 = B.hits[slot], o4 = ro[slot], d4 = rd[slot];
            pid = f2b(o4.w);
            const int prim = int(f2b(h4.x));
            const F3 ray_o = F3{o4.x, o4.y, o4.z};
            ray_d = F3{d4.x, d4.y, d4.z};
            const float4 v0 = S.tri_verts[3 * size_t(prim)], v1 = S.tri_verts[3 * size_t(prim) + 1], v2 = S.tri_verts[3 * size_t(prim) + 2];
            const uint32_t flags = f2b(v0.w);
            const int material = int(f2b(v1.w)), light = int(f2b(v2.w));
            if (flags & 1u) {
                float t;
                F3 od, ph;
                const DSphere &sp = S.spheres[S.prim_shape[prim]];
                sphere_test(sp, ray_o, ray_d, IILE_INF, &t, &od, &ph);
                sphere_interaction<TEX>(sp, od, ph, &is);
            } else {
                triangle_interaction(S, prim, flags, F3{v0.x, v0.y, v0.z}, F3{v1.x, v1.y, v1.z}, F3{v2.x, v2.y, v2.z}, ray_d, h4.y, h4.z,
                                     h4.w, &is);
            }
            int px = 0, py = 0;
            uint32_t kk = 0;
            path_pixel(S, P, pid, &px, &py, &kk);
            stream = pixel_stream(S, P, px, py, kk);
            // isect.ComputeScatteringFunctions(ray, arena): the differentials of the camera ray at depth 0, of the reflected ray
            // (left by the vertex before, below) further on; computed for every hit when reflected rays carry them, else only
            // where a texture is looked up
            const DMaterial &m0 = S.materials[material];
            const bool mat_tex = m0.kd_tex >= 0 || m0.ks_tex >= 0 || m0.kr_tex >= 0 || m0.kt_tex >= 0 || m0.bump_tex >= 0 || m0.rough_tex >= 0 || m0.sigma_tex >= 0 || m0.opacity_tex >= 0 || m0.rough_tex_v >= 0;
            if (TEX && S.textured_materials && (mat_tex || B.dir_RD)) {
                if (depth == 0) {
                    const float4 cs = B.beta[pid];  // pFilm, pLens left by k_direct_generate
                    rdiff = camera_differentials(S, cs.x, cs.y, cs.z, cs.w, ray_o, ray_d);
                    has_diff = true;
                } else if (B.dir_RD) {
                    const float4 r0 = B.dir_RD[pid];
                    has_diff = r0.w != 0.f;
                    if (has_diff) {
                        const float4 r1 = B.dir_RD[size_t(B.dir_paths) + pid], r2 = B.dir_RD[2 * size_t(B.dir_paths) + pid],
                                     r3 = B.dir_RD[3 * size_t(B.dir_paths) + pid];
                        rdiff = RayDiff{F3{r0.x, r0.y, r0.z}, F3{r1.x, r1.y, r1.z}, F3{r2.x, r2.y, r2.z}, F3{r3.x, r3.y, r3.z}};
                    }
                }
                if (has_diff) td = compute_differentials(is, rdiff, &dpdx, &dpdy);
            }
            if (TEX && S.textured_materials && mat_tex) {
                if (m0.bump_tex >= 0) bump(S, m0.bump_tex, td, &is);
                bsdf = make_bsdf<true>(textured_material(S, m0, is, td), is);
            } else {
                bsdf = make_bsdf<true>(m0, is);
            }
            // L += isect.Le(wo)
            if (light >= 0) {
                const F3 Le = area_light_L(S.lights[light], is.n, -ray_d);
                B.dir_E[size_t(depth) * B.dir_paths + pid] = make_float4(Le.x, Le.y, Le.z, 0);
            }
            lit_surface = n_nonspec(bsdf) > 0;
        }
        // UniformSampleAllLights (integrator.cpp:54-83): every light nSamples times, sample k from entry k of the two arrays the
        // pixel's stream filled first for it (array 2 c: uLight, 2 c + 1: uScattering, c = depth * n_lights + light: the c-th pair
        // of Get2DArray calls; pixel sample 0 reads the arrays' first nSamples entries)
        int sample_slot = depth * P.direct_total_samples;  // result slot of (depth, light, k) in D
        for (int li = 0; li < S.n_lights; ++li) {
          const int c = depth * S.n_lights + li;
          DPcg ra{0, 1}, rb{0, 1};
          if (valid && lit_surface) {
              ra = pcg_at(stream, P.direct_jump, 2 * c);
              rb = pcg_at(stream, P.direct_jump, 2 * c + 1);
          }
          for (int ks = 0; ks < P.direct_nsamples[li]; ++ks, ++sample_slot) {
            bool emit_nee = false;
            F3 so = F3{0, 0, 0}, sd = F3{0, 0, 1}, mo = F3{0, 0, 0}, md = F3{0, 0, 1}, A = F3{0, 0, 0}, Bc = F3{0, 0, 0};
            uint32_t nee_flags = 0;
            if (valid && lit_surface) {
                const DLight &lt = S.lights[li];
                const float ul0 = pcg_float(ra), ul1 = pcg_float(ra), us0 = pcg_float(rb), us1 = pcg_float(rb);
                nee_flags = direct_light_request(S, lt, is, bsdf, ul0, ul1, us0, us1, so, sd, A, mo, md, Bc);
                emit_nee = nee_flags != 0;
            }
            const uint32_t eslot = out_take(nee_out, &B.counts[kCntNee + depth], emit_nee, pad_nee);
            const bool emit_mis = emit_nee && (nee_flags & NEE_HAS_MIS) != 0;
            const uint32_t mslot = out_take(mis_out, &B.counts[kCntMis + depth], emit_mis, pad_mis);
            if (emit_nee) {
                // the record's "path" is the slot of D this light sample's result belongs in: k_shadow stores L[that] = 0 + 1 * Ld
                const uint32_t dslot = uint32_t(sample_slot) * B.dir_paths + pid;
                B.nee[eslot] = make_float4(so.x, so.y, so.z, 1.f);
                B.nee[plane + eslot] = make_float4(sd.x, sd.y, sd.z, b2f(nee_flags));
                if (nee_flags & NEE_HAS_MIS) {
                    B.nee[4 * plane + eslot] = make_float4(A.x, A.y, A.z, b2f(dslot));
                    B.nee[5 * plane + eslot] = make_float4(Bc.x, Bc.y, Bc.z, b2f(uint32_t(li)));
                    B.nee[6 * plane + eslot] = make_float4(1.f, 1.f, 1.f, b2f(dslot));
                } else {  // (the record format of k_shade: without a MIS ray, beta * ((0 + A) / lightPdf) ready made — here 1 * A)
                    const F3 pre = F3{1.f, 1.f, 1.f} * (F3{0, 0, 0} + A);
                    B.nee[4 * plane + eslot] = make_float4(pre.x, pre.y, pre.z, b2f(dslot));
                }
            }
            if (emit_mis) {
                B.nee[2 * plane + mslot] = make_float4(mo.x, mo.y, mo.z, b2f(eslot));
                B.nee[3 * plane + mslot] = make_float4(md.x, md.y, md.z, b2f(uint32_t(li)));
            }
          }
        }
        // SpecularReflect (directprogressiveintegrator.cpp:134-190): BSDF::Sample_f(wo, &wi, Get2D(), &pdf, BSDF_REFLECTION |
        // BSDF_SPECULAR) finds a SpecularReflection lobe or nothing (its sample is not used; pdf = 1)
        bool alive = false;
        F3 next_o = F3{0, 0, 0}, next_d = F3{0, 0, 1};
        if (valid && depth + 1 < 5 && bsdf.has_spec && bsdf.mtype != kMatGlass) {
            const F3 wo_w = is.wo;
            const F3 wo = to_local(bsdf, wo_w);
            if (wo.z != 0) {
                const F3 wi_l = F3{-wo.x, -wo.y, wo.z};
                const float fr = bsdf.mtype == kMatMirror ? 1.f : fr_dielectric(wi_l.z, 1.f, bsdf.eta);
                const F3 f = sdiv(F3{fr, fr, fr} * bsdf.kr, fabsf(wi_l.z));
                const F3 wi = to_world(bsdf, wi_l);
                const float ad = absdot(wi, is.sn);
                if (!is_black(f) && ad != 0.f) {
                    B.dir_F[size_t(depth) * B.dir_paths + pid] = make_float4(f.x, f.y, f.z, ad);
                    next_o = offset_ray_origin(is.p, is.perr, is.n, wi);
                    next_d = wi;
                    alive = true;
                    if (TEX && B.dir_RD) {
                        // the reflected ray's differentials (directprogressiveintegrator.cpp:165-184), for the next vertex
                        float4 o0 = make_float4(0, 0, 0, 0), o1 = o0, o2 = o0, o3 = o0;
                        if (has_diff) {
                            const F3 ns = is.sn;
                            const F3 rxo = is.p + dpdx, ryo = is.p + dpdy;
                            const F3 dndx = is.dndu * td.dudx + is.dndv * td.dvdx;
                            const F3 dndy = is.dndu * td.dudy + is.dndv * td.dvdy;
                            const F3 dwodx = -rdiff.rxd - wo_w, dwody = -rdiff.ryd - wo_w;
                            const float dDNdx = dot(dwodx, ns) + dot(wo_w, dndx);
                            const float dDNdy = dot(dwody, ns) + dot(wo_w, dndy);
                            const F3 rxd = wi - dwodx + 2.f * (dot(wo_w, ns) * dndx + dDNdx * ns);
                            const F3 ryd = wi - dwody + 2.f * (dot(wo_w, ns) * dndy + dDNdy * ns);
                            o0 = make_float4(rxo.x, rxo.y, rxo.z, 1.f);
                            o1 = make_float4(ryo.x, ryo.y, ryo.z, 0);
                            o2 = make_float4(rxd.x, rxd.y, rxd.z, 0);
                            o3 = make_float4(ryd.x, ryd.y, ryd.z, 0);
                        }
                        B.dir_RD[pid] = o0;
                        B.dir_RD[size_t(B.dir_paths) + pid] = o1;
                        B.dir_RD[2 * size_t(B.dir_paths) + pid] = o2;
                        B.dir_RD[3 * size_t(B.dir_paths) + pid] = o3;
                    }
                }
            }
        }
        const uint32_t nslot = out_take(ray_out, &B.counts[kCntRay + depth + 1], alive, pad_ray);
        if (alive) {
            no[nslot] = make_float4(next_o.x, next_o.y, next_o.z, b2f(pid));
            nd[nslot] = make_float4(next_d.x, next_d.y, next_d.z, IILE_INF);
        }
    }
    out_flush(ray_out, pad_ray);
    out_flush(nee_out, pad_nee);
    out_flush(mis_out, pad_mis);
}

// A ray that leaves the scene returns the infinite lights' radiance (directprogressiveintegrator.cpp:29-32: `for (const auto
// &light : scene.lights) L += light->Le(ray)`, at ANY depth — the path integrator adds it only at the camera vertex or after a
// specular bounce): it lands in E[depth], where a hit vertex has its emitted light. Launched for scenes with an infinite light.
__global__ __launch_bounds__(kBlock) void k_direct_miss(DScene S, PassBuffers B, int depth) {
    const uint32_t count = B.counts[kCntRay + depth];
    const float4 *ro = B.ray_o[depth & 1], *rd = B.ray_d[depth & 1];
    for (uint32_t slot = blockIdx.x * kBlock + threadIdx.x; slot < count; slot += gridDim.x * kBlock) {
        const uint32_t pid = f2b(ro[slot].w);
        if (pid == kInvalid) continue;
        if (int(f2b(B.hits[slot].x)) >= 0) continue;
        const float4 d4 = rd[slot];
        const F3 d = F3{d4.x, d4.y, d4.z};
        F3 L = F3{0, 0, 0};
        for (int l = 0; l < S.n_lights; ++l)
            if (S.lights[l].type == kLightInfinite) L = L + inf_le(S, S.lights[l], d);
        B.dir_E[size_t(depth) * B.dir_paths + pid] = make_float4(L.x, L.y, L.z, 0);
    }
}

// Li folded from the deepest vertex back (the recursion returns in that order), the render loop's guards
// (directprogressiveintegrator.cpp:104-127), IisptFilmMonitor::add_n_samples (iisptfilmmonitor.cpp:47-72: doubles)
__global__ __launch_bounds__(kBlock) void k_direct_fold(DScene S, PassDesc P, PassBuffers B, double *film_rgbw) {
    const int fw = S.crop_x1 - S.crop_x0;
    // one thread per pixel slot; the launch's passes of that pixel are added to the film monitor one after the other, in pass
    // order (add_n_samples runs once per pass in the reference: the order of the double additions is part of the result)
    const uint32_t kc = uint32_t(P.kc);
    for (uint32_t pt = blockIdx.x * kBlock + threadIdx.x; pt < P.n_paths / kc; pt += gridDim.x * kBlock)
    for (uint32_t pass = 0; pass < kc; ++pass) {
        const uint32_t pid = pt * kc + pass;
        int px = 0, py = 0;
        uint32_t k = 0;
        if (!(path_pixel(S, P, pid, &px, &py, &k) && px >= S.crop_x0 && px < S.crop_x1 && py >= S.crop_y0 && py < S.crop_y1)) continue;
        F3 Lnext = F3{0, 0, 0};
        // (levels that cannot exist — no specular lobe in the scene: everything past the camera vertex — would add zeros only:
        //  they are neither stored nor read)
        for (int d = P.direct_levels - 1; d >= 0; --d) {
            const float4 e4 = B.dir_E[size_t(d) * B.dir_paths + pid];
            F3 L = F3{0, 0, 0};
            L = L + F3{e4.x, e4.y, e4.z};  // L += isect.Le(wo)
            F3 all = F3{0, 0, 0};          // UniformSampleAllLights' own L(0.f)
            int sample_slot = d * P.direct_total_samples;
            for (int li = 0; li < S.n_lights; ++li) {
                F3 Ld = F3{0, 0, 0};
                for (int ks = 0; ks < P.direct_nsamples[li]; ++ks, ++sample_slot) {  // Ld += EstimateDirect(.., uScatteringArray[k], .., uLightArray[k], ..)
                    const float4 d4 = B.L[size_t(sample_slot) * B.dir_paths + pid];
                    Ld = Ld + F3{d4.x, d4.y, d4.z};
                }
                all = all + sdiv(Ld, float(P.direct_nsamples[li]));  // L += Ld / nSamples
            }
            if (S.n_lights > 0) L = L + all;
            if (d + 1 < 5) {
                const float4 f4 = B.dir_F[size_t(d) * B.dir_paths + pid];
                F3 R = F3{0, 0, 0};
                if (f4.w != 0.f) R = sdiv(F3{f4.x, f4.y, f4.z} * Lnext * f4.w, 1.f);  // f * Li(..) * AbsDot(wi, ns) / pdf
                L = L + R;
                L = L + F3{0, 0, 0};  // SpecularTransmit
            }
            Lnext = L;
        }
        F3 L = Lnext;
        const float y = lum_y(L);
        if (is_nan(L.x) || is_nan(L.y) || is_nan(L.z))
            L = F3{0, 0, 0};
        else if (double(y) < -1e-5)
            L = F3{0, 0, 0};
        else if (is_inf(y))
            L = F3{0, 0, 0};
        double *out = film_rgbw + 4 * (size_t(py - S.crop_y0) * fw + (px - S.crop_x0));
        out[0] += double(L.x);
        out[1] += double(L.y);
        out[2] += double(L.z);
        out[3] += 1.0;
    }
}

// ---------------------------------------------------------------------------
// Scenes with glass: DirectProgressiveIntegrator::Li is a TREE there. Its BSDF is built with allowMultipleLobes = false
// (interaction.h:130-133), so GlassMaterial adds a SpecularReflection(R, FresnelDielectric(1, eta)) and a
// SpecularTransmission(T, 1, eta, Radiance) lobe (glass.cpp:62-90) and SpecularReflect and SpecularTransmit both recurse. The
// pixel's RandomSampler stream is consumed in the recursion's depth-first order — the sample arrays run out after the first
// five vertices VISITED (UniformSampleAllLights then falls back to Get2D draws, one sample per light, integrator.cpp:66-72), so
// which numbers a vertex gets depends on the whole subtree to its left — which a breadth-first wavefront does not know. One
// thread per pixel therefore walks its own tree depth first, with an explicit stack of five levels, tracing every ray on the
// spot (traverse()), in the reference's own order of draws and additions. Slow next to the wavefront (divergent, register
// heavy); it is the corner the wavefront cannot do, not the product's path for anything else. Held to the oracle's recursion
// bit for bit and, through it, to the analytic slab (tests/test_iispt_direct.py).
namespace {
// The walk below is one large kernel; with everything inlined hipcc 7.2 builds 250 KB of code at 256 VGPRs and ~280 spilled
// SGPRs — the regime in which it produced wrong code for the one-kernel IISPT gather (iispt.hip's header) and a memory fault
// here. Its heavy pieces are therefore real calls (speed is not what this kernel is for).
#define TREE_CALL __device__ __noinline__
TREE_CALL bool tree_trace(const DScene &S, F3 ro, F3 rd, float tmax, bool any_hit, lds_int *stack, int *spill, uint32_t spill_stride, HitRec *h) {
    TraceStats st = {0, 0, 0, 0};
    h->t = h->b0 = h->b1 = h->b2 = 0;
    h->prim = -1;
    if (any_hit) return traverse<true, false>(S, ro, rd, tmax, stack, spill, spill_stride, h, &st);
    return traverse<false, false>(S, ro, rd, tmax, stack, spill, spill_stride, h, &st);
}
// the SurfaceInteraction of a hit (and the light / material of its primitive)
template <bool TEX>
TREE_CALL void tree_interaction(const DScene &S, int prim, F3 ro, F3 rd, float b0, float b1, float b2, Isect *is, int *material, int *light) {
    const float4 v0 = S.tri_verts[3 * size_t(prim)], v1 = S.tri_verts[3 * size_t(prim) + 1], v2 = S.tri_verts[3 * size_t(prim) + 2];
    const uint32_t flags = f2b(v0.w);
    *material = int(f2b(v1.w));
    *light = int(f2b(v2.w));
    if (flags & 1u) {
        float t;
        F3 od, ph;
        const DSphere &sp = S.spheres[S.prim_shape[prim]];
        sphere_test(sp, ro, rd, IILE_INF, &t, &od, &ph);
        sphere_interaction<TEX>(sp, od, ph, is);
    } else {
        triangle_interaction(S, prim, flags, F3{v0.x, v0.y, v0.z}, F3{v1.x, v1.y, v1.z}, F3{v2.x, v2.y, v2.z}, rd, b0, b1, b2, is);
    }
}
template <bool TEX>
TREE_CALL void tree_bsdf(const DScene &S, int material, const TexDiff &td, Isect *is, Bsdf *bsdf) {
    const DMaterial &m0 = S.materials[material];
    const bool mat_tex = m0.kd_tex >= 0 || m0.ks_tex >= 0 || m0.kr_tex >= 0 || m0.kt_tex >= 0 || m0.bump_tex >= 0 || m0.rough_tex >= 0 || m0.sigma_tex >= 0 || m0.opacity_tex >= 0 || m0.rough_tex_v >= 0;
    if (TEX && S.textured_materials && mat_tex) {
        if (m0.bump_tex >= 0) bump(S, m0.bump_tex, td, is);
        *bsdf = make_bsdf<true>(textured_material(S, m0, *is, td), *is);
    } else {
        *bsdf = make_bsdf<true>(m0, *is);
    }
}
TREE_CALL uint32_t tree_light_request(const DScene &S, int li, const Isect &is, const Bsdf &bsdf, float ul0, float ul1, float us0, float us1, F3 *so,
                                      F3 *sd, F3 *A, F3 *mo, F3 *md, F3 *Bc) {
    return direct_light_request(S, S.lights[li], is, bsdf, ul0, ul1, us0, us1, *so, *sd, *A, *mo, *md, *Bc);
}
// did the BSDF-sampled ray of EstimateDirect end on light `li`, on its emitting side? (k_mis + k_mis_lit of the wavefront)
TREE_CALL bool tree_mis_lit(const DScene &S, int li, bool hit, const HitRec &hm, F3 mo, F3 md) {
    const DLight &lt = S.lights[li];
    if (!hit) return lt.type == kLightInfinite;  // `else Li = light.Le(ray)`, integrator.cpp:209-210
    const float4 w0 = S.tri_verts[3 * size_t(hm.prim)], w1 = S.tri_verts[3 * size_t(hm.prim) + 1], w2 = S.tri_verts[3 * size_t(hm.prim) + 2];
    if (int(f2b(w2.w)) != li) return false;  // lightIsect.primitive->GetAreaLight() == &light
    Isect lis;
    if (f2b(w0.w) & 1u) {
        float th;
        F3 od, ph;
        const DSphere &sp = S.spheres[lt.sphere];
        sphere_test(sp, mo, md, IILE_INF, &th, &od, &ph);
        sphere_interaction(sp, od, ph, &lis);
    } else {
        triangle_interaction(S, hm.prim, f2b(w0.w), F3{w0.x, w0.y, w0.z}, F3{w1.x, w1.y, w1.z}, F3{w2.x, w2.y, w2.z}, md, hm.b0, hm.b1, hm.b2, &lis);
    }
    return lt.two_sided || dot(lis.n, -md) > 0;
}
// NT: the number of BSDF_TRANSMISSION | BSDF_SPECULAR lobes a BSDF of the scene can hold — 1 (glass) or 2 (an uber material's
// pass-through and its Kt lobe): SpecularTransmit's u[0] picks among them AFTER the reflection subtree has drawn its samples, so
// every lobe's transmitted ray is made when the vertex is shaded and the choice is taken when the reflection is back
template <int NT>
struct TreeLevel {
    F3 L;                 // Le + direct light (+ the reflection's share once it is back)
    F3 f_r, f_t[NT];      // SpecularReflect / SpecularTransmit: f of the lobe's sample
    float ad_r, ad_t[NT]; // AbsDot(wi, ns); 0 = that recursion does not happen
    F3 t_o[NT], t_d[NT];  // the transmitted ray, made when the vertex is shaded, traced after the reflection subtree
    RayDiff t_rd[NT];
    bool t_has_diff[NT];
    int n_t;              // matchingComps of SpecularTransmit's Sample_f; slot 0 holds the chosen lobe from stage 2 on
    float pdf_t;          // 1 / matchingComps
    int stage;            // 1: the reflection subtree is being walked, 2: the transmission subtree
};
}  // namespace

template <bool TEX, int NT>
__global__ __launch_bounds__(kBlock, 1) void k_direct_tree(DScene S, PassDesc P, PassBuffers B, double *film_rgbw) {
    __shared__ int lds_stack[kWavesPerBlock][2 * kLdsStackDepth][64];
    lds_int *my_stack = (lds_int *)&lds_stack[threadIdx.x >> 6][0][threadIdx.x & 63];
    const uint32_t spill_stride = gridDim.x * kBlock;
    int *my_spill = B.spill + blockIdx.x * kBlock + threadIdx.x;
    const int fw = S.crop_x1 - S.crop_x0;
    for (uint32_t pid = blockIdx.x * kBlock + threadIdx.x; pid < P.n_paths; pid += gridDim.x * kBlock) {
        int px = 0, py = 0;
        uint32_t kk = 0;
        if (!(path_pixel(S, P, pid, &px, &py, &kk) && px >= S.crop_x0 && px < S.crop_x1 && py >= S.crop_y0 && py < S.crop_y1)) continue;
        const DPcg stream = pixel_stream(S, P, px, py, kk);
        DPcg rng = pcg_at(stream, P.direct_jump, P.direct_arrays);  // behind the arrays StartPixel filled: the camera sample
        const float u0 = pcg_float(rng), u1 = pcg_float(rng);
        (void)pcg_float(rng);
        const float l0 = pcg_float(rng), l1 = pcg_float(rng);
        F3 ro, rd;
        float tmax;
        camera_ray(S, float(px) + u0, float(py) + u1, l0, l1, &ro, &rd, &tmax);
        RayDiff rdiff = RayDiff{F3{0, 0, 0}, F3{0, 0, 0}, F3{0, 0, 1}, F3{0, 0, 1}};
        bool has_diff = false;
        if (TEX && S.n_textures > 0) {
            rdiff = camera_differentials(S, float(px) + u0, float(py) + u1, l0, l1, ro, rd);
            has_diff = true;
        }
        TreeLevel<NT> lv[5];
        int depth = 0, visited = 0;  // visited: vertices shaded so far = pairs of sample arrays used up / n_lights
        F3 ret = F3{0, 0, 0};
        bool descend = true;         // true: (ro, rd) is a ray to trace at `depth`; false: `ret` is what the subtree below returned
        for (;;) {
            if (descend) {
                // ---- Li(ray) at `depth`: intersect, shade, decide the two recursions
                HitRec h;
                const bool found = tree_trace(S, ro, rd, depth == 0 ? tmax : IILE_INF, false, my_stack, my_spill, spill_stride, &h);
                if (!found) {  // `for (const auto &light : scene.lights) L += light->Le(ray)`
                    F3 L = F3{0, 0, 0};
                    for (int l = 0; l < S.n_lights; ++l)
                        if (S.lights[l].type == kLightInfinite) L = L + inf_le(S, S.lights[l], rd);
                    ret = L;
                    descend = false;
                    --depth;
                    if (depth < 0) break;
                    continue;
                }
                Isect is;
                int material = 0, light = -1;
                tree_interaction<TEX>(S, h.prim, ro, rd, h.b0, h.b1, h.b2, &is, &material, &light);
                TexDiff td = TexDiff{0, 0, 0, 0};
                F3 dpdx = F3{0, 0, 0}, dpdy = F3{0, 0, 0};
                if (TEX && has_diff) td = compute_differentials(is, rdiff, &dpdx, &dpdy);
                Bsdf bsdf;
                tree_bsdf<TEX>(S, material, td, &is, &bsdf);
                TreeLevel<NT> &me = lv[depth];
                me.L = F3{0, 0, 0};
                if (light >= 0) me.L = me.L + area_light_L(S.lights[light], is.n, -rd);  // L += isect.Le(wo)
                // ---- UniformSampleAllLights, integrator.cpp:54-83
                if (S.n_lights > 0) {
                    F3 all = F3{0, 0, 0};
                    for (int li = 0; li < S.n_lights; ++li) {
                        const bool arrays = visited < 5;  // Get2DArray hands out the requested arrays, then nullptr
                        const int n = arrays ? P.direct_nsamples[li] : 1;
                        DPcg ra{0, 1}, rb{0, 1};
                        if (arrays) {
                            const int c = visited * S.n_lights + li;
                            ra = pcg_at(stream, P.direct_jump, 2 * c);
                            rb = pcg_at(stream, P.direct_jump, 2 * c + 1);
                        }
                        F3 Ld = F3{0, 0, 0};
                        for (int ks = 0; ks < n; ++ks) {
                            float ul0, ul1, us0, us1;
                            if (arrays) {
                                ul0 = pcg_float(ra), ul1 = pcg_float(ra), us0 = pcg_float(rb), us1 = pcg_float(rb);
                            } else {  // `Point2f uLight = sampler.Get2D(); Point2f uScattering = sampler.Get2D();`
                                ul0 = pcg_float(rng), ul1 = pcg_float(rng), us0 = pcg_float(rng), us1 = pcg_float(rng);
                            }
                            F3 so = F3{0, 0, 0}, sd = F3{0, 0, 1}, mo = F3{0, 0, 0}, md = F3{0, 0, 1}, A = F3{0, 0, 0}, Bc = F3{0, 0, 0};
                            uint32_t nf = 0;
                            if (n_nonspec(bsdf) > 0) nf = tree_light_request(S, li, is, bsdf, ul0, ul1, us0, us1, &so, &sd, &A, &mo, &md, &Bc);
                            // one EstimateDirect: (0 + [unoccluded] A) + [the BSDF-sampled ray ended on this light] Bc — the sums k_shadow forms
                            F3 Le1 = F3{0, 0, 0};
                            if (nf & NEE_HAS_SHADOW) {
                                HitRec hs;
                                if (!tree_trace(S, so, sd, 1 - kShadowEpsilon, true, my_stack, my_spill, spill_stride, &hs)) Le1 = Le1 + A;
                            }
                            if (nf & NEE_HAS_MIS) {
                                HitRec hm;
                                const bool hit = tree_trace(S, mo, md, IILE_INF, false, my_stack, my_spill, spill_stride, &hm);
                                if (tree_mis_lit(S, li, hit, hm, mo, md)) Le1 = Le1 + Bc;
                            }
                            if (arrays)
                                Ld = Ld + (F3{0, 0, 0} + F3{1.f, 1.f, 1.f} * sdiv(Le1, 1.f));  // the slot k_shadow leaves: L_old (0) + beta (1) * Ld / lightPdf (1)
                            else
                                all = all + Le1;  // `L += EstimateDirect(...)`
                        }
                        if (arrays) all = all + sdiv(Ld, float(n));  // `L += Ld / nSamples`
                    }
                    me.L = me.L + all;
                }
                ++visited;
                me.ad_r = 0.f;
                for (int q = 0; q < NT; ++q) me.ad_t[q] = 0.f, me.t_has_diff[q] = false;
                me.n_t = 0;
                me.pdf_t = 1.f;
                me.stage = 0;
                if (depth + 1 >= 5) {  // `if (depth + 1 < maxDepth)`: no recursion below the fifth vertex
                    ret = me.L;
                    descend = false;
                    --depth;
                    if (depth < 0) break;
                    continue;
                }
                // ---- SpecularReflect / SpecularTransmit decided now (their Get2D samples are drawn in order but never used: one
                // matching lobe each); the transmitted ray waits until the reflection subtree is back
                const F3 wo_w = is.wo, ns = is.sn;
                const F3 wo = to_local(bsdf, wo_w);
                F3 dndx = F3{0, 0, 0}, dndy = F3{0, 0, 0}, dwodx = F3{0, 0, 0}, dwody = F3{0, 0, 0};
                float dDNdx = 0, dDNdy = 0;
                if (TEX && has_diff) {
                    dndx = is.dndu * td.dudx + is.dndv * td.dvdx;
                    dndy = is.dndu * td.dudy + is.dndv * td.dvdy;
                    dwodx = -rdiff.rxd - wo_w, dwody = -rdiff.ryd - wo_w;
                    dDNdx = dot(dwodx, ns) + dot(wo_w, dndx);
                    dDNdy = dot(dwody, ns) + dot(wo_w, dndy);
                }
                F3 r_o = F3{0, 0, 0}, r_d = F3{0, 0, 1};
                RayDiff r_rd = rdiff;
                const bool refl_lobe = bsdf.has_spec && !(bsdf.mtype == kMatGlass && is_black(bsdf.kr));
                if (refl_lobe && wo.z != 0) {
                    const F3 wi_l = F3{-wo.x, -wo.y, wo.z};
                    const float fr = bsdf.mtype == kMatMirror ? 1.f : fr_dielectric(wi_l.z, 1.f, bsdf.eta);
                    const F3 f = sdiv(F3{fr, fr, fr} * bsdf.kr, fabsf(wi_l.z));
                    const F3 wi = to_world(bsdf, wi_l);
                    const float ad = absdot(wi, ns);
                    if (!is_black(f) && ad != 0.f) {
                        me.f_r = f;
                        me.ad_r = ad;
                        r_o = offset_ray_origin(is.p, is.perr, is.n, wi);
                        r_d = wi;
                        if (TEX && has_diff) {
                            r_rd.rxo = is.p + dpdx, r_rd.ryo = is.p + dpdy;
                            r_rd.rxd = wi - dwodx + 2.f * (dot(wo_w, ns) * dndx + dDNdx * ns);
                            r_rd.ryd = wi - dwody + 2.f * (dot(wo_w, ns) * dndy + dDNdy * ns);
                        }
                    }
                }
                // the BSDF_TRANSMISSION | BSDF_SPECULAR lobes in the order the material added them: uber's pass-through, glass's
                // SpecularTransmission (allowMultipleLobes = false) or uber's Kt lobe
                const bool glass_t = bsdf.has_spec && bsdf.mtype == kMatGlass && !is_black(bsdf.kt);
                for (int lobe = 0; lobe < 2; ++lobe) {
                    const bool present = lobe == 0 ? bsdf.has_t0 : (glass_t || bsdf.has_t1);
                    if (!present || me.n_t >= NT) continue;
                    const int q = me.n_t++;
                    if (wo.z == 0) continue;
                    // SpecularTransmission::Sample_f, reflection.cpp:154-170
                    const F3 T = lobe == 0 ? bsdf.t0 : bsdf.kt;
                    const float eta_a = 1.f, eta_b = lobe == 0 ? 1.f : bsdf.eta;
                    const bool entering = wo.z > 0;
                    const float eta_i = entering ? eta_a : eta_b, eta_t = entering ? eta_b : eta_a;
                    const F3 n = (wo.z < 0.f) ? -F3{0, 0, 1} : F3{0, 0, 1};
                    const float eta = eta_i / eta_t;
                    const float cos_i = dot(n, wo);
                    const float sin2_i = mx(0.f, 1 - cos_i * cos_i);
                    const float sin2_t = eta * eta * sin2_i;
                    if (!(sin2_t >= 1)) {
                        const float cos_t = sqrtf(1 - sin2_t);
                        const F3 wi_l = eta * -wo + (eta * cos_i - cos_t) * n;
                        F3 ft = T * (1.f - fr_dielectric(wi_l.z, eta_a, eta_b));
                        ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
                        const F3 f = sdiv(ft, fabsf(wi_l.z));
                        const F3 wi = to_world(bsdf, wi_l);
                        const float ad = absdot(wi, ns);
                        if (!is_black(f) && ad != 0.f) {
                            me.f_t[q] = f;
                            me.ad_t[q] = ad;
                            me.t_o[q] = offset_ray_origin(is.p, is.perr, is.n, wi);
                            me.t_d[q] = wi;
                            if (TEX && has_diff) {  // directprogressiveintegrator.cpp:203-233; `Float eta = bsdf.eta`: BSDF::eta
                                me.t_has_diff[q] = true;
                                float e2 = bsdf.path_eta;
                                const F3 w = -wo_w;
                                if (dot(wo_w, ns) < 0) e2 = 1.f / e2;
                                const float mu = e2 * dot(w, ns) - dot(wi, ns);
                                const float dmudx = (e2 - (e2 * e2 * dot(w, ns)) / dot(wi, ns)) * dDNdx;
                                const float dmudy = (e2 - (e2 * e2 * dot(w, ns)) / dot(wi, ns)) * dDNdy;
                                me.t_rd[q].rxo = is.p + dpdx, me.t_rd[q].ryo = is.p + dpdy;
                                me.t_rd[q].rxd = wi + e2 * dwodx - (mu * dndx + dmudx * ns);
                                me.t_rd[q].ryd = wi + e2 * dwody - (mu * dndy + dmudy * ns);
                            }
                        }
                    }
                }
                // SpecularReflect: its Get2D first, then the subtree (or nothing: `return Spectrum(0.f)`)
                (void)pcg_float(rng);
                (void)pcg_float(rng);
                if (me.ad_r != 0.f) {
                    me.stage = 1;
                    ro = r_o, rd = r_d;
                    if (TEX) rdiff = r_rd;
                    // (has_diff stays as it is: a reflected ray has differentials iff its parent had)
                    ++depth;
                    descend = true;
                    continue;
                }
                ret = F3{0, 0, 0};
                me.stage = 1;
                descend = false;
                // (falls through to the return handling below with depth unchanged)
            } else {
                // a subtree returned into lv[depth]
            }
            // ---- back in lv[depth] with `ret`
            TreeLevel<NT> &me = lv[depth];
            if (me.stage == 1) {
                // L += SpecularReflect(...) = f * Li(rd) * AbsDot(wi, ns) / pdf (1), or 0
                F3 R = F3{0, 0, 0};
                if (me.ad_r != 0.f) R = sdiv(me.f_r * ret * me.ad_r, 1.f);
                me.L = me.L + R;
                // SpecularTransmit: Get2D, then its subtree. BSDF::Sample_f picks `comp = min(floor(u[0] * matchingComps),
                // matchingComps - 1)` (reflection.cpp:733-735) and, the lobe being specular, only divides the pdf by matchingComps
                const float ut = pcg_float(rng);
                (void)pcg_float(rng);
                if (NT > 1 && me.n_t > 1) {
                    const int c0 = int(floorf(ut * float(me.n_t))), comp = c0 < me.n_t - 1 ? c0 : me.n_t - 1;
                    me.pdf_t = 1.f / float(me.n_t);
                    if (comp != 0) {
                        me.f_t[0] = me.f_t[comp], me.ad_t[0] = me.ad_t[comp], me.t_o[0] = me.t_o[comp], me.t_d[0] = me.t_d[comp];
                        me.t_rd[0] = me.t_rd[comp], me.t_has_diff[0] = me.t_has_diff[comp];
                    }
                }
                me.stage = 2;
                if (me.ad_t[0] != 0.f) {
                    ro = me.t_o[0], rd = me.t_d[0];
                    if (TEX && me.t_has_diff[0]) rdiff = me.t_rd[0];
                    ++depth;
                    descend = true;
                    continue;
                }
                ret = F3{0, 0, 0};
            }
            // stage 2: L += SpecularTransmit(...)
            {
                F3 T = F3{0, 0, 0};
                if (me.ad_t[0] != 0.f) T = sdiv(me.f_t[0] * ret * me.ad_t[0], me.pdf_t);
                me.L = me.L + T;
                ret = me.L;
                descend = false;
                --depth;
                if (depth < 0) break;
            }
        }
        F3 L = ret;
        const float y = lum_y(L);
        if (is_nan(L.x) || is_nan(L.y) || is_nan(L.z))
            L = F3{0, 0, 0};
        else if (double(y) < -1e-5)
            L = F3{0, 0, 0};
        else if (is_inf(y))
            L = F3{0, 0, 0};
        double *out = film_rgbw + 4 * (size_t(py - S.crop_y0) * fw + (px - S.crop_x0));
        out[0] += double(L.x);
        out[1] += double(L.y);
        out[2] += double(L.z);
        out[3] += 1.0;
    }
}

void launch_direct_tree(const DScene &S, const PassDesc &P, const PassBuffers &B, double *film_rgbw, const LaunchCfg &cfg) {
    const dim3 grid(grid_blocks(P.n_paths, cfg.n_cus, 4));
    const bool tex = S.textured_materials || S.n_textures > 0;
    if (S.has_uber_trans) {
        if (tex)
            hipLaunchKernelGGL((k_direct_tree<true, 2>), grid, dim3(kBlock), 0, cfg.stream, S, P, B, film_rgbw);
        else
            hipLaunchKernelGGL((k_direct_tree<false, 2>), grid, dim3(kBlock), 0, cfg.stream, S, P, B, film_rgbw);
    } else if (tex)
        hipLaunchKernelGGL((k_direct_tree<true, 1>), grid, dim3(kBlock), 0, cfg.stream, S, P, B, film_rgbw);
    else
        hipLaunchKernelGGL((k_direct_tree<false, 1>), grid, dim3(kBlock), 0, cfg.stream, S, P, B, film_rgbw);
}

void launch_direct_generate(const DScene &S, const PassDesc &P, const PassBuffers &B, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_direct_generate, dim3(grid_blocks(P.n_paths, cfg.n_cus, 8)), dim3(kBlock), 0, cfg.stream, S, P, B);
}
void launch_direct_shade(const DScene &S, const PassDesc &P, const PassBuffers &B, int depth, uint32_t max_rays, const LaunchCfg &cfg) {
    const dim3 grid(grid_blocks(max_rays, cfg.n_cus, 2));
    if (S.textured_materials)
        hipLaunchKernelGGL((k_direct_shade<true>), grid, dim3(kBlock), 0, cfg.stream, S, P, B, depth, B.queue_cap);
    else
        hipLaunchKernelGGL((k_direct_shade<false>), grid, dim3(kBlock), 0, cfg.stream, S, P, B, depth, B.queue_cap);
}
void launch_direct_miss(const DScene &S, const PassBuffers &B, int depth, uint32_t max_rays, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_direct_miss, dim3(grid_blocks(max_rays, cfg.n_cus, 8)), dim3(kBlock), 0, cfg.stream, S, B, depth);
}
void launch_direct_fold(const DScene &S, const PassDesc &P, const PassBuffers &B, double *film_rgbw, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_direct_fold, dim3(grid_blocks(P.n_paths / uint32_t(P.kc), cfg.n_cus, 8)), dim3(kBlock), 0, cfg.stream, S, P, B, film_rgbw);
}

}  // namespace iile
