// finalize.cpp — camera matrices, film bounds, Halton tables and the flattened
// iile_scene_desc. Citations are relative to /root/reference/src.
#include <cstdio>

#include "host_scene.h"

namespace iile {
namespace {

// PCG32 with the reference's default state/stream (core/rng.h:62-64, 143-156).
struct Pcg32 {
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    uint32_t next() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = uint32_t(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = uint32_t(old >> 59u);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31));
    }
    uint32_t bounded(uint32_t b) {  // core/rng.h:76-82
        uint32_t threshold = (~b + 1u) % b;
        while (true) {
            uint32_t r = next();
            if (r >= threshold) return r % b;
        }
    }
};

// samplers/halton.cpp:46-63
void extended_gcd(uint64_t a, uint64_t b, int64_t *x, int64_t *y) {
    if (b == 0) {
        *x = 1;
        *y = 0;
        return;
    }
    int64_t d = int64_t(a / b), xp, yp;
    extended_gcd(b, a % b, &xp, &yp);
    *x = yp;
    *y = xp - (d * yp);
}
uint64_t multiplicative_inverse(int64_t a, int64_t n) {
    int64_t x, y;
    extended_gcd(uint64_t(a), uint64_t(n), &x, &y);
    int64_t r = x - (x / n) * n;  // Mod(), core/pbrt.h:310-314
    return uint64_t(r < 0 ? r + n : r);
}

}  // namespace

// Digit permutations for the scrambled radical inverse
// (core/lowdiscrepancy.cpp:2490-2504, core/sampling.h:150-157). The reference
// shuffles all 1000 prime bases from one default-seeded PCG32 stream; the
// stream is consumed base by base, so a prefix of the bases yields a prefix of
// the table. PathIntegrator with maxdepth d touches at most 5 + 8*d dimensions
// (7 per bounce plus one Russian-roulette sample per bounce past the third).
void build_halton_tables(HostScene *scene) {
    const int n_dims = std::max(64, 5 + 8 * (scene->max_depth + 1) + 2);
    scene->primes.clear();
    scene->prime_sums.clear();
    int sum = 0;
    for (int cand = 2; int(scene->primes.size()) < n_dims; ++cand) {
        bool is_prime = true;
        for (int d = 2; d * d <= cand; ++d)
            if (cand % d == 0) {
                is_prime = false;
                break;
            }
        if (!is_prime) continue;
        scene->primes.push_back(cand);
        scene->prime_sums.push_back(sum);
        sum += cand;
    }
    scene->perms.assign(sum, 0);
    Pcg32 rng;
    uint16_t *p = scene->perms.data();
    for (int i = 0; i < n_dims; ++i) {
        const int count = scene->primes[i];
        for (int j = 0; j < count; ++j) p[j] = uint16_t(j);
        for (int j = 0; j < count; ++j) {
            int other = j + int(rng.bounded(uint32_t(count - j)));
            std::swap(p[j], p[other]);
        }
        p += count;
    }
}

bool finalize_scene(HostScene *s, std::string *err) {
    if (s->prims.empty()) {
        if (err) *err = "scene has no primitives";
        return false;
    }
    build_bvh(s);
    build_halton_tables(s);
    if (s->sampler_name == "sobol") {
        // GlobalSampler(RoundUpPow2(samplesPerPixel)), samplers/sobol.h:59-64 — before anything reads the sample count
        int64_t r = 1;
        while (r < s->spp) r <<= 1;
        if (r != s->spp) fprintf(stderr, "Warning: Non power-of-two sample count rounded up to %lld for SobolSampler.\n", (long long)r);
        s->spp = int(r);
    }

    iile_scene_desc &d = s->desc;
    std::memset(&d, 0, sizeof(d));
    d.n_nodes = int(s->nodes.size());
    d.nodes = s->nodes.data();
    d.n_prims = int(s->prims.size());
    d.prim_flags = s->o_flags.data();
    d.prim_material = s->o_material.data();
    d.prim_light = s->o_light.data();
    d.prim_shape = s->o_shape.data();
    d.prim_alpha = s->o_alpha.empty() ? nullptr : s->o_alpha.data();
    d.tri_p = s->o_tri_p.data();
    d.tri_n = s->o_tri_n.data();
    d.tri_uv = s->o_tri_uv.data();
    d.n_spheres = int(s->spheres.size());
    d.spheres = s->spheres.data();
    d.n_materials = int(s->materials.size());
    d.materials = s->materials.data();
    // Light::Preprocess of DistantLight (distant.cpp:63-65): Scene::WorldBound().BoundingSphere
    // (scene.h:56, geometry.h:808-811); the world bound is the BVH root's
    if (!s->nodes.empty()) {
        const iile_bvh_node &root = s->nodes[0];
        const V3 pmin(root.bmin[0], root.bmin[1], root.bmin[2]), pmax(root.bmax[0], root.bmax[1], root.bmax[2]);
        const V3 c = div(pmin + pmax, 2.f);
        const bool inside = c.x >= pmin.x && c.x <= pmax.x && c.y >= pmin.y && c.y <= pmax.y && c.z >= pmin.z && c.z <= pmax.z;
        const float radius = inside ? length(c - pmax) : 0.f;
        for (iile_light &lt : s->lights)
            if (lt.type == IILE_LIGHT_DISTANT || lt.type == IILE_LIGHT_INFINITE) lt.world_radius = radius;
    }
    for (size_t i = 0; i < s->o_light.size(); ++i)  // a triangle emitter's primitive, in BVH order
        if (s->o_light[i] >= 0 && s->lights[size_t(s->o_light[i])].type == IILE_LIGHT_AREA_TRIANGLE)
            s->lights[size_t(s->o_light[i])].prim = int(i);
    d.n_lights = int(s->lights.size());
    d.lights = s->lights.data();

    // image textures: concatenate the pyramids
    s->o_textures.clear();
    s->o_texels.clear();
    for (const HostTexture &ht : s->textures) {
        iile_texture t = ht.t;
        const int64_t base = int64_t(s->o_texels.size() / 3);
        for (int l = 0; l < t.n_levels; ++l) t.level_offset[l] += base;
        s->o_texels.insert(s->o_texels.end(), ht.texels.begin(), ht.texels.end());
        s->o_textures.push_back(t);
    }
    d.n_env_dist = int64_t(s->env_dist.size());
    d.env_dist = s->env_dist.data();
    d.n_textures = int(s->o_textures.size());
    d.textures = s->o_textures.data();
    d.n_texels = int64_t(s->o_texels.size() / 3);
    d.texels = s->o_texels.data();
    ewa_weight_lut(d.ewa_lut);

    // Film, core/film.cpp:45-82
    iile_film_desc &f = d.film;
    f.xres = s->xres;
    f.yres = s->yres;
    f.crop_x0 = int(std::ceil(s->xres * s->crop[0]));
    f.crop_y0 = int(std::ceil(s->yres * s->crop[2]));
    f.crop_x1 = int(std::ceil(s->xres * s->crop[1]));
    f.crop_y1 = int(std::ceil(s->yres * s->crop[3]));
    f.filter_rx = s->filter_rx;
    f.filter_ry = s->filter_ry;
    f.scale = s->film_scale;
    f.max_sample_luminance = s->max_sample_luminance;
    // Film::GetSampleBounds, core/film.cpp:76-82
    f.samp_x0 = int(std::floor(float(f.crop_x0) + 0.5f - f.filter_rx));
    f.samp_y0 = int(std::floor(float(f.crop_y0) + 0.5f - f.filter_ry));
    f.samp_x1 = int(std::ceil(float(f.crop_x1) - 0.5f + f.filter_rx));
    f.samp_y1 = int(std::ceil(float(f.crop_y1) - 0.5f + f.filter_ry));

    // Film::filterTable, film.cpp:65-74, over the filters of src/filters/{box,gaussian,mitchell,sinc,triangle}
    {
        const float rx = s->filter_rx, ry = s->filter_ry, p0 = s->filter_p0, p1 = s->filter_p1;
        const std::string &fn = s->filter_name;
        const float exp_x = std::exp(-p0 * rx * rx), exp_y = std::exp(-p0 * ry * ry);  // gaussian.h:53-54
        auto gaussian = [&](float dd, float expv) { return std::max(0.f, float(std::exp(-p0 * dd * dd) - expv)); };
        auto mitchell = [&](float x) {  // mitchell.h:53-63
            const float B = p0, C = p1;
            x = std::abs(2 * x);
            if (x > 1)
                return ((-B - 6 * C) * x * x * x + (6 * B + 30 * C) * x * x + (-12 * B - 48 * C) * x + (8 * B + 24 * C)) * (1.f / 6.f);
            return ((12 - 9 * B - 6 * C) * x * x * x + (-18 + 12 * B + 6 * C) * x * x + (6 - 2 * B)) * (1.f / 6.f);
        };
        auto sinc = [](float x) {  // sinc.h:53-57
            x = std::abs(x);
            if (x < 1e-5) return 1.f;
            return std::sin(kPi * x) / (kPi * x);
        };
        auto windowed_sinc = [&](float x, float radius) {  // sinc.h:58-63
            x = std::abs(x);
            if (x > radius) return 0.f;
            float lanczos = sinc(x / p0);
            return sinc(x) * lanczos;
        };
        const float inv_rx = 1.f / rx, inv_ry = 1.f / ry;  // Filter::invRadius
        int offset = 0;
        for (int y = 0; y < 16; ++y)
            for (int x = 0; x < 16; ++x, ++offset) {
                const float px = (x + 0.5f) * rx / 16, py = (y + 0.5f) * ry / 16;
                float v;
                if (fn == "gaussian")
                    v = gaussian(px, exp_x) * gaussian(py, exp_y);
                else if (fn == "mitchell")
                    v = mitchell(px * inv_rx) * mitchell(py * inv_ry);
                else if (fn == "sinc")
                    v = windowed_sinc(px, rx) * windowed_sinc(py, ry);
                else if (fn == "triangle")
                    v = std::max(0.f, rx - std::abs(px)) * std::max(0.f, ry - std::abs(py));
                else
                    v = 1.f;
                d.film_filter_table[offset] = v;
            }
        d.film_filter_wide = (fn == "box" && rx == 0.5f && ry == 0.5f) ? 0 : 1;
    }

    // Camera, cameras/perspective.cpp:297-330 and core/camera.h:90-111
    float frame = s->frame_aspect > 0 ? s->frame_aspect : float(s->xres) / float(s->yres);
    float sw[4];  // pMin.x, pMax.x, pMin.y, pMax.y
    if (frame > 1.f) {
        sw[0] = -frame;
        sw[1] = frame;
        sw[2] = -1.f;
        sw[3] = 1.f;
    } else {
        sw[0] = -1.f;
        sw[1] = 1.f;
        sw[2] = -1.f / frame;
        sw[3] = 1.f / frame;
    }
    if (s->has_screen_window)
        for (int i = 0; i < 4; ++i) sw[i] = s->screen_window[i];
    Xform camera_to_screen = xf_perspective(s->fov, 1e-2f, 1000.f);
    Xform screen_to_raster = xf_scale(float(s->xres), float(s->yres), 1) *
                             xf_scale(1 / (sw[1] - sw[0]), 1 / (sw[2] - sw[3]), 1) *
                             xf_translate(V3(-sw[0], -sw[3], 0));
    Xform raster_to_screen = inverse(screen_to_raster);
    Xform raster_to_camera = inverse(camera_to_screen) * raster_to_screen;
    std::memcpy(d.camera.raster_to_camera, raster_to_camera.m.m, sizeof(float) * 16);
    std::memcpy(d.camera.camera_to_world, s->camera_to_world.m.m, sizeof(float) * 16);
    {  // PerspectiveCamera ctor, perspective.cpp:58-62
        V3 o = raster_to_camera.point(V3(0, 0, 0));
        V3 dx = raster_to_camera.point(V3(1, 0, 0)) - o, dy = raster_to_camera.point(V3(0, 1, 0)) - o;
        d.camera.dx_camera[0] = dx.x, d.camera.dx_camera[1] = dx.y, d.camera.dx_camera[2] = dx.z;
        d.camera.dy_camera[0] = dy.x, d.camera.dy_camera[1] = dy.y, d.camera.dy_camera[2] = dy.z;
    }
    d.camera.lens_radius = s->lens_radius;
    d.camera.focal_distance = s->focal_distance;
    d.camera.shutter_open = s->shutter_open;
    d.camera.shutter_close = s->shutter_close;

    // HaltonSampler ctor, samplers/halton.cpp:65-93 (kMaxResolution = 128)
    iile_halton &h = d.halton;
    h.spp = s->spp;
    int res[2] = {f.samp_x1 - f.samp_x0, f.samp_y1 - f.samp_y0};
    for (int i = 0; i < 2; ++i) {
        int base = (i == 0) ? 2 : 3;
        int scale = 1, exp = 0;
        while (scale < std::min(res[i], 128)) {
            scale *= base;
            ++exp;
        }
        h.base_scales[i] = scale;
        h.base_exponents[i] = exp;
    }
    h.sample_stride = h.base_scales[0] * h.base_scales[1];
    h.mult_inverse[0] = int(multiplicative_inverse(h.base_scales[1], h.base_scales[0]));
    h.mult_inverse[1] = int(multiplicative_inverse(h.base_scales[0], h.base_scales[1]));
    h.n_dims = int(s->primes.size());
    h.perms = s->perms.data();
    h.primes = s->primes.data();
    h.prime_sums = s->prime_sums.data();
    h.n_perms = int(s->perms.size());
    h.sample_at_pixel_center = s->sample_at_pixel_center ? 1 : 0;

    // SobolSampler ctor, samplers/sobol.h:57-75
    iile_sobol &sb = d.sobol;
    std::memset(&sb, 0, sizeof(sb));
    if (s->sampler_name == "sobol") {
        auto round_up_pow2 = [](int64_t v) {  // pbrt.h:348-357
            int64_t r = 1;
            while (r < v) r <<= 1;
            return r;
        };
        sb.enabled = 1;
        sb.spp = int(round_up_pow2(s->spp));  // (set before the Halton state above was filled: see finalize_scene)
        sb.resolution = int(round_up_pow2(std::max(res[0], res[1])));
        int lg = 0;
        while ((1 << lg) < sb.resolution) ++lg;
        sb.log2_resolution = lg;
        if (lg < 1 || lg > 16 || (uint64_t(sb.spp) << (2 * lg)) > (uint64_t(1) << 32)) {
            *err = "Sampler \"sobol\": pixelsamples x resolution^2 exceeds the 32-bit sample index of the GPU path";
            return false;
        }
        sb.n_dims = std::min(sobol_num_dimensions(), 128);
        s->sobol_matrices.resize(size_t(sb.n_dims) * 32);
        for (int dim = 0; dim < sb.n_dims; ++dim) {
            uint32_t cols[52];
            sobol_columns32(dim, cols);
            std::memcpy(&s->sobol_matrices[size_t(dim) * 32], cols, 32 * sizeof(uint32_t));
        }
        sb.matrices32 = s->sobol_matrices.data();
        uint64_t vdc[52], inv[52];
        sobol_vdc(lg, vdc, inv);
        for (int c = 0; c < 32; ++c) sb.vdc[c] = uint32_t(vdc[c]), sb.vdc_inv[c] = uint32_t(inv[c]);
    }

    d.integrator.max_depth = s->max_depth;
    d.integrator.rr_threshold = s->rr_threshold;
    {   // pixelBounds = Intersect(camera->film->GetSampleBounds(), Bounds2i{{pb[0], pb[2]}, {pb[1], pb[3]}}), path.cpp:217-227
        int32_t *pb = d.integrator.pixel_bounds;
        pb[0] = d.film.samp_x0, pb[1] = d.film.samp_y0, pb[2] = d.film.samp_x1, pb[3] = d.film.samp_y1;
        if (s->has_pixel_bounds && !s->integrator_iispt) {
            const int *g = s->pixel_bounds_given;
            // (Bounds2i's two-point constructor orders the corners: geometry.h:690-693)
            const int gx0 = std::min(g[0], g[1]), gx1 = std::max(g[0], g[1]), gy0 = std::min(g[2], g[3]), gy1 = std::max(g[2], g[3]);
            pb[0] = std::max(pb[0], gx0), pb[1] = std::max(pb[1], gy0), pb[2] = std::min(pb[2], gx1), pb[3] = std::min(pb[3], gy1);
            if ((pb[2] - pb[0]) * (pb[3] - pb[1]) == 0) std::fprintf(stderr, "Error: Degenerate \"pixelbounds\" specified.\n");
            // (empty bounds — no pixel is inside — in one form that is not "all zero": iile_scene_create reads all zero as "not given")
            if (pb[2] <= pb[0] || pb[3] <= pb[1]) pb[0] = pb[1] = 0, pb[2] = pb[3] = -1;
        }
    }
    d.integrator.light_strategy = s->light_strategy == "uniform" ? IILE_LIGHTS_UNIFORM : (s->light_strategy == "power" ? IILE_LIGHTS_POWER : IILE_LIGHTS_SPATIAL);
    // Light::Power().y() of every light (ComputeLightPowerDistribution, integrator.cpp:217-225)
    for (int i = 0; i < IILE_MAX_LIGHTS; ++i) d.integrator.light_power[i] = 0;
    for (size_t i = 0; i < s->lights.size() && i < size_t(IILE_MAX_LIGHTS); ++i) {
        const iile_light &lt = s->lights[i];
        float pw[3] = {0, 0, 0};
        const float wr = lt.world_radius;
        for (int c = 0; c < 3; ++c) {
            const float L = lt.lemit[c];
            switch (lt.type) {
            case IILE_LIGHT_DIFFUSE_AREA: {  // diffuse.cpp:64-66 with Sphere::Area (sphere.cpp:217)
                const iile_sphere &sp = s->spheres[size_t(lt.sphere)];
                const float area = sp.phi_max * sp.radius * (sp.zmax - sp.zmin);
                pw[c] = (lt.two_sided ? 2 : 1) * L * area * kPi;
                break;
            }
            case IILE_LIGHT_AREA_TRIANGLE: {  // Triangle::Area, triangle.cpp:546-552
                const float *tp = &s->o_tri_p[9 * size_t(lt.prim)];
                const V3 p0(tp[0], tp[1], tp[2]), p1(tp[3], tp[4], tp[5]), p2(tp[6], tp[7], tp[8]);
                const float area = float(0.5 * length(cross(p1 - p0, p2 - p0)));
                pw[c] = (lt.two_sided ? 2 : 1) * L * area * kPi;
                break;
            }
            case IILE_LIGHT_POINT: pw[c] = 4 * kPi * L; break;  // point.cpp:55
            case IILE_LIGHT_SPOT: pw[c] = L * 2 * kPi * (1 - .5f * (lt.cos_falloff_start + lt.cos_total_width)); break;  // spot.cpp:75-77
            case IILE_LIGHT_DISTANT: pw[c] = L * kPi * wr * wr; break;  // distant.cpp:61-63
            default: {  // infinite.cpp:86-90: Pi r^2 * Lmap->Lookup((.5, .5), .5)
                float rgb[3];
                mip_lookup_width(s->textures[size_t(lt.env_tex)], .5f, .5f, .5f, rgb);
                pw[c] = kPi * wr * wr * rgb[c];
            }
            }
        }
        d.integrator.light_power[i] = 0.212671f * pw[0] + 0.715160f * pw[1] + 0.072169f * pw[2];
    }

    // IISPT probe pass: CreateHemisphericCamera's film (hemispheric.cpp:131-147) and CreateIISPTdIntegrator's
    // sampler and depth (iispt_d.cpp:492-527)
    {
        iile_probe_setup &pr = d.probe;
        std::memset(&pr, 0, sizeof(pr));
        pr.hemi_size = 32;
        pr.max_depth = 3;
        iile_film_desc &pf = pr.film;
        pf.xres = pf.yres = pr.hemi_size;
        pf.crop_x0 = pf.crop_y0 = 0;
        pf.crop_x1 = pf.crop_y1 = pr.hemi_size;
        pf.filter_rx = pf.filter_ry = 2.f;
        pf.scale = 1.f;
        pf.max_sample_luminance = std::numeric_limits<float>::infinity();
        pf.samp_x0 = int(std::floor(float(pf.crop_x0) + 0.5f - pf.filter_rx));
        pf.samp_y0 = int(std::floor(float(pf.crop_y0) + 0.5f - pf.filter_ry));
        pf.samp_x1 = int(std::ceil(float(pf.crop_x1) - 0.5f + pf.filter_rx));
        pf.samp_y1 = int(std::ceil(float(pf.crop_y1) - 0.5f + pf.filter_ry));
        const float alpha = 2.f, exp_r = std::exp(-alpha * 2.f * 2.f);
        auto gaussian = [&](float dd) { return std::max(0.f, float(std::exp(-alpha * dd * dd) - exp_r)); };
        int offset = 0;
        for (int y = 0; y < 16; ++y)
            for (int x = 0; x < 16; ++x, ++offset)
                pr.filter_table[offset] = gaussian((x + 0.5f) * 2.f / 16) * gaussian((y + 0.5f) * 2.f / 16);
        const int pres[2] = {pf.samp_x1 - pf.samp_x0, pf.samp_y1 - pf.samp_y0};
        for (int i = 0; i < 2; ++i) {
            const int base = (i == 0) ? 2 : 3;
            int scale = 1, exp = 0;
            while (scale < std::min(pres[i], 128)) {
                scale *= base;
                ++exp;
            }
            pr.base_scales[i] = scale;
            pr.base_exponents[i] = exp;
        }
        pr.sample_stride = pr.base_scales[0] * pr.base_scales[1];
        pr.mult_inverse[0] = int(multiplicative_inverse(pr.base_scales[1], pr.base_scales[0]));
        pr.mult_inverse[1] = int(multiplicative_inverse(pr.base_scales[0], pr.base_scales[1]));
    }
    return true;
}

}  // namespace iile
