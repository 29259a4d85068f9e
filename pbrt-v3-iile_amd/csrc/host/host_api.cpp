// host_api.cpp — C ABI of libiile_host.so (see include/iile_host.h).
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <exception>

#include "../../../include/iile_host.h"
#include "host_scene.h"

struct iile_host_scene {
    iile::HostScene s;
};

namespace {
thread_local std::string g_err;
}

extern "C" {

const char *iile_host_last_error(void) { return g_err.c_str(); }

int iile_host_load_pbrt(const char *path, const iile_host_overrides *ov, iile_host_scene **out) {
    if (!path || !out) {
        g_err = "iile_host_load_pbrt: null argument";
        return 1;
    }
    iile_host_scene *hs = nullptr;
    // no C++ exception may cross the C boundary (a malformed file can make a vector throw length_error / bad_alloc)
    try {
        hs = new iile_host_scene;
        std::string err;
        if (!iile::load_pbrt_file(path, &hs->s, &err)) {
            g_err = err;
            delete hs;
            return 2;
        }
        if (ov) {
            if (ov->quick_render) {
                hs->s.xres = std::max(1, hs->s.xres / 4);
                hs->s.yres = std::max(1, hs->s.yres / 4);
                hs->s.spp = 1;
                // Light::nSamples of area and infinite lights: max(1, nSamples / 4) (diffuse.cpp:143, infinite.cpp:183)
                for (iile_light &lt : hs->s.lights)
                    if (lt.n_samples > 1) lt.n_samples = std::max(1, lt.n_samples / 4);
            }
            if (ov->xres > 0) hs->s.xres = ov->xres;
            if (ov->yres > 0) hs->s.yres = ov->yres;
            if (ov->spp > 0) hs->s.spp = ov->spp;
            if (ov->max_depth > 0) hs->s.max_depth = ov->max_depth;
            if (ov->sampler == IILE_SAMPLER_HALTON) hs->s.sampler_name = "halton";
            if (ov->sampler == IILE_SAMPLER_SOBOL) hs->s.sampler_name = "sobol", hs->s.sample_at_pixel_center = false;
            static const char *const kSplit[] = {nullptr, "sah", "hlbvh", "middle", "equal"};
            if (ov->accel_split >= IILE_SPLIT_SAH && ov->accel_split <= IILE_SPLIT_EQUAL) hs->s.accel_split = kSplit[ov->accel_split];
            hs->s.bvh_hook = ov->bvh_build;
        }
        if (!iile::finalize_scene(&hs->s, &err)) {
            g_err = err;
            delete hs;
            return 3;
        }
    } catch (const std::exception &e) {
        g_err = std::string("iile_host_load_pbrt: ") + e.what();
        delete hs;
        return 4;
    }
    *out = hs;
    return 0;
}

const iile_scene_desc *iile_host_scene_desc(const iile_host_scene *scene) {
    return scene ? &scene->s.desc : nullptr;
}

const iile_film_desc *iile_host_scene_film(const iile_host_scene *scene) {
    return scene ? &scene->s.desc.film : nullptr;
}

const char *iile_host_scene_film_filename(const iile_host_scene *scene) { return scene ? scene->s.film_filename.c_str() : ""; }

int iile_host_scene_light(const iile_host_scene *scene, int32_t index, iile_light *out) {
    if (!scene || !out || index < 0 || size_t(index) >= scene->s.lights.size()) {
        g_err = "iile_host_scene_light: null argument or light index out of range";
        return 1;
    }
    *out = scene->s.lights[size_t(index)];
    return 0;
}

int iile_host_scene_get_info(const iile_host_scene *scene, iile_host_scene_info *info) {
    if (!scene || !info) {
        g_err = "iile_host_scene_get_info: null argument";
        return 1;
    }
    const iile::HostScene &s = scene->s;
    info->n_prims = int(s.prims.size());
    info->n_spheres = int(s.spheres.size());
    info->n_triangles = info->n_prims - info->n_spheres;
    info->n_meshes = s.n_meshes;
    info->n_nodes = int(s.nodes.size());
    info->n_interior_nodes = s.n_interior;
    info->n_leaf_nodes = s.n_leaf;
    info->n_materials = int(s.materials.size());
    info->n_lights = int(s.lights.size());
    info->xres = s.xres;
    info->yres = s.yres;
    info->spp = s.spp;
    info->max_depth = s.max_depth;
    info->probe_hemi_size = s.desc.probe.hemi_size;
    info->integrator = s.integrator_iispt ? IILE_INTEGRATOR_IISPT : IILE_INTEGRATOR_PATH;
    return 0;
}

void iile_host_scene_free(iile_host_scene *scene) { delete scene; }

int iile_host_sobol_matrices(int32_t n_dims, uint32_t *m32, uint64_t *m64) {
    if (n_dims < 0 || n_dims > iile::sobol_num_dimensions()) {
        g_err = "iile_host_sobol_matrices: dimension count out of range";
        return 1;
    }
    for (int d = 0; d < n_dims; ++d) {
        if (m32) iile::sobol_columns32(d, m32 + 52 * size_t(d));
        if (m64) iile::sobol_columns64(d, reinterpret_cast<uint64_t *>(m64) + 52 * size_t(d));
    }
    return 0;
}
int iile_host_sobol_vdc(int32_t log2_resolution, uint64_t *vdc52, uint64_t *vdc_inv52) {
    if (!vdc52 || !vdc_inv52 || !iile::sobol_vdc(log2_resolution, reinterpret_cast<uint64_t *>(vdc52), reinterpret_cast<uint64_t *>(vdc_inv52))) {
        g_err = "iile_host_sobol_vdc: log2_resolution must lie in 1 .. 16";
        return 1;
    }
    return 0;
}

// Film::to_rgb_array, /root/reference/src/core/film.cpp:187-225, with the
// XYZ->RGB matrix of src/core/spectrum.h:56-60. No splats on this path.
int iile_host_film_to_rgb(const iile_film_desc *film, const float *xyzw, float *rgb) {
    if (!film || !xyzw || !rgb) {
        g_err = "iile_host_film_to_rgb: null argument";
        return 1;
    }
    const long n = long(film->crop_x1 - film->crop_x0) * long(film->crop_y1 - film->crop_y0);
    for (long i = 0; i < n; ++i) {
        const float *p = xyzw + 4 * i;
        float r = 3.240479f * p[0] - 1.537150f * p[1] - 0.498535f * p[2];
        float g = -0.969256f * p[0] + 1.875991f * p[1] + 0.041556f * p[2];
        float b = 0.055648f * p[0] - 0.204043f * p[1] + 1.057311f * p[2];
        float w = p[3];
        if (w != 0) {
            float inv = 1.f / w;
            r = std::max(0.f, r * inv);
            g = std::max(0.f, g * inv);
            b = std::max(0.f, b * inv);
        }
        // splatScale * splatRGB adds +0 here (film.cpp:209-215)
        r += 0.f;
        g += 0.f;
        b += 0.f;
        rgb[3 * i] = r * film->scale;
        rgb[3 * i + 1] = g * film->scale;
        rgb[3 * i + 2] = b * film->scale;
    }
    return 0;
}

int iile_host_write_pfm(const char *path, const float *rgb, int32_t width, int32_t height) {
    FILE *fp = fopen(path, "wb");
    if (!fp) {
        g_err = std::string("cannot open ") + path;
        return 1;
    }
    // little-endian host: negative scale; scanlines bottom-to-top
    fprintf(fp, "PF\n%d %d\n-1.0\n", width, height);
    for (int y = height - 1; y >= 0; --y) fwrite(rgb + 3 * size_t(y) * width, sizeof(float), 3 * size_t(width), fp);
    fclose(fp);
    return 0;
}

int iile_host_write_exr(const char *path, const float *rgb, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t total_w,
                        int32_t total_h) {
    if (!path || !rgb) {
        g_err = "iile_host_write_exr: null argument";
        return 1;
    }
    std::string err;
    try {
        if (!iile::write_exr(path, rgb, x0, y0, x1, y1, total_w, total_h, &err)) {
            g_err = err;
            return 2;
        }
    } catch (const std::exception &e) {
        g_err = std::string("iile_host_write_exr: ") + e.what();
        return 3;
    }
    return 0;
}

int iile_host_write_image(const char *path, const iile_film_desc *film, const float *rgb) {
    if (!path || !film || !rgb) {
        g_err = "iile_host_write_image: null argument";
        return 1;
    }
    const std::string p(path);
    auto ends_with = [&](const char *ext) {
        const size_t n = std::strlen(ext);
        if (p.size() < n) return false;
        for (size_t i = 0; i < n; ++i)
            if (std::tolower(static_cast<unsigned char>(p[p.size() - n + i])) != ext[i]) return false;
        return true;
    };
    if (ends_with(".exr"))
        return iile_host_write_exr(path, rgb, film->crop_x0, film->crop_y0, film->crop_x1, film->crop_y1, film->xres, film->yres);
    if (ends_with(".pfm")) return iile_host_write_pfm(path, rgb, film->crop_x1 - film->crop_x0, film->crop_y1 - film->crop_y0);
    g_err = "Can't determine image file type from suffix of filename \"" + p + "\" (.exr and .pfm are written)";
    return 2;
}

int iile_host_read_image(const char *path, int32_t *width, int32_t *height, float *rgb) {
    std::vector<float> data;
    int w = 0, h = 0;
    std::string err;
    try {  // (no exception may cross the C boundary: a damaged header can ask for more memory than there is)
        if (!path || !width || !height) {
            g_err = "iile_host_read_image: null argument";
            return 1;
        }
        if (!iile::read_image(path, &data, &w, &h, &err)) {
            g_err = err;
            return 1;
        }
    } catch (const std::exception &e) {
        g_err = std::string("iile_host_read_image: ") + e.what();
        return 2;
    }
    *width = w;
    *height = h;
    if (rgb) std::memcpy(rgb, data.data(), data.size() * sizeof(float));
    return 0;
}

int iile_host_scene_filter_table(const iile_host_scene *scene, float *table256) {
    const iile_scene_desc &d = *iile_host_scene_desc(scene);
    std::memcpy(table256, d.film_filter_table, sizeof(d.film_filter_table));
    return d.film_filter_wide;
}

int iile_host_scene_texture(const iile_host_scene *scene, int32_t index, iile_texture *out) {
    const iile_scene_desc &d = *iile_host_scene_desc(scene);
    if (index < 0 || index >= d.n_textures) {
        g_err = "texture index out of range";
        return 1;
    }
    *out = d.textures[index];
    return 0;
}

int iile_host_scene_texture_level(const iile_host_scene *scene, int32_t index, int32_t level, float *rgb) {
    const iile_scene_desc &d = *iile_host_scene_desc(scene);
    if (index < 0 || index >= d.n_textures || level < 0 || level >= d.textures[index].n_levels) {
        g_err = "texture index / level out of range";
        return 1;
    }
    const iile_texture &t = d.textures[index];
    std::memcpy(rgb, d.texels + 3 * t.level_offset[level], sizeof(float) * 3 * size_t(t.level_w[level]) * t.level_h[level]);
    return 0;
}

}  // extern "C"
