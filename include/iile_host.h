/*
 * iile_host.h — C ABI of libiile_host.so: host-side scene preparation and film
 * finalisation around the GPU path. No HIP dependency; loads without a GPU.
 *
 * In the reference these steps live in the C++ host that stays on the CPU:
 *   - scene description -> Scene:   src/core/parser.cpp:712+, src/core/api.cpp:1371-1430,
 *                                   1632-1692 (pbrtShape / pbrtWorldEnd / MakeScene)
 *   - BVH build:                    src/accelerators/bvh.cpp:183-402, 640-658
 *   - Halton tables:                src/samplers/halton.cpp:65-93,
 *                                   src/core/lowdiscrepancy.cpp:2490-2504
 *   - camera matrices:              src/cameras/perspective.cpp:50-72, src/core/camera.h:90-111
 *   - film normalisation + output:  src/core/film.cpp:187-235 (to_rgb_array / WriteImage)
 * The GPU box receives only this repository, so the path carries its own
 * minimal versions of them.
 *
 * All functions return 0 on success, non-zero on error (message via
 * iile_host_last_error()); thread-compatible, not thread-safe.
 */
#ifndef IILE_HOST_H
#define IILE_HOST_H

#include "iile_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct iile_host_scene iile_host_scene;

/* Values <= 0 keep what the scene file says. Equivalent of editing the Film /
 * Sampler / Integrator lines of the .pbrt file (BASELINE.json configs override
 * resolution and pixelsamples of scenes/killeroo-simple.pbrt this way). */
typedef struct iile_host_overrides {
    int32_t xres, yres;
    int32_t spp;
    int32_t max_depth;
    int32_t sampler; /* IILE_SAMPLER_*: which sampler renders the frame. SOBOL is what the fork's path integrator does
                        under IILE_PATH_SAMPLES_OVERRIDE = spp (src/integrators/path.cpp:202-212) */
    int32_t accel_split; /* IILE_SPLIT_*: BVHAccel's "splitmethod" (src/accelerators/bvh.cpp:740-760) in place of the file's */
    /* Builder for the "hlbvh" split method in place of the host's (csrc/host/bvh_build.cpp): iile_bvh_build_hlbvh of
     * libiile_gpu.so has this signature (the last argument receives NULL). libiile_host itself stays free of HIP. */
    int (*bvh_build)(int32_t n_prims, const float *bounds6, int32_t max_prims_in_node, iile_bvh_node *nodes_out,
                     int32_t *n_nodes_out, int32_t *order_out, void *stats);
    int32_t quick_render; /* pbrt --quick (PbrtOptions.quickRender): a quarter of the file's resolution per axis (film.cpp:284-285)
                             and one pixel sample (halton.cpp:136, sobol.cpp:69), a quarter of the light samples of area and infinite lights
                             (diffuse.cpp:143, infinite.cpp:183); explicit xres / yres / spp above still win */
} iile_host_overrides;
#define IILE_SPLIT_KEEP 0
#define IILE_SPLIT_SAH 1
#define IILE_SPLIT_HLBVH 2
#define IILE_SPLIT_MIDDLE 3
#define IILE_SPLIT_EQUAL 4
#define IILE_SAMPLER_KEEP 0
#define IILE_SAMPLER_HALTON 1
#define IILE_SAMPLER_SOBOL 2

typedef struct iile_host_scene_info {
    int32_t n_prims, n_triangles, n_spheres, n_meshes;
    int32_t n_nodes, n_interior_nodes, n_leaf_nodes;
    int32_t n_materials, n_lights;
    int32_t xres, yres, spp, max_depth;
    int32_t probe_hemi_size; /* side of the IISPT probe films iile_render_probes writes (iile_scene_desc::probe) */
    int32_t integrator;      /* IILE_INTEGRATOR_*: the file's Integrator directive ("path" when it has none is pbrt's default) */
} iile_host_scene_info;
#define IILE_INTEGRATOR_PATH 0
#define IILE_INTEGRATOR_IISPT 1

/* Parse a .pbrt file (plus its Includes), tessellate, build the BVH and the
 * sampler tables. Stands where ParseFile + pbrtWorldEnd's MakeScene /
 * MakeIntegrator stand (src/main/pbrt.cpp:97-219, src/core/api.cpp:1632-1660). */
int iile_host_load_pbrt(const char *path, const iile_host_overrides *ov, iile_host_scene **out);
/* The flattened scene; valid until iile_host_scene_free. */
const iile_scene_desc *iile_host_scene_desc(const iile_host_scene *scene);
/* &desc->film, for bindings that do not mirror the whole iile_scene_desc. */
const iile_film_desc *iile_host_scene_film(const iile_host_scene *scene);
int iile_host_scene_get_info(const iile_host_scene *scene, iile_host_scene_info *info);
/* Film "string filename" of the scene file ("pbrt.exr" when it names none, src/core/api.cpp MakeFilm / film.cpp:262). */
const char *iile_host_scene_film_filename(const iile_host_scene *scene);
void iile_host_scene_free(iile_host_scene *scene);

/* Film::to_rgb_array (src/core/film.cpp:187-225) on a film of
 * {X, Y, Z, filterWeightSum} float4 pixels over the cropped pixel bounds:
 * XYZ->RGB, divide by weight, clamp at 0, multiply by scale. rgb: 3 floats/pixel. */
int iile_host_film_to_rgb(const iile_film_desc *film, const float *film_xyzw, float *rgb);
/* PFM writer (src/core/imageio.cpp WriteImagePFM: bottom-to-top scanlines). */
int iile_host_write_pfm(const char *path, const float *rgb, int32_t width, int32_t height);
/* WriteImageEXR (src/core/imageio.cpp:180-214) without the OpenEXR library: R, G, B as 16-bit half, ZIP-compressed
 * scan-line blocks; the (x1 - x0) x (y1 - y0) pixels are the data window at (x0, y0) of a total_w x total_h display
 * window (Film::WriteImage passes the cropped pixel bounds and the full resolution). */
int iile_host_write_exr(const char *path, const float *rgb, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t total_w,
                        int32_t total_h);
/* WriteImage (src/core/imageio.cpp:84-136) for a whole film: .exr or .pfm by the file name's extension. */
int iile_host_write_image(const char *path, const iile_film_desc *film, const float *rgb);

/* ReadImage (src/core/imageio.cpp:60-82) for .exr (scan-line files; ZIP, ZIPS, RLE or uncompressed; values arrive as halfs,
 * as through Imf::RgbaInputFile) / .pfm / .png / .tga: RGB floats, row 0 = top scanline. Call with
 * rgb == NULL to get the size, then with a buffer of 3 * width * height floats. */
int iile_host_read_image(const char *path, int32_t *width, int32_t *height, float *rgb);
/* The MIP pyramid built for image texture `index` of a loaded scene (ImageTexture::GetTexture + MIPMap's
 * constructor, src/textures/imagemap.cpp:53-101, src/core/mipmap.h:111-208). */
int iile_host_scene_texture(const iile_host_scene *scene, int32_t index, iile_texture *out);
/* Copies light `index` of a loaded scene (Light::nSamples, type, emission ... as iile_scene_desc::lights holds them). */
int iile_host_scene_light(const iile_host_scene *scene, int32_t index, iile_light *out);
/* Copies level `level` (level_w x level_h RGB texels, row 0 = bottom scanline) of that texture. */
int iile_host_scene_texture_level(const iile_host_scene *scene, int32_t index, int32_t level, float *rgb);

/* Film::filterTable of the scene's pixel filter (src/core/film.cpp:65-74): 16 x 16 floats; returns whether the
 * filter is wider than the one-pixel box (iile_scene_desc::film_filter_wide). */
int iile_host_scene_filter_table(const iile_host_scene *scene, float *table256);

/* The Sobol' generator matrices the host builds (csrc/host/sobol.cpp): SobolMatrices32 / SobolMatrices64 of
 * src/core/sobolmatrices.cpp for dimensions [0, n_dims): n_dims * 52 entries each (either pointer may be NULL); and
 * VdCSobolMatrices[log2_resolution - 1] / VdCSobolMatricesInv[log2_resolution - 1], 52 entries each (unused ones 0). */
int iile_host_sobol_matrices(int32_t n_dims, uint32_t *m32, uint64_t *m64);
int iile_host_sobol_vdc(int32_t log2_resolution, uint64_t *vdc52, uint64_t *vdc_inv52);

const char *iile_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* IILE_HOST_H */
