// host_scene.h — in-memory scene produced by the .pbrt loader; owns the arrays
// that iile_scene_desc points into.
#pragma once
#include <string>
#include <vector>

#include "../../../include/iile_scene.h"
#include "hmath.h"

namespace iile {

// One entry per GeometricPrimitive, in creation order (pbrtShape order,
// /root/reference/src/core/api.cpp:1371-1430), before the BVH reorders them.
struct HostPrim {
    uint32_t flags = 0;
    int32_t material = 0;
    int32_t light = -1;
    int32_t shape = 0;  // sphere index or mesh id
    V3 p[3];
    V3 n[3];
    float uv[6] = {0, 0, 0, 0, 0, 0};
    int32_t alpha = IILE_ALPHA_NONE, shadow_alpha = IILE_ALPHA_NONE;
    Bounds3 world_bound;
};

// One ImageTexture: its mapping / filtering options and its MIP pyramid (mipmap.cpp)
struct HostTexture {
    iile_texture t;
    std::vector<float> texels;  // RGB, all levels; level_offset[] is relative to this texture until finalize
};

struct HostScene {
    std::vector<HostPrim> prims;  // creation order
    std::vector<HostTexture> textures;
    std::vector<iile_sphere> spheres;
    std::vector<iile_material> materials;
    std::vector<iile_light> lights;
    int n_meshes = 0;

    // camera / film / sampler / integrator options gathered before WorldBegin
    Xform camera_to_world;
    std::string camera_name = "perspective";
    float fov = 90.f, lens_radius = 0.f, focal_distance = 1e6f;
    float shutter_open = 0.f, shutter_close = 1.f;
    float frame_aspect = -1.f;
    bool has_screen_window = false;
    float screen_window[4] = {0, 0, 0, 0};
    int xres = 1280, yres = 720;
    float crop[4] = {0, 1, 0, 1};
    float film_scale = 1.f, film_diagonal = 35.f;
    float max_sample_luminance = std::numeric_limits<float>::infinity();
    std::string film_filename = "pbrt.exr";
    std::string filter_name = "box";
    float filter_rx = 0.5f, filter_ry = 0.5f;
    float filter_p0 = 0.f, filter_p1 = 0.f;  // gaussian: alpha; mitchell: B, C; sinc: tau
    std::string sampler_name = "halton";
    int spp = 16;
    bool sample_at_pixel_center = false;
    std::string integrator_name = "path";
    int max_depth = 5;
    bool integrator_iispt = false;   // Integrator "iispt" (MakeIntegrator, src/core/api.cpp:1738-1760)
    float rr_threshold = 1.f;
    bool has_pixel_bounds = false;   // "pixelbounds" of the path integrator: x0, x1, y0, y1 as given (path.cpp:216-229)
    int pixel_bounds_given[4] = {0, 0, 0, 0};
    std::string light_strategy = "spatial";
    std::string accel_split = "sah";
    int max_node_prims = 4;
    // optional builder for split method "hlbvh" (iile_host_overrides::bvh_build)
    int (*bvh_hook)(int32_t, const float *, int32_t, iile_bvh_node *, int32_t *, int32_t *, void *) = nullptr;

    // ---- flattened output (filled by finalize) ----
    std::vector<iile_bvh_node> nodes;
    std::vector<uint32_t> o_flags;
    std::vector<int32_t> o_material, o_light, o_shape;
    std::vector<float> o_tri_p, o_tri_n, o_tri_uv;
    std::vector<int32_t> o_alpha;
    std::vector<uint16_t> perms;
    std::vector<int32_t> primes, prime_sums;
    std::vector<uint32_t> sobol_matrices;  // [n_dims][32] when the sampler is "sobol"
    std::vector<float> env_dist;  // Distribution2D tables of the infinite lights
    std::vector<iile_texture> o_textures;
    std::vector<float> o_texels;
    int n_interior = 0, n_leaf = 0;
    iile_scene_desc desc;
};

// pbrt_loader.cpp
bool load_pbrt_file(const std::string &path, HostScene *scene, std::string *err);
// finalize.cpp: BVH build + camera + halton tables + desc
bool finalize_scene(HostScene *scene, std::string *err);
// loopsubdiv.cpp
void loop_subdivide(int n_levels, const std::vector<int> &indices, const std::vector<V3> &P,
                    std::vector<int> *out_indices, std::vector<V3> *out_P, std::vector<V3> *out_N);
// bvh_build.cpp
void build_bvh(HostScene *scene);
// halton_tables.cpp
void build_halton_tables(HostScene *scene);

// sobol.cpp
int sobol_num_dimensions();
bool sobol_columns64(int dim, uint64_t cols[52]);
bool sobol_columns32(int dim, uint32_t cols[52]);
bool sobol_vdc(int log2_resolution, uint64_t vdc[52], uint64_t inv[52]);

// plymesh.cpp
bool load_ply(const std::string &path, std::vector<V3> *P, std::vector<V3> *N, std::vector<float> *uv,
              std::vector<int> *indices, std::string *err);

// imageio.cpp
bool read_image(const std::string &path, std::vector<float> *rgb, int *w, int *h, std::string *err);
bool image_is_8bit(const std::string &path);
// exr.cpp: ReadImageEXR / WriteImageEXR (src/core/imageio.cpp:138-214) for scan-line files, without the OpenEXR library
bool read_exr(const std::string &path, std::vector<float> *rgb, int *w, int *h, std::string *err);
bool write_exr(const std::string &path, const float *rgb, int x0, int y0, int x1, int y1, int total_w, int total_h, std::string *err);
// mipmap.cpp
bool build_image_texture(const std::vector<float> &rgb, int width, int height, float scale, bool gamma, bool as_float,
                         HostTexture *out, std::string *err);
// MIPMap::Lookup(st, width) (trilinear, mipmap.h:233-250) on a built pyramid
void mip_lookup_width(const HostTexture &t, float s, float tt, float width, float rgb[3]);
// MIPMap's constructor alone on prepared texels (row-major, width x height RGB), wrap mode from out->t.wrap
bool build_mip_pyramid(const std::vector<float> &rgb, int width, int height, HostTexture *out, std::string *err);
// InfiniteAreaLight's constructor (infinite.cpp:42-84): Lmap from `rgb` (already times L) into `tex`, the
// Distribution2D tables appended to `dist`; returns their size through w, h and their offset
bool build_environment_light(const std::vector<float> &rgb, int width, int height, HostTexture *tex, std::vector<float> *dist,
                             int *dist_w, int *dist_h, int64_t *dist_offset, std::string *err);
void ewa_weight_lut(float *lut);
float inverse_gamma_correct(float value);

}  // namespace iile
