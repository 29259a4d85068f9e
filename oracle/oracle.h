/*
 * oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE, not product).
 *
 * The oracle is a scalar CPU restatement of the reference's path-tracing hot
 * path (PathIntegrator::Li, BVHAccel::Intersect/IntersectP, the
 * SamplerIntegrator::Render tile loop, HaltonSampler, FilmTile) consuming the
 * same flattened iile_scene_desc as the GPU library. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (libiile_gpu.so) never links or calls it.
 */
#ifndef IILE_ORACLE_H
#define IILE_ORACLE_H

#include <stdint.h>

#include "../include/iile_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

/* trig_mode: how sin/cos/acos on the path are evaluated.
 *   ORACLE_TRIG_LIBM     — call the host libm exactly where the reference does
 *                          (float overloads; double in microfacet.cpp:241-246).
 *                          This is the mode that is pinned against the reference.
 *   ORACLE_TRIG_PORTABLE — a fixed double-precision polynomial evaluation that the
 *                          HIP kernels restate operation for operation, so device
 *                          and oracle agree bit for bit. */
enum { ORACLE_TRIG_LIBM = 0, ORACLE_TRIG_PORTABLE = 1 };

typedef struct oracle_stats {
    uint64_t camera_rays;      /* integrator.cpp:286 */
    uint64_t regular_rays;     /* Scene::Intersect calls, scene.cpp:45-50 */
    uint64_t shadow_rays;      /* Scene::IntersectP calls, scene.cpp:52-57 */
    uint64_t tri_tests;        /* triangle.cpp:45 nTests (Intersect + IntersectP) */
    uint64_t tri_hits;         /* nHits */
    uint64_t sphere_tests;
    uint64_t nodes_closest;    /* BVH nodes visited by Intersect */
    uint64_t nodes_any;        /* BVH nodes visited by IntersectP */
    uint64_t nee_evals;        /* path.cpp:122 totalPaths */
    uint64_t zero_radiance;    /* path.cpp:126 */
    uint64_t path_length[8];   /* histogram of `bounces` at ReportValue, path.cpp:192 */
    int32_t max_stack_depth;
    int32_t threads;
    double seconds;            /* wall time of the render loop only */
} oracle_stats;

/* SamplerIntegrator::Render restated: tiles of 16x16 over the sample bounds,
 * samples k in [k_begin,k_end) (0,-1 => all spp), only tiles with
 * iile_tile_owner(tx, ty, tile_nranks) == tile_rank (iile_scene.h). film_xyzw: {X,Y,Z,weightSum} per pixel
 * of the cropped pixel bounds (Film::Pixel after MergeFilmTile). */
int oracle_render(const iile_scene_desc *scene, int trig_mode, int n_threads, int k_begin, int k_end,
                  int tile_rank, int tile_nranks, float *film_xyzw, oracle_stats *stats);

/* ---- unit-level entry points for known-answer and parity tests ---- */
int64_t oracle_halton_index(const iile_scene_desc *scene, int px, int py, int64_t k);
float oracle_halton_sample(const iile_scene_desc *scene, int64_t index, int dim);
/* the scene's sampler, HaltonSampler or SobolSampler (samplers/sobol.cpp:42-59): sample index of (pixel, k); dimension
 * `dim` of a sample for current pixel (px, py) */
int64_t oracle_sample_index(const iile_scene_desc *scene, int px, int py, int64_t k);
float oracle_sample_dimension(const iile_scene_desc *scene, int64_t index, int dim, int px, int py);
/* core/lowdiscrepancy.h:93-126, 229-288 on caller-supplied matrices */
uint32_t oracle_reverse_bits32(uint32_t n);
uint32_t oracle_multiply_generator(const uint32_t *C, uint32_t a);
float oracle_sample_generator_matrix(const uint32_t *C, uint32_t a, uint32_t scramble);
void oracle_gray_code_sample(const uint32_t *C, uint32_t n, uint32_t scramble, float *p);
float oracle_sobol_sample_float(const uint32_t *m32, int64_t a, int dimension, uint32_t scramble);
double oracle_sobol_sample_double(const uint64_t *m64, int64_t a, int dimension, uint64_t scramble);
uint64_t oracle_sobol_interval_to_index(const uint64_t *vdc, const uint64_t *vdc_inv, uint32_t m, uint64_t frame, int px, int py);
float oracle_radical_inverse(int base_index, uint64_t a);
float oracle_scrambled_radical_inverse(const iile_scene_desc *scene, int base_index, uint64_t a);
/* same function with a caller-supplied digit permutation of `base` entries */
float oracle_scrambled_radical_inverse_perm(int base, const uint16_t *perm, uint64_t a);
void oracle_camera_ray(const iile_scene_desc *scene, float pfilm_x, float pfilm_y, float plens_x,
                       float plens_y, float *o3, float *d3);
/* closest hit for n rays: prim[i] = -1 on miss; tb[4*i..] = {t, b0, b1, b2}
 * (for spheres b* are 0). */
void oracle_intersect(const iile_scene_desc *scene, int n, const float *o, const float *d,
                      const float *tmax, int32_t *prim, float *tb);
void oracle_intersect_p(const iile_scene_desc *scene, int n, const float *o, const float *d,
                        const float *tmax, int32_t *hit);
/* Per-sample radiance Li for n (pixel, k) pairs; L: 3 floats each after the
 * NaN/negative/inf guards of integrator.cpp:293-314; nrays (optional): 2 per
 * sample {regular, shadow}. */
void oracle_li(const iile_scene_desc *scene, int trig_mode, int n, const int32_t *px, const int32_t *py,
               const int32_t *k, float *L, int32_t *nrays);
/* BSDF probes in the local shading frame of material `mat` with ns=ng=(0,0,1),
 * ss=(1,0,0): f (3), pdf and Sample_f -> {wi(3), f(3), pdf}. */
void oracle_bsdf_eval(const iile_scene_desc *scene, int trig_mode, int mat, const float *wo3,
                      const float *wi3, float *f3, float *pdf);
void oracle_bsdf_sample(const iile_scene_desc *scene, int trig_mode, int mat, const float *wo3,
                        const float *u2, float *wi3, float *f3, float *pdf);
/* n samples / pdf evaluations for one outgoing direction (chi-square test of src/tests/bsdfs.cpp) */
void oracle_bsdf_sample_batch(const iile_scene_desc *scene, int trig_mode, int mat, const float *wo3, int n,
                              const float *u2n, float *wi3n, float *pdfn);
void oracle_bsdf_pdf_batch(const iile_scene_desc *scene, int trig_mode, int mat, const float *wo3, int n,
                           const float *wi3n, float *pdfn);
/* One IISPT probe: HemisphericCamera at `pos` looking along `dir` rendered by IISPTdIntegrator::RenderView
 * (iispt_d.cpp, hemispheric.cpp, iisptrenderrunner.cpp:316-346) with scene->probe's film / sampler / depth. Outputs
 * [y][x] in raster coordinates: intensity (hemi^2 x 3), camera-space normals (x 3), distances. Returns 3 for a
 * degenerate direction. */
int oracle_render_probe(const iile_scene_desc *scene, int trig_mode, const float *pos3, const float *dir3, float *intensity_rgb,
                        float *normals_xyz, float *distance);
/* ImageTexture::Evaluate of texture `tex` at n (u, v) with differentials {dudx, dvdx, dudy, dvdy} */
void oracle_texture_eval(const iile_scene_desc *scene, int trig_mode, int tex, int n, const float *uv2, const float *duv4,
                         float *rgb3);
/* first hit of the camera ray through film point (pfx, pfy): {u, v, du/dx, dv/dx, du/dy, dv/dy} as
 * ComputeDifferentials leaves them (interaction.cpp:103-149); returns 0 when the ray escapes */
int oracle_hit_geometry(const iile_scene_desc *scene, int trig_mode, const float *o3, const float *d3, float *out24);
int oracle_camera_hit_differentials(const iile_scene_desc *scene, int trig_mode, float pfx, float pfy, float *out6);
/* Distribution1D (sampling.h:55-109) over func[0..n), n <= 8: mode 0 SampleDiscrete, mode 1 SampleContinuous */
int oracle_distribution1d(const float *func, int n, int mode, float u, float *value, float *pdf);
float oracle_log(int trig_mode, float x);
void oracle_sincos(int trig_mode, float x, float *s, float *c);
void oracle_sincos_d(int trig_mode, double x, double *s, double *c);
float oracle_acos(int trig_mode, float x);
float oracle_atan2(int trig_mode, float y, float x);
double oracle_atan2_d(double y, double x); /* the portable double evaluation itself */

/* Property tests of the reference (src/tests/fp_tests.cpp, src/tests/shapes.cpp), run against the
 * restatement's own float machinery. Each returns the number of violated expectations.
 *   oracle_check_next_float   FloatingPoint.NextUpDownFloat (:29-47): vs nextafterf
 *   oracle_check_efloat       EFloat.Add/Sub/Mul/Div (:201-270): the interval contains the result
 *                             computed from precise values chosen inside the operands' intervals
 *   oracle_check_reintersect  Triangle.Reintersect / FullSphere.Reintersect (shapes.cpp:154-208,
 *                             :374-436): rays spawned from a hit (SpawnRay / SpawnRayTo) never hit
 *                             the primitive they leave. stats = {hits found, spawned rays tested}. */
/* Sphere.SolidAngle (shapes.cpp:316-348): the solid angle a sphere subtends from point p, once by
 * Shape::SolidAngle (shape.cpp:89-102: Sphere::Sample + IntersectP) and once by uniform sphere
 * sampling (mcSolidAngle, shapes.cpp:318-329), both with Halton points (0, 1). */
void oracle_sphere_solid_angle(const iile_scene_desc *scene, int sphere, const float *p3, int n_samples,
                               double *by_sampling, double *by_uniform_directions);
/* Triangle.Sampling (shapes.cpp:210-271) for the area light `light` (a triangle or a sphere emitter):
 * the solid angle it subtends from p as sum 1 / (n pdf) over Shape::Sample(ref, u) with Halton (0, 1)
 * points, and by uniform-direction Monte Carlo with the same points. */
void oracle_light_solid_angle(const iile_scene_desc *scene, int light, const float *p3, int n_samples,
                              double *by_sampling, double *by_uniform_directions);
int64_t oracle_check_next_float(int iters, uint64_t seed);
int64_t oracle_check_efloat(int iters, uint64_t seed);
int64_t oracle_check_reintersect(const iile_scene_desc *scene, int n, const float *o, const float *d, int n_out,
                                 uint64_t seed, int64_t *stats);

/* The IISPT render runner's gather (src/integrators/iisptrenderrunner.cpp:248-596): the hemi points of a task
 * (valid flag, aux ray origin and direction: where the probe cameras go), and the per-pixel loop over the predicted
 * hemispheres nn_films [hemi point][hemi][hemi][3] -> {f_beta * L, weight} per film pixel of the task. */
int oracle_iispt_hemi_points(const iile_scene_desc *scene, int trig_mode, const iile_iispt_task *task, uint8_t *valid, float *pos3, float *dir3);
int oracle_iispt_gather(const iile_scene_desc *scene, int trig_mode, const iile_iispt_task *task, const uint8_t *valid, const float *pos3,
                        const float *dir3, const float *nn_films, float *out_rgbw);
/* iile_tile_owner of iile_scene.h (a static inline there), exported so that tests can call the header's own definition */
int oracle_tile_owner(int tx, int ty, int nranks);

/* The IISPT integrator's direct pass (DirectProgressiveIntegrator driven by IisptRenderRunner::run_direct; oracle_path.cpp,
 * "DIRECT pass") accumulated into a film monitor of doubles {sum r, g, b, weight} per pixel of the cropped pixel bounds, and
 * IisptFilmMonitor::merge_into + to_intensity_film. */
int oracle_iispt_direct(const iile_scene_desc *scene, int trig_mode, int n_passes, int first_pass, int n_threads, double *film_rgbw);
void oracle_iispt_merge(int64_t n_pixels, const double *direct_rgbw, const double *indirect_rgbw, float *out_rgb);

/* BVHAccel(prims, maxPrimsInNode, SplitMethod::HLBVH) as one thread builds it (oracle_bvh.cpp; src/accelerators/bvh.cpp:
 * 107-181, 404-658): nodes_out (room for 2 * n_prims) in flattenBVHTree's order, order_out[p] = number of the primitive at
 * position p of BVHAccel::primitives after the build, codes_out (optional) the sorted Morton codes. 0 = ok, 1 = a CHECK of
 * the reference would abort, 2 = bad arguments. */
int oracle_bvh_hlbvh(int32_t n_prims, const float *bounds6, int32_t max_prims_in_node, iile_bvh_node *nodes_out, int32_t *n_nodes_out,
                     int32_t *order_out, uint32_t *codes_out);

#ifdef __cplusplus
}
#endif
#endif
