/*
 * iile_gpu.h — C ABI of libiile_gpu.so: the MI355X (gfx950) path-tracing
 * integrator behind the reference's Integrator plugin surface.
 *
 * The reference has no binary plugin ABI; its surface for this path is
 *   class Integrator { virtual void Render(const Scene &scene) = 0; }
 *                                         src/core/integrator.h:53-58
 *   class SamplerIntegrator : Integrator  { Render(); virtual Li(...); Preprocess(); }
 *                                         src/core/integrator.h:77-107
 *   PathIntegrator *CreatePathIntegrator(const ParamSet&, std::shared_ptr<Sampler>,
 *                                        std::shared_ptr<const Camera>)
 *                                         src/integrators/path.h:70-72
 * called from pbrtWorldEnd as `integrator->Render(*scene)` (src/core/api.cpp:1650-1662).
 * The BVH arrays are private to BVHAccel (src/accelerators/bvh.h:90-94), so a
 * replacement cannot read the built tree: the host flattens the scene into an
 * iile_scene_desc (iile_scene.h) and hands it over once.
 *
 *   iile_scene_create   <->  MakeScene + integrator construction (api.cpp:1694-1747):
 *                            uploads BVH, triangles, spheres, materials, lights,
 *                            camera matrices and Halton tables to HBM
 *   iile_render         <->  SamplerIntegrator::Render(scene) (integrator.cpp:227-331)
 *                            minus WriteImage: runs generate/extend/shade/connect
 *                            wavefront kernels and leaves the merged film
 *                            ({X,Y,Z,filterWeightSum} per pixel == Film::Pixel after
 *                            MergeFilmTile, film.cpp:135-148) in `film_xyzw`
 *   iile_scene_destroy  <->  ~Scene / ~Integrator
 *
 * Errors: the reference reports through Error()/LOG(FATAL) (src/core/error.h:54-55);
 * here every call returns 0 on success or a non-zero code, message in
 * iile_last_error(). There is NO CPU fallback: without a HIP device every
 * compute entry point fails with IILE_ERR_NO_DEVICE.
 * Threading: thread-compatible, not thread-safe (one scene per host thread).
 */
#ifndef IILE_GPU_H
#define IILE_GPU_H

#include "iile_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct iile_scene iile_scene;

enum {
    IILE_OK = 0,
    IILE_ERR_ARG = 1,
    IILE_ERR_NO_DEVICE = 2,
    IILE_ERR_HIP = 3,
    IILE_ERR_UNSUPPORTED = 4,
    IILE_ERR_TIMEOUT = 5      /* libiile_dist: the other ranks did not answer within the communicator's deadline; it was aborted */
};

typedef struct iile_render_params {
    int32_t k_begin, k_end;         /* sample indices [k_begin, k_end); k_end <= 0: all pixelsamples */
    int32_t tile_rank, tile_nranks; /* this call renders the 16x16 tiles (tx, ty) with iile_tile_owner(tx, ty, nranks)
                                       == rank (iile_scene.h); nranks <= 0: all tiles (single GPU) */
    int32_t spp_per_pass;           /* 0 = passes sized to the workspace budget. A pass renders all samples of a range
                                       of tiles; > 0 asks for passes of about (owned pixels x spp_per_pass) paths (tests) */
    int32_t collect_stats;          /* 1: instrumented kernels (ray / node / triangle counters) */
    int32_t time_kernels;           /* 1: bracket every kernel with HIP events (iile_stats::ms_*). The NEE kernels of a
                                       bounce run on a second stream beside the next bounce's k_extend / k_shade, so
                                       these durations overlap and sum to more than ms_total; 2: the same with every
                                       kernel on `stream`, one after the other (each kernel alone on the GPU) */
    int32_t film_on_device;         /* 1: film_xyzw is a device pointer (stays in HBM) */
    void *stream;                   /* hipStream_t to launch on; NULL = the null stream */
} iile_render_params;

typedef struct iile_stats {
    /* counters named after the reference's STAT_COUNTERs (scene.cpp:45-47,
     * triangle.cpp:45, path.cpp:45-46, integrator.cpp:48); filled when collect_stats */
    uint64_t camera_rays, closest_rays, shadow_rays;
    uint64_t nodes_closest, nodes_any, tri_tests, tri_hits, sphere_tests;
    uint64_t nee_evals, zero_radiance;
    uint64_t path_length[8];
    /* timings in milliseconds; per-kernel sums filled when time_kernels */
    double ms_total;
    double ms_generate, ms_extend, ms_shade, ms_connect, ms_film;
    int32_t n_extend_launches, n_connect_launches /* per NEE kernel */, n_shade_launches, n_passes;
    uint64_t n_paths;             /* camera samples rendered by this call */
    uint64_t workspace_bytes;     /* HBM held by the wavefront queues */
    /* the extend kernel alone (main-path closest-hit rays): inputs of its roofline */
    uint64_t ext_rays, ext_nodes, ext_tri_tests, ext_sphere_tests;
    uint64_t any_tri_tests;       /* triangle tests of the shadow (any-hit) kernel */
    double ms_shadow, ms_mis, ms_resolve; /* the NEE kernels (ms_resolve: k_mis_lit); ms_connect is their sum */
    /* filled by every render that returns stats: the MIS rays (EstimateDirect's BSDF-sampled rays) the call really
     * traced. The instrumented build traces all of them (== closest_rays - ext_rays); the plain build skips those its
     * shade kernel proves unable to end on the sampled light (exact: see k_shade) */
    uint64_t mis_rays_traced;
    uint64_t ext_rays_traced;   /* extension rays k_extend traced (every build; the uninstrumented pass of a scene without specular
                                   lobes or infinite lights does not trace the rays of bounce maxDepth, which can add nothing) */
} iile_stats;

int iile_device_count(void);
const char *iile_last_error(void);

/* Device memory for hosts built without hipcc (the C++ host keeps the film in HBM between iile_render and the
 * multi-GPU merge of iile_dist.h): select the GPU of this process, allocate / free / read back a buffer. */
int iile_device_select(int32_t device);
int iile_device_alloc(uint64_t bytes, void **out_dev);
void iile_device_free(void *dev);
int iile_device_download(void *dst_host, const void *src_dev, uint64_t bytes, void *stream); /* waits for `stream` */
int iile_device_upload(void *dst_dev, const void *src_host, uint64_t bytes, void *stream);   /* waits for `stream` */
int iile_device_zero(void *dev, uint64_t bytes, void *stream);                               /* queued on `stream` */
/* A HIP stream of the caller's own (non-blocking: it does not synchronise with the null stream) for every `stream` argument of this
 * header; iile_stream_wait returns when everything queued on it has finished. */
int iile_stream_create(void **out_stream);
int iile_stream_wait(void *stream);
void iile_stream_destroy(void *stream);

int iile_scene_create(const iile_scene_desc *desc, iile_scene **out);
void iile_scene_destroy(iile_scene *scene);

/* film_xyzw: 4 floats per pixel of the cropped pixel bounds, row-major.
 * With tile sharding each rank's film holds its own tiles' contributions
 * (zeros elsewhere); ranks are combined with one sum-reduction. */
int iile_render(iile_scene *scene, const iile_render_params *params, float *film_xyzw, iile_stats *stats);
/* iile_render(film_on_device = 1, stats = NULL) returns as soon as its kernels are queued on `stream`: it cannot know yet whether
 * the exact finish of whole-number film positions (FilmTile's one-pixel halo) ran out of the room the frame was
 * sized for. iile_render_status waits for `stream` and returns IILE_ERR_UNSUPPORTED if it did (that film is wrong), IILE_OK
 * otherwise; the next iile_render on the scene makes the same check on entry, so the error cannot go unseen. The reference has
 * no such state: SamplerIntegrator::Render returns when the film is complete (integrator.cpp:331-339). */
int iile_render_status(iile_scene *scene, void *stream);
/* Test probe: force the capacity (pixel hits per pass and tile sums per render) of the exact film finish for the scene's next
 * renders; 0 = sized from the frame again. */
int iile_test_patch_capacity(iile_scene *scene, uint32_t capacity);

/* ---- kernel-level entry points (parity tests; host pointers, synchronous) ---- */
/* BVHAccel::Intersect on n rays. prim[i] = -1 on miss; tb[4i..] = {t, b0, b1, b2}.
 * With stats != NULL the instrumented traversal runs (visit counters, binary steps); with
 * stats == NULL the traversal of the uninstrumented render kernels (four-wide steps). */
int iile_trace_closest(iile_scene *scene, int32_t n, const float *o3, const float *d3, const float *tmax,
                       int32_t *prim, float *tb, iile_stats *stats);
/* BVHAccel::IntersectP on n rays. */
int iile_trace_any(iile_scene *scene, int32_t n, const float *o3, const float *d3, const float *tmax,
                   int32_t *hit, iile_stats *stats);
/* The scene's sampler (HaltonSampler, or SobolSampler when iile_scene_desc::sobol is enabled):
 * index_out[i] = GetIndexForSample(k) of pixel i; out[i*ndims + d] =
 * SampleDimension(index, dim0 + d). */
int iile_halton_samples(iile_scene *scene, int32_t n, const int32_t *px, const int32_t *py, const int32_t *k,
                        int32_t dim0, int32_t ndims, float *out, uint32_t *index_out);
/* PerspectiveCamera::GenerateRayDifferential (origin / direction only). plens may be NULL. */
int iile_camera_rays(iile_scene *scene, int32_t n, const float *pfilm2, const float *plens2, float *o3, float *d3);
/* Radiance of n individual camera samples through the full wavefront pipeline,
 * after the NaN / negative / inf guards; nrays (optional): {closest, shadow} per sample. */
int iile_li_samples(iile_scene *scene, int32_t n, const int32_t *px, const int32_t *py, const int32_t *k,
                    float *L3, int32_t *nrays2);
/* BSDF::f / Pdf (out: 4 floats {f.rgb, pdf}) and BSDF::Sample_f (out: 7 floats
 * {wi.xyz, f.rgb, pdf}) of material `mat` in the canonical frame ns=ng=+z, ss=+x. */
int iile_bsdf_eval(iile_scene *scene, int32_t n, int32_t mat, const float *wo3, const float *wi3, float *out4);
int iile_bsdf_sample(iile_scene *scene, int32_t n, int32_t mat, const float *wo3, const float *u2, float *out7);
/* The IISPT probe pass: for each of n probes, what iisptrenderrunner.cpp:316-346 obtains from
 * CreateHemisphericCamera(hemi, hemi, pos, dir) + IISPTdIntegrator::RenderView + get_intensity_film /
 * get_normal_film / get_distance_film (src/integrators/iispt_d.cpp:66-470, src/cameras/hemispheric.cpp) — the three
 * inputs of the IISPT network — rendered as one batched wavefront pass. pos3 / dir3: the probe origins and
 * directions (the runner passes the spawned ray of the surface normal). Outputs per probe hemi x hemi pixels, [y][x]
 * in the probe camera's raster coordinates (the reference's ImageFilm keeps row hemi - 1 - y): intensity RGB,
 * camera-space normals, distances (-1 where the ray escaped). Film, sampler and depth come from
 * iile_scene_desc::probe. outputs_on_device != 0: the three output pointers are device memory (the images stay in
 * HBM for the network); pos3 / dir3 are host memory either way. `stream`: every copy and kernel of the call is queued on it (NULL =
 * the null stream); the call returns when they have finished. */
int iile_render_probes(iile_scene *scene, int32_t n_probes, const float *pos3, const float *dir3, float *intensity_rgb,
                       float *normals_xyz, float *distance, int32_t outputs_on_device, iile_stats *stats, void *stream);
/* The IISPT integrator's DIRECT pass: what IisptRenderRunner::run_direct
 * (src/integrators/iisptrenderrunner.cpp:601-633) leaves in film_monitor_direct — n_passes calls of
 * DirectProgressiveIntegrator::RenderOnePass (src/integrators/directprogressiveintegrator.cpp:60-150: one camera sample per
 * pixel; Li = emitted light + UniformSampleAllLights + the mirror recursion, five levels deep) with the 16-samples-per-pixel
 * RandomSampler of src/integrators/iispt.cpp:813-816, added by IisptFilmMonitor::add_n_samples
 * (src/integrators/iisptfilmmonitor.cpp:47-72) into film_rgbw[(y * w + x) * 4] = {sum r, sum g, sum b, sum of ray weights},
 * DOUBLES, over the film's cropped pixel bounds. The reference consumes ONE random stream per thread, and which thread renders
 * which pass is a race; here pass p (= first_pass + i) is seeded 6284 + 17 p — the seed a runner thread with that number would
 * clone its sampler with — and every pixel of it has its own PCG32 stream, consumed in the reference's per-pixel order (kernels_direct.hip).
 * Up to four passes run in one set of launches (path id = (pixel, pass): one pass of one sample per pixel leaves most of a persistent
 * traversal grid idle); each keeps its seed and records, and a pixel's passes are added to the monitor in pass order — the result is
 * the same, bit for bit, whatever the grouping. accumulate == 0: the film is zeroed first. Every light is sampled Light::nSamples times per vertex (iile_light::n_samples;
 * UniformSampleAllLights, integrator.cpp:54-83), infinite lights included (escaped rays return Le at every depth,
 * directprogressiveintegrator.cpp:29-32); the NEE record planes are sized for pixels x (sum of the lights' nSamples) records
 * per level, at most 64 light samples per vertex. Reflected rays carry differentials in textured scenes
 * (directprogressiveintegrator.cpp:165-184). Scenes with glass — where Li branches into a reflection and a transmission recursion
 * at every glass vertex (allowMultipleLobes = false: glass.cpp:62-90) and the sampler's stream follows the recursion's depth-first
 * order — are rendered by one thread per pixel walking its tree (k_direct_tree), the others by the wavefront; an uber material's
 * specular transmissions (opacity < 1, Kt) likewise, SpecularTransmit's u[0] choosing between the two lobes. Specular spheres
 * in textured scenes carry Sphere::Intersect's dndu / dndv (src/shapes/sphere.cpp:122-143) into the reflected differentials. */
typedef struct iile_direct_params {
    int32_t n_passes, first_pass;
    int32_t accumulate;
    int32_t film_on_device; /* 1: film_rgbw is a device pointer */
    void *stream;
} iile_direct_params;
int iile_render_direct(iile_scene *scene, const iile_direct_params *params, double *film_rgbw);
/* The IISPT render runner around the probe pass and the network (src/integrators/iisptrenderrunner.cpp:216-596),
 * one task (iile_iispt_task, iile_scene.h) at a time:
 *   iile_iispt_hemi_points  for every hemi point of the task (row by row): whether it has a probe (find_intersection found
 *                           a surface that scatters, :632-757) and the aux ray its HemisphericCamera is placed on
 *                           (:299-312) — feed pos3 / dir3 of the valid ones to iile_render_probes, the images to the network;
 *   iile_iispt_gather       the per-pixel loop (:414-596): compute_fpixel_weights (:961-1039) and sample_hemisphere
 *                           (:142-178) over the four neighbouring predicted hemispheres. nn_films: per hemi point (valid or
 *                           not) hemi x hemi RGB, row 0 the top scanline as the network emits it (device memory when
 *                           nn_on_device); out_rgbw: per film pixel of the task, row-major, {f_beta * L, weight} as handed
 *                           to IisptFilmMonitor::add_n_samples (zeros where the runner records nothing). With
 *                           out_on_device the call returns once its kernels are queued on the null stream (use the output
 *                           on that stream, or synchronise); with a host output it returns when the pixels are there.
 * Both keep their device buffers in the scene (grown on demand), so a frame's hundreds of calls allocate nothing.
 * Needs the scene's Halton sampler (the runner clones the scene's sampler) and probe setup. */
int iile_iispt_hemi_points(iile_scene *scene, const iile_iispt_task *task, uint8_t *valid, float *pos3, float *dir3);
int iile_iispt_gather(iile_scene *scene, const iile_iispt_task *task, const uint8_t *valid, const float *pos3, const float *dir3,
                      const float *nn_films, int32_t nn_on_device, float *out_rgbw, int32_t out_on_device);
/* The same for n_tasks tasks in one set of launches (a 100 x 100-pixel task alone fills a sixth of the chip): every array is
 * the per-task arrays of the single-task calls, task after task — hemi points (valid, pos3, dir3, nn_films) in the tasks'
 * own row-by-row order, film pixels (out_rgbw) row-major per task. Results are those of the single-task calls, bit for bit.
 * `stream`: every copy and kernel of the call is queued on it (NULL = the null stream, which the single-task calls use). The whole
 * indirect pass — these two, iile_render_probes, iile_iispt_net_predict, iile_iispt_film_add — given ONE stream is ordered by that
 * stream alone (the scene's scratch block is shared by the IISPT calls: calls on one scene go on one stream, or the caller orders
 * them). hemi_points returns once its host arrays are written; gather with out_on_device returns with its kernels queued. */
int iile_iispt_hemi_points_batch(iile_scene *scene, const iile_iispt_task *tasks, int32_t n_tasks, uint8_t *valid, float *pos3, float *dir3,
                                 void *stream);
int iile_iispt_gather_batch(iile_scene *scene, const iile_iispt_task *tasks, int32_t n_tasks, const uint8_t *valid, const float *pos3,
                            const float *dir3, const float *nn_films, int32_t nn_on_device, float *out_rgbw, int32_t out_on_device, void *stream);
/* The two film monitors of IISPTIntegrator::render_normal_2 (src/integrators/iispt.cpp:357-446), kept in HBM as {r, g, b, weight}
 * double sums per film pixel (IisptPixel, src/integrators/iisptpixel.h):
 *   iile_iispt_film_add    IisptFilmMonitor::add_n_samples (src/integrators/iisptfilmmonitor.cpp:47-72) for every pixel of n_tasks
 *                          tasks in one launch: out_rgbw_dev is what iile_iispt_gather_batch left for these tasks (device memory),
 *                          film_rgbw_dev the film_w x film_h monitor. The tasks of one call must not overlap (tasks of one sweep of
 *                          IisptScheduleMonitor never do). Waits for `stream`'s earlier work, then queues its kernel there.
 *   iile_iispt_film_merge  film_monitor_direct->merge_into(film_monitor_indirect) + to_intensity_film (:231-275, :158-196): both
 *                          monitors normalised (sums over the weight where it is positive), added, as float RGB. Queued on `stream`. */
int iile_iispt_film_add(iile_scene *scene, const iile_iispt_task *tasks, int32_t n_tasks, const float *out_rgbw_dev, double *film_rgbw_dev,
                        int32_t film_w, int32_t film_h, void *stream);
int iile_iispt_film_merge(const double *direct_rgbw_dev, const double *indirect_rgbw_dev, int64_t n_pixels, float *rgb_dev, void *stream);
/* The IISPT network itself: `IISPTNet.forward` of ml/iispt_net.py:8-109 in eval mode, as the child process
 * of ml/main_stdio_net.py:44-106 runs it once per probe (`net(torch_img)`, one CPU thread, fp32) — here over a whole batch
 * of probes with hand-written gfx950 kernels (csrc/device/iispt_net.hip: implicit-GEMM 3 x 3 convolutions on the bf16 matrix
 * pipe over split operands, fp32 accumulation; max-pool and bilinear upsampling as streaming kernels of their own, the
 * concatenation free in the loads; bias, LeakyReLU and the BatchNorm affine in the accumulator registers before the stores). Agreement with the reference module on the fixture of
 * tests/golden/iispt_net_fixture.npz: within 1e-4 of the largest output (tests/test_iispt_nn.py).
 *   iile_iispt_net_create   takes the tensors of the reference's `state_dict()` (host memory, the checkpoint's own shapes:
 *                           Conv2d [out][in][k][k], ConvTranspose2d [in][out][k][k]) in forward order: convolutions
 *                           encoder0.0, encoder0.2, encoder1.1, encoder1.4, encoder2.1, encoder2.4, encoder3.1, encoder3.4,
 *                           decoder0.0, decoder0.3, decoder1.0, decoder1.3, decoder2.0, decoder2.2, decoder2.4; BatchNorm2d
 *                           encoder1.3, encoder2.3, encoder3.3, decoder0.2, decoder1.2 (weight, bias, running_mean, running_var)
 *   iile_iispt_net_forward  in_dev: (n, 7, 32, 32) floats as `read_input` (ml/main_stdio_net.py:47-72) builds them; out_dev:
 *                           (n, 3, 32, 32) as `output_to_stdout` (:77-86) reads them; both DEVICE memory. The kernels are
 *                           queued on `stream` (NULL = the null stream) and the call returns; activations live in a workspace
 *                           the object owns (1.19 MiB per probe of a batch, at most max_batch probes at a time; <= 0: 8192 — no
 *                           faster beyond. If the device cannot hold that many the batch is halved until it can: a speed matter only).
 *                           layer_out_dev != NULL (tests): also copies the NHWC output of convolution `layer` (0..13) there. */
typedef struct iile_iispt_net iile_iispt_net;
typedef struct iile_iispt_net_weights {
    const float *conv_weight[15], *conv_bias[15];
    const float *bn_weight[5], *bn_bias[5], *bn_mean[5], *bn_var[5];
    float bn_eps;   /* BatchNorm2d's eps (1e-5) */
} iile_iispt_net_weights;
int iile_iispt_net_create(const iile_iispt_net_weights *weights, iile_iispt_net **out);
/* The same from a flat file (hosts without Python; binding.save_net_weights writes it from a checkpoint's state_dict): the 8 bytes
 * "IILENET1", BatchNorm2d's eps, then the float32 tensors in the order above, convolutions {weight, bias} first. */
int iile_iispt_net_load(const char *path, iile_iispt_net **out);
int iile_iispt_net_forward(iile_iispt_net *net, const float *in_dev, float *out_dev, int32_t n, int32_t max_batch, void *stream,
                           float *layer_out_dev, int32_t layer);
/* The network with the two transforms IisptRenderRunner applies around it, as one call over a batch of rendered probes:
 * normalizeMapsDownstream (src/integrators/iisptrenderrunner.cpp:1041-1092) -> IISPTNet -> transformMapsUpstream (:1095-1133).
 * intensity_dev / normals_dev: (n, 32, 32, 3), distance_dev: (n, 32, 32), raster order as iile_render_probes leaves them (device
 * memory); pred_dev: (n, 32, 32, 3) predicted intensity, rescaled per channel to the rendered probe's mean — film_rows != 0: rows in
 * ImageFilm order (row 0 = the top scanline: what iile_iispt_gather reads as nn_films), else raster order. Means are double sums,
 * log(1.0 + v) / exp(v) - 1.0 are evaluated in double as the reference does, everything else in float. slot_of_probe_dev: null, or
 * n indices (device memory) — probe i's prediction is then written to image slot_of_probe_dev[i] of pred_dev instead of image i (the
 * integrator renders probes for the valid hemi points only and hands the gather one image per hemi point). Queued on `stream`. */
int iile_iispt_net_predict(iile_iispt_net *net, const float *intensity_dev, const float *normals_dev, const float *distance_dev,
                           float *pred_dev, const int32_t *slot_of_probe_dev, int32_t n, int32_t film_rows, int32_t max_batch, void *stream);
void iile_iispt_net_destroy(iile_iispt_net *net);
/* BVHAccel's HLBVH build (src/accelerators/bvh.cpp:404-472: Morton codes :413-427, RadixSort :133-181, treelets and
 * emitLBVH :434-452, 555-618, buildUpperSAH :474-553) and flattenBVHTree (:640-658) — SURVEY.md §8 f4. bounds6: per
 * primitive WorldBound() as {min xyz, max xyz} (host memory); nodes_out: room for 2 * n_prims nodes; order_out[i] = the
 * number of the primitive at position i of BVHAccel::primitives after the build. The tree is the one the reference
 * builds with one thread (treelets in index order); Morton codes, the sort, the treelets, buildUpperSAH (level by level, a wavefront
 * per span) and the flattening all run on the device. The signature doubles as the `bvh_build` hook of
 * iile_host_overrides (include/iile_host.h). */
typedef struct iile_bvh_build_stats {
    float ms_total, ms_morton, ms_sort, ms_treelets, ms_upper, ms_flatten, ms_download;
    int32_t n_treelets, n_nodes, n_interior, n_leaf;
} iile_bvh_build_stats;
int iile_bvh_build_hlbvh(int32_t n_prims, const float *bounds6, int32_t max_prims_in_node, iile_bvh_node *nodes_out,
                         int32_t *n_nodes_out, int32_t *order_out, iile_bvh_build_stats *stats);
/* Test probe: the traversal kernels' records of a flattened tree as iile_scene_create packs them on the device —
 * wide16: 16 floats per interior node (two child boxes, refs, axis), wide4_32: 32 floats (four grandchild boxes as six
 * SoA planes, refs, axes); *nested = every child box lies inside its parent's. Layout: DESIGN.md §3. */
int iile_bvh_pack_probe(int32_t n_nodes, const iile_bvh_node *nodes, int32_t n_interior, float *wide16, float *wide4_32,
                        int32_t *nested);
/* How this build packs the refs of a four-wide record: 0 = the plain ref and an axes word of its own; 2 = ref << 2 | split
 * axis (of the node, its first child, its second child) — one vector load less per interior step. */
int32_t iile_wide_ref_shift(void);
/* ImageTexture<RGBSpectrum, Spectrum>::Evaluate (src/textures/imagemap.h:87-94) of image texture `tex` at n
 * surface points given by (u, v) and the screen-space differentials {du/dx, dv/dx, du/dy, dv/dy}. */
int iile_texture_eval(iile_scene *scene, int32_t tex, int32_t n, const float *uv2, const float *duv4, float *rgb3);
/* portable sin / cos / acos used on the device: out[3i..] = {sin x, cos x, acos clamp(x)} */
int iile_trig_probe(int32_t n, const float *x, float *out3);

#ifdef __cplusplus
}
#endif
#endif /* IILE_GPU_H */
