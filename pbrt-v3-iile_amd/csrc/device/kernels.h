// kernels.h — launch interface between api.hip and kernels.hip.
#pragma once
#include <string>

#include "../../../include/iile_scene.h"
#include "dscene.h"

namespace iile {

// One render pass = all pixels of the owned tiles x a chunk of sample indices.
struct PassDesc {
    // tile ownership (SamplerIntegrator::Render's 16x16 tiles, integrator.cpp:235-237)
    int n_tiles_x, n_tiles_y, tile_rank, tile_nranks, n_owned_tiles;
    // the rank's tiles in tile index order (its "slots") and the inverse map (-1: another rank's tile); both null
    // when one rank owns every tile (slot == tile index). Ownership: iile_tile_owner (iile_scene.h).
    const int *tile_of_slot, *slot_of_tile;
    // a pass renders ALL samples [k0, k0 + kc) of the owned tiles [slot0, slot0 + n_pass_tiles): whole tiles, so that
    // every FilmTile sum is complete when the pass ends (path id = ((slot - slot0) * 256 + pixel) * kc + k - k0)
    int slot0, n_pass_tiles;
    int k0, kc;
    uint32_t n_paths;  // n_pass_tiles * 256 * kc, or the explicit list length
    // explicit path list (kernel-level tests); null for tile enumeration
    const int *list_px, *list_py, *list_k;
    // IISPT probe batch: n_owned_tiles = n_probes * tiles per probe, tile slot = probe * tiles + tile
    int probe_mode, probe_tiles, probe_stx;  // probe pass: storage tiles per probe / per row (16 x 16 slots over the film's pixel bounds)
    const DProbeCam *probe_cams;
    // camera rays are generated inside the first k_extend and rebuilt in the first k_shade: no k_generate, no ray
    // queue for bounce 0 (run_pass decides; needs L cleared and counts[kCntRay] set beforehand)
    int gen_fused;
    // The path loop breaks right after intersecting the ray of bounce maxDepth (path.cpp:104), and that intersection adds
    // emitted light — from the surface hit or, for an escaped ray, from the infinite lights — only after a specular bounce
    // (path.cpp:91-101). Set by run_pass for uninstrumented passes: 1 = k_shade of bounce maxDepth - 1 keeps only the
    // continuations sampled from a specular lobe; 2 = the scene has no specular lobe: no continuation is sampled there and
    // bounce maxDepth is not launched at all.
    int skip_last_bounce;
    // the IISPT direct pass (kernels_direct.hip): pass seed 6284 + 17 * pass, number of 2D sample arrays a pixel's stream
    // starts with (5 levels x lights x 2), and the PCG32 jump table {A, G} per array start + one for the camera sample
    uint32_t direct_seed;
    int direct_arrays;
    const unsigned long long *direct_jump;
    // Light::nSamples per light (directprogressiveintegrator.cpp:9-18: nLightSamples; killeroo-simple's area light says 8) and
    // their sum: UniformSampleAllLights makes that many EstimateDirect calls per vertex, each with a result slot of its own
    int direct_nsamples[8];
    int direct_total_samples;
    int direct_levels;         // vertices along Li's recursion that can exist: 5 with a specular lobe in the scene, else 1
};
static_assert(kMaxLights <= 8, "PassDesc::direct_nsamples");

// Queue arrays (ray_o/ray_d/hits/shade_q/nee) hold `queue_cap` slots: the paths of a
// pass plus the padding that block-reserved appends leave behind (kernels.hip).
// shade-queue entries carry the shading class above the slot number
constexpr int kSlotBits = 28;
constexpr int kCntWords = 128;  // words of PassBuffers::counts (layout in kcommon.h)
struct PassBuffers {
    uint32_t queue_cap;
    float4 *L;          // [n_paths] radiance so far (xyz)
    float4 *beta;       // [n_paths] throughput (xyz), w = bitcast sampler dimension
    uint32_t *hindex;   // [n_paths] Halton index of the sample
    float *eta_scale;   // [n_paths] etaScale of path.cpp:81 (touched only when the scene has glass)
    float4 *ray_o[2];   // ping-pong ray queues
    float4 *ray_d[2];
    // the path's state travels with its ray from bounce 1 on (dense, in queue order) instead of being fetched by path id
    // (scattered: 64-byte sectors for 16 + 4 bytes): ray_s = (throughput xyz, Halton index), and the ray's direction record
    // carries sampler dimension | specularBounce << 16 in .w where tMax (always infinite there) stood
    float4 *ray_s[2];
    float4 *hits;       // [n_paths]
    float4 *mis_hit;    // [queue_cap] k_mis -> k_mis_lit: where a MIS ray met an emitter (indexed by NEE record)
    // second set of the NEE arrays (odd bounces): k_shade of bounce b + 1 writes its records while k_shadow of bounce b
    // still reads (run_pass); null when the workspace holds one set
    float4 *nee_alt, *mis_hit_alt;
    uint8_t *nee_mis_alt;
    float4 *nee;        // 7 planes of queue_cap float4
    uint8_t *nee_mis;   // [queue_cap] k_mis: area light index + 1 the MIS ray ended on, 0 = none;
                        // after k_mis_lit: 1 = it reached the sampled light on its emitting side
    uint32_t *shade_q;  // [n_paths] queue slots whose ray hit something (input of shade)
    uint32_t *counts;   // kCntWords words: queue sizes and chunk cursors per bounce (layout in kernels.hip)
    DCounters *counters;
    uint32_t *nray_out; // optional [2*n_paths] per-path {closest, shadow} ray counts (tests)
    int *spill;         // [kSpillStackDepth][max grid threads] overflow of the LDS traversal stacks
    float4 *aux;        // probe pass only: [n_paths] camera-space normal (xyz) and distance (w) of the first hit
    // camera samples whose film position is a whole number after float rounding (they also land in a neighbouring
    // pixel, film.h:159-166): {px, py, k, pFilm.x, pFilm.y, path id} records, appended by the generation code, count in [0]
    uint32_t *flag_count;
    float *flag_rec;    // [kMaxFlagged * 6]
    // the IISPT direct pass: per vertex depth d < 5 and path, emitted light E[d * dir_paths + path] and the mirror lobe's
    // {f, |cos|} F[d * dir_paths + path]; the lights' direct light lands in L[(d * n_lights + light) * dir_paths + path]
    float4 *dir_E, *dir_F;
    // ... and, in textured scenes with specular lobes, the differentials of the ray that reaches the path's current vertex
    // (SpecularReflect, directprogressiveintegrator.cpp:165-184): {rxOrigin, has}, {ryOrigin}, {rxDirection}, {ryDirection} at
    // [plane * dir_paths + path]; null otherwise
    float4 *dir_RD;
    uint32_t dir_paths;
};
constexpr uint32_t kMaxFlagged = 1u << 20;

struct FilmBuffers {
    float4 *tile_rgbw;  // [n_owned_tiles*256] per-pixel RGB contribSum + weight of own samples
    float4 *k0_rgbv;    // [n_owned_tiles*256] guarded radiance of sample k=0 (xyz), w = splat mask bits
    float4 *film_xyzw;  // [crop_w*crop_h] output {X,Y,Z,weightSum}
    // filters wider than one pixel: every sample of the frame, [owned pixel slot][k]
    float4 *wide_L;     // guarded radiance (rgb)
    float2 *wide_pf;    // pFilm
};

// The exact finish of film pixels reached by samples with whole-number film positions, on the device (kernels.hip
// "exact film finish"; DESIGN.md section 4 "Whole-number film positions"): hit records {destination pixel, flagged sample},
// a hash table destination -> list of its hits, and the exact FilmTile sums ("entries") the passes of a frame produce.
struct PatchDev {
    uint32_t *counters;   // [0] hits of this pass, [1] entries of this render, [2] error bits (1: list overflow, 2: table full)
    uint4 *hits;          // {destination film index, flagged sample, next hit of the same destination, -}
    uint32_t *keys;       // open-addressing table over film indices (0xffffffff: empty) ...
    uint32_t *heads;      // ... and the head of each one's list (0xffffffff: none)
    uint32_t table_mask;
    uint4 *ent_a;         // {film index, tile, nonplain, next entry of the same film index}
    float4 *ent_b;        // {r, g, b, w}: what that tile's FilmTile holds for that pixel
    uint32_t cap_hits, cap_entries;
};
constexpr uint32_t kPatchNil = 0xffffffffu;

struct LaunchCfg {
    int n_cus;
    hipStream_t stream;
    bool count_stats;
    int trav_blocks_per_cu = 0;   // persistent traversal blocks per CU; 0 = default_trav_blocks_per_cu()
};

// api.hip: records the message iile_last_error() returns, hands back `code`
int api_fail(int code, const std::string &msg);
// bvh_build.hip: two-wide (4 float4) and four-wide (8 float4) records per interior node of a flattened tree in HBM
// (record index = rank of the node among the interior nodes in depth-first order, or d_remap[rank] when a record order is
// given: a permutation of [0, n_interior)); *nested_out = every child box lies inside its parent's
int pack_wide_records(const iile_bvh_node *d_nodes, int n_nodes, int n_interior, float4 *d_wide, float4 *d_wide4, int *nested_out,
                      const int *d_remap = nullptr);

// slots a queue needs for n_paths paths
uint32_t queue_capacity(uint32_t n_paths, int n_cus);
void launch_generate(const DScene &S, const PassDesc &P, const PassBuffers &B, const LaunchCfg &cfg);
void launch_extend(const DScene &S, const PassDesc &P, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg);
void launch_shade(const DScene &S, const PassDesc &P, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg);
void launch_shadow(const DScene &S, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg);
void launch_mis(const DScene &S, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg);
void launch_light_distributions(const DScene &S, const float *samples, float *out, const LaunchCfg &cfg);
constexpr int kLightDistFloats = 2 * kMaxLights + 2;  // per voxel
void launch_miss(const DScene &S, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg);
void launch_mis_lit(const DScene &S, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg);
void launch_film_accumulate(const DScene &S, const PassDesc &P, const PassBuffers &B, const FilmBuffers &F,
                            const LaunchCfg &cfg);
void launch_film_store(const DScene &S, const PassDesc &P, const PassBuffers &B, const FilmBuffers &F, int k_begin, int n_samples,
                       const LaunchCfg &cfg);
void launch_film_gather(const DScene &S, const PassDesc &P, const FilmBuffers &F, int n_samples, const LaunchCfg &cfg);
// the probe pass's film store + gather + finish in one kernel; false (nothing launched): the probe film is too large for it
bool launch_probe_film(const DScene &S, const PassDesc &P, const PassBuffers &B, int n_probes, float *intensity, float *normals, float *distance,
                       const LaunchCfg &cfg);
void launch_direct_generate(const DScene &S, const PassDesc &P, const PassBuffers &B, const LaunchCfg &cfg);
void launch_direct_shade(const DScene &S, const PassDesc &P, const PassBuffers &B, int depth, uint32_t max_rays, const LaunchCfg &cfg);
void launch_direct_tree(const DScene &S, const PassDesc &P, const PassBuffers &B, double *film_rgbw, const LaunchCfg &cfg);
void launch_direct_miss(const DScene &S, const PassBuffers &B, int depth, uint32_t max_rays, const LaunchCfg &cfg);
void launch_direct_fold(const DScene &S, const PassDesc &P, const PassBuffers &B, double *film_rgbw, const LaunchCfg &cfg);
void launch_probe_finish(const DScene &S, const PassDesc &P, const PassBuffers &B, const FilmBuffers &F, int n_probes,
                         float *intensity, float *normals, float *distance, const LaunchCfg &cfg);
void launch_film_resolve(const DScene &S, const PassDesc &P, const FilmBuffers &F, const LaunchCfg &cfg);

// ---- the IISPT runner's gather (iispt.hip) ----
struct DHemiCam {  // a hemi point's HemisphericCamera as the gather uses it (cameras/hemispheric.h:23-82)
    M44 c2w;       // CameraToWorld
    float w2c[9];  // upper 3 x 3 of WorldToCamera's matrix (row major)
    float look[3], origin[3];
    int valid;
};
// Per-item records of the gather's small wavefront pipeline (float4 planes of n_items each; layout in iispt.hip)
struct IisptItems {
    float4 *ro, *rd, *beta, *hit, *pf;
    uint32_t *idx;
    uint32_t *n_active;
    int n_items, n_hemi, nx;
};
// One task of a batch as the kernels see it (an array of these in HBM, one per blockIdx.y): the task, its slice of the item
// planes, and where its hemi-point outputs / cameras / predicted hemispheres / film pixels are
struct IisptJob {
    iile_iispt_task T;
    IisptItems I;
    int ny;
    uint8_t *valid;  // hemi_out
    float *pos3, *dir3;
    const DHemiCam *cams;  // gather
    const float *nn_films;
    float4 *out;
};
// camera samples + find_intersection for every item (hemi points, then film pixels) of every job; synchronises cfg.stream
void launch_iispt_first_hits(const DScene &S, const IisptJob *jobs, int n_jobs, int max_items, uint32_t *n_active, int *spill, const LaunchCfg &cfg);
void launch_iispt_hemi_out(const DScene &S, const IisptJob *jobs, int n_jobs, int max_hemi, const LaunchCfg &cfg);
void launch_iispt_gather(const DScene &S, const IisptJob *jobs, int n_jobs, int max_pixels, const float *jac, const LaunchCfg &cfg);
void launch_iispt_film_add(const int4 *rects, const uint32_t *first, int n_tasks, int max_pixels, const float4 *out, double *film, int film_w, hipStream_t stream);
void launch_iispt_film_merge(const double *a, const double *b, float *rgb, long long n, hipStream_t stream);

// kernel-level entry points for parity tests
void launch_trace(const DScene &S, int n, const float4 *ro, const float4 *rd, float4 *hits, int any_hit,
                  DCounters *counters, int *spill, const LaunchCfg &cfg);
// number of ints of the HBM spill array: kSpillStackDepth x the largest traversal grid
uint32_t max_traversal_threads(int n_cus);
int default_trav_blocks_per_cu();
void launch_halton(const DScene &S, int n, const int *px, const int *py, const int *k, int dim0, int ndims,
                   float *out, uint32_t *index_out, const LaunchCfg &cfg);
void launch_camera(const DScene &S, int n, const float *pfilm, const float *plens, float *o, float *d,
                   const LaunchCfg &cfg);
void launch_bsdf_probe(const DScene &S, int n, int mat, const float *wo, const float *wi_or_u, int sample,
                       float *out, const LaunchCfg &cfg);
void launch_trig_probe(int n, const float *x, float *out, const LaunchCfg &cfg);
// one pass's flagged samples -> entries (after the pass's k_film_accumulate); all entries -> film (after k_film_resolve)
void launch_patch_pass(const DScene &S, const PassDesc &P, const PassBuffers &B, const FilmBuffers &F, const PatchDev &D, const LaunchCfg &cfg);
void launch_patch_merge(const DScene &S, const PassDesc &P, const FilmBuffers &F, const PatchDev &D, const LaunchCfg &cfg);
void launch_patch_own(const DScene &S, const float4 *L, int n, const uint32_t *local_slot, const uint32_t *range3, const uint32_t *flag_pid,
                      int kc, float4 *out, const LaunchCfg &cfg);
void launch_gather4(const float4 *src, const uint32_t *idx, int n, float4 *out, const LaunchCfg &cfg);
void launch_scatter4(float4 *dst, const uint32_t *idx, int n, const float4 *in, const LaunchCfg &cfg);
void launch_texture_probe(const DScene &S, int n, int tex, const float *uv, const float *duv, float *out, const LaunchCfg &cfg);

}  // namespace iile
