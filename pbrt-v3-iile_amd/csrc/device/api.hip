// api.hip — C ABI of libiile_gpu.so (see include/iile_gpu.h): scene upload,
// wavefront scheduling of the kernels in kernels.hip, film download.
//
// There is no CPU fallback anywhere in this file: every compute entry point
// needs a HIP device and fails loudly without one.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <limits>
#include <map>
#include <numeric>
#include <vector>

#include "../../../include/iile_gpu.h"
#include "kernels.h"

using namespace iile;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
}  // namespace
namespace iile {
int api_fail(int code, const std::string &msg) { return fail(code, msg); }  // the other translation units' errors
}  // namespace iile
namespace {
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(IILE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

struct EventPair {
    hipEvent_t a, b;
    int kind;  // 0 generate, 1 extend, 2 shade, 3 shadow, 4 film, 5 mis
};

}  // namespace

struct iile_scene {
    DScene ds;
    std::vector<void *> allocs;
    int n_cus = 256;
    int max_depth = 5;
    int spp = 1;
    int light_samples[8] = {1, 1, 1, 1, 1, 1, 1, 1};  // Light::nSamples (iile_light::n_samples), the direct pass's nLightSamples
    // wavefront workspace, grown on demand and kept across renders
    uint32_t ws_paths = 0;
    uint64_t ws_bytes = 0;
    PassBuffers pb;
    void *ws_block = nullptr;
    uint32_t *nray_buf = nullptr;
    uint32_t nray_cap = 0;
    // film workspace
    uint32_t film_tiles = 0;
    uint32_t film_pixels = 0;
    FilmBuffers fb;
    iile_probe_setup probe;
    const uint32_t *probe_pixel_offsets = nullptr;
    const float *probe_filter_table = nullptr;
    uint32_t *flag_count = nullptr;  // whole-number film positions (PassBuffers::flag_count / flag_rec)
    float *flag_rec = nullptr;
    float *flag_host = nullptr;      // pinned staging for the records (count first)
    size_t flag_host_floats = 0;
    hipStream_t aux_stream = nullptr;  // copies the flagged list to the host beside a running pass
    hipEvent_t ev_flags = nullptr;     // recorded after the first k_extend of a pass: the list is complete
    void *probe_block = nullptr;  // cameras + aux + outputs of the last probe batch
    size_t probe_block_bytes = 0;
    void *film_block = nullptr;
    void *wide_block = nullptr;
    uint64_t film_wide = 0;
    int *spill = nullptr;  // HBM overflow of the LDS traversal stacks
    // the NEE kernels of bounce b (k_mis, k_mis_lit, k_shadow) run on a stream of their own beside k_extend of bounce
    // b + 1: each fills the other's tail (run_pass)
    hipStream_t nee_stream = nullptr;
    int *spill_nee = nullptr;
    hipEvent_t ev_shade[16] = {}, ev_nee[16] = {};
    std::vector<EventPair> events;
    size_t events_used = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    // tile ownership of the last sharded render (iile_tile_owner): the rank's tiles in index order and the inverse
    std::vector<int> tile_of_slot, slot_of_tile;
    int *d_tile_tables = nullptr;  // tile_of_slot, then slot_of_tile
    size_t d_tile_tables_ints = 0;
    int map_key[4] = {0, 0, 0, 0};  // {n_tiles_x, n_tiles_y, rank, nranks} the tables were built for
    int slot_of(int tile) const { return slot_of_tile.empty() ? tile : slot_of_tile[size_t(tile)]; }
    // the exact film finish on the device (kernels.hip): hit / entry records and the hash table, allocated at the first render
    PatchDev patch{};
    void *patch_block = nullptr;
    void *film_add_buf = nullptr;        // iile_iispt_film_add's task table (rectangles, first pixels)
    size_t film_add_cap = 0;
    uint32_t patch_cap_override = 0;     // iile_test_patch_capacity: hits / entries capacity forced by a test
    bool overflow_unchecked = false;     // an asynchronous render left patch.counters[2] unread
    hipStream_t overflow_stream = nullptr;
    // grow-only device scratch of the host-side film finish (IILE_DEBUG_HOST_FILM_FINISH: index lists in, gathered values out)
    char *scratch = nullptr;
    size_t scratch_cap = 0, scratch_used = 0;
};

namespace {

template <typename T>
int upload(iile_scene *sc, const T *host, size_t n, const T **dev) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    sc->allocs.push_back(p);
    if (n) HIP_TRY(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<const T *>(p);
    return IILE_OK;
}

int ensure_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(IILE_ERR_NO_DEVICE,
                    "no HIP device available: libiile_gpu has no CPU fallback (hipGetDeviceCount: " +
                        std::string(e == hipSuccess ? "0 devices" : hipGetErrorString(e)) + ")");
    return IILE_OK;
}

int ensure_workspace(iile_scene *sc, uint32_t n_paths) {
    if (n_paths <= sc->ws_paths) return IILE_OK;
    if (sc->ws_block) {
        HIP_TRY(hipFree(sc->ws_block));
        sc->ws_block = nullptr;
        sc->ws_paths = 0;
    }
    const size_t n = n_paths;
    const size_t cap = queue_capacity(n_paths, sc->n_cus);
    // per path: L, beta (float4), hindex; per queue slot: ray_o[2], ray_d[2], hits, 2 x {nee[7], mis_hit} (float4), shade_q
    const size_t f4 = sizeof(float4);
    size_t bytes = 2 * n * f4 + 2 * n * sizeof(uint32_t) + 23 * cap * f4 + cap * sizeof(uint32_t) + 2 * cap +
                   kCntWords * sizeof(uint32_t) + sizeof(DCounters) + 16384;
    void *blk = nullptr;
    const auto t_alloc = std::chrono::steady_clock::now();
    HIP_TRY(hipMalloc(&blk, bytes));
    if (std::getenv("IILE_TIMING"))   // (a fresh process's large allocation can wait seconds for memory another process has just freed)
        fprintf(stderr, "iile timing: workspace of %.1f GiB for %u paths allocated in %.3f s\n", double(bytes) / 1073741824.0, n_paths,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_alloc).count());
    sc->ws_block = blk;
    sc->ws_bytes = bytes;
    char *p = static_cast<char *>(blk);
    auto take = [&](size_t b) {
        char *r = p;
        p += (b + 255) & ~size_t(255);
        return r;
    };
    PassBuffers &B = sc->pb;
    B.queue_cap = uint32_t(cap);
    B.L = reinterpret_cast<float4 *>(take(n * f4));
    B.beta = reinterpret_cast<float4 *>(take(n * f4));
    B.ray_o[0] = reinterpret_cast<float4 *>(take(cap * f4));
    B.ray_o[1] = reinterpret_cast<float4 *>(take(cap * f4));
    B.ray_d[0] = reinterpret_cast<float4 *>(take(cap * f4));
    B.ray_d[1] = reinterpret_cast<float4 *>(take(cap * f4));
    B.ray_s[0] = reinterpret_cast<float4 *>(take(cap * f4));
    B.ray_s[1] = reinterpret_cast<float4 *>(take(cap * f4));
    B.hits = reinterpret_cast<float4 *>(take(cap * f4));
    B.nee = reinterpret_cast<float4 *>(take(7 * cap * f4));
    B.mis_hit = reinterpret_cast<float4 *>(take(cap * f4));
    B.nee_alt = reinterpret_cast<float4 *>(take(7 * cap * f4));
    B.mis_hit_alt = reinterpret_cast<float4 *>(take(cap * f4));
    B.nee_mis_alt = reinterpret_cast<uint8_t *>(take(cap));
    B.hindex = reinterpret_cast<uint32_t *>(take(n * sizeof(uint32_t)));
    B.eta_scale = reinterpret_cast<float *>(take(n * sizeof(float)));
    B.shade_q = reinterpret_cast<uint32_t *>(take(cap * sizeof(uint32_t)));
    B.nee_mis = reinterpret_cast<uint8_t *>(take(cap));
    B.counts = reinterpret_cast<uint32_t *>(take(kCntWords * sizeof(uint32_t)));
    B.counters = reinterpret_cast<DCounters *>(take(sizeof(DCounters)));
    B.nray_out = nullptr;
    B.spill = sc->spill;
    sc->ws_paths = n_paths;
    return IILE_OK;
}

// Room for the exact film finish (kernels.hip), sized from the frame: a flagged camera sample reaches at most three more
// pixels (hits), every hit becomes at most one tile sum (entries, kept for the whole render). paths_per_pass / samples_in_render
// bound the flagged samples of a pass / of the render; the flagged list itself holds kMaxFlagged records per pass.
int ensure_patch(iile_scene *sc, uint64_t paths_per_pass, uint64_t samples_in_render) {
    PatchDev &D = sc->patch;
    uint64_t want_hits = 3 * std::min<uint64_t>(kMaxFlagged, paths_per_pass);
    uint64_t want_entries = 3 * std::min<uint64_t>(kMaxFlagged, samples_in_render);
    want_hits = std::max<uint64_t>(want_hits, 4096);
    want_entries = std::max<uint64_t>(want_entries, 4096);
    if (sc->patch_cap_override) want_hits = want_entries = sc->patch_cap_override;   // iile_test_patch_capacity
    if (sc->patch_block && want_hits == D.cap_hits && want_entries == D.cap_entries) return IILE_OK;
    if (sc->patch_block && !sc->patch_cap_override && want_hits <= D.cap_hits && want_entries <= D.cap_entries) return IILE_OK;
    if (sc->patch_block) HIP_TRY(hipFree(sc->patch_block));
    sc->patch_block = nullptr;
    D.cap_hits = uint32_t(want_hits);
    D.cap_entries = uint32_t(want_entries);
    uint32_t table = 1024;
    while (table < 2 * D.cap_entries) table <<= 1;
    D.table_mask = table - 1;
    const size_t bytes = 256 + size_t(D.cap_hits) * sizeof(uint4) + size_t(table) * 8 + size_t(D.cap_entries) * (sizeof(uint4) + sizeof(float4));
    void *blk = nullptr;
    if (hipMalloc(&blk, bytes) != hipSuccess) return fail(IILE_ERR_HIP, "out of device memory for the exact film finish");
    sc->patch_block = blk;
    char *p = static_cast<char *>(blk);
    D.counters = reinterpret_cast<uint32_t *>(p);
    p += 256;
    D.hits = reinterpret_cast<uint4 *>(p);
    p += size_t(D.cap_hits) * sizeof(uint4);
    D.keys = reinterpret_cast<uint32_t *>(p);
    p += size_t(table) * 4;
    D.heads = reinterpret_cast<uint32_t *>(p);
    p += size_t(table) * 4;
    D.ent_a = reinterpret_cast<uint4 *>(p);
    p += size_t(D.cap_entries) * sizeof(uint4);
    D.ent_b = reinterpret_cast<float4 *>(p);
    return IILE_OK;
}

// The exact film finish reports running out of room through patch.counters[2]; a render that returns before its stream has
// drained (film on the device, no statistics) cannot look. Whoever waits next does: iile_render_status, or the next iile_render.
int check_pending_overflow(iile_scene *sc) {
    if (!sc->overflow_unchecked || !sc->patch_block) return IILE_OK;
    HIP_TRY(hipStreamSynchronize(sc->overflow_stream));
    uint32_t pc[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(pc, sc->patch.counters, sizeof(pc), hipMemcpyDeviceToHost));
    sc->overflow_unchecked = false;
    if (pc[2] != 0)
        return fail(IILE_ERR_UNSUPPORTED, "the exact film finish of the previous asynchronous iile_render ran out of room (camera samples with whole-number "
                                          "film positions: more than 2^20 in one pass, or more pixel hits / tile sums than the frame was sized for): "
                                          "that film is wrong");
    return IILE_OK;
}

int ensure_film(iile_scene *sc, uint32_t n_tiles, uint32_t n_pixels, uint64_t n_wide = 0) {
    if (n_wide > sc->film_wide) {  // the frame's samples for a wide pixel filter: 24 B each
        if (sc->wide_block) HIP_TRY(hipFree(sc->wide_block));
        sc->wide_block = nullptr;
        sc->film_wide = 0;
        void *w = nullptr;
        if (hipMalloc(&w, size_t(n_wide) * 24 + 256) != hipSuccess)
            return fail(IILE_ERR_HIP, "out of device memory for the sample store of a wide pixel filter (24 B per camera sample)");
        sc->wide_block = w;
        sc->film_wide = n_wide;
        sc->fb.wide_L = reinterpret_cast<float4 *>(w);
        sc->fb.wide_pf = reinterpret_cast<float2 *>(static_cast<char *>(w) + size_t(n_wide) * 16);
    }
    if (n_tiles <= sc->film_tiles && n_pixels <= sc->film_pixels) return IILE_OK;
    if (sc->film_block) {
        HIP_TRY(hipFree(sc->film_block));
        sc->film_block = nullptr;
    }
    n_tiles = std::max(n_tiles, sc->film_tiles);
    n_pixels = std::max(n_pixels, sc->film_pixels);
    const size_t tile_bytes = size_t(n_tiles) * 256 * sizeof(float4);
    const size_t bytes = 2 * tile_bytes + size_t(n_pixels) * sizeof(float4) + 1024;
    void *blk = nullptr;
    HIP_TRY(hipMalloc(&blk, bytes));
    sc->film_block = blk;
    char *p = static_cast<char *>(blk);
    sc->fb.tile_rgbw = reinterpret_cast<float4 *>(p);
    sc->fb.k0_rgbv = reinterpret_cast<float4 *>(p + tile_bytes);
    sc->fb.film_xyzw = reinterpret_cast<float4 *>(p + 2 * tile_bytes);
    sc->film_tiles = n_tiles;
    sc->film_pixels = n_pixels;
    return IILE_OK;
}

int get_events(iile_scene *sc, int kind, EventPair **out) {
    if (sc->events_used == sc->events.size()) {
        EventPair ep;
        HIP_TRY(hipEventCreate(&ep.a));
        HIP_TRY(hipEventCreate(&ep.b));
        sc->events.push_back(ep);
    }
    *out = &sc->events[sc->events_used++];
    (*out)->kind = kind;
    return IILE_OK;
}

// The rank's share of SamplerIntegrator::Render's tile grid (iile_tile_owner, iile_scene.h) as two tables in HBM:
// slot -> tile (tile index order) and tile -> slot. One rank owning everything needs none (slot == tile).
int ensure_tile_map(iile_scene *sc, PassDesc *P, hipStream_t stream) {
    const int ntx = P->n_tiles_x, nty = P->n_tiles_y, n_tiles = ntx * nty, rank = P->tile_rank, nranks = P->tile_nranks;
    P->tile_of_slot = P->slot_of_tile = nullptr;
    if (nranks <= 1) {
        sc->tile_of_slot.clear();
        sc->slot_of_tile.clear();
        sc->map_key[3] = 0;
        P->n_owned_tiles = n_tiles;
        return IILE_OK;
    }
    const int key[4] = {ntx, nty, rank, nranks};
    if (std::memcmp(key, sc->map_key, sizeof(key)) != 0 || sc->slot_of_tile.size() != size_t(n_tiles)) {
        sc->tile_of_slot.clear();
        sc->slot_of_tile.assign(size_t(n_tiles), -1);
        for (int t = 0; t < n_tiles; ++t)
            if (iile_tile_owner(t % ntx, t / ntx, nranks) == rank) {
                sc->slot_of_tile[size_t(t)] = int(sc->tile_of_slot.size());
                sc->tile_of_slot.push_back(t);
            }
        const size_t ints = sc->tile_of_slot.size() + sc->slot_of_tile.size();
        if (ints > sc->d_tile_tables_ints) {
            if (sc->d_tile_tables) HIP_TRY(hipFree(sc->d_tile_tables));
            sc->d_tile_tables = nullptr;
            sc->d_tile_tables_ints = 0;
            HIP_TRY(hipMalloc(&sc->d_tile_tables, std::max<size_t>(ints, 1) * sizeof(int)));
            sc->d_tile_tables_ints = ints;
        }
        // (synchronous copies from pageable vectors: a few KB, once per change of the sharding)
        HIP_TRY(hipStreamSynchronize(stream));
        if (!sc->tile_of_slot.empty())
            HIP_TRY(hipMemcpy(sc->d_tile_tables, sc->tile_of_slot.data(), sc->tile_of_slot.size() * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(sc->d_tile_tables + sc->tile_of_slot.size(), sc->slot_of_tile.data(), sc->slot_of_tile.size() * sizeof(int),
                          hipMemcpyHostToDevice));
        std::memcpy(sc->map_key, key, sizeof(key));
    }
    P->n_owned_tiles = int(sc->tile_of_slot.size());
    P->tile_of_slot = sc->d_tile_tables;
    P->slot_of_tile = sc->d_tile_tables + sc->tile_of_slot.size();
    return IILE_OK;
}

void copy_counters(const DCounters &c, iile_stats *st) {
    st->camera_rays = c.camera_rays;
    st->closest_rays = c.closest_rays;
    st->shadow_rays = c.shadow_rays;
    st->nodes_closest = c.nodes_closest;
    st->nodes_any = c.nodes_any;
    st->tri_tests = c.tri_tests;
    st->tri_hits = c.tri_hits;
    st->sphere_tests = c.sphere_tests;
    st->nee_evals = c.nee_evals;
    st->zero_radiance = c.zero_radiance;
    for (int i = 0; i < 8; ++i) st->path_length[i] = c.path_length[i];
    st->ext_rays = c.ext_rays;
    st->ext_nodes = c.ext_nodes;
    st->ext_tri_tests = c.ext_tri_tests;
    st->ext_sphere_tests = c.ext_sphere_tests;
    st->any_tri_tests = c.any_tri_tests;
    st->mis_rays_traced = c.mis_traced;
    st->ext_rays_traced = c.ext_traced;
}

// Enqueue one wavefront pass on cfg.stream.
int run_pass(iile_scene *sc, const DScene &S, int max_depth, const PassDesc &P_in, const LaunchCfg &cfg, bool timed, bool one_stream = false) {
    PassBuffers &B = sc->pb;
    PassDesc P = P_in;
    // camera rays made inside the first extend / shade (see PassDesc::gen_fused) where nothing else reads queue 0
    P.gen_fused = !cfg.count_stats && !P.list_px && !S.has_infinite && !S.probe_mode && !B.nray_out;
    HIP_TRY(hipMemsetAsync(B.counts, 0, kCntWords * sizeof(uint32_t), cfg.stream));
    auto timed_launch_on = [&](hipStream_t stream, int kind, auto &&fn) -> int {
        EventPair *ep = nullptr;
        if (timed) {
            int rc = get_events(sc, kind, &ep);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ep->a, stream));
        }
        fn();
        if (timed) HIP_TRY(hipEventRecord(ep->b, stream));
        return IILE_OK;
    };
    auto timed_launch = [&](int kind, auto &&fn) -> int { return timed_launch_on(cfg.stream, kind, fn); };
    // Two streams: the shadow / MIS rays of bounce b and the extension rays of bounce b + 1 both hang on k_shade of bounce
    // b and on nothing else of each other (k_shadow accumulates into L, k_extend reads the ray queue), so they run side
    // by side and each fills the idle compute units of the other's tail; k_shade of bounce b + 1 waits for both (it
    // overwrites the NEE records, and may add emitted light to L after the NEE contribution of bounce b as path.cpp
    // does). Not with infinite lights (k_miss adds to L between the two) and not in the instrumented pass.
    const bool two_streams = sc->nee_stream && !one_stream && !cfg.count_stats && !S.has_infinite && max_depth < 15;
    // Without specular lobes k_shade touches L at bounce 0 only, and with the NEE arrays doubled (even / odd bounces) it
    // need not wait for k_shadow of the bounce before: the NEE stream then trails the main one by up to a bounce.
    const bool nee_doubled = two_streams && !S.has_specular && B.nee_alt;
    LaunchCfg cfg_nee = cfg;
    if (two_streams) cfg_nee.stream = sc->nee_stream;
    auto buffers_of = [&](int bounce, bool nee_side) {
        PassBuffers X = B;
        if (nee_doubled && (bounce & 1)) X.nee = B.nee_alt, X.mis_hit = B.mis_hit_alt, X.nee_mis = B.nee_mis_alt;
        if (two_streams && nee_side) X.spill = sc->spill_nee;
        return X;
    };
    int rc = IILE_OK;
    if (P.gen_fused) {
        // no k_generate: queue 0 is the dense range of path ids (and the first k_extend zeroes each path's L)
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&B.counts[0]), int(P.n_paths), 1, cfg.stream));
    } else {
        rc = timed_launch(0, [&] { launch_generate(S, P, B, cfg); });
        if (rc) return rc;
    }
    // bounces 0 .. maxDepth: the path loop exits at `bounces >= maxDepth` after
    // intersecting (path.cpp:104), so maxDepth + 1 extend launches are needed — to reproduce the reference's ray count.
    // The radiance does not need the last of them unless a specular bounce or an infinite light can add emitted light
    // at that vertex (path.cpp:91-101): the uninstrumented pass of a scene with neither leaves the bounce out.
    // With specular lobes (or an infinite light) around, the rays that leave the last shaded vertex through a specular
    // lobe are the only ones whose intersection can still add something: the others are dropped there (1); without
    // either, the whole bounce is (2).
    P.skip_last_bounce = 0;
    if (!cfg.count_stats && max_depth >= 1 && !S.probe_mode && !B.nray_out)
        P.skip_last_bounce = S.has_specular ? 1 : 2;
    const int last_bounce = (P.skip_last_bounce == 2) ? max_depth - 1 : max_depth;
    for (int b = 0; b <= last_bounce; ++b) {
        rc = timed_launch(1, [&] { launch_extend(S, P, B, b, B.queue_cap, cfg); });
        if (rc) return rc;
        if (b == 0 && B.flag_count && sc->ev_flags) HIP_TRY(hipEventRecord(sc->ev_flags, cfg.stream));
        if (S.has_infinite) {  // escaped rays see the infinite lights (path.cpp:97-99)
            rc = timed_launch(6, [&] { launch_miss(S, B, b, B.queue_cap, cfg); });
            if (rc) return rc;
        }
        if (two_streams && !nee_doubled && b > 0) HIP_TRY(hipStreamWaitEvent(cfg.stream, sc->ev_nee[b - 1], 0));
        if (nee_doubled && b > 1) HIP_TRY(hipStreamWaitEvent(cfg.stream, sc->ev_nee[b - 2], 0));  // its records are overwritten now
        const PassBuffers B_shade = buffers_of(b, false), B_nee = buffers_of(b, true);
        rc = timed_launch(2, [&] { launch_shade(S, P, B_shade, b, B.queue_cap, cfg); });
        if (rc) return rc;
        if (b < max_depth) {
            if (two_streams) {
                HIP_TRY(hipEventRecord(sc->ev_shade[b], cfg.stream));
                HIP_TRY(hipStreamWaitEvent(cfg_nee.stream, sc->ev_shade[b], 0));
            }
            // MIS rays first: the shadow kernel finishes each record (L += beta * Ld)
            rc = timed_launch_on(cfg_nee.stream, 5, [&] { launch_mis(S, B_nee, b, B.queue_cap, cfg_nee); });
            if (rc) return rc;
            rc = timed_launch_on(cfg_nee.stream, 6, [&] { launch_mis_lit(S, B_nee, b, B.queue_cap, cfg_nee); });
            if (rc) return rc;
            rc = timed_launch_on(cfg_nee.stream, 3, [&] { launch_shadow(S, B_nee, b, B.queue_cap, cfg_nee); });
            if (rc) return rc;
            if (two_streams) HIP_TRY(hipEventRecord(sc->ev_nee[b], cfg_nee.stream));
        }
    }
    if (two_streams && max_depth > 0) HIP_TRY(hipStreamWaitEvent(cfg.stream, sc->ev_nee[max_depth - 1], 0));
    HIP_TRY(hipGetLastError());
    return IILE_OK;
}

int collect_times(iile_scene *sc, iile_stats *st) {
    for (size_t i = 0; i < sc->events_used; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, sc->events[i].a, sc->events[i].b));
        switch (sc->events[i].kind) {
        case 0: st->ms_generate += ms; break;
        case 1: st->ms_extend += ms; st->n_extend_launches++; break;
        case 2: st->ms_shade += ms; st->n_shade_launches++; break;
        case 3: st->ms_connect += ms; st->ms_shadow += ms; st->n_connect_launches++; break;
        case 5: st->ms_connect += ms; st->ms_mis += ms; break;
        case 6: st->ms_connect += ms; st->ms_resolve += ms; break;
        default: st->ms_film += ms; break;
        }
    }
    return IILE_OK;
}

}  // namespace

namespace {
template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t n) {
        HIP_TRY(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
        return IILE_OK;
    }
    int put(const T *h, size_t n) {
        int rc = alloc(n);
        if (rc) return rc;
        if (n) HIP_TRY(hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice));
        return IILE_OK;
    }
    int get(T *h, size_t n) {
        if (n) HIP_TRY(hipMemcpy(h, p, n * sizeof(T), hipMemcpyDeviceToHost));
        return IILE_OK;
    }
};
}  // namespace

namespace {
// Device scratch for the finish: reserve once per use (may reallocate: nothing of an earlier use is live), then carve.
int scratch_reserve(iile_scene *sc, size_t bytes, hipStream_t stream) {
    sc->scratch_used = 0;
    if (bytes <= sc->scratch_cap) return IILE_OK;
    HIP_TRY(hipStreamSynchronize(stream));
    if (sc->scratch) HIP_TRY(hipFree(sc->scratch));
    sc->scratch = nullptr;
    sc->scratch_cap = 0;
    const size_t want = std::max<size_t>(2 * bytes, size_t(1) << 20);
    HIP_TRY(hipMalloc(&sc->scratch, want));
    sc->scratch_cap = want;
    return IILE_OK;
}
template <typename T>
T *scratch_take(iile_scene *sc, size_t n) {
    char *p = sc->scratch + sc->scratch_used;
    sc->scratch_used += (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~size_t(255);
    return reinterpret_cast<T *>(p);
}
template <typename T>
size_t scratch_bytes(size_t n) {
    return (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~size_t(255);
}
template <typename T>
int scratch_put(iile_scene *sc, const std::vector<T> &h, hipStream_t stream, T **dev) {
    *dev = scratch_take<T>(sc, h.size());
    if (!h.empty()) HIP_TRY(hipMemcpyAsync(*dev, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, stream));
    return IILE_OK;
}

}  // namespace

extern "C" {

const char *iile_last_error(void) { return g_err.c_str(); }

int iile_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int iile_device_select(int32_t device) {
    int rc = ensure_device();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    return IILE_OK;
}
int iile_device_alloc(uint64_t bytes, void **out_dev) {
    if (!out_dev) return fail(IILE_ERR_ARG, "iile_device_alloc: null argument");
    int rc = ensure_device();
    if (rc) return rc;
    HIP_TRY(hipMalloc(out_dev, std::max<size_t>(size_t(bytes), 1)));
    return IILE_OK;
}
void iile_device_free(void *dev) {
    if (dev) (void)hipFree(dev);
}
int iile_device_download(void *dst_host, const void *src_dev, uint64_t bytes, void *stream) {
    if (!dst_host || !src_dev) return fail(IILE_ERR_ARG, "iile_device_download: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(dst_host, src_dev, size_t(bytes), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return IILE_OK;
}

int iile_device_upload(void *dst_dev, const void *src_host, uint64_t bytes, void *stream) {
    if (!dst_dev || !src_host) return fail(IILE_ERR_ARG, "iile_device_upload: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(dst_dev, src_host, size_t(bytes), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));   // (the host buffer may be reused on return)
    return IILE_OK;
}
// A stream of the caller's own for hosts built without hipcc (the C++ IISPT host runs its whole indirect pass on one): a
// non-blocking stream, i.e. one that does not synchronise with the null stream.
int iile_stream_create(void **out_stream) {
    if (!out_stream) return fail(IILE_ERR_ARG, "iile_stream_create: null argument");
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out_stream = s;
    return IILE_OK;
}
int iile_stream_wait(void *stream) {
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return IILE_OK;
}
void iile_stream_destroy(void *stream) {
    if (stream) (void)hipStreamDestroy(static_cast<hipStream_t>(stream));
}
int iile_device_zero(void *dev, uint64_t bytes, void *stream) {
    if (!dev) return fail(IILE_ERR_ARG, "iile_device_zero: null argument");
    HIP_TRY(hipMemsetAsync(dev, 0, size_t(bytes), static_cast<hipStream_t>(stream)));
    return IILE_OK;
}

static_assert(kLightDiffuseArea == IILE_LIGHT_DIFFUSE_AREA && kLightPoint == IILE_LIGHT_POINT &&
                  kLightSpot == IILE_LIGHT_SPOT && kLightDistant == IILE_LIGHT_DISTANT &&
                  kLightAreaTriangle == IILE_LIGHT_AREA_TRIANGLE && kLightInfinite == IILE_LIGHT_INFINITE,
              "light type codes");
static_assert(kMatMatte == IILE_MAT_MATTE && kMatPlastic == IILE_MAT_PLASTIC && kMatUber == IILE_MAT_UBER &&
                  kMatMirror == IILE_MAT_MIRROR && kMatGlass == IILE_MAT_GLASS,
              "material type codes");
int iile_scene_create(const iile_scene_desc *d, iile_scene **out) {
    if (!d || !out) return fail(IILE_ERR_ARG, "iile_scene_create: null argument");
    int rc = ensure_device();
    if (rc) return rc;
    // ---- what the device path supports -------------------------------------
    if (d->n_prims >= (1 << 24)) return fail(IILE_ERR_UNSUPPORTED, "more than 2^24 primitives");
    if (d->n_spheres > kMaxSpheres || d->n_materials > kMaxMaterials || d->n_lights > kMaxLights)
        return fail(IILE_ERR_UNSUPPORTED, "too many spheres / materials / lights");
    for (int i = 0; i < d->n_lights; ++i) {
        const iile_light &l = d->lights[i];
        if (l.type == IILE_LIGHT_DIFFUSE_AREA) {
            if (l.sphere < 0 || l.sphere >= d->n_spheres) return fail(IILE_ERR_ARG, "area light without a sphere");
        } else if (l.type == IILE_LIGHT_AREA_TRIANGLE) {
            if (l.prim < 0 || l.prim >= d->n_prims || (d->prim_flags[l.prim] & IILE_PRIM_SPHERE) || d->prim_light[l.prim] != i)
                return fail(IILE_ERR_ARG, "triangle area light without its triangle");
        } else if (l.type != IILE_LIGHT_POINT && l.type != IILE_LIGHT_SPOT && l.type != IILE_LIGHT_DISTANT &&
                   l.type != IILE_LIGHT_INFINITE) {
            return fail(IILE_ERR_UNSUPPORTED, "unsupported light type");
        }
    }
    for (int i = 0; i < d->n_materials; ++i)
        if (d->materials[i].type < IILE_MAT_MATTE || d->materials[i].type > IILE_MAT_GLASS)
            return fail(IILE_ERR_UNSUPPORTED, "unsupported material type");
    if (d->halton.n_dims > kMaxHaltonDims) return fail(IILE_ERR_UNSUPPORTED, "too many Halton dimensions");
    const int need_dims = 5 + 8 * d->integrator.max_depth + 1;
    if (d->halton.n_dims < need_dims) return fail(IILE_ERR_ARG, "Halton table covers too few dimensions for maxdepth");
    if (d->integrator.max_depth > 14) return fail(IILE_ERR_UNSUPPORTED, "maxdepth > 14");
    if ((double(d->halton.spp) + 1) * double(d->halton.sample_stride) >= 4294967296.0)
        return fail(IILE_ERR_UNSUPPORTED, "Halton index exceeds 32 bits (pixelsamples too large)");
    if (!(d->film.filter_rx > 0) || !(d->film.filter_ry > 0) || d->film.filter_rx > 16 || d->film.filter_ry > 16)
        return fail(IILE_ERR_UNSUPPORTED, "pixel filter radius must lie in (0, 16]");
    if (!d->film_filter_wide && (d->film.filter_rx != 0.5f || d->film.filter_ry != 0.5f))
        return fail(IILE_ERR_ARG, "film_filter_wide must be set for any filter but the box of radius 0.5");

    iile_scene *sc = new iile_scene;
    std::memset(&sc->ds, 0, sizeof(sc->ds));
    std::memset(&sc->pb, 0, sizeof(sc->pb));
    std::memset(&sc->fb, 0, sizeof(sc->fb));
    DScene &S = sc->ds;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        sc->n_cus = prop.multiProcessorCount;

    auto bail = [&](int code) {
        iile_scene_destroy(sc);
        return code;
    };
    // BVH: the depth-first LinearBVHNode array (bvh.cpp:640-658) is validated here and re-packed on the device
    // (bvh_build.hip, pack_wide_records) into the two-wide records {children[0] box, children[1] box, refs, axis} of the
    // instrumented kernels and the four-wide records of dpath.h trav_interior4. A reference is the interior record
    // index, or ~firstPrimitive for a leaf child.
    std::vector<uint32_t> last_in_leaf(size_t(d->n_prims), 0);
    {
        const int n = d->n_nodes;
        int n_interior = 0;
        for (int i = 0; i < n; ++i) {
            const iile_bvh_node &nd = d->nodes[i];
            if (nd.nprims > 0) {
                if (nd.offset < 0 || nd.offset + nd.nprims > d->n_prims) return bail(fail(IILE_ERR_ARG, "bad leaf range"));
                last_in_leaf[size_t(nd.offset) + nd.nprims - 1] = 16u;
                continue;
            }
            if (i + 1 >= n || nd.offset <= i || nd.offset >= n) return bail(fail(IILE_ERR_ARG, "bad BVH child index"));
            ++n_interior;
        }
        const iile_bvh_node *d_nodes = nullptr;
        rc = upload(sc, d->nodes, size_t(std::max(n, 0)), &d_nodes);
        if (rc) return bail(rc);
        float4 *wide = nullptr, *wide4 = nullptr;
        if (hipMalloc(&wide, 4 * size_t(std::max(n_interior, 1)) * sizeof(float4)) != hipSuccess)
            return bail(fail(IILE_ERR_HIP, "out of device memory for the BVH records"));
        sc->allocs.push_back(wide);
        if (hipMalloc(&wide4, 8 * size_t(std::max(n_interior, 1)) * sizeof(float4)) != hipSuccess)
            return bail(fail(IILE_ERR_HIP, "out of device memory for the BVH records"));
        sc->allocs.push_back(wide4);
        // The four-wide step never tests the two children themselves; that is exact because a child's box lies inside
        // its parent's (Union in recursiveBuild is exact). pack_wide_records verifies it for the tree we were handed; a
        // tree that violates it is traversed with binary steps only.
        S.boxes_nested = 1;
        // (where the records sit in memory is free — a reference is a record slot — and worth nothing: depth-first rank 476.2 ms,
        // scattered 476.0 on the room, profiles/r04_ab_traversal_scheduling.txt)
        rc = pack_wide_records(d_nodes, n, n_interior, wide, wide4, &S.boxes_nested, nullptr);
        if (rc) return bail(rc);
        // the four-wide step addresses its records with 32-bit byte offsets and gives two bits of every ref to a split axis:
        // leaf refs ~prim must survive the shift
        if (n_interior >= (1 << 25) || (kRefShift && d->n_prims >= (1 << 28))) S.boxes_nested = 0;
        S.wide = wide;
        S.wide4 = wide4;
        if (n > 0) {
            for (int c = 0; c < 3; ++c) {
                S.root_box[c] = d->nodes[0].bmin[c];
                S.root_box[3 + c] = d->nodes[0].bmax[c];
            }
            S.root_ref = d->nodes[0].nprims == 0 ? 0 : ~d->nodes[0].offset;  // the root is interior rank 0
        }
        // The top of the four-wide tree, breadth first, for the traversal kernels' LDS copies (dpath.h, load_wide4): the
        // records are read back once, the references among the chosen ones become kTopFlag | slot, each copy keeps its own
        // record index (the binary fallback step needs it) in the word behind its axes.
        S.top4 = nullptr;
        S.n_top = 0;
        S.root_ref_top = S.root_ref;
        int want_top = kMaxTop;
        if (n_interior > 0 && S.boxes_nested && want_top > 0 && S.root_ref >= 0) {
            std::vector<float4> all(8 * size_t(n_interior));
            if (hipMemcpy(all.data(), wide4, all.size() * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess)
                return bail(fail(IILE_ERR_HIP, "reading back the BVH records failed"));
            std::vector<int> order;       // record index per slot
            std::map<int, int> slot_of;   // record index -> slot
            order.push_back(S.root_ref);
            slot_of[S.root_ref] = 0;
            for (size_t at = 0; at < order.size() && int(order.size()) < want_top; ++at) {
                const float4 refs = all[8 * size_t(order[at]) + 6];
                const float rf[4] = {refs.x, refs.y, refs.z, refs.w};
                for (int j = 0; j < 4 && int(order.size()) < want_top; ++j) {
                    int r;
                    std::memcpy(&r, &rf[j], sizeof(r));
                    r >>= kRefShift;  // (the low bits are a split axis)
                    // (an empty slot — the second one of a leaf child — holds no box: its planes are +-inf and its ref is unused)
                    const float bmin_x = (&all[8 * size_t(order[at]) + 0].x)[j];
                    if (r < 0 || r >= n_interior || !(bmin_x < std::numeric_limits<float>::infinity()) || slot_of.count(r)) continue;
                    slot_of[r] = int(order.size());
                    order.push_back(r);
                }
            }
            std::vector<float4> top(8 * order.size());
            for (size_t sl = 0; sl < order.size(); ++sl) {
                for (int q = 0; q < 8; ++q) top[8 * sl + q] = all[8 * size_t(order[sl]) + q];
                float *refs = &top[8 * sl + 6].x;
                for (int j = 0; j < 4; ++j) {
                    int raw;
                    std::memcpy(&raw, &refs[j], sizeof(raw));
                    const int r = raw >> kRefShift;
                    const float bmin_x = (&top[8 * sl + 0].x)[j];
                    if (r >= 0 && r < n_interior && bmin_x < std::numeric_limits<float>::infinity() && slot_of.count(r)) {
                        const int tagged = int(uint32_t(kTopFlag | slot_of[r]) << kRefShift) | (raw & ((1 << kRefShift) - 1));
                        std::memcpy(&refs[j], &tagged, sizeof(raw));
                    }
                }
                std::memcpy(&top[8 * sl + 7].y, &order[sl], sizeof(int));
            }
            const float4 *d_top = nullptr;
            rc = upload(sc, top.data(), top.size(), &d_top);
            if (rc) return bail(rc);
            S.top4 = d_top;
            S.n_top = int(order.size());
            S.root_ref_top = kTopFlag | 0;
        }
    }
    // primitives: gather into 48-byte vertex records + normal / uv records
    {
        const size_t n = size_t(d->n_prims);
        std::vector<float4> verts(3 * n + 3), norms(3 * n);  // one pad record, flagged last-in-leaf
        std::vector<float2> uvs(3 * n);
        for (size_t i = 0; i < n; ++i) {
            const float *p = d->tri_p + 9 * i, *nn = d->tri_n + 9 * i, *uv = d->tri_uv + 6 * i;
            // flag word: bits 0..3 iile_scene.h, bit 4 last primitive of its leaf, bits 5..7 shading
            // class (material type, +4 for a sphere), bits 8..11 area light index + 1
            const int mt = d->prim_material[i] >= 0 ? d->materials[d->prim_material[i]].type : 3;
            const uint32_t cls = uint32_t(mt < 0 ? 3 : (mt > 3 ? 3 : mt)) | ((d->prim_flags[i] & 1u) ? 4u : 0u);
            const bool masked = (d->prim_flags[i] & IILE_PRIM_HAS_ALPHA) && d->prim_alpha &&
                                (d->prim_alpha[2 * i] != IILE_ALPHA_NONE || d->prim_alpha[2 * i + 1] != IILE_ALPHA_NONE);
            if (masked) S.has_alpha = 1;  // bit 12 of the flag word
            uint32_t w[3] = {(d->prim_flags[i] & 15u) | (masked ? 4096u : 0u) | last_in_leaf[i] | (cls == 7u ? 6u : cls) << 5 | (uint32_t(d->prim_light[i] + 1) << 8),
                             uint32_t(d->prim_material[i]), uint32_t(d->prim_light[i])};
            for (int k = 0; k < 3; ++k) {
                float wf;
                std::memcpy(&wf, &w[k], 4);
                verts[3 * i + k] = make_float4(p[3 * k], p[3 * k + 1], p[3 * k + 2], wf);
                norms[3 * i + k] = make_float4(nn[3 * k], nn[3 * k + 1], nn[3 * k + 2], 0.f);
                uvs[3 * i + k] = make_float2(uv[2 * k], uv[2 * k + 1]);
            }
            if ((d->prim_flags[i] & IILE_PRIM_SPHERE) && (d->prim_light[i] >= 0) &&
                d->lights[d->prim_light[i]].sphere != d->prim_shape[i])
                return bail(fail(IILE_ERR_ARG, "light / sphere cross reference is inconsistent"));
        }
        {
            const uint32_t last = 16u;
            float lf;
            std::memcpy(&lf, &last, 4);
            verts[3 * n] = verts[3 * n + 1] = verts[3 * n + 2] = make_float4(0, 0, 0, lf);
        }
        rc = upload(sc, verts.data(), verts.size(), &S.tri_verts);
        if (rc) return bail(rc);
        rc = upload(sc, norms.data(), norms.size(), &S.tri_norms);
        if (rc) return bail(rc);
        rc = upload(sc, uvs.data(), uvs.size(), &S.tri_uv);
        if (rc) return bail(rc);
        rc = upload(sc, d->prim_shape, n, &S.prim_shape);
        if (rc) return bail(rc);
        if (S.has_alpha) {
            std::vector<int2> masks(n);
            for (size_t i = 0; i < n; ++i) {
                masks[i] = make_int2(d->prim_alpha[2 * i], d->prim_alpha[2 * i + 1]);
                for (int m : {masks[i].x, masks[i].y})
                    if (m >= d->n_textures || m < IILE_ALPHA_ZERO) return bail(fail(IILE_ERR_ARG, "alpha mask refers to a texture that does not exist"));
            }
            rc = upload(sc, masks.data(), masks.size(), &S.prim_alpha);
            if (rc) return bail(rc);
        }
    }
    {
        std::vector<DSphere> sp(d->n_spheres);
        for (int i = 0; i < d->n_spheres; ++i) {
            const iile_sphere &s = d->spheres[i];
            std::memcpy(sp[i].o2w.m, s.o2w, 64);
            std::memcpy(sp[i].o2w_inv.m, s.o2w_inv, 64);
            sp[i].radius = s.radius;
            sp[i].zmin = s.zmin;
            sp[i].zmax = s.zmax;
            sp[i].theta_min = s.theta_min;
            sp[i].theta_max = s.theta_max;
            sp[i].phi_max = s.phi_max;
            sp[i].reverse_orientation = s.reverse_orientation;
            sp[i].swaps_handedness = s.swaps_handedness;
            // xf_point(o2w, (0, 0, 0)) (dmath.h), operation by operation: products with zero included, the division by w as
            // a multiplication with its reciprocal (this file is compiled without contraction or fast math, host side too)
            const float *m = s.o2w;
            const float zero = 0.f;
            float c[4];
            for (int r = 0; r < 4; ++r) c[r] = m[4 * r] * zero + m[4 * r + 1] * zero + m[4 * r + 2] * zero + m[4 * r + 3];
            if (c[3] != 1) {
                const float inv = 1.f / c[3];
                for (int r = 0; r < 3; ++r) c[r] = c[r] * inv;
            }
            for (int r = 0; r < 3; ++r) sp[i].center[r] = c[r];
            sp[i].pad_ = 0.f;
        }
        rc = upload(sc, sp.data(), sp.size(), &S.spheres);
        if (rc) return bail(rc);
        std::vector<DMaterial> mats(d->n_materials);
        for (int i = 0; i < d->n_materials; ++i) {
            const iile_material &m = d->materials[i];
            mats[i].type = m.type;
            for (int c = 0; c < 3; ++c) {
                mats[i].kd[c] = m.kd[c];
                mats[i].ks[c] = m.ks[c];
            }
            mats[i].alpha = m.alpha;
            mats[i].alpha_y = m.type == IILE_MAT_GLASS ? m.alpha_v : (m.type == IILE_MAT_UBER && m.rough_tex_v != -2) ? m.alpha_v : m.alpha;
            for (int c = 0; c < 3; ++c) mats[i].kr[c] = m.kr[c];
            for (int c = 0; c < 3; ++c) mats[i].kt[c] = m.kt[c];
            const bool oren_nayar = m.type == IILE_MAT_MATTE && m.sigma != 0;
            mats[i].on_a = oren_nayar ? m.on_a : 1.f;
            mats[i].on_b = oren_nayar ? m.on_b : 0.f;
            if (m.type == IILE_MAT_GLASS) S.has_glass = 1;
            for (int c = 0; c < 3; ++c) mats[i].opacity[c] = m.type == IILE_MAT_UBER ? m.opacity[c] : 1.f;
            if (m.type == IILE_MAT_UBER) {
                // uber.cpp:53-61, 94-99: a SpecularTransmission lobe exists if 1 - opacity or opacity x Kt is not black (an image for Kt: may be)
                bool trans = (m.kt_tex >= 0 || m.opacity_tex >= 0) && d->n_textures > 0;
                for (int c = 0; c < 3; ++c) {
                    const float op = m.opacity[c] > 0.f ? m.opacity[c] : 0.f;
                    trans = trans || (-op + 1.f) > 0.f || op * (m.kt[c] > 0.f ? m.kt[c] : 0.f) != 0.f;
                }
                if (trans) S.has_glass = 1, S.has_uber_trans = 1;   // (etaScale is tracked: path.cpp:151-157)
            }
            mats[i].eta = m.eta;
            mats[i].kd_tex = d->n_textures > 0 ? m.kd_tex : -1;
            mats[i].ks_tex = d->n_textures > 0 ? m.ks_tex : -1;
            mats[i].kr_tex = d->n_textures > 0 ? m.kr_tex : -1;
            mats[i].kt_tex = d->n_textures > 0 ? m.kt_tex : -1;
            mats[i].opacity_tex = (d->n_textures > 0 && m.type == IILE_MAT_UBER) ? m.opacity_tex : -1;
            mats[i].rough_tex_v = m.type == IILE_MAT_UBER ? ((m.rough_tex_v >= 0 && d->n_textures == 0) ? -1 : m.rough_tex_v) : -2;
            mats[i].bump_tex = d->n_textures > 0 ? m.bump_tex : -1;
            mats[i].rough_tex = d->n_textures > 0 ? m.rough_tex : -1;
            mats[i].sigma_tex = d->n_textures > 0 ? m.sigma_tex : -1;
            mats[i].remap_roughness = m.remap_roughness;
            for (int t : {mats[i].kd_tex, mats[i].ks_tex, mats[i].kr_tex, mats[i].kt_tex, mats[i].bump_tex, mats[i].rough_tex, mats[i].sigma_tex, mats[i].opacity_tex, mats[i].rough_tex_v})
                if (t >= 0) S.textured_materials = 1;
            for (int t : {mats[i].kd_tex, mats[i].ks_tex, mats[i].kr_tex, mats[i].kt_tex, mats[i].bump_tex, mats[i].rough_tex, mats[i].sigma_tex, mats[i].opacity_tex, mats[i].rough_tex_v})
                if (t >= d->n_textures) return bail(fail(IILE_ERR_ARG, "material refers to a texture that does not exist"));
        }
        rc = upload(sc, mats.data(), mats.size(), &S.materials);
        if (rc) return bail(rc);
        // image textures: the host-built pyramids, texels widened to float4 (one 16-byte load each)
        S.n_textures = d->n_textures;
        if (d->n_textures > 0) {
            std::vector<DTexture> tx(d->n_textures);
            for (int i = 0; i < d->n_textures; ++i) {
                const iile_texture &t = d->textures[i];
                if (t.n_levels < 1 || t.n_levels > kMaxTexLevels) return bail(fail(IILE_ERR_ARG, "texture with a bad level count"));
                tx[i].n_levels = t.n_levels;
                tx[i].wrap = t.wrap;
                tx[i].trilinear = t.trilinear;
                tx[i].max_aniso = t.max_aniso;
                tx[i].su = t.su, tx[i].sv = t.sv, tx[i].du = t.du, tx[i].dv = t.dv;
                for (int l = 0; l < kMaxTexLevels; ++l) {
                    tx[i].level_w[l] = l < t.n_levels ? t.level_w[l] : 1;
                    tx[i].level_h[l] = l < t.n_levels ? t.level_h[l] : 1;
                    tx[i].level_offset[l] = l < t.n_levels ? t.level_offset[l] : 0;
                    if (l < t.n_levels && (t.level_offset[l] < 0 || t.level_offset[l] + int64_t(t.level_w[l]) * t.level_h[l] > d->n_texels))
                        return bail(fail(IILE_ERR_ARG, "texture level outside the texel array"));
                }
            }
            rc = upload(sc, tx.data(), tx.size(), &S.textures);
            if (rc) return bail(rc);
            std::vector<float4> tex4(size_t(d->n_texels));
            for (int64_t i = 0; i < d->n_texels; ++i)
                tex4[size_t(i)] = make_float4(d->texels[3 * i], d->texels[3 * i + 1], d->texels[3 * i + 2], 0.f);
            rc = upload(sc, tex4.data(), tex4.size(), &S.texels);
            if (rc) return bail(rc);
            rc = upload(sc, d->ewa_lut, size_t(IILE_EWA_LUT_SIZE), &S.ewa_lut);
            if (rc) return bail(rc);
            // (a sphere's (u, v) = (phi / phiMax, (theta - thetaMin) / (thetaMax - thetaMin)), sphere.cpp:107-109: sphere_interaction)
        }
        std::vector<DLight> lts(d->n_lights);
        S.all_lights_infinite = d->n_lights > 0 ? 1 : 0;
        for (int i = 0; i < d->n_lights; ++i) {
            for (int c = 0; c < 3; ++c) lts[i].lemit[c] = d->lights[i].lemit[c];
            lts[i].two_sided = d->lights[i].two_sided;
            lts[i].sphere = d->lights[i].sphere;
            lts[i].type = d->lights[i].type;
            for (int c = 0; c < 3; ++c) lts[i].pos[c] = d->lights[i].pos[c];
            for (int c = 0; c < 9; ++c) lts[i].w2l[c] = d->lights[i].w2l[c];
            lts[i].cos_total_width = d->lights[i].cos_total_width;
            lts[i].cos_falloff_start = d->lights[i].cos_falloff_start;
            lts[i].world_radius = d->lights[i].world_radius;
            lts[i].prim = d->lights[i].prim;
            std::memcpy(lts[i].l2w, d->lights[i].l2w, sizeof(lts[i].l2w));
            lts[i].env_tex = d->lights[i].env_tex;
            lts[i].dist_w = d->lights[i].dist_w;
            lts[i].dist_h = d->lights[i].dist_h;
            lts[i].dist_offset = d->lights[i].dist_offset;
            if (d->lights[i].type != IILE_LIGHT_INFINITE) S.all_lights_infinite = 0;
            if (d->lights[i].type == IILE_LIGHT_INFINITE) {
                S.has_infinite = 1;
                const iile_light &il = d->lights[i];
                if (il.env_tex < 0 || il.env_tex >= d->n_textures || il.dist_w < 1 || il.dist_h < 1 || il.dist_offset < 0 ||
                    il.dist_offset + int64_t(2 * il.dist_w + 2) * il.dist_h + 2 * il.dist_h + 2 > d->n_env_dist)
                    return bail(fail(IILE_ERR_ARG, "infinite light: bad environment map / distribution reference"));
            }
        }
        rc = upload(sc, lts.data(), lts.size(), &S.lights);
        if (rc) return bail(rc);
        rc = upload(sc, d->env_dist, size_t(d->n_env_dist), &S.env_dist);
        if (rc) return bail(rc);
    }
    // Halton: permutations + per-dimension constants
    {
        const iile_halton &h = d->halton;
        rc = upload(sc, h.perms, size_t(h.n_perms), &S.perms);
        if (rc) return bail(rc);
        std::vector<DHaltonDim> dims(h.n_dims);
        for (int i = 0; i < h.n_dims; ++i) {
            const uint32_t base = uint32_t(h.primes[i]);
            dims[i].base = base;
            dims[i].perm_offset = uint32_t(h.prime_sums[i]);
            const float inv_base = 1.f / float(int(base));
            dims[i].inv_base = inv_base;
            dims[i].perm0_term = inv_base * h.perms[h.prime_sums[i]] / (1 - inv_base);
            dims[i].base_d = double(base);
            dims[i].inv_base_d = 1.0 / double(base);
        }
        rc = upload(sc, dims.data(), dims.size(), &S.hdims);
        if (rc) return bail(rc);
        S.n_hdims = h.n_dims;
        S.n_perms = h.n_perms;
        S.base_scale0 = h.base_scales[0];
        S.base_scale1 = h.base_scales[1];
        S.base_exp0 = h.base_exponents[0];
        S.base_exp1 = h.base_exponents[1];
        S.sample_stride = h.sample_stride;
        S.mult_inv0 = h.mult_inverse[0];
        S.mult_inv1 = h.mult_inverse[1];
        S.sample_center = h.sample_at_pixel_center;
        sc->spp = h.spp;
        // HaltonSampler::GetIndexForSample's per-pixel offset (halton.cpp:96-122) depends only on
        // the pixel modulo kMaxResolution = 128: tabulated once (integer arithmetic, exact)
        auto pixel_offsets = [](const int32_t *scales, const int32_t *exps, int32_t stride, const int32_t *mult_inv) {
            std::vector<uint32_t> offs(128 * 128, 0u);
            if (stride > 1) {
                for (int pmy = 0; pmy < 128; ++pmy)
                    for (int pmx = 0; pmx < 128; ++pmx) {
                        uint32_t inv = uint32_t(pmx), idx0 = 0, idx1 = 0;  // InverseRadicalInverse<2>, <3>
                        for (int i = 0; i < exps[0]; ++i) {
                            idx0 = idx0 * 2 + (inv & 1);
                            inv >>= 1;
                        }
                        inv = uint32_t(pmy);
                        for (int i = 0; i < exps[1]; ++i) {
                            idx1 = idx1 * 3 + inv % 3;
                            inv /= 3;
                        }
                        const unsigned long long off =
                            (unsigned long long)idx0 * (unsigned long long)(stride / scales[0]) * (unsigned long long)mult_inv[0] +
                            (unsigned long long)idx1 * (unsigned long long)(stride / scales[1]) * (unsigned long long)mult_inv[1];
                        offs[pmy * 128 + pmx] = uint32_t(off % (unsigned long long)stride);
                    }
            }
            return offs;
        };
        const std::vector<uint32_t> offs = pixel_offsets(h.base_scales, h.base_exponents, h.sample_stride, h.mult_inverse);
        // the IISPT probe pass has its own film and sampler (iile_probe_setup)
        sc->probe = d->probe;
        if (d->probe.hemi_size > 0 && d->probe.sample_stride > 0) {
            const std::vector<uint32_t> poffs =
                pixel_offsets(d->probe.base_scales, d->probe.base_exponents, d->probe.sample_stride, d->probe.mult_inverse);
            rc = upload(sc, poffs.data(), poffs.size(), &sc->probe_pixel_offsets);
            if (rc) return bail(rc);
            rc = upload(sc, d->probe.filter_table, size_t(256), &sc->probe_filter_table);
            if (rc) return bail(rc);
        }
        rc = upload(sc, offs.data(), offs.size(), &S.pixel_offsets);
        if (rc) return bail(rc);
    }
    // SobolSampler in place of the Halton sampler (iile_sobol): matrices [n_dims][32], then vdc[32], vdc_inv[32]
    S.sobol = 0;
    if (d->sobol.enabled) {
        const iile_sobol &sb = d->sobol;
        if (sb.n_dims < need_dims || sb.n_dims > 256 || !sb.matrices32 || sb.log2_resolution < 1 || sb.log2_resolution > 16 ||
            sb.resolution != (1 << sb.log2_resolution) || (uint64_t(sb.spp) << (2 * sb.log2_resolution)) > (uint64_t(1) << 32))
            return bail(fail(IILE_ERR_ARG, "iile_sobol: bad dimension count / resolution, or sample indices beyond 32 bits"));
        if (sb.resolution < d->film.samp_x1 - d->film.samp_x0 || sb.resolution < d->film.samp_y1 - d->film.samp_y0)
            return bail(fail(IILE_ERR_ARG, "iile_sobol: resolution smaller than the sample bounds"));
        std::vector<uint32_t> tab(size_t(sb.n_dims) * 32 + 64);
        std::memcpy(tab.data(), sb.matrices32, size_t(sb.n_dims) * 32 * sizeof(uint32_t));
        std::memcpy(tab.data() + size_t(sb.n_dims) * 32, sb.vdc, 32 * sizeof(uint32_t));
        std::memcpy(tab.data() + size_t(sb.n_dims) * 32 + 32, sb.vdc_inv, 32 * sizeof(uint32_t));
        rc = upload(sc, tab.data(), tab.size(), &S.sobol_mat);
        if (rc) return bail(rc);
        S.sobol_vdc = S.sobol_mat + size_t(sb.n_dims) * 32;
        {
            auto byte_tables = [](const uint32_t *cols, int n_cols, uint32_t *out) {  // out[4][256]
                for (int q = 0; q < 4; ++q)
                    for (int v = 0; v < 256; ++v) {
                        uint32_t x = 0;
                        for (int j = 0; j < 8; ++j)
                            if (((v >> j) & 1) && 8 * q + j < n_cols) x ^= cols[8 * q + j];
                        out[q * 256 + v] = x;
                    }
            };
            std::vector<uint32_t> bt(size_t(sb.n_dims) * 1024 + 2048);
            for (int dd = 0; dd < sb.n_dims; ++dd) byte_tables(sb.matrices32 + size_t(dd) * 32, 32, bt.data() + size_t(dd) * 1024);
            const int m2 = 2 * sb.log2_resolution;
            byte_tables(sb.vdc, 32 - m2, bt.data() + size_t(sb.n_dims) * 1024);          // bits of the sample number k
            byte_tables(sb.vdc_inv, m2, bt.data() + size_t(sb.n_dims) * 1024 + 1024);    // bits of the pixel word b
            rc = upload(sc, bt.data(), bt.size(), &S.sobol_bt);
            if (rc) return bail(rc);
            S.sobol_vdc_bt = S.sobol_bt + size_t(sb.n_dims) * 1024;
        }
        S.sobol = 1;
        S.sobol_log2res = sb.log2_resolution;
        S.sobol_res = sb.resolution;
        S.sobol_dims = sb.n_dims;
        sc->spp = sb.spp;
    }
    S.n_nodes = d->n_nodes;
    S.n_prims = d->n_prims;
    S.n_spheres = d->n_spheres;
    S.n_materials = d->n_materials;
    S.n_lights = d->n_lights;
    std::memcpy(S.raster_to_camera.m, d->camera.raster_to_camera, 64);
    std::memcpy(S.camera_to_world.m, d->camera.camera_to_world, 64);
    S.lens_radius = d->camera.lens_radius;
    S.focal_distance = d->camera.focal_distance;
    for (int c = 0; c < 3; ++c) {
        S.dx_camera[c] = d->camera.dx_camera[c];
        S.dy_camera[c] = d->camera.dy_camera[c];
    }
    S.diff_scale = 1 / std::sqrt(float(d->halton.spp));  // integrator.cpp:284-285
    S.filter_wide = d->film_filter_wide;
    rc = upload(sc, d->film_filter_table, size_t(256), &S.filter_table);
    if (rc) return bail(rc);
    const iile_film_desc &f = d->film;
    S.xres = f.xres;
    S.yres = f.yres;
    S.crop_x0 = f.crop_x0;
    S.crop_y0 = f.crop_y0;
    S.crop_x1 = f.crop_x1;
    S.crop_y1 = f.crop_y1;
    S.samp_x0 = f.samp_x0;
    S.samp_y0 = f.samp_y0;
    S.samp_x1 = f.samp_x1;
    S.samp_y1 = f.samp_y1;
    {   // "pixelbounds" (iile_integrator::pixel_bounds; all zero = not given, as a caller that never heard of the field leaves it)
        const int32_t *pb = d->integrator.pixel_bounds;
        const bool given = pb[0] != 0 || pb[1] != 0 || pb[2] != 0 || pb[3] != 0;
        S.pb_x0 = given ? std::max(pb[0], f.samp_x0) : f.samp_x0, S.pb_y0 = given ? std::max(pb[1], f.samp_y0) : f.samp_y0;
        S.pb_x1 = given ? std::min(pb[2], f.samp_x1) : f.samp_x1, S.pb_y1 = given ? std::min(pb[3], f.samp_y1) : f.samp_y1;
        S.pb_set = (S.pb_x0 != f.samp_x0 || S.pb_y0 != f.samp_y0 || S.pb_x1 != f.samp_x1 || S.pb_y1 != f.samp_y1) ? 1 : 0;
    }
    S.filter_rx = f.filter_rx;
    S.filter_ry = f.filter_ry;
    S.max_sample_luminance = f.max_sample_luminance;
    S.max_depth = d->integrator.max_depth;
    S.rr_threshold = d->integrator.rr_threshold;
    sc->max_depth = d->integrator.max_depth;
    for (int i = 0; i < d->n_lights && i < 8; ++i) sc->light_samples[i] = std::max(1, int(d->lights[i].n_samples));
    {
        void *p = nullptr;
        if (hipMalloc(&p, size_t(max_traversal_threads(sc->n_cus)) * sizeof(int)) != hipSuccess)
            return bail(fail(IILE_ERR_HIP, "hipMalloc(spill) failed"));
        sc->allocs.push_back(p);
        sc->spill = static_cast<int *>(p);
        if (hipMalloc(&p, size_t(max_traversal_threads(sc->n_cus)) * sizeof(int)) != hipSuccess)
            return bail(fail(IILE_ERR_HIP, "hipMalloc(spill) failed"));
        sc->allocs.push_back(p);
        sc->spill_nee = static_cast<int *>(p);
        if (hipStreamCreateWithFlags(&sc->nee_stream, hipStreamNonBlocking) != hipSuccess)
            return bail(fail(IILE_ERR_HIP, "hipStreamCreate failed"));
        for (int i = 0; i < 16; ++i)
            if (hipEventCreateWithFlags(&sc->ev_shade[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&sc->ev_nee[i], hipEventDisableTiming) != hipSuccess)
                return bail(fail(IILE_ERR_HIP, "hipEventCreate failed"));
    }
    if (hipEventCreate(&sc->ev_begin) != hipSuccess || hipEventCreate(&sc->ev_end) != hipSuccess)
        return bail(fail(IILE_ERR_HIP, "hipEventCreate failed"));
    {
        void *p = nullptr;
        if (hipMalloc(&p, 256 + size_t(kMaxFlagged) * 6 * sizeof(float)) != hipSuccess) return bail(fail(IILE_ERR_HIP, "hipMalloc(flag records) failed"));
        sc->allocs.push_back(p);
        sc->flag_count = static_cast<uint32_t *>(p);
        sc->flag_rec = reinterpret_cast<float *>(static_cast<char *>(p) + 256);
        if (hipStreamCreateWithFlags(&sc->aux_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&sc->ev_flags, hipEventDisableTiming) != hipSuccess)
            return bail(fail(IILE_ERR_HIP, "hipStreamCreate / hipEventCreate failed"));
    }
    // More than one light: tabulate the spatial light distribution (lightdistrib.cpp:91-299) for
    // every voxel of its grid — up to 64 per axis, cubes along the longest one.
    S.light_dist = nullptr;
    S.light_nv[0] = S.light_nv[1] = S.light_nv[2] = 1;
    if (d->n_lights > 1 && d->integrator.light_strategy != IILE_LIGHTS_SPATIAL) {
        // UniformLightDistribution / PowerLightDistribution (lightdistrib.cpp:65-82, integrator.cpp:217-225): one
        // Distribution1D for every point — a grid of a single voxel, built here (sampling.h:57-69)
        std::vector<float> tab(size_t(kLightDistStride), 0.f);
        const int n = d->n_lights;
        float *func = tab.data(), *cdf = tab.data() + kMaxLights;
        for (int i = 0; i < n; ++i) func[i] = d->integrator.light_strategy == IILE_LIGHTS_UNIFORM ? 1.f : d->integrator.light_power[i];
        cdf[0] = 0;
        for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / n;
        const float func_int = cdf[n];
        if (func_int == 0)
            for (int i = 1; i < n + 1; ++i) cdf[i] = float(i) / float(n);
        else
            for (int i = 1; i < n + 1; ++i) cdf[i] /= func_int;
        tab[2 * kMaxLights + 1] = func_int;
        rc = upload(sc, tab.data(), tab.size(), &S.light_dist);
        if (rc) return bail(rc);
    } else if (d->n_lights > 1 && d->n_nodes > 0) {
        const float diag[3] = {S.root_box[3] - S.root_box[0], S.root_box[4] - S.root_box[1], S.root_box[5] - S.root_box[2]};
        const int me = (diag[0] > diag[1] && diag[0] > diag[2]) ? 0 : (diag[1] > diag[2] ? 1 : 2);  // MaximumExtent
        const float bmax = diag[me];
        for (int i = 0; i < 3; ++i) S.light_nv[i] = std::max(1, int(std::round(diag[i] / bmax * 64)));
        // RadicalInverse(0..4, i), i < 128 (lowdiscrepancy.cpp:389-444): bases 2, 3, 5, 7, 11
        std::vector<float> samples(128 * 5);
        const int bases[5] = {2, 3, 5, 7, 11};
        for (int i = 0; i < 128; ++i)
            for (int b = 0; b < 5; ++b) {
                if (b == 0) {
                    uint64_t v = uint64_t(i), r = 0;  // ReverseBits64(a) * 0x1p-64
                    for (int k = 0; k < 64; ++k) r |= ((v >> k) & 1ull) << (63 - k);
                    samples[5 * i] = float(double(r) * 0x1p-64);
                    continue;
                }
                const float inv_base = 1.f / float(bases[b]);
                uint64_t a = uint64_t(i), rev = 0;
                float inv_base_n = 1;
                while (a) {
                    const uint64_t next = a / uint64_t(bases[b]);
                    rev = rev * uint64_t(bases[b]) + (a - next * uint64_t(bases[b]));
                    inv_base_n *= inv_base;
                    a = next;
                }
                samples[5 * i + b] = std::min(float(rev) * inv_base_n, 0x1.fffffep-1f);
            }
        const float *dsamples = nullptr;
        rc = upload(sc, samples.data(), samples.size(), &dsamples);
        if (rc) return bail(rc);
        const size_t n_vox = size_t(S.light_nv[0]) * S.light_nv[1] * S.light_nv[2];
        void *p = nullptr;
        if (hipMalloc(&p, n_vox * kLightDistFloats * sizeof(float)) != hipSuccess)
            return bail(fail(IILE_ERR_HIP, "hipMalloc(light distributions) failed"));
        sc->allocs.push_back(p);
        S.light_dist = static_cast<const float *>(p);
        LaunchCfg cfg{sc->n_cus, nullptr, false};
        launch_light_distributions(S, dsamples, static_cast<float *>(p), cfg);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess)
            return bail(fail(IILE_ERR_HIP, "light distribution kernel failed"));
    }
    // which build of k_shade the scene needs (kernels.hip launch_shade)
    S.extended_features = 0;
    S.has_specular = 0;
    for (int i = 0; i < d->n_materials; ++i)
        if (d->materials[i].type != IILE_MAT_MATTE && d->materials[i].type != IILE_MAT_PLASTIC) S.has_specular = 1;
    for (int i = 0; i < d->n_materials; ++i)
        if ((d->materials[i].type != IILE_MAT_MATTE && d->materials[i].type != IILE_MAT_PLASTIC) ||
            (d->materials[i].type == IILE_MAT_MATTE && d->materials[i].sigma != 0))
            S.extended_features = 1;
    if (d->n_lights > 1 || (d->n_lights == 1 && d->lights[0].type != IILE_LIGHT_DIFFUSE_AREA)) S.extended_features = 1;
    if (S.has_infinite) S.extended_features = 1;
    *out = sc;
    return IILE_OK;
}

void iile_scene_destroy(iile_scene *sc) {
    if (!sc) return;
    for (void *p : sc->allocs) (void)hipFree(p);
    if (sc->ws_block) (void)hipFree(sc->ws_block);
    if (sc->film_block) (void)hipFree(sc->film_block);
    if (sc->d_tile_tables) (void)hipFree(sc->d_tile_tables);
    if (sc->patch_block) (void)hipFree(sc->patch_block);
    if (sc->film_add_buf) (void)hipFree(sc->film_add_buf);
    if (sc->scratch) (void)hipFree(sc->scratch);
    if (sc->wide_block) (void)hipFree(sc->wide_block);
    if (sc->flag_host) (void)hipHostFree(sc->flag_host);
    if (sc->aux_stream) (void)hipStreamDestroy(sc->aux_stream);
    if (sc->ev_flags) (void)hipEventDestroy(sc->ev_flags);
    if (sc->nee_stream) (void)hipStreamDestroy(sc->nee_stream);
    for (int i = 0; i < 16; ++i) {
        if (sc->ev_shade[i]) (void)hipEventDestroy(sc->ev_shade[i]);
        if (sc->ev_nee[i]) (void)hipEventDestroy(sc->ev_nee[i]);
    }
    if (sc->probe_block) (void)hipFree(sc->probe_block);
    if (sc->nray_buf) (void)hipFree(sc->nray_buf);
    for (EventPair &e : sc->events) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    if (sc->ev_begin) (void)hipEventDestroy(sc->ev_begin);
    if (sc->ev_end) (void)hipEventDestroy(sc->ev_end);
    delete sc;
}

namespace {
// The one-pixel box film's exact finish. k_film_accumulate / k_film_resolve give every pixel the sum of its own
// samples (plus the k = 0 zero-offset splats) — all there is unless a sample's film position is a whole number, when
// FilmTile::AddSample (film.h:159-188) also adds it to a neighbouring pixel, in sample order inside its tile. The
// generation code listed those samples (rare: ~1e-4 of them where pixel coordinates pass 1024); here every pixel
// they touch is recomputed the way the reference sums it: per contributing tile, own samples and neighbours'
// samples merged in generation order (pixel-major, then k), tiles added in index order (MergeFilmTile, film.cpp:135-148;
// the reference merges in completion order, the oracle and this in index order). Radiances come from re-rendering
// the few paths involved through the explicit-list pass (bitwise the same values).
struct Flagged {
    int px, py, k;
    float pfx, pfy;
    uint32_t pid;
    int tile, pix;  // tile index and row-major pixel rank inside the tile: generation order = (tile, pix, k)
    bool plain_k0;  // sample 0 with exact zero offsets only: k_film_resolve adds it to its neighbours by itself
};
struct PatchDest {
    uint32_t film_index;
    int qx, qy, tile, pix;
    bool own_in_pass, need_own;  // its own tile is rendered by this pass; it receives from a pixel generated before it there
    uint32_t own_slot;           // index into tile_rgbw
    size_t first, last;          // its range of `hits`
    size_t own_first;            // first of its own samples in the gather list
};
// What patch_prepare works out on the host for one pass. It needs only the pass's list of flagged samples, complete once
// the pass's first k_extend has run — i.e. while the GPU is still busy with the rest of the pass.
struct PatchPlan {
    bool active = false;
    int k_begin = 0, k_end = 0, slot0 = 0;
    std::vector<Flagged> fl;
    std::vector<std::pair<uint32_t, int>> hits;  // (destination film pixel, flagged sample) by destination, then generation order
    std::vector<PatchDest> dests;
    std::vector<uint32_t> list_pid;              // path ids (of this pass) whose radiance is needed
    std::vector<int> list_of_flag;
};
// One exact FilmTile sum: what tile `tile` adds to film pixel `film_index` (rgb contribSum, weight sum)
struct PatchEntry {
    uint32_t film_index;
    int tile;
    float r, g, b, w;
    bool nonplain;  // involves a sample that k_film_resolve does not place
};
struct PatchTimer {
    bool on = false;   // (laps of the host-side film finish to stderr: flip when debugging IILE_DEBUG_HOST_FILM_FINISH)
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[patch] %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

int patch_prepare(iile_scene *sc, const DScene &S, const PassDesc &Pf, hipStream_t copy_stream, PatchPlan *plan) {
    PatchTimer tm;
    plan->active = false;
    plan->k_begin = Pf.k0, plan->k_end = Pf.k0 + Pf.kc, plan->slot0 = Pf.slot0;
    uint32_t n_flag = 0;
    if (sc->flag_host_floats < 64) {  // pinned staging: count first, then the records
        void *hp = nullptr;
        HIP_TRY(hipHostMalloc(&hp, (size_t(1) << 16) * sizeof(float), hipHostMallocDefault));
        if (sc->flag_host) (void)hipHostFree(sc->flag_host);
        sc->flag_host = static_cast<float *>(hp);
        sc->flag_host_floats = size_t(1) << 16;
    }
    HIP_TRY(hipMemcpyAsync(sc->flag_host, sc->flag_count, sizeof(uint32_t), hipMemcpyDeviceToHost, copy_stream));
    HIP_TRY(hipStreamSynchronize(copy_stream));
    std::memcpy(&n_flag, sc->flag_host, sizeof(n_flag));
    std::vector<Flagged> &fl = plan->fl;
    std::vector<std::pair<uint32_t, int>> &hits = plan->hits;
    std::vector<PatchDest> &dests = plan->dests;
    fl.clear(), hits.clear(), dests.clear(), plan->list_pid.clear(), plan->list_of_flag.clear();
    if (n_flag == 0) return IILE_OK;
    if (n_flag > kMaxFlagged) return fail(IILE_ERR_UNSUPPORTED, "more than 2^20 camera samples with whole-number film positions in one pass");
    if (sc->flag_host_floats < size_t(n_flag) * 6) {
        void *hp = nullptr;
        const size_t want = size_t(n_flag) * 6 * 2;
        HIP_TRY(hipHostMalloc(&hp, want * sizeof(float), hipHostMallocDefault));
        if (sc->flag_host) (void)hipHostFree(sc->flag_host);
        sc->flag_host = static_cast<float *>(hp);
        sc->flag_host_floats = want;
    }
    HIP_TRY(hipMemcpyAsync(sc->flag_host, sc->flag_rec, size_t(n_flag) * 6 * sizeof(float), hipMemcpyDeviceToHost, copy_stream));
    HIP_TRY(hipStreamSynchronize(copy_stream));
    const float *rec = sc->flag_host;
    tm.lap("download records");
    const int ntx = Pf.n_tiles_x;
    auto tile_of = [&](int x, int y, int *pix) {
        const int tx = (x - S.samp_x0) / 16, ty = (y - S.samp_y0) / 16;
        *pix = (y - S.samp_y0 - ty * 16) * 16 + (x - S.samp_x0 - tx * 16);
        return ty * ntx + tx;
    };
    fl.resize(n_flag);
    for (uint32_t i = 0; i < n_flag; ++i) {
        uint32_t u[6];
        std::memcpy(u, rec + 6 * size_t(i), sizeof(u));
        Flagged &f = fl[i];
        f.px = int(u[0]), f.py = int(u[1]), f.k = int(u[2] & 0x3fffffffu);
        f.pfx = rec[6 * size_t(i) + 3], f.pfy = rec[6 * size_t(i) + 4];
        f.pid = u[5];
        f.tile = tile_of(f.px, f.py, &f.pix);
        const bool zero_x = (u[2] >> 30) & 1u, zero_y = (u[2] >> 31) & 1u;
        const bool whole_x = f.pfx == float(f.px) || f.pfx == float(f.px + 1), whole_y = f.pfy == float(f.py) || f.pfy == float(f.py + 1);
        f.plain_k0 = f.k == 0 && (!whole_x || zero_x) && (!whole_y || zero_y);
    }
    auto gen_key = [&](int i) { return (uint64_t(uint32_t(fl[i].tile)) << 40) | (uint64_t(uint32_t(fl[i].pix)) << 32) | uint64_t(uint32_t(fl[i].k)); };
    // every (destination pixel, flagged sample of another pixel that lands in it)
    const int fw = S.crop_x1 - S.crop_x0;
    hits.reserve(size_t(n_flag) * 2);
    const float r = 0.5f;
    for (int i = 0; i < int(fl.size()); ++i) {
        const Flagged &f = fl[i];
        const int tx = f.tile % ntx, ty = f.tile / ntx;
        const int sx0 = S.samp_x0 + tx * 16, sy0 = S.samp_y0 + ty * 16;
        const int sx1 = std::min(sx0 + 16, S.samp_x1), sy1 = std::min(sy0 + 16, S.samp_y1);
        // Film::GetFilmTile bounds of the sample's tile, film.cpp:92-103
        const int fx0 = std::max(int(std::ceil(float(sx0) - 0.5f - r)), S.crop_x0), fx1 = std::min(int(std::floor(float(sx1) - 0.5f + r)) + 1, S.crop_x1);
        const int fy0 = std::max(int(std::ceil(float(sy0) - 0.5f - r)), S.crop_y0), fy1 = std::min(int(std::floor(float(sy1) - 0.5f + r)) + 1, S.crop_y1);
        const float dxf = f.pfx - 0.5f, dyf = f.pfy - 0.5f;
        const int ax0 = std::max(int(std::ceil(dxf - r)), fx0), ax1 = std::min(int(std::floor(dxf + r)) + 1, fx1);
        const int ay0 = std::max(int(std::ceil(dyf - r)), fy0), ay1 = std::min(int(std::floor(dyf + r)) + 1, fy1);
        for (int y = ay0; y < ay1; ++y)
            for (int x = ax0; x < ax1; ++x)
                if (x != f.px || y != f.py) hits.emplace_back(uint32_t(y - S.crop_y0) * uint32_t(fw) + uint32_t(x - S.crop_x0), i);
    }
    if (hits.empty()) return IILE_OK;
    std::sort(hits.begin(), hits.end(), [&](const std::pair<uint32_t, int> &a, const std::pair<uint32_t, int> &b) {
        return a.first != b.first ? a.first < b.first : gen_key(a.second) < gen_key(b.second);
    });
    tm.lap("destinations");
    // which radiances are needed: every flagged sample that lands somewhere else, and all own samples of a pixel that
    // receives from a pixel generated before it in its own tile
    const int n_k = Pf.kc;
    plan->list_of_flag.assign(fl.size(), -1);
    for (size_t a = 0; a < hits.size();) {
        size_t b = a;
        while (b < hits.size() && hits[b].first == hits[a].first) ++b;
        if (Pf.n_pass_tiles == Pf.n_owned_tiles) {
            // the pass is the whole frame: a pixel reached only by samples k_film_resolve places itself needs nothing
            bool only_plain = true;
            for (size_t h = a; h < b; ++h) only_plain = only_plain && fl[hits[h].second].plain_k0;
            if (only_plain) {
                a = b;
                continue;
            }
        }
        PatchDest d;
        d.film_index = hits[a].first;
        d.first = a, d.last = b;
        d.qx = S.crop_x0 + int(d.film_index % uint32_t(fw));
        d.qy = S.crop_y0 + int(d.film_index / uint32_t(fw));
        const bool in_bounds = d.qx >= S.samp_x0 && d.qx < S.samp_x1 && d.qy >= S.samp_y0 && d.qy < S.samp_y1;
        d.tile = in_bounds ? tile_of(d.qx, d.qy, &d.pix) : -1;
        d.own_in_pass = false;
        d.own_slot = 0;
        if (in_bounds && sc->slot_of(d.tile) >= 0) {
            const int slot = sc->slot_of(d.tile);
            d.own_in_pass = slot >= Pf.slot0 && slot < Pf.slot0 + Pf.n_pass_tiles;
            d.own_slot = uint32_t(slot) * 256u + uint32_t(d.pix);
        }
        d.need_own = false;
        d.own_first = 0;
        for (size_t h = a; h < b; ++h) {
            const int i = hits[h].second;
            if (plan->list_of_flag[i] < 0) {
                plan->list_of_flag[i] = int(plan->list_pid.size());
                plan->list_pid.push_back(fl[i].pid);
            }
            if (d.own_in_pass && fl[i].tile == d.tile && fl[i].pix < d.pix) d.need_own = true;
        }
        if (!(d.qx >= S.pb_x0 && d.qx < S.pb_x1 && d.qy >= S.pb_y0 && d.qy < S.pb_y1)) d.need_own = false;   // (no samples of its own: k_patch_dests)
        dests.push_back(d);
        a = b;
    }
    (void)n_k;  // (the own samples of need_own pixels are summed on the device: k_patch_own)
    if (dests.empty()) return IILE_OK;
    tm.lap("lists");
    if (tm.on) std::fprintf(stderr, "[patch] pass at slot %d: %u flagged samples, %zu pixels reached, %zu radiances to gather\n", Pf.slot0, n_flag, dests.size(), plan->list_pid.size());
    plan->active = true;
    return IILE_OK;
}

// After the pass's film accumulation: the exact sum every tile of this pass adds to every pixel reached by a flagged sample.
// Everything runs on the render's stream (no null-stream work, no device-wide synchronisation).
int patch_pass_finish(iile_scene *sc, const DScene &S, PatchPlan *plan, std::vector<PatchEntry> *entries, hipStream_t stream) {
    if (!plan->active) return IILE_OK;
    PatchTimer tm;
    const std::vector<Flagged> &fl = plan->fl;
    const std::vector<std::pair<uint32_t, int>> &hits = plan->hits;
    const int n_k = plan->k_end - plan->k_begin;
    std::vector<uint32_t> own_idx;
    // need_own pixels: their own-tile sum is taken on the device (own samples never leave HBM)
    std::vector<uint32_t> no_local, no_range, no_pid;
    for (const PatchDest &d : plan->dests) {
        if (d.own_in_pass && !d.need_own) own_idx.push_back(d.own_slot);
        if (d.need_own) {
            no_local.push_back(d.own_slot - uint32_t(plan->slot0) * 256u);
            no_range.push_back(uint32_t(no_pid.size()));
            uint32_t split = 0;
            bool split_set = false;
            for (size_t h = d.first; h < d.last; ++h) {
                const Flagged &f = fl[hits[h].second];
                if (f.tile != d.tile) continue;
                if (!split_set && f.pix > d.pix) {
                    split = uint32_t(no_pid.size());
                    split_set = true;
                }
                no_pid.push_back(f.pid);
            }
            no_range.push_back(split_set ? split : uint32_t(no_pid.size()));
            no_range.push_back(uint32_t(no_pid.size()));
        }
    }
    std::vector<float4> L, own, no_sum;
    L.resize(plan->list_pid.size());
    own.resize(own_idx.size());
    no_sum.resize(no_local.size());
    LaunchCfg cfg{sc->n_cus, stream, false};
    int rc;
    {
        rc = scratch_reserve(sc, scratch_bytes<uint32_t>(plan->list_pid.size()) + scratch_bytes<float4>(L.size()) +
                                     scratch_bytes<uint32_t>(own_idx.size()) + scratch_bytes<float4>(own.size()) +
                                     scratch_bytes<uint32_t>(no_local.size()) + scratch_bytes<uint32_t>(no_range.size()) +
                                     scratch_bytes<uint32_t>(no_pid.size()) + scratch_bytes<float4>(no_sum.size()), stream);
        if (rc) return rc;
        uint32_t *di = nullptr, *dj = nullptr, *d1 = nullptr, *d2 = nullptr, *d3 = nullptr;
        if ((rc = scratch_put(sc, plan->list_pid, stream, &di))) return rc;
        float4 *dv = scratch_take<float4>(sc, L.size()), *dw = nullptr, *dn = nullptr;
        launch_gather4(sc->pb.L, di, int(plan->list_pid.size()), dv, cfg);
        if (!own_idx.empty()) {
            if ((rc = scratch_put(sc, own_idx, stream, &dj))) return rc;
            dw = scratch_take<float4>(sc, own.size());
            launch_gather4(sc->fb.tile_rgbw, dj, int(own_idx.size()), dw, cfg);
        }
        if (!no_local.empty()) {
            if ((rc = scratch_put(sc, no_local, stream, &d1)) || (rc = scratch_put(sc, no_range, stream, &d2)) ||
                (rc = scratch_put(sc, no_pid, stream, &d3)))
                return rc;
            dn = scratch_take<float4>(sc, no_sum.size());
            launch_patch_own(S, sc->pb.L, int(no_local.size()), d1, d2, d3, n_k, dn, cfg);
        }
        HIP_TRY(hipGetLastError());
        if (!L.empty()) HIP_TRY(hipMemcpyAsync(L.data(), dv, L.size() * sizeof(float4), hipMemcpyDeviceToHost, stream));
        if (!own.empty()) HIP_TRY(hipMemcpyAsync(own.data(), dw, own.size() * sizeof(float4), hipMemcpyDeviceToHost, stream));
        if (!no_sum.empty()) HIP_TRY(hipMemcpyAsync(no_sum.data(), dn, no_sum.size() * sizeof(float4), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    tm.lap("gathers");
    for (float4 &v : L) {  // guard_radiance (kernels.hip)
        const float y = 0.212671f * v.x + 0.715160f * v.y + 0.072169f * v.z;
        if (std::isnan(v.x) || std::isnan(v.y) || std::isnan(v.z) || double(y) < -1e-5 || std::isinf(y)) v.x = v.y = v.z = 0.f;
        const float y2 = 0.212671f * v.x + 0.715160f * v.y + 0.072169f * v.z;
        if (y2 > S.max_sample_luminance) {
            const float sc2 = S.max_sample_luminance / y2;
            v.x *= sc2, v.y *= sc2, v.z *= sc2;
        }
    }
    size_t own_at = 0, no_at = 0;
    for (const PatchDest &d : plan->dests) {
        float4 own_sum = make_float4(0, 0, 0, 0);
        if (d.own_in_pass && !d.need_own) own_sum = own[own_at++];
        float4 own_tile_sum = make_float4(0, 0, 0, 0);  // need_own: the finished sum of its own tile (k_patch_own)
        if (d.need_own) own_tile_sum = no_sum[no_at++];
        // the hits of one destination are in generation order: runs of equal tile, ascending
        for (size_t a = d.first; a < d.last;) {
            const int t = fl[hits[a].second].tile;
            size_t b = a;
            while (b < d.last && fl[hits[b].second].tile == t) ++b;
            PatchEntry e;
            e.film_index = d.film_index;
            e.tile = t;
            e.nonplain = false;
            float rr = 0, gg = 0, bb = 0, ww = 0;
            auto add = [&](const float4 &v) {
                rr += v.x * 1.f * 1.f;
                gg += v.y * 1.f * 1.f;
                bb += v.z * 1.f * 1.f;
                ww += 1.f;
            };
            if (d.own_in_pass && t == d.tile && d.need_own) {
                rr = own_tile_sum.x, gg = own_tile_sum.y, bb = own_tile_sum.z, ww = own_tile_sum.w;
            } else {
                if (d.own_in_pass && t == d.tile) rr = own_sum.x, gg = own_sum.y, bb = own_sum.z, ww = own_sum.w;
                for (size_t h = a; h < b; ++h) add(L[size_t(plan->list_of_flag[hits[h].second])]);
            }
            for (size_t h = a; h < b; ++h) e.nonplain = e.nonplain || !fl[hits[h].second].plain_k0;
            e.r = rr, e.g = gg, e.b = bb, e.w = ww;
            entries->push_back(e);
            a = b;
        }
    }
    tm.lap("tile sums");
    return IILE_OK;
}

// After k_film_resolve: every pixel reached by a sample the resolve kernel does not place is rebuilt from its tiles'
// exact sums, added in tile index order (Film::MergeFilmTile, film.cpp:135-148).
int patch_merge(iile_scene *sc, const DScene &S, const PassDesc &Pf, std::vector<PatchEntry> *entries, float4 *film_dev, uint64_t *n_patched,
                hipStream_t stream) {
    *n_patched = 0;
    if (entries->empty()) return IILE_OK;
    PatchTimer tm;
    std::sort(entries->begin(), entries->end(), [](const PatchEntry &a, const PatchEntry &b) {
        return a.film_index != b.film_index ? a.film_index < b.film_index : a.tile < b.tile;
    });
    const int fw = S.crop_x1 - S.crop_x0, ntx = Pf.n_tiles_x;
    struct Group {
        size_t first, last;
        int own_tile;        // the pixel's own (owned) tile if no entry covers it, else -1
        uint32_t own_slot;
    };
    std::vector<Group> groups;
    std::vector<uint32_t> own_idx;
    for (size_t a = 0; a < entries->size();) {
        size_t b = a;
        bool nonplain = false;
        while (b < entries->size() && (*entries)[b].film_index == (*entries)[a].film_index) {
            if ((*entries)[b].nonplain) nonplain = true;
            ++b;
        }
        if (nonplain) {
            Group g{a, b, -1, 0u};
            const uint32_t fi = (*entries)[a].film_index;
            const int qx = S.crop_x0 + int(fi % uint32_t(fw)), qy = S.crop_y0 + int(fi / uint32_t(fw));
            if (qx >= S.samp_x0 && qx < S.samp_x1 && qy >= S.samp_y0 && qy < S.samp_y1) {
                const int tx = (qx - S.samp_x0) / 16, ty = (qy - S.samp_y0) / 16, t = ty * ntx + tx;
                bool covered = false;
                for (size_t h = a; h < b; ++h) covered = covered || (*entries)[h].tile == t;
                if (!covered && sc->slot_of(t) >= 0) {
                    g.own_tile = t;
                    g.own_slot = uint32_t(sc->slot_of(t)) * 256u + uint32_t((qy - S.samp_y0 - ty * 16) * 16 + (qx - S.samp_x0 - tx * 16));
                    own_idx.push_back(g.own_slot);
                }
            }
            groups.push_back(g);
        }
        a = b;
    }
    if (groups.empty()) return IILE_OK;
    LaunchCfg cfg{sc->n_cus, stream, false};
    int rc;
    std::vector<float4> own(own_idx.size());
    if (!own_idx.empty()) {
        if ((rc = scratch_reserve(sc, scratch_bytes<uint32_t>(own_idx.size()) + scratch_bytes<float4>(own.size()), stream))) return rc;
        uint32_t *di = nullptr;
        if ((rc = scratch_put(sc, own_idx, stream, &di))) return rc;
        float4 *dv = scratch_take<float4>(sc, own.size());
        launch_gather4(sc->fb.tile_rgbw, di, int(own_idx.size()), dv, cfg);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(own.data(), dv, own.size() * sizeof(float4), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    std::vector<uint32_t> out_idx(groups.size());
    std::vector<float4> out_val(groups.size());
    size_t own_at = 0;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        const Group &g = groups[gi];
        float4 o = make_float4(0, 0, 0, 0);
        auto add_tile = [&](float rr, float gg, float bb, float ww) {
            o.x += 0.412453f * rr + 0.357580f * gg + 0.180423f * bb;  // RGBToXYZ, spectrum.h:62-66
            o.y += 0.212671f * rr + 0.715160f * gg + 0.072169f * bb;
            o.z += 0.019334f * rr + 0.119193f * gg + 0.950227f * bb;
            o.w += ww;
        };
        bool own_added = g.own_tile < 0;
        float4 own_sum = make_float4(0, 0, 0, 0);
        if (g.own_tile >= 0) own_sum = own[own_at++];
        for (size_t h = g.first; h < g.last; ++h) {
            const PatchEntry &e = (*entries)[h];
            if (!own_added && g.own_tile < e.tile) {
                add_tile(own_sum.x, own_sum.y, own_sum.z, own_sum.w);
                own_added = true;
            }
            add_tile(e.r, e.g, e.b, e.w);
        }
        if (!own_added) add_tile(own_sum.x, own_sum.y, own_sum.z, own_sum.w);
        out_idx[gi] = (*entries)[g.first].film_index;
        out_val[gi] = o;
    }
    {
        if ((rc = scratch_reserve(sc, scratch_bytes<uint32_t>(out_idx.size()) + scratch_bytes<float4>(out_val.size()), stream))) return rc;
        uint32_t *di = nullptr;
        float4 *dv = nullptr;
        if ((rc = scratch_put(sc, out_idx, stream, &di)) || (rc = scratch_put(sc, out_val, stream, &dv))) return rc;
        launch_scatter4(film_dev, di, int(out_idx.size()), dv, cfg);
        HIP_TRY(hipGetLastError());
        // out_idx / out_val are pageable host vectors that die with this scope: the copies must have been taken
        HIP_TRY(hipStreamSynchronize(stream));
    }
    tm.lap("merge + scatter");
    *n_patched = out_idx.size();
    return IILE_OK;
}
}  // namespace

int iile_render(iile_scene *sc, const iile_render_params *prm, float *film_xyzw, iile_stats *stats) {
    if (!sc || !prm || !film_xyzw) return fail(IILE_ERR_ARG, "iile_render: null argument");
    int rc = ensure_device();
    if (rc) return rc;
    rc = check_pending_overflow(sc);   // of an earlier render that returned before its stream had drained
    if (rc) return rc;
    const DScene &S = sc->ds;
    int k_begin = prm->k_begin, k_end = prm->k_end;
    if (k_end <= 0) {
        k_begin = 0;
        k_end = sc->spp;
    }
    if (k_begin < 0 || k_begin >= k_end) return fail(IILE_ERR_ARG, "iile_render: empty sample range");
    int rank = prm->tile_rank, nranks = prm->tile_nranks;
    if (nranks <= 0) {
        rank = 0;
        nranks = 1;
    }
    if (rank < 0 || rank >= nranks) return fail(IILE_ERR_ARG, "iile_render: tile_rank out of range");
    hipStream_t stream = static_cast<hipStream_t>(prm->stream);
    LaunchCfg cfg{sc->n_cus, stream, prm->collect_stats != 0};
    const bool timed = prm->time_kernels != 0;

    PassDesc P;
    std::memset(&P, 0, sizeof(P));
    P.n_tiles_x = (S.samp_x1 - S.samp_x0 + 15) / 16;
    P.n_tiles_y = (S.samp_y1 - S.samp_y0 + 15) / 16;
    P.tile_rank = rank;
    P.tile_nranks = nranks;
    rc = ensure_tile_map(sc, &P, stream);
    if (rc) return rc;
    const uint64_t pix_slots = uint64_t(P.n_owned_tiles) * 256;
    const int n_samples = k_end - k_begin;
    // A pass renders all samples of a range of owned tiles; the range is bounded by the workspace budget (~410 B per
    // path incl. queue padding). `spp_per_pass` (tests) asks for passes of about that many samples per pixel's worth
    // of paths: n_owned_tiles * spp_per_pass / n_samples tiles each.
    uint64_t max_paths;
    {
        double budget_mb = 65536;  // 64 GiB of the 288 GB: one pass covers 1080p x 64 spp
        if (const char *e = std::getenv("IILE_WORKSPACE_MB")) budget_mb = std::max(64.0, atof(e));
        max_paths = std::min<uint64_t>(uint64_t(budget_mb * 1048576.0 / 410.0), 200000000ull);  // queue slots must fit kSlotBits
        if (prm->spp_per_pass > 0) max_paths = std::min<uint64_t>(max_paths, std::max<uint64_t>(1, pix_slots * uint64_t(prm->spp_per_pass)));
    }
    const uint64_t paths_per_tile = uint64_t(256) * uint64_t(n_samples);
    if (paths_per_tile > 200000000ull) return fail(IILE_ERR_UNSUPPORTED, "more than 781 250 samples per pixel in one render: split the sample range");
    const int tiles_per_pass = int(std::max<uint64_t>(1, std::min<uint64_t>(uint64_t(std::max(P.n_owned_tiles, 1)), max_paths / paths_per_tile)));
    const uint32_t fw = uint32_t(S.crop_x1 - S.crop_x0), fh = uint32_t(S.crop_y1 - S.crop_y0);

    if (pix_slots) {
        rc = ensure_workspace(sc, uint32_t(uint64_t(tiles_per_pass) * paths_per_tile));
        if (rc) return rc;
    }
    rc = ensure_film(sc, uint32_t(P.n_owned_tiles), fw * fh, S.filter_wide ? pix_slots * uint64_t(n_samples) : 0);
    if (rc) return rc;
    sc->pb.nray_out = nullptr;
    sc->events_used = 0;
    iile_stats st;
    std::memset(&st, 0, sizeof(st));
    // whole-number film positions are listed for the one-pixel box film (the sample store of wider filters handles them).
    // Their exact finish runs on the device, on this stream, without a host wait (kernels.hip "exact film finish"); the
    // host-side version of rounds 1-3 stays behind IILE_DEBUG_HOST_FILM_FINISH as an A/B witness (bench.py refuses to run with it).
    PatchPlan plan;
    std::vector<PatchEntry> entries;
    const bool host_finish = std::getenv("IILE_DEBUG_HOST_FILM_FINISH") != nullptr;
    if (!S.filter_wide && !host_finish) {
        rc = ensure_patch(sc, uint64_t(tiles_per_pass) * paths_per_tile, uint64_t(P.n_owned_tiles) * paths_per_tile);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(sc->patch.counters, 0, 16, stream));
    }
    const size_t patch_table_bytes = (size_t(sc->patch.table_mask) + 1) * 8;  // keys, then heads: one memset of 0xff
    sc->pb.flag_count = S.filter_wide ? nullptr : sc->flag_count;
    sc->pb.flag_rec = sc->flag_rec;
    struct FlagGuard {  // error returns below must not leave the list armed for the kernel-level entry points
        PassBuffers *pb;
        ~FlagGuard() { pb->flag_count = nullptr; }
    } flag_guard{&sc->pb};

    HIP_TRY(hipEventRecord(sc->ev_begin, stream));
    if (pix_slots) HIP_TRY(hipMemsetAsync(sc->pb.counters, 0, sizeof(DCounters), stream));
    // (the two planes are sized for the largest share this scene has rendered: k0_rgbv starts film_tiles * 256 records behind tile_rgbw,
    //  not pix_slots — each is cleared where it lies; a pixel outside "pixelbounds" keeps these zeros)
    if (pix_slots) HIP_TRY(hipMemsetAsync(sc->fb.tile_rgbw, 0, size_t(pix_slots) * sizeof(float4), stream));
    if (pix_slots) HIP_TRY(hipMemsetAsync(sc->fb.k0_rgbv, 0, size_t(pix_slots) * sizeof(float4), stream));
    P.k0 = k_begin;
    P.kc = n_samples;
    for (int slot0 = 0; slot0 < P.n_owned_tiles; slot0 += tiles_per_pass) {
        P.slot0 = slot0;
        P.n_pass_tiles = std::min(tiles_per_pass, P.n_owned_tiles - slot0);
        P.n_paths = uint32_t(uint64_t(P.n_pass_tiles) * paths_per_tile);
        if (sc->pb.flag_count) HIP_TRY(hipMemsetAsync(sc->flag_count, 0, sizeof(uint32_t), stream));
        rc = run_pass(sc, S, sc->max_depth, P, cfg, timed, prm->time_kernels == 2);
        if (rc) return rc;
        EventPair *ep = nullptr;
        if (timed) {
            rc = get_events(sc, 4, &ep);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ep->a, stream));
        }
        if (S.filter_wide)
            launch_film_store(S, P, sc->pb, sc->fb, k_begin, n_samples, cfg);
        else
            launch_film_accumulate(S, P, sc->pb, sc->fb, cfg);
        if (timed) HIP_TRY(hipEventRecord(ep->b, stream));
        if (sc->pb.flag_count && !host_finish) {
            // this pass's flagged samples -> exact FilmTile sums (entries) for the pixels they reach
            HIP_TRY(hipMemsetAsync(sc->patch.counters, 0, sizeof(uint32_t), stream));  // hits are per pass; entries add up
            HIP_TRY(hipMemsetAsync(sc->patch.keys, 0xff, patch_table_bytes, stream));
            launch_patch_pass(S, P, sc->pb, sc->fb, sc->patch, cfg);
        } else if (sc->pb.flag_count) {
            // the pass's list of whole-number film positions is final after its first k_extend: fetch and sort it on the
            // host while the GPU works through the rest of the pass, then take the exact tile sums once it is done
            HIP_TRY(hipStreamWaitEvent(sc->aux_stream, sc->ev_flags, 0));
            rc = patch_prepare(sc, S, P, sc->aux_stream, &plan);
            if (rc) return rc;
            // (its index lists are built on the host while the pass is still running; its gathers queue up behind the pass)
            rc = patch_pass_finish(sc, S, &plan, &entries, stream);
            if (rc) return rc;
        }
        st.n_passes++;
        st.n_paths += P.n_paths;
    }
    FilmBuffers F = sc->fb;
    if (prm->film_on_device) F.film_xyzw = reinterpret_cast<float4 *>(film_xyzw);
    if (S.filter_wide) {
        EventPair *ep = nullptr;
        if (timed) {  // counted with the film kernels (ms_film)
            rc = get_events(sc, 4, &ep);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ep->a, stream));
        }
        launch_film_gather(S, P, F, n_samples, cfg);
        if (timed) HIP_TRY(hipEventRecord(ep->b, stream));
    } else {
        launch_film_resolve(S, P, F, cfg);
        HIP_TRY(hipGetLastError());
        if (!host_finish && sc->pb.flag_count && pix_slots) {
            HIP_TRY(hipMemsetAsync(sc->patch.keys, 0xff, patch_table_bytes, stream));
            launch_patch_merge(S, P, F, sc->patch, cfg);
            HIP_TRY(hipGetLastError());
        }
        if (!entries.empty()) {
            HIP_TRY(hipStreamSynchronize(stream));
            uint64_t n_patched = 0;
            rc = patch_merge(sc, S, P, &entries, F.film_xyzw, &n_patched, stream);
            if (rc) return rc;
        }
    }
    sc->pb.flag_count = nullptr;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sc->ev_end, stream));
    if (!prm->film_on_device) {
        HIP_TRY(hipMemcpyAsync(film_xyzw, F.film_xyzw, size_t(fw) * fh * sizeof(float4), hipMemcpyDeviceToHost, stream));
    }
    // The film is complete once the stream drains. Statistics need the drain; a device-resident film without stats
    // stays asynchronous on `stream` past the last pass (the exact film finish waits for each pass on that stream, and
    // only on it: nothing here touches the null stream or synchronises the device).
    if (!(stats || !prm->film_on_device) && pix_slots && !S.filter_wide && !host_finish) {
        sc->overflow_unchecked = true;   // nobody waits here: iile_render_status / the next iile_render reads the error word
        sc->overflow_stream = stream;
    }
    if (stats || !prm->film_on_device) {
        HIP_TRY(hipStreamSynchronize(stream));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, sc->ev_begin, sc->ev_end));
        st.ms_total = ms;
        if (timed) {
            rc = collect_times(sc, &st);
            if (rc) return rc;
        }
        if (pix_slots && !S.filter_wide && !host_finish) {  // did the exact film finish run out of room? (checked wherever the host waits anyway)
            uint32_t pc[4] = {0, 0, 0, 0};
            HIP_TRY(hipMemcpyAsync(pc, sc->patch.counters, sizeof(pc), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (pc[2] != 0)
                return fail(IILE_ERR_UNSUPPORTED, "the exact film finish ran out of room (camera samples with whole-number film positions: more than "
                                                  "2^20 in one pass, or more pixel hits / tile sums than the frame was sized for)");
        }
        if (pix_slots) {
            DCounters c;
            HIP_TRY(hipMemcpyAsync(&c, sc->pb.counters, sizeof(c), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (prm->collect_stats)
                copy_counters(c, &st);
            else
                st.mis_rays_traced = c.mis_traced, st.ext_rays_traced = c.ext_traced;
#if defined(IILE_SHADE_STAMPS) || defined(IILE_TRAV_STAMPS) || defined(IILE_SHADOW_STAMPS) || defined(IILE_TRAV_ITERSTATS)
            // diagnostic builds (tools/shade_stamps.py, tools/trav_stamps.py): per-section wave cycles ride out in the path-length histogram
            if (!prm->collect_stats)
                for (int i = 0; i < 8; ++i) st.path_length[i] = c.path_length[i];
#endif
        }
        st.workspace_bytes = sc->ws_bytes;
        if (stats) *stats = st;
    }
    return IILE_OK;
}

// ---- IISPT direct pass (kernels_direct.hip) -------------------------------------
int iile_render_direct(iile_scene *sc, const iile_direct_params *prm, double *film_rgbw) {
    if (!sc || !prm || !film_rgbw || prm->n_passes < 0 || prm->first_pass < 0) return fail(IILE_ERR_ARG, "iile_render_direct: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    DScene S = sc->ds;
    // reflected rays carry differentials in textured scenes (SpecularReflect, directprogressiveintegrator.cpp:165-184), built from the
    // hit's dpdu / dpdv and shading.dndu / dndv (triangles: triangle_interaction; spheres: sphere_interaction<true>)
    const bool reflect_diffs = S.textured_materials && S.has_specular;
    if (S.filter_wide) return fail(IILE_ERR_UNSUPPORTED, "iile_render_direct: the direct pass is defined for the one-pixel box film");
    // ("pixelbounds" belongs to the path integrator: the IISPT runner hands DirectProgressiveIntegrator the film's bounds, iisptrenderrunner.cpp:608-613)
    S.pb_x0 = S.samp_x0, S.pb_y0 = S.samp_y0, S.pb_x1 = S.samp_x1, S.pb_y1 = S.samp_y1, S.pb_set = 0;
    // Glass: DirectProgressiveIntegrator::Li builds its BSDF with allowMultipleLobes = false (interaction.h:130-133), GlassMaterial
    // then adds a SpecularReflection and a SpecularTransmission lobe (glass.cpp:62-90) and both recursions fire — Li is a tree,
    // walked depth first by one thread per pixel (k_direct_tree) instead of the wavefront below.
    const bool tree = S.has_glass != 0;
    S.diff_scale = 0.25f;  // ScaleDifferentials(1 / sqrt(16)): the RandomSampler's samples per pixel
    hipStream_t stream = static_cast<hipStream_t>(prm->stream);
    LaunchCfg cfg{sc->n_cus, stream, false};
    PassDesc P;
    std::memset(&P, 0, sizeof(P));
    P.n_tiles_x = (S.samp_x1 - S.samp_x0 + 15) / 16;
    P.n_tiles_y = (S.samp_y1 - S.samp_y0 + 15) / 16;
    P.tile_rank = 0;
    P.tile_nranks = 1;
    rc = ensure_tile_map(sc, &P, stream);
    if (rc) return rc;
    P.slot0 = 0;
    P.n_pass_tiles = P.n_owned_tiles;
    P.k0 = 0;
    // One launch renders `batch` passes of the frame at once: path id = (pixel slot, pass of the batch) — the "sample of the pixel"
    // coordinate of the path tracer's enumeration (path_pixel). A pass of one sample per pixel leaves most of a persistent
    // traversal grid without a second ray (2 M paths over 393 k lanes); four at a time run at the path integrator's rates. Each
    // pass keeps its own seed and its own records; the fold adds a pixel's passes in pass order. Glass (one thread per pixel
    // walking a tree) stays at one pass per launch.
    int total_samples_pre = 0;
    for (int l = 0; l < std::max(S.n_lights, 0) && l < 8; ++l) total_samples_pre += std::max(1, sc->light_samples[l]);
    const uint64_t pixels64 = uint64_t(P.n_owned_tiles) * 256;
    int batch = (S.has_glass != 0) ? 1 : std::max(1, std::min(prm->n_passes, 4));
    while (batch > 1 && pixels64 * uint64_t(batch) * uint64_t(std::max(total_samples_pre, 1)) > 100000000ull) --batch;   // NEE records per level (8 passes at a time measured no faster than 4)
    P.kc = batch;
    const uint64_t n_paths64 = pixels64 * uint64_t(batch);
    if (n_paths64 > 200000000ull) return fail(IILE_ERR_UNSUPPORTED, "iile_render_direct: frame too large for one pass");
    P.n_paths = uint32_t(n_paths64);
    const uint32_t fw = uint32_t(S.crop_x1 - S.crop_x0), fh = uint32_t(S.crop_y1 - S.crop_y0);
    const size_t film_bytes = size_t(fw) * fh * 4 * sizeof(double);
    if (P.n_paths == 0 || fw == 0 || fh == 0) return IILE_OK;
    // k_direct_shade appends one NEE record (and at most one MIS ray) per LIGHT and hit (UniformSampleAllLights), where the path
    // integrator's k_shade appends one per hit: the record planes are sized for paths x lights (a workspace sized for the
    // paths alone overflowed from 4 lights on at 1080p; found by the round-3 advisor)
    // UniformSampleAllLights takes Light::nSamples samples of every light (directprogressiveintegrator.cpp:9-18, integrator.cpp:54-83)
    const int n_lights = std::max(S.n_lights, 0), n_arrays = 5 * n_lights * 2;
    int total_samples = 0;
    for (int l = 0; l < n_lights && l < 8; ++l) {
        P.direct_nsamples[l] = std::max(1, sc->light_samples[l]);
        total_samples += P.direct_nsamples[l];
    }
    P.direct_total_samples = total_samples;
    if (total_samples > 64)
        return fail(IILE_ERR_UNSUPPORTED, "iile_render_direct: " + std::to_string(total_samples) + " light samples per vertex (the lights' nsamples summed; at most 64)");
    const uint64_t n_records64 = n_paths64 * uint64_t(std::max(total_samples, 1));
    if (n_records64 > 400000000ull)
        return fail(IILE_ERR_UNSUPPORTED, "iile_render_direct: pixels x light samples = " + std::to_string(n_records64) + " NEE records per level exceed one pass");
    rc = ensure_workspace(sc, tree ? 1024u : uint32_t(n_records64));  // (the per-pixel tree walk queues nothing)
    if (rc) return rc;
    sc->pb.nray_out = nullptr;
    sc->pb.flag_count = nullptr;
    // E, F (5 levels) and D (5 levels x light samples) of every path, the PCG jump table, the film
    const size_t np = P.n_paths, vec = sizeof(float4);
    const int levels = S.has_specular ? 5 : 1;  // Li recurses through specular lobes only (the pass loop below stops likewise)
    P.direct_levels = levels;
    const size_t d_bytes = tree ? vec : std::max<size_t>(size_t(levels) * size_t(total_samples) * np * vec, vec), ef_bytes = tree ? vec : size_t(levels) * np * vec;
    const size_t jump_bytes = (size_t(n_arrays) + 1) * 2 * sizeof(unsigned long long);
    const size_t rd_bytes = (reflect_diffs && !tree) ? size_t(4) * np * vec : 0;
    DevBuf<char> block;
    {
        void *p = nullptr;
        if (hipMalloc(&p, d_bytes + 2 * ef_bytes + rd_bytes + jump_bytes + (prm->film_on_device ? 0 : film_bytes) + 1024) != hipSuccess)
            return fail(IILE_ERR_HIP, "out of device memory for the direct pass (" + std::to_string((d_bytes + 2 * ef_bytes) >> 20) + " MiB of per-vertex records)");
        block.p = static_cast<char *>(p);
    }
    char *at = block.p;
    float4 *D = reinterpret_cast<float4 *>(at);
    at += d_bytes;
    float4 *E = reinterpret_cast<float4 *>(at);
    at += ef_bytes;
    float4 *F = reinterpret_cast<float4 *>(at);
    at += ef_bytes;
    float4 *RD = rd_bytes ? reinterpret_cast<float4 *>(at) : nullptr;
    at += rd_bytes;
    unsigned long long *jump_dev = reinterpret_cast<unsigned long long *>(at);
    at += (jump_bytes + 255) & ~size_t(255);
    double *film_dev = prm->film_on_device ? film_rgbw : reinterpret_cast<double *>(at);
    {   // the stream at every array's first entry: array i (of light (i / 2) % n_lights) holds 16 x nSamples entries of two
        // floats (RandomSampler::StartPixel, random.cpp:62-72; Request2DArray(nLightSamples[j]) twice per level and light)
        std::vector<unsigned long long> jump(size_t(n_arrays + 1) * 2);
        const unsigned long long a = 0x5851f42d4c957f2dULL;
        unsigned long long A = 1, G = 0;
        for (int i = 0; i <= n_arrays; ++i) {
            jump[2 * size_t(i)] = A;
            jump[2 * size_t(i) + 1] = G;
            const int draws = i < n_arrays ? 32 * P.direct_nsamples[(i / 2) % std::max(n_lights, 1)] : 0;
            for (int s = 0; s < draws; ++s) {  // one more draw: state' = a state + inc
                G = G * a + 1;
                A = A * a;
            }
        }
        HIP_TRY(hipMemcpyAsync(jump_dev, jump.data(), jump_bytes, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));  // (the vector dies with this scope)
    }
    if (!prm->accumulate) HIP_TRY(hipMemsetAsync(film_dev, 0, film_bytes, stream));
    else if (!prm->film_on_device) HIP_TRY(hipMemcpyAsync(film_dev, film_rgbw, film_bytes, hipMemcpyHostToDevice, stream));
    PassBuffers B = sc->pb;
    B.L = D;
    B.dir_E = E;
    B.dir_F = F;
    B.dir_RD = RD;
    B.dir_paths = P.n_paths;
    B.spill = sc->spill;
    P.direct_arrays = n_arrays;
    P.direct_jump = jump_dev;
    for (int i = 0; i < prm->n_passes; i += batch) {
        const int nb = std::min(batch, prm->n_passes - i);   // (the last launch may hold fewer passes: same buffers, fewer paths)
        P.kc = nb;
        P.n_paths = uint32_t(pixels64 * uint64_t(nb));
        B.dir_paths = P.n_paths;
        P.direct_seed = uint32_t(6284 + 17 * (prm->first_pass + i));
        if (tree) {
            launch_direct_tree(S, P, B, film_dev, cfg);
            HIP_TRY(hipGetLastError());
            continue;
        }
        HIP_TRY(hipMemsetAsync(B.counts, 0, kCntWords * sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(D, 0, d_bytes + 2 * ef_bytes, stream));
        launch_direct_generate(S, P, B, cfg);
        for (int d = 0; d < 5; ++d) {
            launch_extend(S, P, B, d, B.queue_cap, cfg);
            if (S.has_infinite) launch_direct_miss(S, B, d, B.queue_cap, cfg);
            launch_direct_shade(S, P, B, d, B.queue_cap, cfg);
            launch_mis(S, B, d, B.queue_cap, cfg);
            launch_mis_lit(S, B, d, B.queue_cap, cfg);
            launch_shadow(S, B, d, B.queue_cap, cfg);
            if (!S.has_specular) break;  // no mirror lobe anywhere: Li never recurses
        }
        launch_direct_fold(S, P, B, film_dev, cfg);
        HIP_TRY(hipGetLastError());
    }
    if (!prm->film_on_device) HIP_TRY(hipMemcpyAsync(film_rgbw, film_dev, film_bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));  // the per-vertex records are freed on return
    return IILE_OK;
}

// ---- IISPT probe pass ---------------------------------------------------------
namespace {
// Inverse(Matrix4x4), transform.cpp:82-141 (Gauss-Jordan, full pivoting; the pivot reciprocal is a double divide)
bool invert4(const float in[16], float out[16]) {
    int indxc[4], indxr[4];
    int ipiv[4] = {0, 0, 0, 0};
    float a[4][4];
    std::memcpy(a, in, sizeof(a));
    for (int i = 0; i < 4; i++) {
        int irow = 0, icol = 0;
        float big = 0.f;
        for (int j = 0; j < 4; j++) {
            if (ipiv[j] == 1) continue;
            for (int k = 0; k < 4; k++) {
                if (ipiv[k] == 0) {
                    if (std::abs(a[j][k]) >= big) {
                        big = std::abs(a[j][k]);
                        irow = j;
                        icol = k;
                    }
                } else if (ipiv[k] > 1)
                    return false;
            }
        }
        ++ipiv[icol];
        if (irow != icol)
            for (int k = 0; k < 4; ++k) std::swap(a[irow][k], a[icol][k]);
        indxr[i] = irow;
        indxc[i] = icol;
        if (a[icol][icol] == 0.f) return false;
        const float pivinv = float(1. / double(a[icol][icol]));
        a[icol][icol] = 1.f;
        for (int j = 0; j < 4; j++) a[icol][j] *= pivinv;
        for (int j = 0; j < 4; j++) {
            if (j == icol) continue;
            const float save = a[j][icol];
            a[j][icol] = 0;
            for (int k = 0; k < 4; k++) a[j][k] -= a[icol][k] * save;
        }
    }
    for (int j = 3; j >= 0; j--)
        if (indxr[j] != indxc[j])
            for (int k = 0; k < 4; k++) std::swap(a[k][indxr[j]], a[k][indxc[j]]);
    std::memcpy(out, a, sizeof(a));
    return true;
}
struct H3 {
    float x, y, z;
};
H3 h_normalize(H3 v) {  // Vector3::operator/ multiplies by the float reciprocal (geometry.h:242-246)
    const float inv = 1.f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return H3{v.x * inv, v.y * inv, v.z * inv};
}
H3 h_cross(H3 a, H3 b) {  // geometry.h:957-963: in double
    const double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return H3{float((ay * bz) - (az * by)), float((az * bx) - (ax * bz)), float((ax * by) - (ay * bx))};
}
// CreateHemisphericCamera (hemispheric.cpp:109-160) over LookAt (transform.cpp:203-236)
bool make_probe_camera(const float *pos, const float *dir, DProbeCam *cam) {
    const H3 up = (dir[0] == 0.0 && dir[1] == 0.0) ? H3{0.f, 1.f, 0.f} : H3{0.f, 0.f, 1.f};
    const H3 look = H3{pos[0] + dir[0], pos[1] + dir[1], pos[2] + dir[2]};
    const H3 d = h_normalize(H3{look.x - pos[0], look.y - pos[1], look.z - pos[2]});
    const H3 c = h_cross(h_normalize(up), d);
    if (std::sqrt(c.x * c.x + c.y * c.y + c.z * c.z) == 0) return false;
    const H3 right = h_normalize(c);
    const H3 new_up = h_cross(d, right);
    const float m[16] = {right.x, new_up.x, d.x, pos[0], right.y, new_up.y, d.y, pos[1], right.z, new_up.z, d.z, pos[2], 0.f, 0.f, 0.f, 1.f};
    std::memcpy(cam->c2w.m, m, sizeof(m));
    float inv[16], minv[16];
    if (!invert4(m, inv) || !invert4(inv, minv)) return false;
    for (int r = 0; r < 3; ++r)
        for (int cidx = 0; cidx < 3; ++cidx) cam->nrm[3 * r + cidx] = minv[4 * cidx + r];  // transpose of mInv
    return true;
}
}  // namespace

int iile_render_probes(iile_scene *sc, int32_t n_probes, const float *pos3, const float *dir3, float *intensity_rgb, float *normals_xyz,
                       float *distance, int32_t outputs_on_device, iile_stats *stats, void *stream_arg) {
    if (!sc || n_probes < 0 || !pos3 || !dir3 || !intensity_rgb || !normals_xyz || !distance)
        return fail(IILE_ERR_ARG, "iile_render_probes: null argument");
    int rc = ensure_device();
    if (rc) return rc;
    const iile_probe_setup &pr = sc->probe;
    if (pr.hemi_size <= 0 || !sc->probe_pixel_offsets) return fail(IILE_ERR_ARG, "iile_render_probes: the scene has no probe setup");
    if (pr.max_depth > 14) return fail(IILE_ERR_UNSUPPORTED, "probe maxdepth > 14");
    iile_stats st;
    std::memset(&st, 0, sizeof(st));
    if (n_probes == 0) {
        if (stats) *stats = st;
        return IILE_OK;
    }
    // the probe's film, sampler and depth in place of the frame's
    DScene S = sc->ds;
    const iile_film_desc &f = pr.film;
    S.probe_mode = 1;
    S.xres = f.xres, S.yres = f.yres;
    S.crop_x0 = f.crop_x0, S.crop_y0 = f.crop_y0, S.crop_x1 = f.crop_x1, S.crop_y1 = f.crop_y1;
    S.samp_x0 = f.samp_x0, S.samp_y0 = f.samp_y0, S.samp_x1 = f.samp_x1, S.samp_y1 = f.samp_y1;
    S.filter_rx = f.filter_rx, S.filter_ry = f.filter_ry;
    S.max_sample_luminance = f.max_sample_luminance;
    S.filter_wide = 1;
    S.filter_table = sc->probe_filter_table;
    S.pixel_offsets = sc->probe_pixel_offsets;
    S.base_scale0 = pr.base_scales[0], S.base_scale1 = pr.base_scales[1];
    S.base_exp0 = pr.base_exponents[0], S.base_exp1 = pr.base_exponents[1];
    S.sample_stride = pr.sample_stride;
    S.mult_inv0 = pr.mult_inverse[0], S.mult_inv1 = pr.mult_inverse[1];
    S.max_depth = pr.max_depth;
    S.sample_center = 0;  // the probes' own sampler: HaltonSampler(1, sampleBounds)
    S.sobol = 0;
    S.lens_radius = 0;
    S.diff_scale = 1.f;  // ScaleDifferentials(1 / sqrt(1 sample per pixel))
    const int need_dims = 5 + 8 * (pr.max_depth + 1) + 2;
    if (S.n_hdims < need_dims) return fail(IILE_ERR_ARG, "Halton table covers too few dimensions for the probe depth");

    PassDesc P;
    std::memset(&P, 0, sizeof(P));
    P.n_tiles_x = (S.samp_x1 - S.samp_x0 + 15) / 16;
    P.n_tiles_y = (S.samp_y1 - S.samp_y0 + 15) / 16;
    P.probe_mode = 1;
    // path slots cover the film's pixel bounds only (no samples are taken elsewhere): 16 x 16 storage tiles over them
    P.probe_stx = (f.crop_x1 - f.crop_x0 + 15) / 16;
    P.probe_tiles = P.probe_stx * ((f.crop_y1 - f.crop_y0 + 15) / 16);
    P.tile_rank = 0;
    P.tile_nranks = 1;
    P.k0 = 0;
    P.kc = 1;
    const uint32_t per_pixels = uint32_t(f.crop_x1 - f.crop_x0) * uint32_t(f.crop_y1 - f.crop_y0);
    const uint64_t slots_per_probe = uint64_t(P.probe_tiles) * 256;
    // probes per pass: bounded by the workspace budget like iile_render's passes
    double budget_mb = 65536;
    if (const char *e = std::getenv("IILE_WORKSPACE_MB")) budget_mb = std::max(64.0, atof(e));
    uint64_t max_paths = std::min<uint64_t>(uint64_t(budget_mb * 1048576.0 / 430.0), 200000000ull);
    const int batch = int(std::max<uint64_t>(1, std::min<uint64_t>(uint64_t(n_probes), max_paths / slots_per_probe)));

    std::vector<DProbeCam> cams;
    cams.resize(size_t(n_probes));
    for (int i = 0; i < n_probes; ++i)
        if (!make_probe_camera(pos3 + 3 * size_t(i), dir3 + 3 * size_t(i), &cams[size_t(i)]))
            return fail(IILE_ERR_ARG, "iile_render_probes: degenerate probe direction (probe " + std::to_string(i) + ")");

    hipStream_t stream = static_cast<hipStream_t>(stream_arg);
    LaunchCfg cfg{sc->n_cus, stream, false};
    const uint64_t batch_paths = uint64_t(batch) * slots_per_probe;
    rc = ensure_workspace(sc, uint32_t(batch_paths));
    if (rc) return rc;
    rc = ensure_film(sc, uint32_t(uint64_t(batch) * P.probe_tiles), uint32_t(uint64_t(batch) * per_pixels), batch_paths);
    if (rc) return rc;
    // cameras + aux + device-side outputs of one batch
    const size_t cam_bytes = (size_t(batch) * sizeof(DProbeCam) + 255) & ~size_t(255);
    const size_t aux_bytes = size_t(batch_paths) * sizeof(float4);
    const size_t out_floats = size_t(batch) * per_pixels * 7;
    const size_t need = cam_bytes + aux_bytes + out_floats * sizeof(float) + 1024;
    if (need > sc->probe_block_bytes) {
        if (sc->probe_block) HIP_TRY(hipFree(sc->probe_block));
        sc->probe_block = nullptr;
        sc->probe_block_bytes = 0;
        HIP_TRY(hipMalloc(&sc->probe_block, need));
        sc->probe_block_bytes = need;
    }
    char *blk = static_cast<char *>(sc->probe_block);
    DProbeCam *d_cams = reinterpret_cast<DProbeCam *>(blk);
    sc->pb.aux = reinterpret_cast<float4 *>(blk + cam_bytes);
    float *d_int = reinterpret_cast<float *>(blk + cam_bytes + aux_bytes);
    float *d_nrm = d_int + size_t(batch) * per_pixels * 3;
    float *d_dist = d_nrm + size_t(batch) * per_pixels * 3;
    sc->pb.nray_out = nullptr;
    sc->events_used = 0;
    HIP_TRY(hipEventRecord(sc->ev_begin, stream));
    for (int first = 0; first < n_probes; first += batch) {
        const int nb = std::min(batch, n_probes - first);
        HIP_TRY(hipMemcpyAsync(d_cams, cams.data() + first, size_t(nb) * sizeof(DProbeCam), hipMemcpyHostToDevice, stream));
        P.probe_cams = d_cams;
        P.n_owned_tiles = nb * P.probe_tiles;
        P.n_paths = uint32_t(uint64_t(nb) * slots_per_probe);
        rc = run_pass(sc, S, pr.max_depth, P, cfg, false);
        if (rc) return rc;
        const size_t px = size_t(nb) * per_pixels, off = size_t(first) * per_pixels;
        // the images stay in HBM for whatever consumes them next (the network), or go through the batch's device block
        float *o_int = outputs_on_device ? intensity_rgb + 3 * off : d_int, *o_nrm = outputs_on_device ? normals_xyz + 3 * off : d_nrm;
        float *o_dist = outputs_on_device ? distance + off : d_dist;
        if (!launch_probe_film(S, P, sc->pb, nb, o_int, o_nrm, o_dist, cfg)) {
            launch_film_store(S, P, sc->pb, sc->fb, 0, 1, cfg);
            launch_film_gather(S, P, sc->fb, 1, cfg);
            launch_probe_finish(S, P, sc->pb, sc->fb, nb, o_int, o_nrm, o_dist, cfg);
        }
        HIP_TRY(hipGetLastError());
        if (!outputs_on_device) {
            HIP_TRY(hipMemcpyAsync(intensity_rgb + 3 * off, d_int, px * 3 * sizeof(float), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(normals_xyz + 3 * off, d_nrm, px * 3 * sizeof(float), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(distance + off, d_dist, px * sizeof(float), hipMemcpyDeviceToHost, stream));
        }
        HIP_TRY(hipStreamSynchronize(stream));
        st.n_passes++;
        st.n_paths += uint64_t(nb) * per_pixels;
    }
    HIP_TRY(hipEventRecord(sc->ev_end, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, sc->ev_begin, sc->ev_end));
    st.ms_total = ms;
    st.workspace_bytes = sc->ws_bytes;
    sc->pb.aux = nullptr;
    if (stats) *stats = st;
    return IILE_OK;
}

// ---- the IISPT runner's gather ----------------------------------------------------
namespace {
int iispt_check(iile_scene *sc, const iile_iispt_task *t, int *nx, int *ny) {
    if (!sc || !t) return fail(IILE_ERR_ARG, "iile_iispt: null argument");
    int rc = ensure_device();
    if (rc) return rc;
    if (t->x1 <= t->x0 || t->y1 <= t->y0 || t->tilesize < 1) return fail(IILE_ERR_ARG, "iile_iispt: empty task or tilesize < 1");
    if (sc->ds.sobol) return fail(IILE_ERR_UNSUPPORTED, "iile_iispt: the runner's camera samples need the scene's Halton sampler");
    if (sc->probe.hemi_size != 32) return fail(IILE_ERR_UNSUPPORTED, "iile_iispt: the gather is built for 32 x 32 hemispheres (iisptHemiSize)");
    *nx = iile_iispt_grid_count(t->x0, t->x1, t->tilesize);
    *ny = iile_iispt_grid_count(t->y0, t->y1, t->tilesize);
    if (uint64_t(t->counter_base) + uint64_t(*nx) * uint64_t(*ny) + uint64_t(t->x1 - t->x0) * uint64_t(t->y1 - t->y0) >= 0x7fffffffull)
        return fail(IILE_ERR_UNSUPPORTED, "iile_iispt: the sampler's pixel counter would pass INT_MAX");
    return IILE_OK;
}
}  // namespace

namespace {
// The items of a task in HBM: hemi points first, then (with_pixels) the film pixels. Everything a call needs on the device is
// carved from the scene's scratch block (grown on demand, kept): the runner makes hundreds of calls per frame.
size_t carve_bytes(size_t n, size_t elem) { return (std::max<size_t>(n, 1) * elem + 255) & ~size_t(255); }
constexpr int kIisptMaxJobs = 1024;  // tasks per launch (blockIdx.y); longer batches run in slices
// The traversal kernel of a slice runs max(tasks, ~6 blocks per CU) blocks and every block owns a column of the stack spill
// array, which is sized for 8 blocks per CU (max_traversal_threads): a slice never holds more tasks than that.
int iispt_slice_jobs(const iile_scene *sc) { return std::max(1, std::min(kIisptMaxJobs, sc->n_cus * 8)); }

// A slice of a batch laid out in the scratch block: per task its items (five float4 planes and the Halton indices, task after
// task), the job array the kernels read, one counter of items still on a specular chain.
struct IisptSlice {
    std::vector<IisptJob> jobs;
    std::vector<size_t> hemi_off, pix_off;  // per task: first hemi point / first film pixel of the slice's concatenated arrays
    size_t n_hemi = 0, n_pix = 0, n_items = 0;
    int max_items = 0, max_hemi = 0, max_pix = 0;
    IisptJob *d_jobs = nullptr;
    uint32_t *n_active = nullptr;
};
int plan_slice(iile_scene *sc, const iile_iispt_task *tasks, int n_tasks, bool with_pixels, IisptSlice *sl) {
    sl->jobs.resize(size_t(n_tasks));
    sl->hemi_off.resize(size_t(n_tasks));
    sl->pix_off.resize(size_t(n_tasks));
    for (int k = 0; k < n_tasks; ++k) {
        int nx = 0, ny = 0;
        const int rc = iispt_check(sc, &tasks[k], &nx, &ny);
        if (rc) return rc;
        IisptJob &J = sl->jobs[size_t(k)];
        std::memset(&J, 0, sizeof(J));
        J.T = tasks[k];
        J.ny = ny;
        const size_t nh = size_t(nx) * ny, np = with_pixels ? size_t(tasks[k].x1 - tasks[k].x0) * size_t(tasks[k].y1 - tasks[k].y0) : 0;
        J.I.n_hemi = int(nh), J.I.n_items = int(nh + np), J.I.nx = nx;
        sl->hemi_off[size_t(k)] = sl->n_hemi, sl->pix_off[size_t(k)] = sl->n_pix;
        sl->n_hemi += nh, sl->n_pix += np, sl->n_items += nh + np;
        sl->max_items = std::max(sl->max_items, int(nh + np)), sl->max_hemi = std::max(sl->max_hemi, int(nh)), sl->max_pix = std::max(sl->max_pix, int(np));
    }
    if (sl->n_items >= 0x7fffffffull) return fail(IILE_ERR_UNSUPPORTED, "iile_iispt: more than 2^31 items in one slice of a batch");
    return IILE_OK;
}
size_t slice_item_bytes(const IisptSlice &sl) {
    return carve_bytes(5 * sl.n_items, sizeof(float4)) + carve_bytes(sl.n_items + 64, sizeof(uint32_t)) + carve_bytes(sl.jobs.size(), sizeof(IisptJob));
}
// carve the item planes and point every job at its part (after scratch_reserve)
void carve_slice(iile_scene *sc, IisptSlice *sl) {
    float4 *planes = scratch_take<float4>(sc, 5 * sl->n_items);
    uint32_t *words = scratch_take<uint32_t>(sc, sl->n_items + 64);
    sl->d_jobs = scratch_take<IisptJob>(sc, sl->jobs.size());
    sl->n_active = words;
    size_t first = 0;
    for (IisptJob &J : sl->jobs) {
        const size_t n = sl->n_items;
        J.I.ro = planes + first, J.I.rd = planes + n + first, J.I.beta = planes + 2 * n + first, J.I.hit = planes + 3 * n + first, J.I.pf = planes + 4 * n + first;
        J.I.idx = words + 64 + first;
        J.I.n_active = words;
        first += size_t(J.I.n_items);
    }
}

int hemi_points_slice(iile_scene *sc, const iile_iispt_task *tasks, int n_tasks, uint8_t *valid, float *pos3, float *dir3, hipStream_t s) {
    IisptSlice sl;
    int rc = plan_slice(sc, tasks, n_tasks, false, &sl);
    if (rc) return rc;
    const size_t n = sl.n_hemi;
    if ((rc = scratch_reserve(sc, slice_item_bytes(sl) + carve_bytes(n, 1) + 2 * carve_bytes(3 * n, sizeof(float)), s))) return rc;
    carve_slice(sc, &sl);
    uint8_t *dv = scratch_take<uint8_t>(sc, n);
    float *dp = scratch_take<float>(sc, 3 * n), *dd = scratch_take<float>(sc, 3 * n);
    for (size_t k = 0; k < sl.jobs.size(); ++k)
        sl.jobs[k].valid = dv + sl.hemi_off[k], sl.jobs[k].pos3 = dp + 3 * sl.hemi_off[k], sl.jobs[k].dir3 = dd + 3 * sl.hemi_off[k];
    // (everything in the caller's stream's order: the copy follows whatever that stream did with the block last, the kernels follow it)
    HIP_TRY(hipMemcpyAsync(sl.d_jobs, sl.jobs.data(), sl.jobs.size() * sizeof(IisptJob), hipMemcpyHostToDevice, s));
    DScene S = sc->ds;
    S.diff_scale = 1.f;  // r.ScaleDifferentials(1.0), iisptrenderrunner.cpp:272
    LaunchCfg cfg{sc->n_cus, s, false};
    launch_iispt_first_hits(S, sl.d_jobs, n_tasks, sl.max_items, sl.n_active, sc->spill, cfg);
    launch_iispt_hemi_out(S, sl.d_jobs, n_tasks, sl.max_hemi, cfg);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(valid, dv, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(pos3, dp, 3 * n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(dir3, dd, 3 * n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));   // (the results are for the host, and `sl` dies with this call)
    return IILE_OK;
}

int gather_slice(iile_scene *sc, const iile_iispt_task *tasks, int n_tasks, const uint8_t *valid, const float *pos3, const float *dir3,
                 const float *nn_films, int32_t nn_on_device, float *out_rgbw, int32_t out_on_device, hipStream_t s) {
    IisptSlice sl;
    int rc = plan_slice(sc, tasks, n_tasks, true, &sl);
    if (rc) return rc;
    const size_t n = sl.n_hemi, n_pix = sl.n_pix;
    const int hemi = sc->probe.hemi_size;
    // the hemi points' cameras: CreateHemisphericCamera (hemispheric.cpp:109-160) — CameraToWorld from LookAt, WorldToCamera
    // its numerical inverse, the look direction and origin as given
    std::vector<DHemiCam> cams(n);
    for (size_t k = 0; k < n; ++k) {
        DHemiCam &hc = cams[k];
        std::memset(&hc, 0, sizeof(hc));
        if (!valid[k]) continue;
        DProbeCam pc;
        float inv[16];
        if (!make_probe_camera(pos3 + 3 * k, dir3 + 3 * k, &pc) || !invert4(pc.c2w.m, inv))
            return fail(IILE_ERR_ARG, "iile_iispt_gather: degenerate hemi point direction (hemi point " + std::to_string(k) + " of the slice)");
        hc.c2w = pc.c2w;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) hc.w2c[3 * r + c] = inv[4 * r + c];
        for (int c = 0; c < 3; ++c) hc.look[c] = dir3[3 * k + c], hc.origin[c] = pos3[3 * k + c];
        hc.valid = 1;
    }
    std::vector<float> jac(static_cast<size_t>(hemi), 0.f);  // IntensityFilm::get_camera_coord_jacobian, intensityfilm.cpp:60-66
    for (int y = 0; y < hemi; ++y) {
        const float abs_vertical_value = float(y) / hemi;
        const float polar_vertical_value = float(M_PI * abs_vertical_value);
        jac[size_t(y)] = std::sin(polar_vertical_value);  // sin(Float): the float overload (sinf), as in the reference
    }
    const size_t per_hemi = size_t(hemi) * hemi * 3, nn_floats = n * per_hemi;
    if ((rc = scratch_reserve(sc, slice_item_bytes(sl) + carve_bytes(n, sizeof(DHemiCam)) + carve_bytes(jac.size(), sizeof(float)) +
                                      (nn_on_device ? 0 : carve_bytes(nn_floats, sizeof(float))) + (out_on_device ? 0 : carve_bytes(n_pix, sizeof(float4))),
                              s)))
        return rc;
    carve_slice(sc, &sl);
    DHemiCam *dc = scratch_take<DHemiCam>(sc, n);
    float *dj = scratch_take<float>(sc, jac.size());
    const float *nn_dev = nn_films;
    float *dnn = nullptr;
    if (!nn_on_device) nn_dev = dnn = scratch_take<float>(sc, nn_floats);
    float4 *out_dev = reinterpret_cast<float4 *>(out_rgbw);
    if (!out_on_device) out_dev = scratch_take<float4>(sc, n_pix);
    for (size_t k = 0; k < sl.jobs.size(); ++k)
        sl.jobs[k].cams = dc + sl.hemi_off[k], sl.jobs[k].nn_films = nn_dev + sl.hemi_off[k] * per_hemi, sl.jobs[k].out = out_dev + sl.pix_off[k];
    // (copies from these short-lived host vectors, in the caller's stream's order: they follow whatever that stream did with the block
    // last — and the network's kernels that wrote nn_films, when the caller queued them on the same stream)
    HIP_TRY(hipMemcpyAsync(sl.d_jobs, sl.jobs.data(), sl.jobs.size() * sizeof(IisptJob), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dc, cams.data(), n * sizeof(DHemiCam), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dj, jac.data(), jac.size() * sizeof(float), hipMemcpyHostToDevice, s));
    if (dnn) HIP_TRY(hipMemcpyAsync(dnn, nn_films, nn_floats * sizeof(float), hipMemcpyHostToDevice, s));
    DScene S = sc->ds;
    S.diff_scale = 1.f;
    LaunchCfg cfg{sc->n_cus, s, false};
    launch_iispt_first_hits(S, sl.d_jobs, n_tasks, sl.max_items, sl.n_active, sc->spill, cfg);   // (waits on the stream between its rounds: the vectors above are consumed)
    launch_iispt_gather(S, sl.d_jobs, n_tasks, sl.max_pix, dj, cfg);
    HIP_TRY(hipGetLastError());
    // Results on the device: the kernels are in the stream's order and the call returns (the caller's next use of the output, on that
    // stream or one that synchronises with it, follows them). Results for the host: the copy waits.
    if (!out_on_device) {
        HIP_TRY(hipMemcpyAsync(out_rgbw, out_dev, n_pix * sizeof(float4), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return IILE_OK;
}
}  // namespace

int iile_iispt_hemi_points_batch(iile_scene *sc, const iile_iispt_task *tasks, int32_t n_tasks, uint8_t *valid, float *pos3, float *dir3, void *stream) {
    if (!sc || !tasks || n_tasks < 1) return fail(IILE_ERR_ARG, "iile_iispt_hemi_points: no task");
    if (!valid || !pos3 || !dir3) return fail(IILE_ERR_ARG, "iile_iispt_hemi_points: null output");
    size_t first = 0;
    const int slice = iispt_slice_jobs(sc);
    for (int k0 = 0; k0 < n_tasks; k0 += slice) {
        const int nk = std::min(slice, n_tasks - k0);
        const int rc = hemi_points_slice(sc, tasks + k0, nk, valid + first, pos3 + 3 * first, dir3 + 3 * first, static_cast<hipStream_t>(stream));
        if (rc) return rc;
        for (int k = k0; k < k0 + nk; ++k)
            first += size_t(iile_iispt_grid_count(tasks[k].x0, tasks[k].x1, tasks[k].tilesize)) * size_t(iile_iispt_grid_count(tasks[k].y0, tasks[k].y1, tasks[k].tilesize));
    }
    return IILE_OK;
}
int iile_iispt_hemi_points(iile_scene *sc, const iile_iispt_task *t, uint8_t *valid, float *pos3, float *dir3) {
    return iile_iispt_hemi_points_batch(sc, t, 1, valid, pos3, dir3, nullptr);
}

int iile_iispt_gather_batch(iile_scene *sc, const iile_iispt_task *tasks, int32_t n_tasks, const uint8_t *valid, const float *pos3, const float *dir3,
                            const float *nn_films, int32_t nn_on_device, float *out_rgbw, int32_t out_on_device, void *stream) {
    if (!sc || !tasks || n_tasks < 1) return fail(IILE_ERR_ARG, "iile_iispt_gather: no task");
    if (!valid || !pos3 || !dir3 || !nn_films || !out_rgbw) return fail(IILE_ERR_ARG, "iile_iispt_gather: null argument");
    const size_t per_hemi = size_t(sc->probe.hemi_size) * sc->probe.hemi_size * 3;
    size_t first_h = 0, first_p = 0;
    const int slice = iispt_slice_jobs(sc);
    for (int k0 = 0; k0 < n_tasks; k0 += slice) {
        const int nk = std::min(slice, n_tasks - k0);
        const int rc = gather_slice(sc, tasks + k0, nk, valid + first_h, pos3 + 3 * first_h, dir3 + 3 * first_h, nn_films + first_h * per_hemi, nn_on_device,
                                    out_rgbw + 4 * first_p, out_on_device, static_cast<hipStream_t>(stream));
        if (rc) return rc;
        for (int k = k0; k < k0 + nk; ++k) {
            first_h += size_t(iile_iispt_grid_count(tasks[k].x0, tasks[k].x1, tasks[k].tilesize)) * size_t(iile_iispt_grid_count(tasks[k].y0, tasks[k].y1, tasks[k].tilesize));
            first_p += size_t(tasks[k].x1 - tasks[k].x0) * size_t(tasks[k].y1 - tasks[k].y0);
        }
    }
    return IILE_OK;
}
int iile_iispt_film_add(iile_scene *sc, const iile_iispt_task *tasks, int32_t n_tasks, const float *out_rgbw_dev, double *film_rgbw_dev,
                        int32_t film_w, int32_t film_h, void *stream) {
    if (!sc || !tasks || n_tasks < 1 || !out_rgbw_dev || !film_rgbw_dev || film_w < 1 || film_h < 1)
        return fail(IILE_ERR_ARG, "iile_iispt_film_add: bad argument");
    if (n_tasks > 65535) return fail(IILE_ERR_UNSUPPORTED, "iile_iispt_film_add: more than 65535 tasks in one call");
    std::vector<int4> rects(static_cast<size_t>(n_tasks));
    std::vector<uint32_t> first(static_cast<size_t>(n_tasks));
    uint64_t at = 0;
    int max_pixels = 1;
    for (int k = 0; k < n_tasks; ++k) {
        const iile_iispt_task &t = tasks[k];
        if (t.x0 < 0 || t.y0 < 0 || t.x1 > film_w || t.y1 > film_h || t.x1 <= t.x0 || t.y1 <= t.y0)
            return fail(IILE_ERR_ARG, "iile_iispt_film_add: a task lies outside the film");
        rects[size_t(k)] = make_int4(t.x0, t.y0, t.x1, t.y1);
        first[size_t(k)] = uint32_t(at);
        const uint64_t n = uint64_t(t.x1 - t.x0) * uint64_t(t.y1 - t.y0);
        at += n;
        max_pixels = std::max<int>(max_pixels, int(std::min<uint64_t>(n, 1u << 30)));
    }
    if (at >= 0xffffffffull) return fail(IILE_ERR_UNSUPPORTED, "iile_iispt_film_add: more than 2^32 pixels in one call");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t bytes = size_t(n_tasks) * (sizeof(int4) + sizeof(uint32_t));
    if (bytes > sc->film_add_cap) {   // a table of its own (the scene's shared scratch may still be read by the gather's kernels)
        HIP_TRY(hipStreamSynchronize(s));
        if (sc->film_add_buf) HIP_TRY(hipFree(sc->film_add_buf));
        sc->film_add_buf = nullptr;
        sc->film_add_cap = 0;
        HIP_TRY(hipMalloc(&sc->film_add_buf, 2 * bytes));
        sc->film_add_cap = 2 * bytes;
    }
    int4 *d_rects = static_cast<int4 *>(sc->film_add_buf);
    uint32_t *d_first = reinterpret_cast<uint32_t *>(d_rects + n_tasks);
    HIP_TRY(hipMemcpyAsync(d_rects, rects.data(), rects.size() * sizeof(int4), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_first, first.data(), first.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));   // (the vectors die with this call; the kernels before have to finish anyway)
    launch_iispt_film_add(d_rects, d_first, n_tasks, max_pixels, reinterpret_cast<const float4 *>(out_rgbw_dev), film_rgbw_dev, film_w, s);
    HIP_TRY(hipGetLastError());
    return IILE_OK;
}

int iile_iispt_film_merge(const double *direct_rgbw_dev, const double *indirect_rgbw_dev, int64_t n_pixels, float *rgb_dev, void *stream) {
    if (!direct_rgbw_dev || !indirect_rgbw_dev || !rgb_dev || n_pixels < 0) return fail(IILE_ERR_ARG, "iile_iispt_film_merge: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    if (n_pixels == 0) return IILE_OK;
    launch_iispt_film_merge(direct_rgbw_dev, indirect_rgbw_dev, rgb_dev, (long long)n_pixels, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return IILE_OK;
}

int iile_iispt_gather(iile_scene *sc, const iile_iispt_task *t, const uint8_t *valid, const float *pos3, const float *dir3, const float *nn_films,
                      int32_t nn_on_device, float *out_rgbw, int32_t out_on_device) {
    return iile_iispt_gather_batch(sc, t, 1, valid, pos3, dir3, nn_films, nn_on_device, out_rgbw, out_on_device, nullptr);
}

// ---- kernel-level entry points ---------------------------------------------

static int trace_common(iile_scene *sc, int32_t n, const float *o3, const float *d3, const float *tmax, int any,
                        std::vector<float4> *hits, iile_stats *stats) {
    if (!sc || n < 0 || !o3 || !d3 || !tmax) return fail(IILE_ERR_ARG, "iile_trace: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    std::vector<float4> ro(n), rd(n);
    for (int i = 0; i < n; ++i) {
        ro[i] = make_float4(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2], 0);
        rd[i] = make_float4(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2], tmax[i]);
    }
    float4 *dro = nullptr, *drd = nullptr, *dh = nullptr;
    DCounters *dc = nullptr;
    const size_t nb = std::max<size_t>(size_t(n), 1) * sizeof(float4);
    HIP_TRY(hipMalloc(&dro, nb));
    HIP_TRY(hipMalloc(&drd, nb));
    HIP_TRY(hipMalloc(&dh, 2 * nb));
    HIP_TRY(hipMalloc(&dc, sizeof(DCounters)));
    HIP_TRY(hipMemset(dc, 0, sizeof(DCounters)));
    HIP_TRY(hipMemcpy(dro, ro.data(), size_t(n) * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(drd, rd.data(), size_t(n) * sizeof(float4), hipMemcpyHostToDevice));
    LaunchCfg cfg{sc->n_cus, nullptr, stats != nullptr};
    if (n) launch_trace(sc->ds, n, dro, drd, dh, any, dc, sc->spill, cfg);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    hits->resize(2 * size_t(n));
    HIP_TRY(hipMemcpy(hits->data(), dh, 2 * size_t(n) * sizeof(float4), hipMemcpyDeviceToHost));
    if (stats) {
        DCounters c;
        HIP_TRY(hipMemcpy(&c, dc, sizeof(c), hipMemcpyDeviceToHost));
        std::memset(stats, 0, sizeof(*stats));
        copy_counters(c, stats);
    }
    (void)hipFree(dro);
    (void)hipFree(drd);
    (void)hipFree(dh);
    (void)hipFree(dc);
    return IILE_OK;
}

int iile_render_status(iile_scene *sc, void *stream) {
    if (!sc) return fail(IILE_ERR_ARG, "iile_render_status: null scene");
    int rc = ensure_device();
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return check_pending_overflow(sc);
}

int iile_test_patch_capacity(iile_scene *sc, uint32_t capacity) {
    if (!sc) return fail(IILE_ERR_ARG, "iile_test_patch_capacity: null scene");
    sc->patch_cap_override = capacity;
    return IILE_OK;
}

int iile_trace_closest(iile_scene *sc, int32_t n, const float *o3, const float *d3, const float *tmax, int32_t *prim,
                       float *tb, iile_stats *stats) {
    if (!prim || !tb) return fail(IILE_ERR_ARG, "iile_trace_closest: null output");
    std::vector<float4> hits;
    int rc = trace_common(sc, n, o3, d3, tmax, 0, &hits, stats);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        uint32_t u;
        std::memcpy(&u, &hits[2 * i].x, 4);
        prim[i] = int32_t(u);
        tb[4 * i] = hits[2 * i].y;
        tb[4 * i + 1] = hits[2 * i + 1].x;
        tb[4 * i + 2] = hits[2 * i + 1].y;
        tb[4 * i + 3] = hits[2 * i + 1].z;
    }
    return IILE_OK;
}

int iile_trace_any(iile_scene *sc, int32_t n, const float *o3, const float *d3, const float *tmax, int32_t *hit,
                   iile_stats *stats) {
    if (!hit) return fail(IILE_ERR_ARG, "iile_trace_any: null output");
    std::vector<float4> hits;
    int rc = trace_common(sc, n, o3, d3, tmax, 1, &hits, stats);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        uint32_t u;
        std::memcpy(&u, &hits[2 * i].x, 4);
        hit[i] = int32_t(u);
    }
    return IILE_OK;
}


int iile_halton_samples(iile_scene *sc, int32_t n, const int32_t *px, const int32_t *py, const int32_t *k,
                        int32_t dim0, int32_t ndims, float *out, uint32_t *index_out) {
    if (!sc || n < 0 || !px || !py || !k || !out || ndims <= 0 || dim0 < 0 || dim0 + ndims > (sc->ds.sobol ? sc->ds.sobol_dims : sc->ds.n_hdims))
        return fail(IILE_ERR_ARG, "iile_halton_samples: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    DevBuf<int> dx, dy, dk;
    DevBuf<float> dout;
    DevBuf<uint32_t> dindex;
    if ((rc = dx.put(px, n)) || (rc = dy.put(py, n)) || (rc = dk.put(k, n)) || (rc = dout.alloc(size_t(n) * ndims)) ||
        (rc = dindex.alloc(n)))
        return rc;
    LaunchCfg cfg{sc->n_cus, nullptr, false};
    if (n) launch_halton(sc->ds, n, dx.p, dy.p, dk.p, dim0, ndims, dout.p, dindex.p, cfg);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if ((rc = dout.get(out, size_t(n) * ndims))) return rc;
    if (index_out && (rc = dindex.get(index_out, n))) return rc;
    return IILE_OK;
}

int iile_camera_rays(iile_scene *sc, int32_t n, const float *pfilm2, const float *plens2, float *o3, float *d3) {
    if (!sc || n < 0 || !pfilm2 || !o3 || !d3) return fail(IILE_ERR_ARG, "iile_camera_rays: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    DevBuf<float> df, dl, dox, ddx;
    if ((rc = df.put(pfilm2, 2 * size_t(n))) || (rc = dox.alloc(3 * size_t(n))) || (rc = ddx.alloc(3 * size_t(n))))
        return rc;
    if (plens2 && (rc = dl.put(plens2, 2 * size_t(n)))) return rc;
    LaunchCfg cfg{sc->n_cus, nullptr, false};
    if (n) launch_camera(sc->ds, n, df.p, plens2 ? dl.p : nullptr, dox.p, ddx.p, cfg);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if ((rc = dox.get(o3, 3 * size_t(n))) || (rc = ddx.get(d3, 3 * size_t(n)))) return rc;
    return IILE_OK;
}

int iile_li_samples(iile_scene *sc, int32_t n, const int32_t *px, const int32_t *py, const int32_t *k, float *L3,
                    int32_t *nrays2) {
    if (!sc || n <= 0 || !px || !py || !k || !L3) return fail(IILE_ERR_ARG, "iile_li_samples: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    DevBuf<int> dx, dy, dk;
    DevBuf<uint32_t> dn;
    if ((rc = dx.put(px, n)) || (rc = dy.put(py, n)) || (rc = dk.put(k, n)) || (rc = dn.alloc(2 * size_t(n)))) return rc;
    rc = ensure_workspace(sc, uint32_t(n));
    if (rc) return rc;
    PassDesc P;
    std::memset(&P, 0, sizeof(P));
    P.n_tiles_x = P.n_tiles_y = 1;
    P.tile_nranks = 1;
    P.kc = 1;
    P.n_paths = uint32_t(n);
    P.list_px = dx.p;
    P.list_py = dy.p;
    P.list_k = dk.p;
    LaunchCfg cfg{sc->n_cus, nullptr, true};
    HIP_TRY(hipMemset(sc->pb.counters, 0, sizeof(DCounters)));
    sc->pb.nray_out = dn.p;
    sc->events_used = 0;
    rc = run_pass(sc, sc->ds, sc->max_depth, P, cfg, false);
    sc->pb.nray_out = nullptr;
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    std::vector<float4> L(n);
    HIP_TRY(hipMemcpy(L.data(), sc->pb.L, size_t(n) * sizeof(float4), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        // guards of SamplerIntegrator::Render (integrator.cpp:293-314)
        float r = L[i].x, g = L[i].y, b = L[i].z;
        float y = 0.212671f * r + 0.715160f * g + 0.072169f * b;
        if (std::isnan(r) || std::isnan(g) || std::isnan(b) || y < -1e-5 || std::isinf(y)) r = g = b = 0.f;
        L3[3 * i] = r;
        L3[3 * i + 1] = g;
        L3[3 * i + 2] = b;
    }
    if (nrays2) {
        std::vector<uint32_t> nr(2 * size_t(n));
        if ((rc = dn.get(nr.data(), nr.size()))) return rc;
        for (size_t i = 0; i < nr.size(); ++i) nrays2[i] = int32_t(nr[i]);
    }
    return IILE_OK;
}

static int bsdf_probe(iile_scene *sc, int32_t n, int32_t mat, const float *wo3, const float *in, size_t in_stride,
                      int sample, float *out, size_t out_stride) {
    if (!sc || n < 0 || !wo3 || !in || !out || mat < 0 || mat >= sc->ds.n_materials)
        return fail(IILE_ERR_ARG, "iile_bsdf: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    DevBuf<float> dwo, din, dout;
    if ((rc = dwo.put(wo3, 3 * size_t(n))) || (rc = din.put(in, in_stride * size_t(n))) ||
        (rc = dout.alloc(out_stride * size_t(n))))
        return rc;
    LaunchCfg cfg{sc->n_cus, nullptr, false};
    if (n) launch_bsdf_probe(sc->ds, n, mat, dwo.p, din.p, sample, dout.p, cfg);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return dout.get(out, out_stride * size_t(n));
}
int iile_bsdf_eval(iile_scene *sc, int32_t n, int32_t mat, const float *wo3, const float *wi3, float *out4) {
    return bsdf_probe(sc, n, mat, wo3, wi3, 3, 0, out4, 4);
}
int iile_bsdf_sample(iile_scene *sc, int32_t n, int32_t mat, const float *wo3, const float *u2, float *out7) {
    return bsdf_probe(sc, n, mat, wo3, u2, 2, 1, out7, 7);
}

int iile_texture_eval(iile_scene *sc, int32_t tex, int32_t n, const float *uv2, const float *duv4, float *rgb3) {
    if (!sc || n < 0 || !uv2 || !duv4 || !rgb3 || tex < 0 || tex >= sc->ds.n_textures)
        return fail(IILE_ERR_ARG, "iile_texture_eval: bad argument");
    DevBuf<float> duv, dd, dout;
    int rc;
    if ((rc = duv.put(uv2, 2 * size_t(n))) || (rc = dd.put(duv4, 4 * size_t(n))) || (rc = dout.alloc(3 * size_t(n)))) return rc;
    LaunchCfg cfg{sc->n_cus, nullptr, false};
    if (n) launch_texture_probe(sc->ds, n, tex, duv.p, dd.p, dout.p, cfg);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return dout.get(rgb3, 3 * size_t(n));
}

int iile_trig_probe(int32_t n, const float *x, float *out3) {
    if (n < 0 || !x || !out3) return fail(IILE_ERR_ARG, "iile_trig_probe: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    DevBuf<float> dx, dout;
    if ((rc = dx.put(x, n)) || (rc = dout.alloc(3 * size_t(n)))) return rc;
    LaunchCfg cfg{256, nullptr, false};
    if (n) launch_trig_probe(n, dx.p, dout.p, cfg);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return dout.get(out3, 3 * size_t(n));
}

}  // extern "C"
