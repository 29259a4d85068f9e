// kernels.hip — hand-written gfx950 kernels of the wavefront path tracer.
//
// Pipeline per pass (one pass = owned tiles x a chunk of sample indices):
//   generate -> [ extend -> shade -> mis -> mis_lit -> shadow ] x (maxDepth + 1) -> film_accumulate
// and, after the last pass, film_resolve.
//
//   generate  HaltonSampler + PerspectiveCamera::GenerateRayDifferential; fills ray queue 0
//   extend    BVHAccel::Intersect over a ray queue (closest hit), LDS traversal stacks
//   shade     the body of PathIntegrator::Li for one bounce: interaction, Le, BSDF, light
//             sampling + BSDF sampling of EstimateDirect (emits one NEE record holding a
//             shadow ray and an MIS ray), next direction, Russian roulette; compacts the
//             surviving paths into the next ray queue with ballot + one atomic per wavefront
//   shadow    BVHAccel::IntersectP for the shadow ray of each NEE record
//   mis       BVHAccel::Intersect for the MIS ray of each NEE record
//   shadow    BVHAccel::IntersectP for its shadow ray; L += beta * Ld
//
// All kernels are persistent grid-stride loops that read their queue length
// from device memory, so a whole pass is enqueued without host synchronisation.
// 64-lane wavefronts throughout: ballots are 64-bit, lane = threadIdx.x & 63.
#include <algorithm>

#include "dpath.h"
#include "kernels.h"

namespace iile {

// Phase scheduling of the traversal kernels: 1 = one step per iteration for the whole
// wavefront, interior or leaf, whichever has more lanes waiting; 0 = strict while-while.
#ifndef IILE_FLAT_EXTEND
#define IILE_FLAT_EXTEND 1
#endif
#ifndef IILE_FLAT_SHADOW
#define IILE_FLAT_SHADOW 1
#endif
#ifndef IILE_FLAT_MIS
#define IILE_FLAT_MIS 1
#endif
#ifndef IILE_TRAV_WAVES
#define IILE_TRAV_WAVES 6  // waves per SIMD = resident blocks per CU of the traversal kernels (<= 80 VGPRs, no scratch)
#endif
#ifndef IILE_VOTE_NUM
#define IILE_VOTE_NUM 4
#define IILE_VOTE_DEN 5
#endif

constexpr int kBlock = 256;            // 4 wavefronts
constexpr int kWavesPerBlock = kBlock / 64;
#ifndef IILE_SHADE_CHUNK
#define IILE_SHADE_CHUNK 512
#endif
constexpr int kShadeChunk = IILE_SHADE_CHUNK;  // hits one k_shade wavefront regroups by shading class at a time
constexpr int kTile = 16;

DEV int lane_id() { return int(threadIdx.x & 63); }
DEV uint32_t lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}
// Wavefront-aggregated append: one atomic per wavefront reserves a contiguous
// run of queue slots; lanes take slots in lane order.
DEV uint32_t wave_append(bool emit, uint32_t *counter) {
    const unsigned long long mask = __ballot(emit);
    if (mask == 0) return 0;
    const uint32_t n = uint32_t(__popcll(mask));
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane_id() == leader) base = atomicAdd(counter, n);
    base = __shfl(base, leader);
    return base + lanes_below(mask);
}
// Block-reserved queue output. A returning atomic on one queue-tail word per
// wavefront per iteration saturates that word (~88 atomics/us on MI355X) long
// before the kernels run out of anything else, so a wavefront instead reserves
// kOutBlock slots at a time with ONE atomic and appends into its private block
// (ballot + mbcnt, no memory traffic). Slots it cannot use — the < 64 left when a
// block runs out, and the tail of its last block — are padded with an INVALID
// record that consumers skip. The queue length a consumer sees is the number of
// reserved slots.
constexpr uint32_t kOutBlock = 1024;
constexpr uint32_t kInvalid = 0xffffffffu;
struct WaveOut {
    uint32_t cur, end;
};
template <typename Pad>
DEV uint32_t out_take(WaveOut &o, uint32_t *counter, bool emit, Pad pad) {
    const unsigned long long mask = __ballot(emit);
    const uint32_t n = uint32_t(__popcll(mask));
    if (n == 0) return 0;
    if (o.end - o.cur < n) {
        const uint32_t left = o.end - o.cur;  // < n <= 64
        if (uint32_t(lane_id()) < left) pad(o.cur + uint32_t(lane_id()));
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(counter, kOutBlock);
        base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
        o.cur = base;
        o.end = base + kOutBlock;
    }
    const uint32_t slot = o.cur + lanes_below(mask);
    o.cur += n;
    return slot;
}
template <typename Pad>
DEV void out_flush(WaveOut &o, Pad pad) {
    for (uint32_t sl = o.cur + uint32_t(lane_id()); sl < o.end; sl += 64) pad(sl);
    o.cur = o.end;
}

DEV unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
DEV void flush_counter(unsigned long long *dst, unsigned long long v) {
    v = wave_sum(v);
    if (lane_id() == 0 && v) atomicAdd(dst, v);
}
static inline int grid_blocks(uint32_t n, int n_cus, int per_cu) {
    long want = (long(n) + kBlock - 1) / kBlock;
    long cap = long(n_cus) * per_cu;
    if (want < 1) want = 1;
    return int(want < cap ? want : cap);
}

// layout of PassBuffers::counts (zeroed once per pass)
constexpr int kCntRay = 0;       // [bounce] rays in the extend queue
constexpr int kCntNee = 16;      // [bounce] NEE records
constexpr int kCntShade = 32;    // [bounce] hits to shade
constexpr int kCntExtHead = 48;  // [bounce] chunk cursor of the extend queue
constexpr int kCntConHead = 64;  // [bounce] chunk cursor of the NEE queue (shadow kernel)
constexpr int kCntMisHead = 80;  // [bounce] chunk cursor of the NEE queue (MIS kernel)
constexpr int kCntShdHead = 96;  // [bounce] chunk cursor of the shade queue
constexpr int kCntMis = 112;     // [bounce] MIS rays (a dense queue of its own: most NEE records have none)

// Persistent-wavefront work feed. A wavefront reserves kChunk consecutive queue
// slots with one atomic and hands them to its lanes as they go idle, so lanes
// whose ray terminated early pick up new rays instead of waiting for the slowest
// lane of the wavefront (the classic while-while + dynamic fetch scheme, sized
// for 64 lanes). 512-slot chunks keep the head word at a few atomics per
// microsecond, far below its ~88/us saturation point.
constexpr uint32_t kChunk = 512;
#ifndef IILE_REFILL_IDLE
#define IILE_REFILL_IDLE 16
#endif
constexpr int kRefillIdle = IILE_REFILL_IDLE;  // refill once this many lanes are idle (or all)
struct WaveFeed {
    uint32_t cur, end;
    bool exhausted;
};
// `warm(first_slot)` is called once per new chunk: the wavefront touches every 128-byte
// line of the chunk's records (lane l -> records first+8l .. first+8l+7), so the per-lane
// refill loads that follow hit L2 instead of paying an HBM round trip each time a few
// lanes go idle.
template <typename Warm>
DEV bool feed_take(WaveFeed &f, uint32_t *head, uint32_t count, bool idle, uint32_t *slot, Warm warm) {
    const unsigned long long mask = __ballot(idle);
    const uint32_t n_idle = uint32_t(__popcll(mask));
    if (f.cur == f.end) {
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(head, kChunk);
        base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
        if (base >= count) {
            f.exhausted = true;
            f.cur = f.end = 0;
            return false;
        }
        f.cur = base;
        f.end = (base + kChunk < count) ? base + kChunk : count;
        warm(base);
    }
    const uint32_t avail = f.end - f.cur;
    const uint32_t take = n_idle < avail ? n_idle : avail;
    const uint32_t rank = lanes_below(mask);
    *slot = f.cur + rank;
    f.cur += take;
    return idle && rank < take;
}

// touch one float4 of every 128-byte line of records [first, first + kChunk) of a float4 plane
static_assert(kChunk == 64 * 8, "one lane per 128-byte line of a chunk");
DEV void warm_plane(const float4 *plane_base, uint32_t first, uint32_t limit) {
    const uint32_t i = first + uint32_t(lane_id()) * 8u;
    if (i < limit) {
        const float v = plane_base[i].x;
        asm volatile("" ::"v"(v));  // keep the load; the value itself is not needed
    }
}

// pid -> (pixel, sample) for tile enumeration: pid = ((tile_slot*256 + pix)*kc + kk)
DEV bool path_pixel(const DScene &S, const PassDesc &P, uint32_t pid, int *px, int *py, uint32_t *k) {
    if (P.list_px) {
        *px = P.list_px[pid];
        *py = P.list_py[pid];
        *k = uint32_t(P.list_k[pid]);
        return true;
    }
    const uint32_t kk = pid % uint32_t(P.kc);
    const uint32_t pt = pid / uint32_t(P.kc);
    const uint32_t pix = pt & 255u, slot = pt >> 8;
    if (P.probe_mode) {
        // every probe has its own film: tile `slot % probe_tiles` of probe `slot / probe_tiles`; RenderView skips the
        // pixels outside the film's pixel bounds (iispt_d.cpp:428-429)
        const int tile = int(slot % uint32_t(P.probe_tiles));
        const int tx = tile % P.n_tiles_x, ty = tile / P.n_tiles_x;
        *px = S.samp_x0 + tx * kTile + int(pix & 15u);
        *py = S.samp_y0 + ty * kTile + int(pix >> 4);
        *k = uint32_t(P.k0) + kk;
        return *px >= S.crop_x0 && *py >= S.crop_y0 && *px < S.crop_x1 && *py < S.crop_y1;
    }
    const int tile = P.tile_of_slot ? P.tile_of_slot[P.slot0 + int(slot)] : P.slot0 + int(slot);
    const int tx = tile % P.n_tiles_x, ty = tile / P.n_tiles_x;
    *px = S.samp_x0 + tx * kTile + int(pix & 15u);
    *py = S.samp_y0 + ty * kTile + int(pix >> 4);
    *k = uint32_t(P.k0) + kk;
    return *px < S.samp_x1 && *py < S.samp_y1;
}

// A film position that is a whole number (u == 0, or float(px) + u rounded to px or px + 1 where the pixel
// coordinate is large) puts the sample into two pixels along that axis under the one-pixel box filter
// (FilmTile::AddSample, film.h:159-166: pixels ceil(pFilm - 1) .. floor(pFilm)). Rare (1080p x 64 spp: ~1e-4 of the
// samples); they are listed here and the pixels they touch are finished exactly by iile_render (api.hip).
DEV void flag_whole_film_position(const PassBuffers &B, uint32_t pid, int px, int py, uint32_t k, float pfx, float pfy, float u0,
                                  float u1) {
    if (!B.flag_count) return;
    if (pfx == float(px) || pfx == float(px + 1) || pfy == float(py) || pfy == float(py + 1)) {
        const uint32_t at = atomicAdd(B.flag_count, 1u);
        if (at < kMaxFlagged) {
            float *r = B.flag_rec + 6 * size_t(at);
            r[0] = b2f(uint32_t(px));
            r[1] = b2f(uint32_t(py));
            r[2] = b2f(k | (u0 == 0.f ? 1u << 30 : 0u) | (u1 == 0.f ? 1u << 31 : 0u));  // + "the offset is an exact zero"
            r[3] = pfx;
            r[4] = pfy;
            r[5] = b2f(pid);  // its path id in the pass that made it
        }
    }
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_generate(DScene S, PassDesc P, PassBuffers B, int count_stats) {
    unsigned long long n_cam = 0;
    for (uint32_t base = blockIdx.x * kBlock; base < P.n_paths; base += gridDim.x * kBlock) {
        const uint32_t pid = base + threadIdx.x;
        bool valid = pid < P.n_paths;
        int px = 0, py = 0;
        uint32_t k = 0;
        if (valid) valid = path_pixel(S, P, pid, &px, &py, &k);
        F3 o = F3{0, 0, 0}, d = F3{0, 0, 1};
        float tmax = 0;
        if (valid) {
            // Sampler::GetCameraSample (sampler.cpp:46-52): dims 0,1 film, 2 time, 3,4 lens
            const uint32_t idx = sample_index(S, px, py, k);
            const float u0 = sample_dimension(S, idx, 0, px, py), u1 = sample_dimension(S, idx, 1, px, py);
            float l0 = 0, l1 = 0;
            if (S.lens_radius > 0) {
                l0 = sample_dimension(S, idx, 3);
                l1 = sample_dimension(S, idx, 4);
            }
            if (P.probe_mode) {
                const uint32_t probe = (pid / uint32_t(P.kc)) / (256u * uint32_t(P.probe_tiles));
                probe_ray(S, P.probe_cams[probe], float(px) + u0, float(py) + u1, &o, &d, &tmax);
                B.aux[pid] = make_float4(0, 0, 0, -1.f);  // no intersection: normal 0, NO_INTERSECTION_DISTANCE
            } else
                camera_ray(S, float(px) + u0, float(py) + u1, l0, l1, &o, &d, &tmax);
            if (!P.list_px && !P.probe_mode) flag_whole_film_position(B, pid, px, py, k, float(px) + u0, float(py) + u1, u0, u1);
            B.hindex[pid] = idx;
            B.L[pid] = make_float4(0, 0, 0, 0);
            if (B.nray_out) {
                B.nray_out[2 * pid] = 0;
                B.nray_out[2 * pid + 1] = 0;
            }
            ++n_cam;
        } else if (pid < P.n_paths) {
            B.L[pid] = make_float4(0, 0, 0, 0);
        }
        // queue 0 is dense (slot == pid); pixel slots outside the sample bounds are INVALID records
        if (pid < P.n_paths) {
            B.ray_o[0][pid] = make_float4(o.x, o.y, o.z, b2f(valid ? pid : kInvalid));
            B.ray_d[0][pid] = make_float4(d.x, d.y, d.z, tmax);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) B.counts[kCntRay] = P.n_paths;
    if (count_stats) flush_counter(&B.counters->camera_rays, n_cam);
}

// ---------------------------------------------------------------------------
// extend: BVHAccel::Intersect for every ray of queue `bounce & 1`; hits are
// appended (ballot-compacted) to the shade queue.
// GEN (first bounce, PassDesc::gen_fused): the queue is the dense range of path ids and a lane makes its camera ray
// itself (what k_generate would have written and this kernel read back: 64 B per path)
template <bool COUNT, bool ALPHA, bool GEN>
__global__ __launch_bounds__(kBlock, IILE_TRAV_WAVES) void k_extend(DScene S, PassDesc P, PassBuffers B, int bounce) {
    __shared__ int lds_stack[kWavesPerBlock][2 * kLdsStackDepth][64];
    const StackRef sr{(lds_int *)&lds_stack[threadIdx.x >> 6][0][threadIdx.x & 63], B.spill,
                      blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock};
    const uint32_t count = B.counts[kCntRay + bounce];
    uint32_t *head = &B.counts[kCntExtHead + bounce];
    const float4 *ro = B.ray_o[bounce & 1], *rd = B.ray_d[bounce & 1];
    TraceStats st = {0, 0, 0, 0};
    unsigned long long n_rays = 0, n_term = 0;
    WaveFeed feed{0, 0, count == 0};
    WaveOut shade_out{0, 0};
    auto pad_shade = [&](uint32_t sl) { B.shade_q[sl] = kInvalid; };
    auto warm = [&](uint32_t first) {
        if (GEN) return;
        warm_plane(ro, first, count);
        warm_plane(rd, first, count);
    };
    Trav t;
    t.have = false;
    t.cur = 0;
    t.sp = 0;
    t.hit_prim = -1;
    bool active = false;
    uint32_t slot = 0;
    float4 gen_d = make_float4(0, 0, 1, 0);  // GEN: the ray direction (the sphere test reads it back)
    while (true) {
        const unsigned long long idle_mask = __ballot(!active);
        if (!feed.exhausted && idle_mask != 0 && (__popcll(idle_mask) >= kRefillIdle || idle_mask == ~0ull)) {
            uint32_t s_new;
            if (feed_take(feed, head, count, !active, &s_new, warm)) {
                slot = s_new;
                if (GEN) {
                    int px = 0, py = 0;
                    uint32_t k = 0;
                    if (path_pixel(S, P, slot, &px, &py, &k)) {  // queue 0 is dense: slot == path id
                        const uint32_t idx = sample_index(S, px, py, k);
                        const float u0 = sample_dimension(S, idx, 0, px, py), u1 = sample_dimension(S, idx, 1, px, py);
                        float l0 = 0, l1 = 0;
                        if (S.lens_radius > 0) {
                            l0 = sample_dimension(S, idx, 3);
                            l1 = sample_dimension(S, idx, 4);
                        }
                        F3 o, d;
                        float tmax;
                        const float pfx = float(px) + u0, pfy = float(py) + u1;
                        flag_whole_film_position(B, slot, px, py, k, pfx, pfy, u0, u1);
                        camera_ray(S, pfx, pfy, l0, l1, &o, &d, &tmax);
                        B.hindex[slot] = idx;
                        // the film position rides in the path's (not yet used) throughput record: the first k_shade
                        // rebuilds the ray from it instead of evaluating the Halton dimensions again
                        reinterpret_cast<float2 *>(&B.beta[slot])[0] = make_float2(pfx, pfy);
                        gen_d = make_float4(d.x, d.y, d.z, tmax);
                        trav_begin<COUNT>(S, t, o, d, tmax, &st);
                        active = true;
                    }
                } else {
                    const float4 o4 = ro[slot], d4 = rd[slot];
                    if (f2b(o4.w) != kInvalid) {
                        trav_begin<COUNT>(S, t, F3{o4.x, o4.y, o4.z}, F3{d4.x, d4.y, d4.z}, d4.w, &st);
                        active = true;
                        if (COUNT) {
                            ++n_rays;
                            if (B.nray_out) B.nray_out[2 * f2b(o4.w)] += 1;
                        }
                    }
                }
            }
        }
        if (__ballot(active) == 0) {
            if (feed.exhausted) break;
            continue;
        }
        // while-while: every lane walks interior records until it stands at a leaf (or is
        // done), then all lanes at a leaf run the primitive tests together. Main-path rays
        // are coherent enough that this beats finer-grained phase scheduling (measured:
        // 96 ms vs 180+ ms per 1080p/64spp step).
#if IILE_FLAT_EXTEND
        // one step per iteration for the whole wavefront, interior or leaf, whichever has more
        // lanes waiting (25.9 ms vs 34.6 ms for strict while-while on the 1080p/64spp step)
        {
            const bool wi = active && t.have && t.cur >= 0;
            const bool wl = active && t.have && t.cur < 0;
            const int n_int = __popcll(__ballot(wi)), n_leaf = __popcll(__ballot(wl));
            if (n_int > 0 && n_int * IILE_VOTE_NUM >= n_leaf * IILE_VOTE_DEN) {
                if (wi) trav_step<COUNT>(S, t, sr, &st);
            } else if (n_leaf > 0) {
                if (wl) trav_leaf<COUNT, ALPHA>(S, t, sr, &st, false, GEN ? &gen_d : &rd[slot]);
            }
        }
#else
        while (active && t.have && t.cur >= 0) trav_step<COUNT>(S, t, sr, &st);
        if (active && t.have) trav_leaf<COUNT, ALPHA>(S, t, sr, &st, false, GEN ? &gen_d : &rd[slot]);
#endif
        const bool fin = active && !t.have;
        const bool is_hit = fin && t.hit_prim >= 0;
        if (fin) {
            B.hits[slot] = make_float4(b2f(uint32_t(hit_index(t.hit_prim))), t.b0, t.b1, t.b2);
            active = false;
            if (COUNT && t.hit_prim < 0) ++n_term;  // the path left the scene: ReportValue(pathLength, bounces)
        }
        const uint32_t pos = out_take(shade_out, &B.counts[kCntShade + bounce], is_hit, pad_shade);
        // entry = queue slot | shading class << 28 (k_shade regroups its block by class)
        if (is_hit) B.shade_q[pos] = slot | (uint32_t(t.hit_prim >> kHitClassShift) & 7u) << kSlotBits;
    }
    out_flush(shade_out, pad_shade);
    if (COUNT) {
        flush_counter(&B.counters->closest_rays, n_rays);
        flush_counter(&B.counters->ext_rays, n_rays);
        flush_counter(&B.counters->ext_nodes, st.nodes);
        flush_counter(&B.counters->ext_tri_tests, st.tris);
        flush_counter(&B.counters->ext_sphere_tests, st.spheres);
        flush_counter(&B.counters->nodes_closest, st.nodes);
        flush_counter(&B.counters->tri_tests, st.tris);
        flush_counter(&B.counters->tri_hits, st.tri_hits);
        flush_counter(&B.counters->sphere_tests, st.spheres);
        flush_counter(&B.counters->path_length[bounce < 7 ? bounce : 7], n_term);
    }
}

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
// 3 waves/SIMD (<= 168 VGPRs, 6 spilled): measured 38.0 ms vs 40.2 ms at 2 waves/SIMD
// TEX: some material takes a parameter from an image texture (implies EXT)
template <bool COUNT, bool EXT, bool TEX>
__global__ __launch_bounds__(kBlock, 3) void k_shade(DScene S, PassDesc P, PassBuffers B, int bounce, uint32_t plane) {
    // digit permutations of the Halton sampler staged in LDS (dynamic shared memory)
    // (or, for SobolSampler, its generator matrices: sobol_column)
    extern __shared__ __attribute__((aligned(16))) uint16_t s_perms_raw[];
    if (S.sobol) {
        for (int i = threadIdx.x; i < S.sobol_dims * 32; i += kBlock) reinterpret_cast<uint32_t *>(s_perms_raw)[i] = S.sobol_mat[i];
    } else {
        for (int i = threadIdx.x; i < S.n_perms; i += kBlock) s_perms_raw[i] = S.perms[i];
    }
    __syncthreads();
    lds_u16 *const s_perms = (lds_u16 *)s_perms_raw;
    const uint32_t count = B.counts[kCntShade + bounce];
    const float4 *ro = B.ray_o[bounce & 1], *rd = B.ray_d[bounce & 1];
    float4 *no = B.ray_o[(bounce + 1) & 1], *nd = B.ray_d[(bounce + 1) & 1];
    unsigned long long n_nee = 0, n_term = 0, n_pdf_tests = 0, n_pdf_hits = 0;
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
        {
            bool emit_nee = false;
            F3 so = F3{0, 0, 0}, sd = F3{0, 0, 1}, mo = F3{0, 0, 0}, md = F3{0, 0, 1};
            F3 A = F3{0, 0, 0}, Bc = F3{0, 0, 0};
            uint32_t nee_flags = 0, nee_light = 0;
            float light_sel_pdf = 1.f;  // lightPdf of UniformSampleOneLight: Ld is divided by it
            if (valid) {
                float4 o4, d4;
                const float4 h4 = B.hits[slot];
                if (bounce == 0 && P.gen_fused) {
                    // the camera ray again, as the first k_extend made it (queue 0 is dense: slot == path id)
                    const float2 cpf = reinterpret_cast<const float2 *>(&B.beta[slot])[0];  // pFilm, left by k_extend
                    float cl0 = 0, cl1 = 0;
                    if (S.lens_radius > 0) {
                        const uint32_t cidx = B.hindex[slot];
                        cl0 = sample_dimension(S, s_perms, cidx, 3);
                        cl1 = sample_dimension(S, s_perms, cidx, 4);
                    }
                    F3 co, cd;
                    float ctm;
                    camera_ray(S, cpf.x, cpf.y, cl0, cl1, &co, &cd, &ctm);
                    o4 = make_float4(co.x, co.y, co.z, b2f(slot));
                    d4 = make_float4(cd.x, cd.y, cd.z, ctm);
                } else {
                    o4 = ro[slot];
                    d4 = rd[slot];
                }
                pid = f2b(o4.w);
                // a path arrives at its first vertex with beta = 1 at sampler dimension 5 (after
                // the camera sample): k_generate does not spend 16 B per path on saying so
                const float4 beta4 = bounce == 0 ? make_float4(1, 1, 1, b2f(5u)) : B.beta[pid];
                beta = F3{beta4.x, beta4.y, beta4.z};
                dim = int(f2b(beta4.w) & 0xffffu);
                const bool prev_specular = EXT && (f2b(beta4.w) >> 16) != 0;  // specularBounce of path.cpp:150
                hidx = B.hindex[pid];
                // The four samples of EstimateDirect (dims dim+1 .. dim+4; dim itself is the
                // 1D sample SampleDiscrete consumes), drawn here while few registers are live.
                // Every path of a bounce normally sits at the same dimension.
                float u_nee[4] = {0, 0, 0, 0};
                if (bounce < S.max_depth) {
                    const int dim_u = __builtin_amdgcn_readfirstlane(dim);
                    sample_dimensions_n<4>(S, s_perms, dim_u + 1, __ballot(dim != dim_u) == 0, dim + 1, hidx, u_nee);
                }
                const int prim = int(f2b(h4.x));
                const F3 ray_o = F3{o4.x, o4.y, o4.z};
                ray_d = F3{d4.x, d4.y, d4.z};
                const float4 v0 = S.tri_verts[3 * size_t(prim)];
                const float4 v1 = S.tri_verts[3 * size_t(prim) + 1];
                const float4 v2 = S.tri_verts[3 * size_t(prim) + 2];
                const uint32_t flags = f2b(v0.w);
                const int material = int(f2b(v1.w)), light = int(f2b(v2.w));
                if (flags & 1u) {
                    // the closest hit was the sphere: redo its (deterministic) root
                    // selection to recover the object-space ray and refined hit point
                    float t;
                    F3 od, ph;
                    const DSphere &sp = S.spheres[S.prim_shape[prim]];
                    sphere_test(sp, ray_o, ray_d, IILE_INF, &t, &od, &ph);
                    sphere_interaction(sp, od, ph, &is);
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
                        if (m0.kd_tex >= 0 || m0.ks_tex >= 0 || m0.kr_tex >= 0 || m0.kt_tex >= 0 || m0.bump_tex >= 0 || m0.rough_tex >= 0 || m0.sigma_tex >= 0) {
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
                    if (n_nonspec(bsdf) > 0) {  // NumComponents(BSDF_ALL & ~BSDF_SPECULAR) > 0, path.cpp:118
                        ++n_nee;
                        // UniformSampleOneLight (integrator.cpp:85-106). One light: it is chosen with pdf 1
                        // (SampleDiscrete still consumes a 1D sample). Several: through the voxel's
                        // distribution of the spatial light distribution (lightdistrib.cpp:134-226,
                        // tabulated at scene creation); a zero pdf returns before any further sample.
                        int li = 0;
                        if (EXT && S.n_lights > 1) {
                            const float ul = sample_dimension(S, s_perms, hidx, dim);
                            li = sample_light(S, is.p, ul, &light_sel_pdf);
                        }
                        if (S.n_lights > 0) ++dim;
                        if (S.n_lights > 0 && light_sel_pdf != 0) {
                            const DLight &lt = S.lights[li];
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
                                                     : sphere_sample(S.spheres[lt.sphere], is, ul0, ul1, &light_pdf);
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
                                // BSDF-sampling half (integrator.cpp:165-213)
                                F3 f2 = bsdf_sample_f(bsdf, is.wo, &wi, us0, us1, &scattering_pdf);
                                f2 = f2 * absdot(wi, is.sn);
                                if (!is_black(f2) && scattering_pdf > 0) {
                                    const float lp = EXT ? shape_pdf(S, lt, is, wi, &n_pdf_tests, &n_pdf_hits)
                                                         : sphere_pdf(S.spheres[lt.sphere], is, wi);
                                    if (lp != 0) {
                                        const float weight = power_heuristic(scattering_pdf, lp);
                                        mo = offset_ray_origin(is.p, is.perr, is.n, wi);
                                        md = wi;
                                        // Li is Lemit when the MIS ray finds this light facing it
                                        Bc = sdiv(f2 * F3{lt.lemit[0], lt.lemit[1], lt.lemit[2]} * weight, scattering_pdf);
                                        // The ray only matters if its closest hit is the sampled light (integrator.cpp:205-209),
                                        // and Sphere::Pdf is the cone's pdf for ANY direction (sphere.cpp:294-306): most of these
                                        // rays point away from the light. The traversal would run Sphere::Intersect on this very
                                        // ray with some tMax <= inf, and every rejection of that test that depends on tMax only
                                        // gets stricter as tMax shrinks (t0.hi > tMax, ts.hi > tMax): a ray the sphere test
                                        // rejects at tMax = inf can never end on the light, whatever else it hits. Those rays are
                                        // not traced by the uninstrumented kernels (the instrumented build traces them all: the
                                        // reference's ray counters are part of parity). Triangle emitters: Shape::Pdf has already
                                        // intersected the triangle with this ray (lp == 0 on a miss).
                                        bool can_reach = true;
                                        if (!COUNT && lt.type == kLightDiffuseArea) {
                                            float t_l;
                                            F3 od_l, ph_l;
                                            can_reach = sphere_test(S.spheres[lt.sphere], mo, md, IILE_INF, &t_l, &od_l, &ph_l);
                                        }
                                        if (can_reach) nee_flags |= NEE_HAS_MIS;
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
            const uint32_t eslot = out_take(nee_out, &B.counts[kCntNee + bounce], emit_nee, pad_nee);
            // the MIS rays go to a dense queue of their own (planes 2 and 3): most records have none
            const bool emit_mis = emit_nee && (nee_flags & NEE_HAS_MIS) != 0;
            const uint32_t mslot = out_take(mis_out, &B.counts[kCntMis + bounce], emit_mis, pad_mis);
            if (emit_nee) {
                B.nee[eslot] = make_float4(so.x, so.y, so.z, light_sel_pdf);
                B.nee[plane + eslot] = make_float4(sd.x, sd.y, sd.z, b2f(nee_flags));
                // flags / light / pid are repeated in the planes each consumer streams anyway
                B.nee[4 * plane + eslot] = make_float4(A.x, A.y, A.z, b2f(nee_flags));
                B.nee[5 * plane + eslot] = make_float4(Bc.x, Bc.y, Bc.z, b2f(nee_light));
                B.nee[6 * plane + eslot] = make_float4(beta.x, beta.y, beta.z, b2f(pid));  // beta before this bounce
            }
            if (emit_mis) {
                B.nee[2 * plane + mslot] = make_float4(mo.x, mo.y, mo.z, b2f(eslot));  // + the record it belongs to
                B.nee[3 * plane + mslot] = make_float4(md.x, md.y, md.z, b2f(nee_light));
            }
        }
        F3 next_o = F3{0, 0, 0}, next_d = F3{0, 0, 1};
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
                        const float eta = bsdf.eta;
                        eta_scale *= (dot(-ray_d, is.n) > 0) ? (eta * eta) : 1 / (eta * eta);
                    }
                    // Russian roulette on rrBeta = beta * etaScale (path.cpp:182-190)
                    const F3 rr_beta = beta * eta_scale;
                    const float mc = max3(rr_beta.x, rr_beta.y, rr_beta.z);
                    if (mc < S.rr_threshold && bounce > 3) {
                        const float q = mx(.05f, 1 - mc);
                        const float ur = sample_dimension(S, s_perms, hidx, dim);
                        ++dim;
                        if (ur < q)
                            alive = false;
                        else
                            beta = sdiv(beta, 1 - q);
                    }
                }
            }
            if (EXT && alive && S.has_glass) B.eta_scale[pid] = eta_scale;
            // sampler dimension | specularBounce << 16
            if (alive) B.beta[pid] = make_float4(beta.x, beta.y, beta.z, b2f(uint32_t(dim) | (sampled_specular ? 0x10000u : 0u)));
        }
        // ReportValue(pathLength, bounces): a path that ends in this iteration leaves the
        // loop with bounces == bounce (not counted on the early `return L`)
        if (COUNT && valid && !alive && !returned_early) ++n_term;
        const uint32_t nslot = out_take(ray_out, &B.counts[kCntRay + bounce + 1], alive, pad_ray);
        if (alive) {
            no[nslot] = make_float4(next_o.x, next_o.y, next_o.z, b2f(pid));
            nd[nslot] = make_float4(next_d.x, next_d.y, next_d.z, IILE_INF);
        }
      }
        __builtin_amdgcn_wave_barrier();  // the next chunk overwrites s_entry
    }
    out_flush(ray_out, pad_ray);
    out_flush(nee_out, pad_nee);
    out_flush(mis_out, pad_mis);
    if (COUNT) {
        flush_counter(&B.counters->nee_evals, n_nee);
        flush_counter(&B.counters->path_length[bounce < 7 ? bounce : 7], n_term);
        flush_counter(&B.counters->tri_tests, n_pdf_tests);  // Triangle::Intersect calls of Shape::Pdf
        flush_counter(&B.counters->tri_hits, n_pdf_hits);
    }
}

// ---------------------------------------------------------------------------
// NEE resolution: two homogeneous kernels over the NEE records of one bounce.
//   k_mis     BVHAccel::Intersect for the MIS ray; records which emitter (if any) it ended on
//   k_shadow  BVHAccel::IntersectP for the shadow ray, then L += beta * Ld
// Both finish a record with a single store, so no load ever stalls their loops.
// (One fused kernel walking each record through both rays measured 60 ms per
// 1080p/64spp step against 7 + 17.5 + 26.4 ms for its parts: any-hit and
// closest-hit lanes in one wavefront keep each other waiting.)


template <bool COUNT, bool ALPHA>
__global__ __launch_bounds__(kBlock, IILE_TRAV_WAVES) void k_shadow(DScene S, PassBuffers B, int bounce, uint32_t plane) {
    __shared__ int lds_stack[kWavesPerBlock][2 * kLdsStackDepth][64];
    const StackRef sr{(lds_int *)&lds_stack[threadIdx.x >> 6][0][threadIdx.x & 63], B.spill,
                      blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock};
    const uint32_t count = B.counts[kCntNee + bounce];
    uint32_t *head = &B.counts[kCntConHead + bounce];
    TraceStats st = {0, 0, 0, 0};
    unsigned long long n_shadow = 0, n_zero = 0;
    WaveFeed feed{0, 0, count == 0};
    Trav t;
    t.have = false;
    t.cur = 0;
    t.sp = 0;
    t.hit_prim = -1;
    bool active = false, occluded = false;
    uint32_t e = 0, pid = 0;
    // The record's two possible outcomes, L + beta * Ld with and without the light sample, are
    // formed when the lane takes the record (its loads ride along with the shadow ray's), so a
    // lane that finishes only stores one of them: no load inside the traversal loop, and no
    // separate pass over the records (a streaming resolve kernel cost 4.9 ms per frame).
    // (L itself is only *consumed* at the store, so its load — the one scattered access of the
    // record — overlaps the ray's first traversal steps instead of holding up the refill.)
    F3 L_old = F3{0, 0, 0}, add_unoccluded = F3{0, 0, 0}, add_occluded = F3{0, 0, 0};
    while (true) {
        const unsigned long long idle_mask = __ballot(!active);
        if (!feed.exhausted && idle_mask != 0 && (__popcll(idle_mask) >= kRefillIdle || idle_mask == ~0ull)) {
            uint32_t e_new;
            if (feed_take(feed, head, count, !active, &e_new, [&](uint32_t first) {
                    warm_plane(B.nee, first, count);
                    warm_plane(B.nee + plane, first, count);
                    warm_plane(B.nee + 4 * size_t(plane), first, count);
                    warm_plane(B.nee + 6 * size_t(plane), first, count);
                })) {
                e = e_new;
                const float4 n1 = B.nee[plane + e];
                const uint32_t flags = f2b(n1.w);
                if (flags != kInvalid) {
                    // Ld = [light sample unoccluded] A + [MIS ray reached the sampled light, facing it] B,
                    // L += beta * Ld (integrator.cpp:150-158, 205-211; path.cpp:123-128)
                    const float4 n0 = B.nee[e], a4 = B.nee[4 * size_t(plane) + e], be = B.nee[6 * size_t(plane) + e];
                    const bool lit = (flags & NEE_HAS_MIS) && B.nee_mis[e] != 0;
                    pid = f2b(be.w);
                    const float4 L4 = B.L[pid];
                    const bool has_shadow = (flags & NEE_HAS_SHADOW) != 0;
                    F3 Ld_u = F3{0, 0, 0}, Ld_o = F3{0, 0, 0};
                    if (has_shadow) Ld_u = Ld_u + F3{a4.x, a4.y, a4.z};
                    if (lit) {
                        const float4 b4 = B.nee[5 * size_t(plane) + e];
                        Ld_u = Ld_u + F3{b4.x, b4.y, b4.z};
                        Ld_o = Ld_o + F3{b4.x, b4.y, b4.z};
                    }
                    // UniformSampleOneLight returns EstimateDirect / lightPdf (n0.w; 1 with a single light)
                    const F3 beta = F3{be.x, be.y, be.z};
                    add_unoccluded = beta * sdiv(Ld_u, n0.w);
                    add_occluded = beta * sdiv(Ld_o, n0.w);
                    L_old = F3{L4.x, L4.y, L4.z};
                    if (has_shadow) {
                        trav_begin<COUNT>(S, t, F3{n0.x, n0.y, n0.z}, F3{n1.x, n1.y, n1.z}, 1 - kShadowEpsilon, &st);
                        active = true;
                        occluded = false;
                        if (COUNT) {
                            ++n_shadow;
                            if (B.nray_out) B.nray_out[2 * pid + 1] += 1;
                        }
                    } else {  // no light sample to test: the record is complete
                        const F3 Ln = L_old + add_occluded;
                        B.L[pid] = make_float4(Ln.x, Ln.y, Ln.z, 0);
                        if (COUNT && is_black(add_occluded)) ++n_zero;
                    }
                }
            }
        }
        if (__ballot(active) == 0) {
            if (feed.exhausted) break;
            continue;
        }
#if IILE_FLAT_SHADOW
        // Shadow rays end at their first hit, so lanes leave at very different times: one
        // step per iteration, interior or leaf, whichever keeps more lanes busy
        // (17.9 ms vs 22.1 ms for strict while-while on the 1080p/64spp step).
        {
            const bool wi = active && t.have && t.cur >= 0;
            const bool wl = active && t.have && t.cur < 0;
            const int n_int = __popcll(__ballot(wi)), n_leaf = __popcll(__ballot(wl));
            if (n_int > 0 && n_int * IILE_VOTE_NUM >= n_leaf * IILE_VOTE_DEN) {
                if (wi) trav_step<COUNT>(S, t, sr, &st);
            } else if (n_leaf > 0) {
                if (wl && trav_leaf<COUNT, ALPHA>(S, t, sr, &st, true, &B.nee[plane + e])) occluded = true;
            }
        }
#else
        while (active && t.have && t.cur >= 0) trav_step<COUNT>(S, t, sr, &st);
        if (active && t.have && trav_leaf<COUNT, ALPHA>(S, t, sr, &st, true, &B.nee[plane + e])) occluded = true;
#endif
        if (active && !t.have) {
            const F3 add = occluded ? add_occluded : add_unoccluded;
            const F3 Ln = L_old + add;  // store only: nothing is loaded here
            B.L[pid] = make_float4(Ln.x, Ln.y, Ln.z, 0);
            if (COUNT && is_black(add)) ++n_zero;
            active = false;
        }
    }
    if (COUNT) {
        flush_counter(&B.counters->shadow_rays, n_shadow);
        flush_counter(&B.counters->zero_radiance, n_zero);
        flush_counter(&B.counters->nodes_any, st.nodes);
        flush_counter(&B.counters->any_tri_tests, st.tris);
        flush_counter(&B.counters->tri_tests, st.tris);
        flush_counter(&B.counters->tri_hits, st.tri_hits);
        flush_counter(&B.counters->sphere_tests, st.spheres);
    }
}

template <bool COUNT, bool ALPHA>
__global__ __launch_bounds__(kBlock, IILE_TRAV_WAVES) void k_mis(DScene S, PassBuffers B, int bounce, uint32_t plane) {
    __shared__ int lds_stack[kWavesPerBlock][2 * kLdsStackDepth][64];
    const StackRef sr{(lds_int *)&lds_stack[threadIdx.x >> 6][0][threadIdx.x & 63], B.spill,
                      blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock};
    // the dense queue of MIS rays k_shade wrote beside the NEE records: (o, record) in plane 2, (d, light) in plane 3
    const uint32_t count = B.counts[kCntMis + bounce];
    uint32_t *head = &B.counts[kCntMisHead + bounce];
    TraceStats st = {0, 0, 0, 0};
    unsigned long long n_closest = 0, n_traced = 0;
    WaveFeed feed{0, 0, count == 0};
    Trav t;
    t.have = false;
    t.cur = 0;
    t.sp = 0;
    t.hit_prim = -1;
    bool active = false;
    uint32_t q = 0, e = 0;
    while (true) {
        const unsigned long long idle_mask = __ballot(!active);
        if (!feed.exhausted && idle_mask != 0 && (__popcll(idle_mask) >= kRefillIdle || idle_mask == ~0ull)) {
            uint32_t q_new;
            if (feed_take(feed, head, count, !active, &q_new, [&](uint32_t first) {
                    warm_plane(B.nee + 2 * size_t(plane), first, count);
                    warm_plane(B.nee + 3 * size_t(plane), first, count);
                })) {
                q = q_new;
                const float4 n2 = B.nee[2 * size_t(plane) + q];
                if (f2b(n2.w) != kInvalid) {
                    const float4 n3 = B.nee[3 * size_t(plane) + q];
                    e = f2b(n2.w);
                    trav_begin<COUNT>(S, t, F3{n2.x, n2.y, n2.z}, F3{n3.x, n3.y, n3.z}, IILE_INF, &st);
                    active = true;
                    ++n_traced;
                    if (COUNT) {
                        ++n_closest;
                        if (B.nray_out) B.nray_out[2 * f2b(B.nee[6 * size_t(plane) + e].w)] += 1;
                    }
                }
            }
        }
        if (__ballot(active) == 0) {
            if (feed.exhausted) break;
            continue;
        }
#if IILE_FLAT_MIS
        {
            const bool wi = active && t.have && t.cur >= 0;
            const bool wl = active && t.have && t.cur < 0;
            const int n_int = __popcll(__ballot(wi)), n_leaf = __popcll(__ballot(wl));
            if (n_int > 0 && n_int * IILE_VOTE_NUM >= n_leaf * IILE_VOTE_DEN) {
                if (wi) trav_step<COUNT>(S, t, sr, &st);
            } else if (n_leaf > 0) {
                if (wl) trav_leaf<COUNT, ALPHA>(S, t, sr, &st, false, &B.nee[3 * size_t(plane) + q]);
            }
        }
#else
        while (active && t.have && t.cur >= 0) trav_step<COUNT>(S, t, sr, &st);
        if (active && t.have) trav_leaf<COUNT, ALPHA>(S, t, sr, &st, false, &B.nee[3 * size_t(plane) + q]);
#endif
        if (active && !t.have) {
            // store only: (area light index + 1) of the primitive the MIS ray ended on, 0 for none
            // 255: the ray escaped (matters to an infinite light only)
            const uint8_t on_light = uint8_t(t.hit_prim < 0 ? (S.has_infinite ? 255 : 0) : (t.hit_prim >> kHitLightShift));
            B.nee_mis[e] = on_light;
            // the rare ray that ends on an emitter leaves its hit for k_mis_lit (the hit records are
            // idle between shade and the next extend)
            if (on_light && t.hit_prim >= 0) B.hits[e] = make_float4(b2f(uint32_t(hit_index(t.hit_prim))), t.b0, t.b1, t.b2);
            active = false;
        }
    }
    flush_counter(&B.counters->mis_traced, n_traced);
    if (COUNT) {
        flush_counter(&B.counters->closest_rays, n_closest);
        flush_counter(&B.counters->nodes_closest, st.nodes);
        flush_counter(&B.counters->tri_tests, st.tris);
        flush_counter(&B.counters->tri_hits, st.tri_hits);
        flush_counter(&B.counters->sphere_tests, st.spheres);
    }
}

// miss: a ray that left the scene at the first vertex or after a specular bounce picks up the
// infinite lights' radiance (path.cpp:91-99). Only launched for scenes that have one: a pass over
// the hit records of the bounce, so the traversal kernels stay as they are.
__global__ __launch_bounds__(kBlock) void k_miss(DScene S, PassBuffers B, int bounce) {
    const uint32_t count = B.counts[kCntRay + bounce];
    const float4 *ro = B.ray_o[bounce & 1], *rd = B.ray_d[bounce & 1];
    for (uint32_t slot = blockIdx.x * kBlock + threadIdx.x; slot < count; slot += gridDim.x * kBlock) {
        const float4 h4 = B.hits[slot];
        if (int(f2b(h4.x)) >= 0) continue;
        const uint32_t pid = f2b(ro[slot].w);
        if (pid == kInvalid) continue;
        const float4 beta4 = bounce == 0 ? make_float4(1, 1, 1, b2f(5u)) : B.beta[pid];
        if (!((bounce == 0 && !S.probe_mode) || (bounce != 0 && (f2b(beta4.w) >> 16) != 0))) continue;  // iispt_d.cpp:124-131
        const float4 d4 = rd[slot];
        const F3 beta = F3{beta4.x, beta4.y, beta4.z}, d = F3{d4.x, d4.y, d4.z};
        const float4 L4 = B.L[pid];
        F3 L = F3{L4.x, L4.y, L4.z};
        for (int l = 0; l < S.n_lights; ++l)
            if (S.lights[l].type == kLightInfinite) L = L + beta * inf_le(S, S.lights[l], d);
        B.L[pid] = make_float4(L.x, L.y, L.z, 0);
    }
}

// mis_lit: k_mis leaves, per record, the (area light index + 1) of the primitive its MIS ray
// ended on. This pass over that byte plane turns it into "the ray reached the *sampled* light,
// on its emitting side" (`lightIsect.primitive->GetAreaLight() == &light`, then
// SurfaceInteraction::Le -> DiffuseAreaLight::L; integrator.cpp:205-209). Only the rare records
// whose ray did end on an emitter are looked at any further.
__global__ __launch_bounds__(kBlock) void k_mis_lit(DScene S, PassBuffers B, int bounce, uint32_t plane) {
    const uint32_t count = B.counts[kCntMis + bounce];
    for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < count; q += gridDim.x * kBlock) {
        const float4 n2 = B.nee[2 * size_t(plane) + q];
        const uint32_t e = f2b(n2.w);  // the NEE record of this MIS ray
        if (e == kInvalid) continue;
        const uint32_t mis = B.nee_mis[e];
        if (mis == 0) continue;
        const float4 n3 = B.nee[3 * size_t(plane) + q];
        bool lit = false;
        {
            const int li = int(f2b(n3.w));
            if (mis == 255u) {
                lit = S.lights[li].type == kLightInfinite;  // `else Li = light.Le(ray)`, integrator.cpp:209-210
            } else if (int(mis) == li + 1) {
                const DLight &lt = S.lights[li];
                const F3 mo = F3{n2.x, n2.y, n2.z}, md = F3{n3.x, n3.y, n3.z};
                Isect lis;
                if (lt.type == kLightAreaTriangle) {
                    const float4 h4 = B.hits[e];  // left by k_mis
                    const int prim = int(f2b(h4.x));
                    const float4 v0 = S.tri_verts[3 * size_t(prim)], v1 = S.tri_verts[3 * size_t(prim) + 1],
                                 v2 = S.tri_verts[3 * size_t(prim) + 2];
                    triangle_interaction(S, prim, f2b(v0.w), F3{v0.x, v0.y, v0.z}, F3{v1.x, v1.y, v1.z},
                                         F3{v2.x, v2.y, v2.z}, md, h4.y, h4.z, h4.w, &lis);
                } else {
                    const DSphere &sp = S.spheres[lt.sphere];
                    float th;
                    F3 od, ph;
                    // the closest hit was this sphere: redo its root selection for the hit point
                    sphere_test(sp, mo, md, IILE_INF, &th, &od, &ph);
                    sphere_interaction(sp, od, ph, &lis);
                }
                lit = lt.two_sided || dot(lis.n, -md) > 0;
            }
        }
        B.nee_mis[e] = lit ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------
// radiance guards of SamplerIntegrator::Render (integrator.cpp:293-314) and the
// luminance clamp of FilmTile::AddSample (film.h:157-158)
DEV F3 guard_radiance(const DScene &S, F3 L) {
    const float y = lum_y(L);
    if (is_nan(L.x) || is_nan(L.y) || is_nan(L.z))
        L = F3{0, 0, 0};
    else if (double(y) < -1e-5)
        L = F3{0, 0, 0};
    else if (is_inf(y))
        L = F3{0, 0, 0};
    const float y2 = lum_y(L);
    if (y2 > S.max_sample_luminance) L = L * (S.max_sample_luminance / y2);
    return L;
}

// film_accumulate: one thread per owned pixel adds this pass's samples, in
// sample order, to the pixel's RGB contribSum (FilmTile::AddSample, film.h:153-193
// with the box filter: weight 1 for the pixel containing pFilm).
__global__ __launch_bounds__(kBlock) void k_film_accumulate(DScene S, PassDesc P, PassBuffers B, FilmBuffers F) {
    const uint32_t n_pix = uint32_t(P.n_pass_tiles) * 256u;
    for (uint32_t lpt = blockIdx.x * kBlock + threadIdx.x; lpt < n_pix; lpt += gridDim.x * kBlock) {
        int px, py;
        uint32_t k;
        const uint32_t pid0 = lpt * uint32_t(P.kc);
        if (!path_pixel(S, P, pid0, &px, &py, &k)) continue;
        const uint32_t pt = uint32_t(P.slot0) * 256u + lpt;  // the pixel's slot among all owned tiles
        float4 acc = F.tile_rgbw[pt];
        // A lane's samples are consecutive in memory (1 KB apart from its neighbour's at 64 spp):
        // eight loads = one whole 128-byte line are issued together, then summed in sample order.
        for (int k8 = 0; k8 < P.kc; k8 += 8) {
            float4 Lb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) Lb[j] = B.L[pid0 + uint32_t(k8 + j < P.kc ? k8 + j : k8)];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = k8 + j;
                if (kk >= P.kc) break;
                const F3 L = guard_radiance(S, F3{Lb[j].x, Lb[j].y, Lb[j].z});
                acc.x += L.x;
                acc.y += L.y;
                acc.z += L.z;
                acc.w += 1.f;
                if (P.k0 + kk == 0) {
                    // a sample whose fractional film offset is exactly 0 also lands in the
                    // left / upper neighbour (support ceil(pd-.5) .. floor(pd+.5))
                    const uint32_t idx = B.hindex[pid0 + kk];
                    uint32_t mask = 0;
                    if (sample_dimension(S, idx, 0, px, py) == 0.f) mask |= 1u;
                    if (sample_dimension(S, idx, 1, px, py) == 0.f) mask |= 2u;
                    F.k0_rgbv[pt] = make_float4(L.x, L.y, L.z, b2f(mask));
                }
            }
        }
        F.tile_rgbw[pt] = acc;
    }
}

// Filters wider than a pixel (gaussian, mitchell, sinc, triangle, larger boxes). A pixel then sums samples of
// neighbouring pixels and tiles, in the order FilmTile::AddSample / MergeFilmTile give: inside a tile in pixel
// (row-major) then sample order, tiles in index order. The neighbouring pixels may belong to tiles of another pass, so
// the frame's samples are kept — k_film_store, 24 B each — and summed once at the end by a gather per film pixel
// (k_film_gather).
__global__ __launch_bounds__(kBlock) void k_film_store(DScene S, PassDesc P, PassBuffers B, FilmBuffers F, int k_begin, int n_samples) {
    for (uint32_t pid = blockIdx.x * kBlock + threadIdx.x; pid < P.n_paths; pid += gridDim.x * kBlock) {
        int px, py;
        uint32_t k;
        if (!path_pixel(S, P, pid, &px, &py, &k)) continue;
        const float4 L4 = B.L[pid];
        const F3 L = guard_radiance(S, F3{L4.x, L4.y, L4.z});
        const uint32_t idx = B.hindex[pid];
        // [tile slot][k][pixel of the tile]: the gather's lanes (neighbouring film pixels) read neighbouring records
        const uint32_t pt = uint32_t(P.slot0) * 256u + pid / uint32_t(P.kc);
        const size_t at = (size_t(pt >> 8) * size_t(n_samples) + size_t(int(k) - k_begin)) * 256u + size_t(pt & 255u);
        F.wide_L[at] = make_float4(L.x, L.y, L.z, 0.f);
        F.wide_pf[at] = make_float2(float(px) + sample_dimension(S, idx, 0, px, py), float(py) + sample_dimension(S, idx, 1, px, py));
    }
}

DEV bool tile_owned(const PassDesc &P, int tx, int ty, uint32_t *slot) {
    if (tx < 0 || ty < 0 || tx >= P.n_tiles_x || ty >= P.n_tiles_y) return false;
    const int t = ty * P.n_tiles_x + tx;
    const int s = P.slot_of_tile ? P.slot_of_tile[t] : t;
    if (s < 0) return false;
    *slot = uint32_t(s);
    return true;
}
DEV void add_xyz(float4 *out, float r, float g, float b, float w) {  // RGBToXYZ, spectrum.h:62-66
    out->x += 0.412453f * r + 0.357580f * g + 0.180423f * b;
    out->y += 0.212671f * r + 0.715160f * g + 0.072169f * b;
    out->z += 0.019334f * r + 0.119193f * g + 0.950227f * b;
    out->w += w;
}

// Probe pass: film + aux images of every probe. Film::to_rgb_array (film.cpp:187-225) on the gathered {X, Y, Z, w},
// and the first hits' normals / distances picked from the paths (1 spp): outputs [probe][y][x].
__global__ __launch_bounds__(kBlock) void k_probe_finish(DScene S, PassDesc P, PassBuffers B, FilmBuffers F, int n_probes, float *intensity,
                                                         float *normals, float *distance) {
    const int fw = S.crop_x1 - S.crop_x0, fh = S.crop_y1 - S.crop_y0;
    const uint32_t per = uint32_t(fw) * uint32_t(fh), n = per * uint32_t(n_probes);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 px = F.film_xyzw[i];
        float rgb[3];
        rgb[0] = 3.240479f * px.x - 1.537150f * px.y - 0.498535f * px.z;  // XYZToRGB, spectrum.h:56-60
        rgb[1] = -0.969256f * px.x + 1.875991f * px.y + 0.041556f * px.z;
        rgb[2] = 0.055648f * px.x - 0.204043f * px.y + 1.057311f * px.z;
        if (px.w != 0) {
            const float inv_wt = 1.f / px.w;
            for (int c = 0; c < 3; ++c) rgb[c] = mx(0.f, rgb[c] * inv_wt);
        }
        for (int c = 0; c < 3; ++c) intensity[3 * size_t(i) + c] = (rgb[c] + 0.f) * 1.f;
        const uint32_t probe = i / per, loc = i % per;
        const int x = S.crop_x0 + int(loc % uint32_t(fw)), y = S.crop_y0 + int(loc / uint32_t(fw));
        const int tx = (x - S.samp_x0) / kTile, ty = (y - S.samp_y0) / kTile;
        const uint32_t slot = probe * uint32_t(P.probe_tiles) + uint32_t(ty * P.n_tiles_x + tx);
        const uint32_t pix = uint32_t((y - S.samp_y0 - ty * kTile) * kTile + (x - S.samp_x0 - tx * kTile));
        const float4 a = B.aux[(size_t(slot) * 256u + pix) * size_t(P.kc)];
        normals[3 * size_t(i)] = a.x;
        normals[3 * size_t(i) + 1] = a.y;
        normals[3 * size_t(i) + 2] = a.z;
        distance[i] = a.w;
    }
}

__global__ __launch_bounds__(kBlock) void k_film_gather(DScene S, PassDesc P, FilmBuffers F, int n_samples) {
    __shared__ float s_table[256];
    for (int j = threadIdx.x; j < 256; j += kBlock) s_table[j] = S.filter_table[j];
    __syncthreads();
    const int fw = S.crop_x1 - S.crop_x0, fh = S.crop_y1 - S.crop_y0;
    const uint32_t per = uint32_t(fw) * uint32_t(fh);
    const uint32_t n = per * (P.probe_mode ? uint32_t(P.n_owned_tiles / P.probe_tiles) : 1u);  // one film per probe
    const float rx = S.filter_rx, ry = S.filter_ry;
    const float inv_rx = 1 / rx, inv_ry = 1 / ry;  // Filter::invRadius
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t probe = i / per, loc = i % per;
        const int x = S.crop_x0 + int(loc % uint32_t(fw)), y = S.crop_y0 + int(loc / uint32_t(fw));
        // sample pixels that can reach (x, y): |q + u - 0.5 - x| <= r with u in [0, 1), i.e. x - r - 0.5 < q <= x + r + 0.5
        // (floor / ceil keep a pixel of slack on either side against the rounding of the sums in AddSample)
        // (a probe's RenderView takes no samples outside the film's pixel bounds: those records do not exist)
        const int lo_x = P.probe_mode ? S.crop_x0 : S.samp_x0, hi_x = P.probe_mode ? S.crop_x1 : S.samp_x1;
        const int lo_y = P.probe_mode ? S.crop_y0 : S.samp_y0, hi_y = P.probe_mode ? S.crop_y1 : S.samp_y1;
        const int qx0 = max(int(floorf(float(x) - rx - 0.5f)), lo_x), qx1 = min(int(ceilf(float(x) + rx + 0.5f)), hi_x - 1);
        const int qy0 = max(int(floorf(float(y) - ry - 0.5f)), lo_y), qy1 = min(int(ceilf(float(y) + ry + 0.5f)), hi_y - 1);
        float4 out = make_float4(0, 0, 0, 0);
        if (qx0 <= qx1 && qy0 <= qy1) {
            const int tx0 = (qx0 - S.samp_x0) / kTile, tx1 = (qx1 - S.samp_x0) / kTile;
            const int ty0 = (qy0 - S.samp_y0) / kTile, ty1 = (qy1 - S.samp_y0) / kTile;
            for (int ty = ty0; ty <= ty1; ++ty)
                for (int tx = tx0; tx <= tx1; ++tx) {  // tile index order
                    uint32_t slot;
                    if (P.probe_mode)
                        slot = probe * uint32_t(P.probe_tiles) + uint32_t(ty * P.n_tiles_x + tx);
                    else if (!tile_owned(P, tx, ty, &slot))
                        continue;
                    // the tile's FilmTile (Film::GetFilmTile, film.cpp:92-103) must hold (x, y)
                    const int sx0 = S.samp_x0 + tx * kTile, sy0 = S.samp_y0 + ty * kTile;
                    const int sx1 = min(sx0 + kTile, S.samp_x1), sy1 = min(sy0 + kTile, S.samp_y1);
                    const int fx0 = max(int(ceilf(float(sx0) - 0.5f - rx)), S.crop_x0), fx1 = min(int(floorf(float(sx1) - 0.5f + rx)) + 1, S.crop_x1);
                    const int fy0 = max(int(ceilf(float(sy0) - 0.5f - ry)), S.crop_y0), fy1 = min(int(floorf(float(sy1) - 0.5f + ry)) + 1, S.crop_y1);
                    if (x < fx0 || x >= fx1 || y < fy0 || y >= fy1) continue;
                    float r = 0, g = 0, b = 0, w = 0;
                    // FilmTile::AddSample's support test and table lookup for this pixel (film.h:159-188)
                    auto add_sample = [&](float2 pf, float4 L) {
                        const float dxf = pf.x - 0.5f, dyf = pf.y - 0.5f;
                        const bool in = x >= max(int(ceilf(dxf - rx)), fx0) && x < min(int(floorf(dxf + rx)) + 1, fx1) &&
                                        y >= max(int(ceilf(dyf - ry)), fy0) && y < min(int(floorf(dyf + ry)) + 1, fy1);
                        if (in) {
                            const float ffx = fabsf((float(x) - dxf) * inv_rx * 16.f), ffy = fabsf((float(y) - dyf) * inv_ry * 16.f);
                            const int ifx = min(int(floorf(ffx)), 15), ify = min(int(floorf(ffy)), 15);
                            const float fwt = s_table[ify * 16 + ifx];
                            r += L.x * 1.f * fwt;
                            g += L.y * 1.f * fwt;
                            b += L.z * 1.f * fwt;
                            w += fwt;
                        }
                    };
                    const int xa = max(qx0, sx0), xb = min(qx1, sx1 - 1);
                    for (int qy = max(qy0, sy0); qy <= min(qy1, sy1 - 1); ++qy) {
                        const size_t row = size_t(slot) * size_t(n_samples) * 256u + size_t((qy - sy0) * kTile - sx0);
                        if (n_samples == 1) {
                            // one sample per pixel (the probe pass): four neighbouring pixels' records in flight
                            for (int q4 = xa; q4 <= xb; q4 += 4) {
                                float2 pfb[4];
                                float4 Lb[4];
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj) {
                                    const size_t at = row + size_t(q4 + jj <= xb ? q4 + jj : q4);
                                    pfb[jj] = F.wide_pf[at];
                                    Lb[jj] = F.wide_L[at];
                                }
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj)
                                    if (q4 + jj <= xb) add_sample(pfb[jj], Lb[jj]);
                            }
                            continue;
                        }
                        for (int qx = xa; qx <= xb; ++qx) {
                            const size_t base = row + size_t(qx);
                            for (int k4 = 0; k4 < n_samples; k4 += 4) {
                                // four records in flight, then summed in sample order
                                float2 pfb[4];
                                float4 Lb[4];
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj) {
                                    const size_t at = base + size_t(k4 + jj < n_samples ? k4 + jj : k4) * 256u;
                                    pfb[jj] = F.wide_pf[at];
                                    Lb[jj] = F.wide_L[at];
                                }
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj)
                                    if (k4 + jj < n_samples) add_sample(pfb[jj], Lb[jj]);
                            }
                        }
                    }
                    add_xyz(&out, r, g, b, w);
                }
        }
        F.film_xyzw[i] = out;
    }
}

// film_resolve: Film::MergeFilmTile (film.cpp:135-148) as a gather. For film
// pixel Q the contributions are grouped by the tile whose FilmTile holds them
// (Q's own tile, then the tiles right / below / diagonal whose k=0 samples
// splat onto Q), each group summed in RGB in pixel order, converted to XYZ and
// added in tile-index order.
__global__ __launch_bounds__(kBlock) void k_film_resolve(DScene S, PassDesc P, FilmBuffers F) {
    const int fw = S.crop_x1 - S.crop_x0, fh = S.crop_y1 - S.crop_y0;
    const uint32_t n = uint32_t(fw) * uint32_t(fh);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int qx = S.crop_x0 + int(i % uint32_t(fw)), qy = S.crop_y0 + int(i / uint32_t(fw));
        float4 out = make_float4(0, 0, 0, 0);
        const int lx = qx - S.samp_x0, ly = qy - S.samp_y0;
        const bool q_in = lx >= 0 && ly >= 0 && qx < S.samp_x1 && qy < S.samp_y1;
        const int tx0 = lx >= 0 ? lx / kTile : -1, ty0 = ly >= 0 ? ly / kTile : -1;
        const int tx1 = (lx + 1) >= 0 ? (lx + 1) / kTile : -1, ty1 = (ly + 1) >= 0 ? (ly + 1) / kTile : -1;
        // sources: S1 = (qx+1, qy) needs mask bit0; S2 = (qx, qy+1) bit1; S3 = (qx+1, qy+1) both
        auto source = [&](int sx, int sy, int tx, int ty, uint32_t need, float *r, float *g, float *b, float *w) {
            if (sx >= S.samp_x1 || sy >= S.samp_y1 || sx < S.samp_x0 || sy < S.samp_y0) return;
            uint32_t slot;
            if (!tile_owned(P, tx, ty, &slot)) return;
            const uint32_t pix = uint32_t((sy - S.samp_y0 - ty * kTile) * kTile + (sx - S.samp_x0 - tx * kTile));
            const float4 v = F.k0_rgbv[slot * 256u + pix];
            if ((f2b(v.w) & need) != need) return;
            *r += v.x;
            *g += v.y;
            *b += v.z;
            *w += 1.f;
        };
        // group 0: Q's own tile
        {
            uint32_t slot;
            float r = 0, g = 0, b = 0, w = 0;
            bool any = false;
            if (q_in && tile_owned(P, tx0, ty0, &slot)) {
                const float4 own = F.tile_rgbw[slot * 256u + uint32_t((ly - ty0 * kTile) * kTile + (lx - tx0 * kTile))];
                r = own.x;
                g = own.y;
                b = own.z;
                w = own.w;
                any = true;
                if (tx1 == tx0) source(qx + 1, qy, tx0, ty0, 1u, &r, &g, &b, &w);
                if (ty1 == ty0) source(qx, qy + 1, tx0, ty0, 2u, &r, &g, &b, &w);
                if (tx1 == tx0 && ty1 == ty0) source(qx + 1, qy + 1, tx0, ty0, 3u, &r, &g, &b, &w);
            }
            if (any) add_xyz(&out, r, g, b, w);
        }
        // group 1: tile to the right in the same tile row
        if (tx1 != tx0 && ty0 >= 0) {
            float r = 0, g = 0, b = 0, w = 0;
            source(qx + 1, qy, tx1, ty0, 1u, &r, &g, &b, &w);
            if (ty1 == ty0) source(qx + 1, qy + 1, tx1, ty0, 3u, &r, &g, &b, &w);
            if (w > 0) add_xyz(&out, r, g, b, w);
        }
        // group 2: tile below in the same tile column
        if (ty1 != ty0 && tx0 >= 0) {
            float r = 0, g = 0, b = 0, w = 0;
            source(qx, qy + 1, tx0, ty1, 2u, &r, &g, &b, &w);
            if (tx1 == tx0) source(qx + 1, qy + 1, tx0, ty1, 3u, &r, &g, &b, &w);
            if (w > 0) add_xyz(&out, r, g, b, w);
        }
        // group 3: diagonal tile
        if (tx1 != tx0 && ty1 != ty0) {
            float r = 0, g = 0, b = 0, w = 0;
            source(qx + 1, qy + 1, tx1, ty1, 3u, &r, &g, &b, &w);
            if (w > 0) add_xyz(&out, r, g, b, w);
        }
        F.film_xyzw[i] = out;
    }
}

// ---------------------------------------------------------------------------
// kernel-level probes for parity tests
template <bool ANY, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace(DScene S, int n, const float4 *ro, const float4 *rd, float4 *hits,
                                                  DCounters *counters, int *SPILL) {
    __shared__ int lds_stack[kWavesPerBlock][2 * kLdsStackDepth][64];
    lds_int *my_stack = (lds_int *)&lds_stack[threadIdx.x >> 6][0][threadIdx.x & 63];
    const uint32_t spill_stride = gridDim.x * kBlock;
    int *my_spill = SPILL + blockIdx.x * kBlock + threadIdx.x;
    TraceStats st = {0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < uint32_t(n); i += gridDim.x * kBlock) {
        const float4 o4 = ro[i], d4 = rd[i];
        HitRec h;
        h.t = h.b0 = h.b1 = h.b2 = 0;
        const bool found = traverse<ANY, COUNT>(S, F3{o4.x, o4.y, o4.z}, F3{d4.x, d4.y, d4.z}, d4.w, my_stack, my_spill,
                                                spill_stride, &h, &st);
        if (ANY) {
            hits[2 * i] = make_float4(b2f(found ? 1u : 0u), 0, 0, 0);
            hits[2 * i + 1] = make_float4(0, 0, 0, 0);
        } else {
            hits[2 * i] = make_float4(b2f(uint32_t(found ? h.prim : -1)), found ? h.t : 0.f, 0, 0);
            hits[2 * i + 1] = make_float4(found ? h.b0 : 0.f, found ? h.b1 : 0.f, found ? h.b2 : 0.f, 0);
        }
    }
    if (COUNT && counters) {
        flush_counter(ANY ? &counters->nodes_any : &counters->nodes_closest, st.nodes);
        flush_counter(&counters->tri_tests, st.tris);
        flush_counter(&counters->tri_hits, st.tri_hits);
        flush_counter(&counters->sphere_tests, st.spheres);
    }
}
__global__ void k_halton(DScene S, int n, const int *px, const int *py, const int *k, int dim0, int ndims, float *out,
                         uint32_t *index_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t idx = sample_index(S, px[i], py[i], uint32_t(k[i]));
    if (index_out) index_out[i] = idx;
    for (int d = 0; d < ndims; ++d) out[size_t(i) * ndims + d] = sample_dimension(S, idx, dim0 + d, px[i], py[i]);
}
__global__ void k_camera(DScene S, int n, const float *pfilm, const float *plens, float *o, float *d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 ro, rd;
    float tm;
    camera_ray(S, pfilm[2 * i], pfilm[2 * i + 1], plens ? plens[2 * i] : 0.f, plens ? plens[2 * i + 1] : 0.f, &ro, &rd,
               &tm);
    o[3 * i] = ro.x;
    o[3 * i + 1] = ro.y;
    o[3 * i + 2] = ro.z;
    d[3 * i] = rd.x;
    d[3 * i + 1] = rd.y;
    d[3 * i + 2] = rd.z;
}
// BSDF in a canonical frame (ns = ng = +z, ss = +x). sample == 0: out = {f.xyz, pdf}
// for (wo, wi); sample == 1: out = {wi.xyz, f.xyz, pdf} for (wo, u).
__global__ void k_bsdf_probe(DScene S, int n, int mat, const float *wo, const float *wi_or_u, int sample, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Isect is;
    is.sn = F3{0, 0, 1};
    is.n = F3{0, 0, 1};
    is.sdpdu = F3{1, 0, 0};
    is.p = is.perr = is.wo = F3{0, 0, 0};
    Bsdf b = make_bsdf(S.materials[mat], is);
    b.ss = F3{1, 0, 0};
    b.ts = cross(b.ns, b.ss);
    const F3 w = F3{wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]};
    if (!sample) {
        const F3 wi = F3{wi_or_u[3 * i], wi_or_u[3 * i + 1], wi_or_u[3 * i + 2]};
        const F3 f = bsdf_f(b, w, wi);
        out[4 * i] = f.x;
        out[4 * i + 1] = f.y;
        out[4 * i + 2] = f.z;
        out[4 * i + 3] = bsdf_pdf(b, w, wi);
    } else {
        F3 wi = F3{0, 0, 0};
        float pdf = 0;
        const F3 f = bsdf_sample_f(b, w, &wi, wi_or_u[2 * i], wi_or_u[2 * i + 1], &pdf);
        out[7 * i] = wi.x;
        out[7 * i + 1] = wi.y;
        out[7 * i + 2] = wi.z;
        out[7 * i + 3] = f.x;
        out[7 * i + 4] = f.y;
        out[7 * i + 5] = f.z;
        out[7 * i + 6] = pdf;
    }
}
__global__ void k_trig_probe(int n, const float *x, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    sincos_f(x[i], &s, &c);
    out[3 * i] = s;
    out[3 * i + 1] = c;
    out[3 * i + 2] = acos_f(clampf(x[i], -1.f, 1.f));
}

// ---------------------------------------------------------------------------
// launchers
constexpr int kTraverseBlocksPerCu = IILE_TRAV_WAVES;  // resident blocks per CU (LDS stacks, VGPRs)
int default_trav_blocks_per_cu() { return kTraverseBlocksPerCu; }
constexpr int kMaxTraverseBlocksPerCu = 8;  // spill columns are sized for this many
uint32_t max_traversal_threads(int n_cus) {
    return uint32_t(n_cus) * kMaxTraverseBlocksPerCu * kBlock * kSpillStackDepth * 2;  // (ref, tMin) per level
}
uint32_t queue_capacity(uint32_t n_paths, int n_cus) {
    // every wavefront that appends can leave < 64 slots per kOutBlock it fills plus one
    // partly filled block behind
    const uint64_t waves = std::min<uint64_t>(uint64_t(n_cus) * kTraverseBlocksPerCu * kWavesPerBlock, n_paths / 64 + 8);
    return uint32_t(std::min<uint64_t>(uint64_t(n_paths) + n_paths / 8 + waves * kOutBlock, 0xffff0000ull));
}
void launch_generate(const DScene &S, const PassDesc &P, const PassBuffers &B, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_generate, dim3(grid_blocks(P.n_paths, cfg.n_cus, 8)), dim3(kBlock), 0, cfg.stream, S, P, B,
                       cfg.count_stats ? 1 : 0);
}
void launch_extend(const DScene &S, const PassDesc &P, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg) {
    const dim3 grid(grid_blocks(max_rays, cfg.n_cus, cfg.trav_blocks_per_cu > 0 ? cfg.trav_blocks_per_cu : kTraverseBlocksPerCu));
    const bool gen = bounce == 0 && P.gen_fused && !cfg.count_stats;
    if (cfg.count_stats)
        hipLaunchKernelGGL((k_extend<true, true, false>), grid, dim3(kBlock), 0, cfg.stream, S, P, B, bounce);
    else if (S.has_alpha) {
        if (gen)
            hipLaunchKernelGGL((k_extend<false, true, true>), grid, dim3(kBlock), 0, cfg.stream, S, P, B, bounce);
        else
            hipLaunchKernelGGL((k_extend<false, true, false>), grid, dim3(kBlock), 0, cfg.stream, S, P, B, bounce);
    } else {
        if (gen)
            hipLaunchKernelGGL((k_extend<false, false, true>), grid, dim3(kBlock), 0, cfg.stream, S, P, B, bounce);
        else
            hipLaunchKernelGGL((k_extend<false, false, false>), grid, dim3(kBlock), 0, cfg.stream, S, P, B, bounce);
    }
}
void launch_shade(const DScene &S, const PassDesc &P, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg) {
#ifndef IILE_SHADE_BLOCKS
#define IILE_SHADE_BLOCKS 3  // = resident blocks per CU at 3 waves/SIMD: the static split has no tail
#endif
    const dim3 grid(grid_blocks(max_rays, cfg.n_cus, IILE_SHADE_BLOCKS));
    const size_t perm_bytes = S.sobol ? size_t(S.sobol_dims) * 32 * sizeof(uint32_t) : (size_t(S.n_perms) * sizeof(uint16_t) + 15) & ~size_t(15);
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
void launch_shadow(const DScene &S, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg) {
    const dim3 grid(grid_blocks(max_rays, cfg.n_cus, cfg.trav_blocks_per_cu > 0 ? cfg.trav_blocks_per_cu : kTraverseBlocksPerCu));
    if (cfg.count_stats)
        hipLaunchKernelGGL((k_shadow<true, true>), grid, dim3(kBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
    else
        {
        if (S.has_alpha)
            hipLaunchKernelGGL((k_shadow<false, true>), grid, dim3(kBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
        else
            hipLaunchKernelGGL((k_shadow<false, false>), grid, dim3(kBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
    }
}
void launch_mis(const DScene &S, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg) {
    const dim3 grid(grid_blocks(max_rays, cfg.n_cus, cfg.trav_blocks_per_cu > 0 ? cfg.trav_blocks_per_cu : kTraverseBlocksPerCu));
    if (cfg.count_stats)
        hipLaunchKernelGGL((k_mis<true, true>), grid, dim3(kBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
    else
        {
        if (S.has_alpha)
            hipLaunchKernelGGL((k_mis<false, true>), grid, dim3(kBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
        else
            hipLaunchKernelGGL((k_mis<false, false>), grid, dim3(kBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
    }
}
void launch_light_distributions(const DScene &S, const float *samples, float *out, const LaunchCfg &cfg) {
    const int n = S.light_nv[0] * S.light_nv[1] * S.light_nv[2];
    hipLaunchKernelGGL(k_light_distributions, dim3((n + 127) / 128), dim3(128), 0, cfg.stream, S, samples, out);
}
void launch_miss(const DScene &S, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg) {
    const dim3 grid(grid_blocks(max_rays, cfg.n_cus, 8));
    hipLaunchKernelGGL(k_miss, grid, dim3(kBlock), 0, cfg.stream, S, B, bounce);
}
void launch_mis_lit(const DScene &S, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg) {
    const dim3 grid(grid_blocks(max_rays, cfg.n_cus, 8));
    hipLaunchKernelGGL(k_mis_lit, grid, dim3(kBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
}
void launch_film_accumulate(const DScene &S, const PassDesc &P, const PassBuffers &B, const FilmBuffers &F,
                            const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_film_accumulate, dim3(grid_blocks(uint32_t(P.n_pass_tiles) * 256u, cfg.n_cus, 8)),
                       dim3(kBlock), 0, cfg.stream, S, P, B, F);
}
void launch_film_store(const DScene &S, const PassDesc &P, const PassBuffers &B, const FilmBuffers &F, int k_begin, int n_samples,
                       const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_film_store, dim3(grid_blocks(P.n_paths, cfg.n_cus, 8)), dim3(kBlock), 0, cfg.stream, S, P, B, F, k_begin, n_samples);
}
void launch_probe_finish(const DScene &S, const PassDesc &P, const PassBuffers &B, const FilmBuffers &F, int n_probes,
                         float *intensity, float *normals, float *distance, const LaunchCfg &cfg) {
    const uint32_t n = uint32_t(S.crop_x1 - S.crop_x0) * uint32_t(S.crop_y1 - S.crop_y0) * uint32_t(n_probes);
    hipLaunchKernelGGL(k_probe_finish, dim3(grid_blocks(n, cfg.n_cus, 8)), dim3(kBlock), 0, cfg.stream, S, P, B, F, n_probes, intensity,
                       normals, distance);
}
void launch_film_gather(const DScene &S, const PassDesc &P, const FilmBuffers &F, int n_samples, const LaunchCfg &cfg) {
    const uint32_t n = uint32_t(S.crop_x1 - S.crop_x0) * uint32_t(S.crop_y1 - S.crop_y0) *
                       (P.probe_mode ? uint32_t(P.n_owned_tiles / P.probe_tiles) : 1u);
    hipLaunchKernelGGL(k_film_gather, dim3(grid_blocks(n, cfg.n_cus, 8)), dim3(kBlock), 0, cfg.stream, S, P, F, n_samples);
}
void launch_film_resolve(const DScene &S, const PassDesc &P, const FilmBuffers &F, const LaunchCfg &cfg) {
    const uint32_t n = uint32_t(S.crop_x1 - S.crop_x0) * uint32_t(S.crop_y1 - S.crop_y0);
    hipLaunchKernelGGL(k_film_resolve, dim3(grid_blocks(n, cfg.n_cus, 8)), dim3(kBlock), 0, cfg.stream, S, P, F);
}
void launch_trace(const DScene &S, int n, const float4 *ro, const float4 *rd, float4 *hits, int any_hit,
                  DCounters *counters, int *spill, const LaunchCfg &cfg) {
    const dim3 grid(grid_blocks(uint32_t(n), cfg.n_cus, kTraverseBlocksPerCu));
    // count_stats selects the instrumented traversal (binary steps) or the one the render
    // kernels run uninstrumented (four-wide steps)
    if (any_hit) {
        if (cfg.count_stats)
            hipLaunchKernelGGL((k_trace<true, true>), grid, dim3(kBlock), 0, cfg.stream, S, n, ro, rd, hits, counters, spill);
        else
            hipLaunchKernelGGL((k_trace<true, false>), grid, dim3(kBlock), 0, cfg.stream, S, n, ro, rd, hits, counters, spill);
    } else {
        if (cfg.count_stats)
            hipLaunchKernelGGL((k_trace<false, true>), grid, dim3(kBlock), 0, cfg.stream, S, n, ro, rd, hits, counters, spill);
        else
            hipLaunchKernelGGL((k_trace<false, false>), grid, dim3(kBlock), 0, cfg.stream, S, n, ro, rd, hits, counters, spill);
    }
}
void launch_halton(const DScene &S, int n, const int *px, const int *py, const int *k, int dim0, int ndims,
                   float *out, uint32_t *index_out, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_halton, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, S, n, px, py, k, dim0, ndims, out,
                       index_out);
}
void launch_camera(const DScene &S, int n, const float *pfilm, const float *plens, float *o, float *d,
                   const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_camera, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, S, n, pfilm, plens, o, d);
}
void launch_bsdf_probe(const DScene &S, int n, int mat, const float *wo, const float *wi_or_u, int sample,
                       float *out, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_bsdf_probe, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, S, n, mat, wo, wi_or_u, sample,
                       out);
}
// ImageTexture::Evaluate for given (u, v) and differentials (test probe)
__global__ void k_texture_probe(DScene S, int n, int tex, const float *uv, const float *duv, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TexDiff td = TexDiff{duv[4 * i], duv[4 * i + 1], duv[4 * i + 2], duv[4 * i + 3]};
    const F3 c = tex_evaluate(S, tex, uv[2 * i], uv[2 * i + 1], td);
    out[3 * i] = c.x;
    out[3 * i + 1] = c.y;
    out[3 * i + 2] = c.z;
}
void launch_texture_probe(const DScene &S, int n, int tex, const float *uv, const float *duv, float *out, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_texture_probe, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, S, n, tex, uv, duv, out);
}
__global__ void k_gather4(const float4 *src, const uint32_t *idx, int n, float4 *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
__global__ void k_scatter4(float4 *dst, const uint32_t *idx, int n, const float4 *in) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = in[i];
}
// One exact FilmTile sum per thread for a pixel that receives flagged samples from pixels generated before it in its
// own tile: those samples first, then its own kc samples, then the flagged samples of later pixels — all in generation
// order, guarded like k_film_accumulate (see patch_pass_finish in api.hip).
__global__ void k_patch_own(DScene S, const float4 *L, int n, const uint32_t *local_slot, const uint32_t *range3, const uint32_t *flag_pid,
                            int kc, float4 *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0, g = 0, b = 0, w = 0;
    auto add = [&](float4 v) {
        const F3 c = guard_radiance(S, F3{v.x, v.y, v.z});
        r += c.x * 1.f * 1.f;
        g += c.y * 1.f * 1.f;
        b += c.z * 1.f * 1.f;
        w += 1.f;
    };
    const uint32_t b0 = range3[3 * i], b1 = range3[3 * i + 1], b2 = range3[3 * i + 2];
    for (uint32_t h = b0; h < b1; ++h) add(L[flag_pid[h]]);
    const uint32_t first = local_slot[i] * uint32_t(kc);
    for (int k = 0; k < kc; ++k) add(L[first + uint32_t(k)]);
    for (uint32_t h = b1; h < b2; ++h) add(L[flag_pid[h]]);
    out[i] = make_float4(r, g, b, w);
}
void launch_patch_own(const DScene &S, const float4 *L, int n, const uint32_t *local_slot, const uint32_t *range3, const uint32_t *flag_pid,
                      int kc, float4 *out, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_patch_own, dim3((n + 63) / 64), dim3(64), 0, cfg.stream, S, L, n, local_slot, range3, flag_pid, kc, out);
}
void launch_gather4(const float4 *src, const uint32_t *idx, int n, float4 *out, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_gather4, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, src, idx, n, out);
}
void launch_scatter4(float4 *dst, const uint32_t *idx, int n, const float4 *in, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_scatter4, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, dst, idx, n, in);
}
void launch_trig_probe(int n, const float *x, float *out, const LaunchCfg &cfg) {
    hipLaunchKernelGGL(k_trig_probe, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, n, x, out);
}

}  // namespace iile
