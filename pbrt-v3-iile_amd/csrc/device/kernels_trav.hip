// kernels_trav.hip — the BVH traversal kernels of the wavefront path tracer (pipeline overview in kernels.hip):
// k_extend (closest hit of the main path, camera rays made in place at bounce 0), k_shadow (any hit + L += beta * Ld),
// k_mis (closest hit of the BSDF-sampled rays), k_trace (kernel-level probe).
#include "kcommon.h"

namespace iile {

// ---------------------------------------------------------------------------
// extend: BVHAccel::Intersect for every ray of queue `bounce & 1`; hits are
// appended (ballot-compacted) to the shade queue.
// GEN (first bounce, PassDesc::gen_fused): the queue is the dense range of path ids and a lane makes its camera ray
// itself (what k_generate would have written and this kernel read back: 64 B per path)
template <bool COUNT, bool ALPHA, bool GEN>
__global__ __launch_bounds__(kTravBlock, IILE_TRAV_WAVES) void k_extend(DScene S, PassDesc P, PassBuffers B, int bounce) {
    __shared__ int lds_stack[kTravWavesPerBlock][2 * kLdsStackDepth][64];
    __shared__ __attribute__((aligned(16))) char lds_top[COUNT ? 16 : kMaxTop * kTopStride];
    StackRef sr{(lds_int *)&lds_stack[threadIdx.x >> 6][0][threadIdx.x & 63], B.spill, blockIdx.x * kTravBlock + threadIdx.x, gridDim.x * kTravBlock};
    sr.root = S.root_ref;
    if (!COUNT && S.n_top > 0) {  // the instrumented build walks the binary records: no four-wide steps, no top
        stage_top_records(S, (lds_char *)lds_top, int(threadIdx.x), kTravBlock);
        __syncthreads();
        sr.top = (lds_char *)lds_top;
        sr.root = S.root_ref_top;
    }
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
#ifdef IILE_TRAV_ITERSTATS
    // diagnostic build only (tools/trav_stamps.py iterstats): what the wavefronts of bounces >= 1 look like at each vote —
    // [0] interior votes, [1] lanes stepping in them, [2] lanes waiting at a leaf meanwhile, [3] leaf votes, [4] lanes stepping,
    // [5] lanes waiting at an interior record meanwhile, [6] idle lanes (no ray) summed over all votes, [7] refills
    unsigned long long iter_stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ITER_STAT(kind, n_go, n_wait)                                            \
    do {                                                                         \
        iter_stat[3 * (kind)] += 1;                                              \
        iter_stat[3 * (kind) + 1] += uint32_t(n_go);                             \
        iter_stat[3 * (kind) + 2] += uint32_t(n_wait);                           \
        iter_stat[6] += uint32_t(64 - (n_go) - (n_wait));                        \
    } while (0)
#else
#define ITER_STAT(kind, n_go, n_wait) \
    do {                              \
    } while (0)
#endif
#ifdef IILE_TRAV_STAMPS
    // diagnostic build only (tools/trav_stamps.py): wave cycles per section of the loop, summed per wavefront and added to
    // DCounters::path_length: the camera-ray build [0] refill + ray generation, [1] interior steps, [2] leaf steps,
    // [3] finish + queue append; the other bounces the same in [4..7]
    unsigned long long stamp_sum[4] = {0, 0, 0, 0};
    unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#define TRAV_STAMP(i)                                                   \
    do {                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        stamp_sum[i] += now_ - stamp_t;                                 \
        stamp_t = now_;                                                 \
    } while (0)
#else
#define TRAV_STAMP(i) \
    do {              \
    } while (0)
#endif
    while (true) {
        TRAV_STAMP(3);
        const unsigned long long idle_mask = __ballot(!active);
        // (the camera-ray build makes its rays here, some 400 instructions each: it waits for more idle lanes than the others)
        if (refill_due(idle_mask, feed, GEN ? IILE_REFILL_IDLE_GEN : kRefillIdle, GEN ? IILE_REFILL_IDLE_GEN_SLOW : IILE_REFILL_IDLE_SLOW)) {
            uint32_t s_new;
#ifdef IILE_TRAV_ITERSTATS
            iter_stat[7] += 1;
#endif
            if (feed_take(feed, head, count, !active, &s_new, warm)) {
                slot = s_new;
                if (GEN) {
                    // the path's radiance starts here (no memset of 2 GB ahead of the pass); the zeros are made on the spot, or
                    // four registers of them live — spilled — through the whole kernel
                    const float z = opaque_zero();
                    B.L[slot] = make_float4(z, z, z, z);
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
                        camera_ray(S, pfx, pfy, l0, l1, &o, &d, &tmax, opaque_zero());
                        // (no PassBuffers::hindex entry: the index rides in the record below, the film kernels work it out again)
                        // the film position rides in the path's (not yet used) throughput record: the first k_shade
                        // rebuilds the ray from it instead of evaluating the Halton dimensions again
                        B.beta[slot] = make_float4(pfx, pfy, b2f(idx), 0.f);  // (+ the Halton index: one record, one load in k_shade)
                        gen_d = make_float4(d.x, d.y, d.z, tmax);
                        trav_begin<COUNT>(S, t, o, d, tmax, &st, sr.root);
                        active = true;
                        ++n_rays;  // (GEN is never an instrumented build)
                    }
                } else {
                    const float4 o4 = ro[slot], d4 = rd[slot];
                    if (f2b(o4.w) != kInvalid) {
                        // (from bounce 1 on the direction record's .w carries path state: those rays have no far end)
                        trav_begin<COUNT>(S, t, F3{o4.x, o4.y, o4.z}, F3{d4.x, d4.y, d4.z}, bounce > 0 ? IILE_INF : d4.w, &st, sr.root);
                        active = true;
                        ++n_rays;
                        if (COUNT && B.nray_out) B.nray_out[2 * f2b(o4.w)] += 1;
                    }
                }
            }
        }
        TRAV_STAMP(0);
        if (__ballot(active) == 0) {
            if (feed.exhausted) break;
            continue;
        }
        // one step per iteration for the whole wavefront, interior or leaf, whichever has more
        // lanes waiting (25.9 ms vs 34.6 ms for strict while-while on the 1080p/64spp step)
        {
            const bool wi = active && t.have && t.cur >= 0;
            const bool wl = active && t.have && t.cur < 0;
            const int n_int = __popcll(__ballot(wi)), n_leaf = __popcll(__ballot(wl));
            if (n_int > 0 && n_int * kVoteNum >= n_leaf * kVoteDen) {
                if (wi) trav_step<COUNT>(S, t, sr, &st);
                ITER_STAT(0, n_int, n_leaf);
                TRAV_STAMP(1);
            } else if (n_leaf > 0) {
                ITER_STAT(1, n_leaf, n_int);
                if (wl) trav_leaf<COUNT, ALPHA>(S, t, sr, &st, false, GEN ? &gen_d : &rd[slot]);
                TRAV_STAMP(2);
            }
        }
        const bool fin = active && !t.have;
        const bool is_hit = fin && t.hit_prim >= 0;
        if (fin) {
            // (a miss's record is read by k_miss alone, which runs only for scenes with an infinite light)
            if (is_hit || S.has_infinite) B.hits[slot] = make_float4(b2f(uint32_t(hit_index(t.hit_prim))), t.b0, t.b1, t.b2);
            active = false;
            if (COUNT && t.hit_prim < 0) ++n_term;  // the path left the scene: ReportValue(pathLength, bounces)
        }
        const uint32_t pos = out_take(shade_out, &B.counts[kCntShade + bounce], is_hit, pad_shade);
        // entry = queue slot | shading class << 28 (k_shade regroups its block by class)
        if (is_hit) B.shade_q[pos] = slot | (uint32_t(t.hit_prim >> kHitClassShift) & 7u) << kSlotBits;
    }
    out_flush(shade_out, pad_shade);
#ifdef IILE_TRAV_STAMPS
    if (!COUNT && (threadIdx.x & 63) == 0)
        for (int i = 0; i < 4; ++i) atomicAdd(&B.counters->path_length[(GEN ? 0 : 4) + i], stamp_sum[i]);
#endif
#ifdef IILE_TRAV_ITERSTATS
    if (!COUNT && !GEN && (threadIdx.x & 63) == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&B.counters->path_length[i], iter_stat[i]);
#endif
    flush_counter(&B.counters->ext_traced, n_rays);  // every build: the uninstrumented pass leaves out rays that cannot matter
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
// NEE resolution: two homogeneous kernels over the NEE records of one bounce.
//   k_mis     BVHAccel::Intersect for the MIS ray; records which emitter (if any) it ended on
//   k_shadow  BVHAccel::IntersectP for the shadow ray, then L += beta * Ld
// Both finish a record with a single store, so no load ever stalls their loops.
// (One fused kernel walking each record through both rays measured 60 ms per
// 1080p/64spp step against 7 + 17.5 + 26.4 ms for its parts: any-hit and
// closest-hit lanes in one wavefront keep each other waiting.)


template <bool COUNT, bool ALPHA>
__global__ __launch_bounds__(kTravBlock, IILE_TRAV_WAVES) void k_shadow(DScene S, PassBuffers B, int bounce, uint32_t plane) {
    __shared__ int lds_stack[kTravWavesPerBlock][2 * kLdsStackDepth][64];
    __shared__ __attribute__((aligned(16))) char lds_top[COUNT ? 16 : kMaxTop * kTopStride];
    StackRef sr{(lds_int *)&lds_stack[threadIdx.x >> 6][0][threadIdx.x & 63], B.spill, blockIdx.x * kTravBlock + threadIdx.x, gridDim.x * kTravBlock};
    sr.root = S.root_ref;
    if (!COUNT && S.n_top > 0) {  // the instrumented build walks the binary records: no four-wide steps, no top
        stage_top_records(S, (lds_char *)lds_top, int(threadIdx.x), kTravBlock);
        __syncthreads();
        sr.top = (lds_char *)lds_top;
        sr.root = S.root_ref_top;
    }
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
#ifdef IILE_SHADOW_STAMPS
    // diagnostic build only (tools/trav_stamps.py shadow): as in k_extend — [0] refill, [1] interior steps, [2] leaf steps, [3] finish
    unsigned long long stamp_sum[4] = {0, 0, 0, 0};
    unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#define SHADOW_STAMP(i)                                                 \
    do {                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        stamp_sum[i] += now_ - stamp_t;                                 \
        stamp_t = now_;                                                 \
    } while (0)
#else
#define SHADOW_STAMP(i) \
    do {                \
    } while (0)
#endif
    while (true) {
        SHADOW_STAMP(3);
        const unsigned long long idle_mask = __ballot(!active);
        if (refill_due(idle_mask, feed, kRefillIdle)) {
            uint32_t e_new;
            if (feed_take(feed, head, count, !active, &e_new, [&](uint32_t first) {
                    warm_plane(B.nee, first, count);
                    warm_plane(B.nee + plane, first, count);
                    warm_plane(B.nee + 4 * size_t(plane), first, count);
                })) {
                e = e_new;
                const float4 n1 = B.nee[plane + e];
                const uint32_t flags = f2b(n1.w);
                if (flags != kInvalid) {
                    // Ld = [light sample unoccluded] A + [MIS ray reached the sampled light, facing it] B,
                    // L += beta * Ld / lightPdf (integrator.cpp:150-158, 205-211; path.cpp:123-128). A record without a MIS
                    // ray carries beta * (A / lightPdf) ready made (k_shade) and no throughput plane.
                    const float4 n0 = B.nee[e], a4 = B.nee[4 * size_t(plane) + e];
                    pid = f2b(a4.w);
#ifdef IILE_SHADOW_DIAG_NO_L_READ   // timing only (wrong film): what the scattered 16-byte read of L costs this kernel (profiles/r06_ab_shadow_L_read.txt)
                    const float4 L4 = make_float4(0, 0, 0, 0);
#else
                    const float4 L4 = B.L[pid];
#endif
                    const bool has_shadow = (flags & NEE_HAS_SHADOW) != 0;
                    if (flags & NEE_HAS_MIS) {
                        const float4 be = B.nee[6 * size_t(plane) + e];
                        const bool lit = B.nee_mis[e] != 0;
                        F3 Ld_u = F3{0, 0, 0}, Ld_o = F3{0, 0, 0};
                        if (has_shadow) Ld_u = Ld_u + F3{a4.x, a4.y, a4.z};
                        if (lit) {
                            const float4 b4 = B.nee[5 * size_t(plane) + e];
                            Ld_u = Ld_u + F3{b4.x, b4.y, b4.z};
                            Ld_o = Ld_o + F3{b4.x, b4.y, b4.z};
                        }
                        // UniformSampleOneLight returns EstimateDirect / lightPdf (n0.w; 1 with a single light: x / 1 is x)
                        const F3 beta = F3{be.x, be.y, be.z};
                        add_unoccluded = beta * sdiv(Ld_u, n0.w);
                        add_occluded = beta * sdiv(Ld_o, n0.w);
                    } else {
                        add_unoccluded = F3{a4.x, a4.y, a4.z};
                        add_occluded = F3{0, 0, 0};  // beta * (0 / lightPdf)
                    }
                    L_old = F3{L4.x, L4.y, L4.z};
                    if (has_shadow) {
                        trav_begin<COUNT>(S, t, F3{n0.x, n0.y, n0.z}, F3{n1.x, n1.y, n1.z}, 1 - kShadowEpsilon, &st, sr.root);
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
        SHADOW_STAMP(0);
        if (__ballot(active) == 0) {
            if (feed.exhausted) break;
            continue;
        }
        // Shadow rays end at their first hit, so lanes leave at very different times: one
        // step per iteration, interior or leaf, whichever keeps more lanes busy
        // (17.9 ms vs 22.1 ms for strict while-while on the 1080p/64spp step).
        {
            const bool wi = active && t.have && t.cur >= 0;
            const bool wl = active && t.have && t.cur < 0;
            const int n_int = __popcll(__ballot(wi)), n_leaf = __popcll(__ballot(wl));
            if (n_int > 0 && n_int * kVoteNum >= n_leaf * kVoteDen) {
                if (wi) trav_step<COUNT, true>(S, t, sr, &st);
                SHADOW_STAMP(1);
            } else if (n_leaf > 0) {
                if (wl && trav_leaf<COUNT, ALPHA>(S, t, sr, &st, true, &B.nee[plane + e])) occluded = true;
                SHADOW_STAMP(2);
            }
        }
        if (active && !t.have) {
            const F3 add = occluded ? add_occluded : add_unoccluded;
            const F3 Ln = L_old + add;  // store only: nothing is loaded here
            // (an occluded light sample without a lit MIS ray adds +0: L stands as it is — a sum of non-negative terms from +0,
            //  never -0 — and the scattered 16-byte store is left out: a sixth of the records)
            if (!is_black(add)) B.L[pid] = make_float4(Ln.x, Ln.y, Ln.z, 0);
            if (COUNT && is_black(add)) ++n_zero;
            active = false;
        }
    }
#ifdef IILE_SHADOW_STAMPS
    if (!COUNT && (threadIdx.x & 63) == 0)
        for (int i = 0; i < 4; ++i) atomicAdd(&B.counters->path_length[i], stamp_sum[i]);
#endif
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

// ANYORDER: every light of the scene is an infinite one, so every ray of this queue ends at its first hit (below) and the
// four-wide records are walked unordered, as k_shadow walks them
template <bool COUNT, bool ALPHA, bool ANYORDER = false>
__global__ __launch_bounds__(kTravBlock, IILE_TRAV_WAVES) void k_mis(DScene S, PassBuffers B, int bounce, uint32_t plane) {
    __shared__ int lds_stack[kTravWavesPerBlock][2 * kLdsStackDepth][64];
    __shared__ __attribute__((aligned(16))) char lds_top[COUNT ? 16 : kMaxTop * kTopStride];
    StackRef sr{(lds_int *)&lds_stack[threadIdx.x >> 6][0][threadIdx.x & 63], B.spill, blockIdx.x * kTravBlock + threadIdx.x, gridDim.x * kTravBlock};
    sr.root = S.root_ref;
    if (!COUNT && S.n_top > 0) {  // the instrumented build walks the binary records: no four-wide steps, no top
        stage_top_records(S, (lds_char *)lds_top, int(threadIdx.x), kTravBlock);
        __syncthreads();
        sr.top = (lds_char *)lds_top;
        sr.root = S.root_ref_top;
    }
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
    bool active = false, first_hit_ends = false;
    uint32_t q = 0, e = 0;
    while (true) {
        const unsigned long long idle_mask = __ballot(!active);
        if (refill_due(idle_mask, feed, kRefillIdle)) {
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
                    // EstimateDirect asks of this ray whether it ends on the sampled light (integrator.cpp:204-211). For an INFINITE light
                    // that is "does it leave the scene": the first primitive it meets settles it (no hit is found later that was not
                    // found first: until one is, nothing prunes but the slab tests, whose outcome does not depend on the order), so
                    // the uninstrumented walk stops there instead of looking for the closest one. Alpha masks as Intersect applies them.
                    first_hit_ends = !COUNT && S.has_infinite && S.lights[f2b(n3.w)].type == kLightInfinite;
                    trav_begin<COUNT>(S, t, F3{n2.x, n2.y, n2.z}, F3{n3.x, n3.y, n3.z}, IILE_INF, &st, sr.root);
                    active = true;
                    ++n_traced;
                    if (COUNT) {
                        ++n_closest;
                        if (B.nray_out) B.nray_out[2 * f2b(B.nee[4 * size_t(plane) + e].w)] += 1;
                    }
                }
            }
        }
        if (__ballot(active) == 0) {
            if (feed.exhausted) break;
            continue;
        }
        {
            const bool wi = active && t.have && t.cur >= 0;
            const bool wl = active && t.have && t.cur < 0;
            const int n_int = __popcll(__ballot(wi)), n_leaf = __popcll(__ballot(wl));
            if (n_int > 0 && n_int * kVoteNum >= n_leaf * kVoteDen) {
                if (wi) trav_step<COUNT, ANYORDER>(S, t, sr, &st);
            } else if (n_leaf > 0) {
                if (wl) trav_leaf<COUNT, ALPHA>(S, t, sr, &st, false, &B.nee[3 * size_t(plane) + q]);
            }
        }
        if (first_hit_ends && active && t.hit_prim >= 0) t.have = false;
        if (active && !t.have) {
            // store only: (area light index + 1) of the primitive the MIS ray ended on, 0 for none
            // 255: the ray escaped (matters to an infinite light only)
            const uint8_t on_light = uint8_t(t.hit_prim < 0 ? (S.has_infinite ? 255 : 0) : (t.hit_prim >> kHitLightShift));
            B.nee_mis[e] = on_light;
            // the rare ray that ends on an emitter leaves its hit for k_mis_lit (the hit records are
            // idle between shade and the next extend)
            if (on_light && t.hit_prim >= 0) B.mis_hit[e] = make_float4(b2f(uint32_t(hit_index(t.hit_prim))), t.b0, t.b1, t.b2);
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


// ---------------------------------------------------------------------------
// kernel-level probe for parity tests
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

// ---------------------------------------------------------------------------
// launchers
constexpr int kTraverseBlocksPerCu = IILE_TRAV_WAVES;  // resident 256-thread blocks per CU (LDS stacks, VGPRs); k_extend / k_shadow / k_mis run the same threads as kTravBlocksPerCu blocks of kTravBlock
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
static dim3 trav_grid(uint32_t n, const LaunchCfg &cfg) {
    const int per_cu256 = cfg.trav_blocks_per_cu > 0 ? cfg.trav_blocks_per_cu : kTraverseBlocksPerCu;
    return dim3(grid_blocks(n, cfg.n_cus, std::max(1, per_cu256 * kBlock / kTravBlock), kTravBlock));
}
void launch_extend(const DScene &S, const PassDesc &P, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg) {
    const dim3 grid = trav_grid(max_rays, cfg);
    const bool gen = bounce == 0 && P.gen_fused && !cfg.count_stats;
    if (cfg.count_stats)
        hipLaunchKernelGGL((k_extend<true, true, false>), grid, dim3(kTravBlock), 0, cfg.stream, S, P, B, bounce);
    else if (S.has_alpha) {
        if (gen)
            hipLaunchKernelGGL((k_extend<false, true, true>), grid, dim3(kTravBlock), 0, cfg.stream, S, P, B, bounce);
        else
            hipLaunchKernelGGL((k_extend<false, true, false>), grid, dim3(kTravBlock), 0, cfg.stream, S, P, B, bounce);
    } else {
        if (gen)
            hipLaunchKernelGGL((k_extend<false, false, true>), grid, dim3(kTravBlock), 0, cfg.stream, S, P, B, bounce);
        else
            hipLaunchKernelGGL((k_extend<false, false, false>), grid, dim3(kTravBlock), 0, cfg.stream, S, P, B, bounce);
    }
}
void launch_shadow(const DScene &S, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg) {
    const dim3 grid = trav_grid(max_rays, cfg);
    if (cfg.count_stats)
        hipLaunchKernelGGL((k_shadow<true, true>), grid, dim3(kTravBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
    else
        {
        if (S.has_alpha)
            hipLaunchKernelGGL((k_shadow<false, true>), grid, dim3(kTravBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
        else
            hipLaunchKernelGGL((k_shadow<false, false>), grid, dim3(kTravBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
    }
}
void launch_mis(const DScene &S, const PassBuffers &B, int bounce, uint32_t max_rays, const LaunchCfg &cfg) {
    const dim3 grid = trav_grid(max_rays, cfg);
    if (cfg.count_stats)
        hipLaunchKernelGGL((k_mis<true, true>), grid, dim3(kTravBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
    else
        {
        if (S.all_lights_infinite) {
            if (S.has_alpha)
                hipLaunchKernelGGL((k_mis<false, true, true>), grid, dim3(kTravBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
            else
                hipLaunchKernelGGL((k_mis<false, false, true>), grid, dim3(kTravBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
        } else if (S.has_alpha)
            hipLaunchKernelGGL((k_mis<false, true>), grid, dim3(kTravBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
        else
            hipLaunchKernelGGL((k_mis<false, false>), grid, dim3(kTravBlock), 0, cfg.stream, S, B, bounce, B.queue_cap);
    }
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

}  // namespace iile
