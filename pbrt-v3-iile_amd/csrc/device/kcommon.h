// kcommon.h — what the kernel translation units share: block geometry, wavefront-aggregated queue appends, the
// persistent work feed, the layout of the per-pass counters, path -> pixel enumeration.
// (kernels_trav.hip: extend / shadow / MIS / trace; kernels_shade.hip: shade, light distributions;
//  kernels.hip: generation, film, probes and the remaining launchers; iispt.hip: the IISPT runner's gather.)
#pragma once
#include <algorithm>

#include "dpath.h"
#include "kernels.h"

namespace iile {

// ---- build parameters of the kernels (every one of them: -DNAME=value through tools/build_variant.sh; the defaults are what is
// tested and benched). Dead ends of earlier rounds are not switches any more: tools/experiments/*.patch restores them.
//   IILE_TRAV_WAVES            waves per SIMD (= resident 256-thread blocks per CU) the traversal kernels are built for
//   IILE_TRAV_BLOCK            threads per block of k_extend / k_shadow / k_mis (a larger block shares one LDS copy of the tree's top)
//   IILE_TOP_RECORDS           four-wide records of the tree's top kept in LDS (dscene.h)
//   IILE_LDS_STACK             traversal stack levels per lane in LDS; deeper levels spill to HBM (dpath.h)
//   IILE_CHUNK                 queue slots a wavefront reserves per atomic on a queue's cursor
//   IILE_SHADE_CHUNK           hits a k_shade wavefront regroups by shading class at a time
//   IILE_REFILL_IDLE / _GEN    idle lanes that trigger a refill (rays that end fast); _GEN: the build that generates camera rays
//   IILE_REFILL_IDLE_SLOW / _GEN_SLOW   the same while the wavefront's lanes go idle slowly (deep trees)
//   IILE_SHADE_WAVES / _TEX    waves per SIMD k_shade's register allocation aims at (untextured / textured build)
//   IILE_DIRECT_SHADE_WAVES    the same for k_direct_shade (kernels_direct.hip)
// Diagnostic builds (tools/*_stamps.py, never shipped): IILE_SHADE_STAMPS, IILE_TRAV_STAMPS, IILE_SHADOW_STAMPS, IILE_TRAV_ITERSTATS.
#ifndef IILE_TRAV_WAVES
#define IILE_TRAV_WAVES 6  // <= 80 VGPRs, no scratch
#endif
// The traversal kernels take ONE step per iteration for the whole wavefront, interior or leaf, whichever has more lanes waiting: an
// interior step when interior lanes x kVoteNum >= leaf lanes x kVoteDen (leaf steps are the dearer ones).
constexpr int kVoteNum = 4, kVoteDen = 5;

constexpr int kBlock = 256;            // 4 wavefronts
constexpr int kWavesPerBlock = kBlock / 64;
#ifndef IILE_TRAV_BLOCK
#define IILE_TRAV_BLOCK 256
#endif
constexpr int kTravBlock = IILE_TRAV_BLOCK;
constexpr int kTravWavesPerBlock = kTravBlock / 64;
constexpr int kTravBlocksPerCu = IILE_TRAV_WAVES * kBlock / kTravBlock;  // same wavefronts per CU whatever the block size
static_assert(kTravBlock % 64 == 0 && kTravBlocksPerCu * kTravBlock == IILE_TRAV_WAVES * kBlock, "IILE_TRAV_BLOCK must divide the CU's traversal threads");
#ifndef IILE_SHADE_CHUNK
#define IILE_SHADE_CHUNK 1024
#endif
constexpr int kShadeChunk = IILE_SHADE_CHUNK;
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
static inline int grid_blocks(uint32_t n, int n_cus, int per_cu, int block = kBlock) {
    long want = (long(n) + block - 1) / block;
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
static_assert(kCntMis + 16 <= kCntWords, "PassBuffers::counts layout");

// Persistent-wavefront work feed. A wavefront reserves kChunk consecutive queue
// slots with one atomic and hands them to its lanes as they go idle, so lanes
// whose ray terminated early pick up new rays instead of waiting for the slowest
// lane of the wavefront (the classic while-while + dynamic fetch scheme, sized
// for 64 lanes). 512-slot chunks keep the head word at a few atomics per
// microsecond, far below its ~88/us saturation point.
#ifndef IILE_CHUNK
#define IILE_CHUNK 512
#endif
constexpr uint32_t kChunk = IILE_CHUNK;
#ifndef IILE_REFILL_IDLE
#define IILE_REFILL_IDLE 32
#endif
#ifndef IILE_REFILL_IDLE_GEN
#define IILE_REFILL_IDLE_GEN 56  // 47.5 -> 47.1 ms against 32: profiles/r03_ab_gen_refill.txt
#endif
#ifndef IILE_REFILL_IDLE_SLOW
#define IILE_REFILL_IDLE_SLOW 12
#endif
#ifndef IILE_REFILL_IDLE_GEN_SLOW
#define IILE_REFILL_IDLE_GEN_SLOW 24
#endif
constexpr int kRefillIdle = IILE_REFILL_IDLE;
constexpr int kRefillSlowRate = 3;   // lanes per vote: below it a wavefront's lanes "go idle slowly"
struct WaveFeed {
    uint32_t cur, end;
    bool exhausted;
    uint32_t votes = 0;  // votes with idle lanes since the last refill
    bool slow = false;   // this wavefront's lanes go idle slowly: refill in small batches
};
// When to refill. A refill runs with only the idle lanes active, so it should wait for a batch of them; but every step it
// waits costs one lane-step per idle lane, and how fast lanes go idle differs by an order of magnitude between workloads:
// killeroo-simple's rays end after ~14 steps (5.8 lanes per vote go idle, 5.5 votes between refills at a batch of 32), the deep
// room's after ~90 (0.6 lanes per vote, 52 votes). A fixed batch of 32 idle lanes — right for the former — left the room's
// wavefronts a quarter empty (profiles/r04_vote_stats.json); 12 is right for the room (432 vs 455 ms) and wrong for killeroo.
// The rule (profiles/r04_ab_refill_rule.txt holds the alternatives measured against it): every wavefront measures the rate at
// which ITS lanes go idle — idle lanes at a refill / votes since the one before — and uses the small batch while that rate is
// below kRefillSlowRate lanes per vote.
DEV bool refill_due(unsigned long long idle_mask, WaveFeed &f, int idle_min, int idle_min_slow = IILE_REFILL_IDLE_SLOW) {
    if (f.exhausted || idle_mask == 0) return false;
    const uint32_t n_idle = uint32_t(__popcll(idle_mask));
    f.votes += 1;
    const uint32_t batch = f.slow ? uint32_t(idle_min_slow) : uint32_t(idle_min);
    const bool due = n_idle >= batch || idle_mask == ~0ull;
    if (due) {
        f.slow = n_idle < uint32_t(kRefillSlowRate) * f.votes;
        f.votes = 0;
    }
    return due;
}

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
static_assert(kChunk <= 64 * 8 && kChunk % 64 == 0, "at most one lane per 128-byte line of a chunk");
DEV void warm_plane(const float4 *plane_base, uint32_t first, uint32_t limit) {
    const uint32_t i = first + uint32_t(lane_id()) * 8u;
    if (i < limit && i < first + kChunk) {
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
        // every probe has its own film. RenderView takes no samples outside the film's pixel bounds (iispt_d.cpp:428-429),
        // so the slots cover those pixels only: storage tile `slot % probe_tiles` of probe `slot / probe_tiles`, laid over
        // the pixel bounds (not the reference's tile grid over the sample bounds, which k_film_gather keeps for its sums)
        const int tile = int(slot % uint32_t(P.probe_tiles));
        const int tx = tile % P.probe_stx, ty = tile / P.probe_stx;
        *px = S.crop_x0 + tx * kTile + int(pix & 15u);
        *py = S.crop_y0 + ty * kTile + int(pix >> 4);
        *k = uint32_t(P.k0) + kk;
        return *px < S.crop_x1 && *py < S.crop_y1;
    }
    const int tile = P.tile_of_slot ? P.tile_of_slot[P.slot0 + int(slot)] : P.slot0 + int(slot);
    const int tx = tile % P.n_tiles_x, ty = tile / P.n_tiles_x;
    *px = S.samp_x0 + tx * kTile + int(pix & 15u);
    *py = S.samp_y0 + ty * kTile + int(pix >> 4);
    *k = uint32_t(P.k0) + kk;
    // (pb = the sample bounds unless "pixelbounds" was given: `if (!InsideExclusive(pixel, pixelBounds)) continue;`, integrator.cpp:272)
    return *px < S.pb_x1 && *py < S.pb_y1 && *px >= S.pb_x0 && *py >= S.pb_y0;
}

// probe pass: the record ((storage slot) * 256 + pixel of the tile) of sample pixel (x, y) of probe `probe`
DEV size_t probe_record(const DScene &S, const PassDesc &P, uint32_t probe, int x, int y) {
    const int rx = x - S.crop_x0, ry = y - S.crop_y0;
    const uint32_t slot = probe * uint32_t(P.probe_tiles) + uint32_t((ry / kTile) * P.probe_stx + rx / kTile);
    return size_t(slot) * 256u + size_t((ry % kTile) * kTile + rx % kTile);
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


}  // namespace iile
