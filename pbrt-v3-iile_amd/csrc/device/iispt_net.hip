// iispt_net.hip — the IISPT network (ml/iispt_net.py:8-109) as hand-written gfx950 kernels behind a C ABI
// (include/iile_gpu.h, iile_iispt_net_*). The reference pipes every probe through a child Python process that runs this
// U-Net in fp32 on one CPU thread (ml/main_stdio_net.py:44-106, 47 ms per probe, Doc.md:55-64); here a batch of thousands
// of probes runs layer by layer over activations that stay in HBM.
//
// Arithmetic. gfx950's fp32 matrix pipe runs at the vector rate (157 TFLOP/s), its 16-bit pipes sixteen times faster, so
// every 3 x 3 convolution is an implicit GEMM on v_mfma_f32_32x32x16_f16 over SPLIT operands: a = a_hi + a_lo with
// a_hi = fp16(a), a_lo = fp16(a - a_hi) — 22 significant bits — and the product a * w is accumulated in fp32 as
// a_hi w_hi + a_hi w_lo + a_lo w_hi (the dropped a_lo w_lo term is 2^-22 of the product). Three matrix instructions per tile
// instead of one. The weights go in times 2^8 (exact; their low halves would be fp16 subnormals otherwise), operands are clamped
// into fp16's range. Against the reference module's output (tests/golden/iispt_net_fixture.npz) and the fp32 module on the CPU the
// network is PER ELEMENT inside north_star's band (|err| <= 1e-4 |want| + 1e-6 max on > 99.9 % of the elements, mean relative
// error 5e-6: the fp32 module's own distance from exact arithmetic), 2e-6 of the largest value at worst — tests/test_iispt_nn.py.
// (Rounds 5's bf16 halves — 16 significant bits — reached 2.4e-5 of the largest value and 96 % of the elements: same instruction
// count, same rate; tools/net_split_emulation.py compares the variants on the CPU.)
//
// Layout. Activations: NHWC fp32, one tensor per layer. Weights: packed once on the host into the matrix instruction's
// B-fragment order, hi and lo, k = (16-channel chunk, tap): 36 KB per (chunk, 64 output channels). A persistent workgroup
// (one per CU; 4 matrix waves + 8 staging waves, k_conv3x3 below) owns a stream of (256-pixel tile, chunk) steps: the staging
// waves bring a step's pixels — with their one-pixel halo, split into fp16 hi / lo rows of 48 bytes — and its packed weights
// into one of two LDS buffers while the matrix waves run the step before out of the other; the matrix waves read LDS only.
// The weights are the matrix instruction's A operand, the pixels its B operand: a lane holds four consecutive channels of its pixel
// per accumulator quad. The sums start at the bias; behind the last chunk of a tile: LeakyReLU(0.2) and the eval-mode BatchNorm2d
// affine in the accumulator registers (parameters in LDS), 16-byte stores. MaxPool2d(2) and Upsample(x2, bilinear) are streaming kernels of their own (k_pool2, k_up2);
// torch.cat costs nothing (a step's 16 channels come from one of two tensors). ConvTranspose2d(k = 3, stride 1, padding 1) is
// the same convolution with the kernel mirrored and its channel axes swapped (done by the packer). Measurements, and the two
// designs this one replaced: DESIGN.md section 4.6.
//
// Timing-only diagnostics (wrong results; tools/net_layers.sh, tools/net_pmc.sh build them as variants):
// -DNET_DIAG_NO_STAGE (the staging waves only keep the barriers), -DNET_DIAG_NO_MFMA, -DNET_DIAG_NO_GLOBAL (no global loads),
// -DNET_DIAG_NO_ARITH / -DNET_DIAG_KEEP_ONLY (the epilogue without its arithmetic / its stores);
// -DNET_DIAG_STAMPS prints where one workgroup's matrix and staging wave spend their cycles (correct results);
// profiles/r05_net_epilogue_study.txt has what they showed.
#include <hip/hip_runtime.h>

#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/iile_gpu.h"

namespace iile {
int api_fail(int code, const std::string &msg);
}

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

enum { PRE_NONE = 0, PRE_CAT = 1 };

// Power-of-two scales of the split operands (exact: they move exponents only). Weights are small (|w| ~ 0.05): w x 2^kWShift keeps
// the LOW half of a weight a normal fp16 number (unscaled it is subnormal with 8 significant bits left: measured, 99.84 % of the
// outputs inside the per-element band instead of 99.996 %, tools/net_split_emulation.py). Activations are O(1..16) and go in as they
// are (kAShift = 0). Every activation tensor in HBM holds 2^(kAShift + kWShift) x the module's value — LeakyReLU, max-pooling and
// bilinear upsampling commute with the factor, bias and BatchNorm shift are uploaded times it, the final 1 x 1 convolution's weights
// divided by it — so the epilogue has no multiply; the staging waves multiply by 2^-kWShift on the way into LDS.
// Range: |activation| < 65504 / 2^kAShift, |weight| < 65504 / 2^kWShift (both clamped, never inf).
constexpr int kAShift = 0, kWShift = 8;
constexpr float kDomain = float(1 << (kAShift + kWShift));   // stored activation / module activation
constexpr float kStageScale = 1.0f / float(1 << kWShift);    // stored activation -> staged operand (2^kAShift x the module's)
constexpr float kF16Max = 65504.0f;

constexpr int kBM = 256;      // pixels per workgroup tile
constexpr int kBN = 64;       // output channels per workgroup tile
constexpr int kChunk = 16;    // input channels per step = one k-step of the matrix instruction per tap
constexpr int kLoaders = 512; // threads of the 8 staging waves (4 stage as fast and leave a wave 256 registers: the experiments' builds)
constexpr int kThreads = 256 + kLoaders;   // 4 matrix waves + 8 staging waves
constexpr int kRowB = kChunk * 2 + 16;   // bytes per staged pixel and plane (48: an odd multiple of 16 -> b128 reads of consecutive pixels hit 16 different bank groups)
constexpr int kKPC = 9;       // k-steps per step (one per tap)
constexpr int kBStep = kKPC * 4 * 64 * 16;   // bytes of packed weights per step and 64-channel n-tile (36 864)

// geometry of one workgroup's pixel tile for H x H images. H = 32: eight rows of one image, with a halo row above and below and a
// halo column either side. H < 32: 256 / H^2 whole images laid out as a G x G grid whose neighbours SHARE their one-pixel
// zero border (H = 4: 16 images in 21 x 21 staged pixels instead of 16 x 6 x 6).
template <int H>
struct Tile {
    static constexpr int ROWS = H >= 32 ? kBM / H : H;            // image rows per tile
    static constexpr int IMGS = H >= 32 ? 1 : kBM / (H * H);      // whole images per tile (H < 32)
    static constexpr int G = IMGS == 1 ? 1 : IMGS == 4 ? 2 : 4;   // images per grid row
    static constexpr int TILES_PER_IMG = H >= 32 ? H / ROWS : 1;
    static constexpr int HP = H >= 32 ? ROWS + 2 : 1 + (H + 1) * G;
    static constexpr int WP = H >= 32 ? H + 2 : 1 + (H + 1) * G;
    static constexpr int NLP = HP * WP;                           // staged pixels per step
    // staged-pixel index of pixel m (0..255) of the tile
    __device__ static int lp_of(int m) {
        if (H >= 32) return (m / H + 1) * WP + m % H + 1;
        int il = m / (H * H), y = (m / H) % H, x = m % H;
        return (1 + (H + 1) * (il / G) + y) * WP + 1 + (H + 1) * (il % G) + x;
    }
    // staged pixel -> (image in tile, y relative to the tile's first row, x); false: a border position (always zero)
    __device__ static bool decode(int lp, int &il, int &y, int &x) {
        int Y = lp / WP, X = lp % WP;
        if (H >= 32) {
            il = 0;
            y = Y - 1;
            x = X - 1;
            return x >= 0 && x < H;   // (rows above / below the image are checked by the caller against y0)
        }
        if (Y % (H + 1) == 0 || X % (H + 1) == 0) return false;
        il = ((Y - 1) / (H + 1)) * G + (X - 1) / (H + 1);
        y = (Y - 1) % (H + 1);
        x = (X - 1) % (H + 1);
        return true;
    }
};

__device__ inline void split_store(f32x4 v, char *hi, char *lo) {
    // a' = a x kStageScale = hi + lo, both fp16 (round to nearest even): 22 significant bits.
    // (a is clamped so that a' stays inside fp16's range: beyond it hi would be inf and lo = a' - inf, a NaN for every sum the pixel feeds)
    constexpr float kClamp = kF16Max / kStageScale;
#ifdef NET_SPLIT_PLAIN   // the same values with conversions and a subtraction (18 vector instructions per four values; A/B witness of the path below)
    v = v * kStageScale;
    v = __builtin_elementwise_min(__builtin_elementwise_max(v, f32x4{-kF16Max, -kF16Max, -kF16Max, -kF16Max}), f32x4{kF16Max, kF16Max, kF16Max, kF16Max});
    const f16x4 h = __builtin_convertvector(v, f16x4);
    const f32x4 r = v - __builtin_convertvector(h, f32x4);
    const f16x4 l = __builtin_convertvector(r, f16x4);
    *reinterpret_cast<uint2 *>(hi) = __builtin_bit_cast(uint2, h);
    *reinterpret_cast<uint2 *>(lo) = __builtin_bit_cast(uint2, l);
#else
    // v_fma_mix: one instruction scales and rounds to fp16 (hi = f16(a s)), one more forms the exact remainder a s - hi in the fused
    // multiply-add and rounds it (lo): 12 vector instructions per four values with the clamps. The staging waves store for ~45 % of a
    // step; the plain sequence cost the network 3 % when the halves became fp16.
    const float s = kStageScale;
    float a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = __builtin_amdgcn_fmed3f(v[q], -kClamp, kClamp);
    uint32_t h[2], l[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[q]) : "v"(a[2 * q]), "s"(s));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[q]) : "v"(a[2 * q + 1]), "s"(s));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l[q]) : "v"(a[2 * q]), "s"(s), "v"(h[q]));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l[q]) : "v"(a[2 * q + 1]), "s"(s), "v"(h[q]));
    }
    *reinterpret_cast<uint2 *>(hi) = make_uint2(h[0], h[1]);
    *reinterpret_cast<uint2 *>(lo) = make_uint2(l[0], l[1]);
#endif
}

__device__ inline f32x4 max4(f32x4 a, f32x4 b) { return __builtin_elementwise_max(a, b); }
// LeakyReLU(0.2) = max(t, 0.2 t) as the bare instruction. fmaxf / elementwise_max quiet their operands first (IEEE maxNum: `v_max x, x, x`
// before every real maximum — and the register allocator runs all of those through ONE temporary, which turns the epilogue's 64
// independent maxima into a serial chain of 128: the disassembly of round 5's kernel). The operands here are sums of products: a
// signalling NaN cannot reach this point.
__device__ inline f32x4 leaky4(f32x4 t) {
    const f32x4 m = 0.2f * t;
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float o;
        asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(t[q]), "v"(m[q]));
        r[q] = o;
    }
    return r;
}
__device__ inline f32x4 lerp4(float wa, f32x4 a, float wb, f32x4 b) { return wa * a + wb * b; }
// the bare maximum of four values (no quieting moves: see leaky4)
__device__ inline f32x4 max4_raw(f32x4 a, f32x4 b) {
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float o;
        asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a[q]), "v"(b[q]));
        r[q] = o;
    }
    return r;
}
// the value of the lane whose number differs in bit 0 (quad_perm [1,0,3,2]), in bit 3 (row_ror:8 inside a row of 16 lanes), in bit 4
// (ds_swizzle, swap of 16-lane halves): register-to-register, no memory
template <int BIT>
__device__ inline f32x4 other_lane4(f32x4 v) {
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float f = v[q];
        const int x = __float_as_int(f);   // (__builtin_bit_cast on the vector ELEMENT reads element 0 four times with this compiler)
        int y;
        if (BIT == 0) y = __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true);
        else if (BIT == 3) y = __builtin_amdgcn_mov_dpp(x, 0x128, 0xf, 0xf, true);
        else y = __builtin_amdgcn_ds_swizzle(x, 0x401F);
        r[q] = __int_as_float(y);
    }
    return r;
}

// nn.Upsample(scale_factor=2, mode="bilinear") (align_corners False): source index and weight of destination d
__device__ inline void up_coord(int d, int n_in, int &i0, int &i1, float &l1) {
    float s = (float(d) + 0.5f) * 0.5f - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = int(s);
    i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
    l1 = s - float(i0);
}

struct ConvArgs {
    const float *in0;      // PRE_NONE: [n][H][H][CIN]; PRE_CAT: the first CIN / 2 channels, [n][H][H][CIN/2]
    const float *in1;      // PRE_CAT: the other CIN / 2 channels (the skip tensor)
    const uint4 *wpack;    // [COUT/64][k-steps][n-tile 2][hi, lo][lane 64] x 16 bytes
    const float *bias, *bn_scale, *bn_shift;   // [COUT]; bn_* only when BNORM
    float *out;            // [n][H][H][COUT]
    float *pool_out;       // POOL: nn.MaxPool2d(2) of `out`, [n][H/2][H/2][COUT] (the next level's input), written by the same epilogue
    int n_img;
};

// One 3 x 3, padding-1 convolution layer with its pre- and post-operations (file header).
//
// Persistent workgroups of 12 waves (kThreads = 768, __launch_bounds__(768, 1): the LDS and register budgets below follow from it),
// one per CU: waves 0-3 issue matrix instructions (each owns 64 pixels x 64 channels of the tile), waves 4-11 stage. The work of a workgroup is a stream of steps (tile, 16-channel chunk). During step s the matrix waves
// run 9 k-steps (one per tap, 12 matrix instructions each) out of LDS buffer s & 1 — A fragments and B fragments both: no
// vector-memory instruction on that side — while the staging waves
//   * convert and store what they loaded during step s - 1 (the A tile of step s + 1, split into fp16 hi / lo; its packed
//     weights, copied as they are) into buffer (s + 1) & 1, and
//   * issue the global loads of step s + 2 into registers, all of them at once: every load has a whole step to land.
// One workgroup barrier per step is the only synchronisation.
template <int H, int CIN, int COUT, int PRE, bool BNORM, bool POOL>
__global__ __launch_bounds__(kThreads, 1) void k_conv3x3(ConvArgs p) {
    using T = Tile<H>;
    constexpr int NCHUNK = CIN / kChunk;
    constexpr int NT = COUT / kBN;
    constexpr int KSTEPS = NCHUNK * kKPC;
    constexpr int PLANE = T::NLP * kRowB;                 // bytes of one plane (hi or lo)
    constexpr int BUF = 2 * PLANE + kBStep;               // bytes of one buffer: A hi, A lo, B
    constexpr int NITEMS = T::NLP * 4;                    // float4 items of an A tile
    constexpr int NA = (NITEMS + kLoaders - 1) / kLoaders;  // ... per staging thread
    constexpr int NB = (kBStep / 16 + kLoaders - 1) / kLoaders;   // 16-byte pieces of B per staging thread
    static_assert(CIN % kChunk == 0 && COUT % kBN == 0, "shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [buffer 2]{A hi [NLP][48], A lo [NLP][48], B [9][4][64][16]}

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave >= 4;
    // the layer's bias and BatchNorm affine behind the two buffers: the epilogue reads them from LDS (a global load there — the first
    // thing a tile's epilogue would wait for — stalls the matrix pipe for its whole round trip, once per tile)
    // (where the buffers leave no room — the 4 x 4 layers with 512 output channels, 16 steps per tile — it stays a global load)
    constexpr bool PAR_LDS = 2 * size_t(BUF) + 3 * size_t(COUT) * 4 <= 160 * 1024;
    float *s_par = reinterpret_cast<float *>(smem + 2 * BUF);   // [bias COUT][scale COUT][shift COUT]
    if (PAR_LDS)
        for (int i = tid; i < COUT; i += kThreads) {
            s_par[i] = p.bias[i];
            s_par[COUT + i] = BNORM ? p.bn_scale[i] : 1.f;
            s_par[2 * COUT + i] = BNORM ? p.bn_shift[i] : 0.f;
        }
    // (the first __syncthreads below — before anybody's first epilogue — publishes them)
    const int n_mtiles = H >= 32 ? p.n_img * T::TILES_PER_IMG : (p.n_img + T::IMGS - 1) / T::IMGS;
    const int n_tiles = ((n_mtiles + 7) / 8) * 8 * NT;   // tile t: group t / (8 NT), n-tile (t % (8 NT)) >> 3, m-tile 8 group + (t & 7)
    const int my_tiles = (n_tiles - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);   // t = blockIdx.x + i gridDim.x: with
    const int n_steps = my_tiles * NCHUNK;   // gridDim.x a multiple of 8, a workgroup's tiles and the n-tiles of a pixel tile share an XCD (speed only)

    auto tile_of = [&](int i, int &mtile, int &ntile) __attribute__((always_inline)) {
        int t = int(blockIdx.x) + i * int(gridDim.x);
        int grp = t / (8 * NT), in_grp = t % (8 * NT);
        mtile = grp * 8 + (in_grp & 7);
        ntile = in_grp >> 3;
    };

    if (loader) {
        // ------------------------------------------------ staging waves ------------------------------------------------
        const int ltid = tid - 256;
        struct Regs {   // what a staging thread holds of one step between its loads and its LDS stores
            f32x4 va[NA];
            u32x4 vb[NB];
        };
        Regs r0, r1;   // steps alternate between the two sets: a step's loads are issued two steps before they are stored (three sets: no faster)

#ifdef NET_DIAG_NO_GLOBAL
#define NET_LD(ptr) (f32x4{float(ltid), 1.f, 2.f, 3.f})
#define NET_LDB(ptr) (u32x4{uint32_t(ltid), 1u, 2u, 3u})
#else
#define NET_LD(ptr) (*reinterpret_cast<const f32x4 *>(ptr))
#define NET_LDB(ptr) (*(ptr))
#endif
        // where this thread's A items sit, once (staged pixel -> image of the tile, row, column is the same every step: only the tile's
        // first image / row and the channel chunk move): element offset from the tile's first pixel, the row and image for the bounds
        int a_rel[NA], a_yl[NA], a_il[NA];   // a_il < 0: a border position (stays zero) or no item
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int it = ltid + kLoaders * u;
            int il = 0, yl = 0, x = 0;
            const bool ok = it < NITEMS && T::decode(it >> 2, il, yl, x);
            constexpr int CS = PRE == PRE_CAT ? CIN / 2 : CIN;
            a_rel[u] = ((il * H + yl) * H + x) * CS + (it & 3) * 4;
            a_yl[u] = yl;
            a_il[u] = ok ? il : -1;
        }
        auto issue = [&](int step, Regs &R) __attribute__((always_inline)) {   // global loads of a step into registers
            int mtile, ntile;
            tile_of(step / NCHUNK, mtile, ntile);
            const int c_base = (step % NCHUNK) * kChunk;
            const int img0 = H >= 32 ? mtile / T::TILES_PER_IMG : mtile * T::IMGS;
            const int y0 = H >= 32 ? (mtile % T::TILES_PER_IMG) * T::ROWS : 0;
            const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(p.wpack) + (size_t(ntile) * KSTEPS + size_t(step % NCHUNK) * kKPC) * 256;
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (ltid + kLoaders * j < kBStep / 16) R.vb[j] = NET_LDB(wsrc + ltid + kLoaders * j);
            // torch.cat((below, skip), 1): which tensor this step's 16 channels come from is uniform per step
            constexpr int CSRC = PRE == PRE_CAT ? CIN / 2 : CIN;
            const bool second = PRE == PRE_CAT && c_base >= CSRC;
            const float *src = (second ? p.in1 : p.in0) + (second ? c_base - CSRC : c_base);
            const float *tile0 = src + (size_t(img0) * H + y0) * H * CSRC;   // the tile's first pixel (uniform)
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                R.va[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int y = y0 + a_yl[u], img = img0 + a_il[u];
                if (a_il[u] < 0 || !(y >= 0 && y < H && img < p.n_img && mtile < n_mtiles)) continue;
                R.va[u] = NET_LD(tile0 + a_rel[u]);
            }
        };
        auto store = [&](int step, Regs &R) __attribute__((always_inline)) {   // registers -> LDS buffer step & 1
            char *s_hi = smem + (step & 1) * BUF, *s_lo = s_hi + PLANE;
            u32x4 *s_b = reinterpret_cast<u32x4 *>(s_hi + 2 * PLANE);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (ltid + kLoaders * j < kBStep / 16) s_b[ltid + kLoaders * j] = R.vb[j];
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const int it = ltid + kLoaders * u;
                if (it >= NITEMS) continue;
                split_store(R.va[u], s_hi + (it >> 2) * kRowB + (it & 3) * 8, s_lo + (it >> 2) * kRowB + (it & 3) * 8);
            }
        };
        // iteration i (beside the matrix waves' step i): request step i + 2 into the set step i used, then store step i + 1
#ifdef NET_DIAG_STAMPS
        unsigned long long lt_issue = 0, lt_store = 0, lt_bar = 0, lt_t = __builtin_amdgcn_s_memtime();
#define NET_LSTAMP(acc_)                                              \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        acc_ += now_ - lt_t;                                          \
        lt_t = now_;                                                  \
    } while (0)
#else
#define NET_LSTAMP(acc_) \
    do {                 \
    } while (0)
#endif
        auto iter = [&](int i, Regs &Ra, Regs &Rb) __attribute__((always_inline)) {
#ifndef NET_DIAG_NO_STAGE
            if (i + 2 < n_steps) issue(i + 2, Ra);
            NET_LSTAMP(lt_issue);
            if (i + 1 < n_steps) store(i + 1, Rb);
            NET_LSTAMP(lt_store);
#endif
            __syncthreads();
            NET_LSTAMP(lt_bar);
        };
#ifndef NET_DIAG_NO_STAGE
        if (n_steps > 0) issue(0, r0);
        if (n_steps > 1) issue(1, r1);
        if (n_steps > 0) store(0, r0);
#endif
        __syncthreads();
        for (int step = 0; step < n_steps; step += 2) {
            iter(step, r0, r1);
            if (step + 1 < n_steps) iter(step + 1, r1, r0);
        }
#ifdef NET_DIAG_STAMPS
        if (blockIdx.x == 17 && ltid == 0)
            printf("k_conv3x3<%d,%d,%d> block 17 staging: %d steps, ticks issue %llu store %llu barrier %llu\n", H, CIN, COUT, n_steps, lt_issue, lt_store, lt_bar);
#endif
        return;
    }

    // ---------------------------------------------------- matrix waves ----------------------------------------------------
    __builtin_amdgcn_s_setprio(1);
    const int r = lane & 31, h = lane >> 5;
    // the two 32-pixel row blocks of this wave: byte offset of each lane's pixel in a plane (tap 0, 0), this lane's k half
    int a_off[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) a_off[mt] = T::lp_of(wave * 64 + mt * 32 + r) * kRowB + h * 16;

    // The accumulators of a tile START at the bias of their channels (where the parameters sit in LDS): the epilogue then has no
    // additions, and the start costs what zeroing did. The epilogue is the one place where the matrix pipe idles by construction — one
    // matrix wave per SIMD runs its vector instructions alone: 8 % of the network's time (timing-only build NET_DIAG_NO_ARITH +
    // NET_DIAG_KEEP_ONLY: 20.7 against 22.6 ms per 8192 probes; a fifth of the 32 x 32 layers, NET_DIAG_STAMPS). Measured and not kept:
    // the same arithmetic woven between the next step's matrix instructions (each vector instruction delays them as much), a second
    // register set stored two pieces per k-step, and the raw sums handed to the staging waves through the LDS buffer the tile's last
    // step has just finished with (those waves are bound by the address path: 58 loads per step) — tools/experiments/r05_net_*.patch.
    f32x16 acc[2][2];
    auto start_tile = [&](int tile_i) __attribute__((always_inline)) {
        int mtile, ntile;
        tile_of(tile_i, mtile, ntile);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
                if (PAR_LDS) b4 = *reinterpret_cast<const f32x4 *>(s_par + ntile * kBN + nt * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[0][nt][4 * g + q] = acc[1][nt][4 * g + q] = b4[q];
            }
    };

    __syncthreads();   // step 0 is staged (and the parameters)
    start_tile(0);
#ifdef NET_DIAG_STAMPS   // timing only: where a matrix wave's time goes (REFCLK ticks): matrix instructions, epilogue, barrier
    unsigned long long st_mfma = 0, st_epi = 0, st_bar = 0, st_t = __builtin_amdgcn_s_memtime();
#define NET_STAMP(acc_)                                              \
    do {                                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        acc_ += now_ - st_t;                                         \
        st_t = now_;                                                 \
    } while (0)
#else
#define NET_STAMP(acc_) \
    do {                \
    } while (0)
#endif

    for (int step = 0; step < n_steps; ++step) {
        const char *s_hi = smem + (step & 1) * BUF, *s_lo = s_hi + PLANE;
        const uint4 *s_b = reinterpret_cast<const uint4 *>(s_hi + 2 * PLANE) + lane;
#ifndef NET_DIAG_NO_MFMA
        f16x8 aq[2][4];    // {m block 0 hi, lo, m block 1 hi, lo}
        uint4 bq[2][4];    // {n half 0 hi, lo, n half 1 hi, lo}
        auto load_ab = [&](int k, f16x8(&da)[4], uint4(&db)[4]) __attribute__((always_inline)) {
            const int toff = ((k / 3 - 1) * T::WP + (k % 3 - 1)) * kRowB;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                da[mt * 2 + 0] = *reinterpret_cast<const f16x8 *>(s_hi + a_off[mt] + toff);
                da[mt * 2 + 1] = *reinterpret_cast<const f16x8 *>(s_lo + a_off[mt] + toff);
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) db[f] = s_b[(k * 4 + f) * 64];
        };
        load_ab(0, aq[0], bq[0]);
#pragma unroll
        for (int k = 0; k < kKPC; ++k) {
            if (k + 1 < kKPC) load_ab(k + 1, aq[(k + 1) & 1], bq[(k + 1) & 1]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f16x8 ah = aq[k & 1][mt * 2], al = aq[k & 1][mt * 2 + 1];
                    const f16x8 bh = __builtin_bit_cast(f16x8, bq[k & 1][nt * 2]), bl = __builtin_bit_cast(f16x8, bq[k & 1][nt * 2 + 1]);
                    // the weights are the instruction's A operand (rows = output channels), the pixels its B operand (columns): a lane
                    // then holds four CONSECUTIVE channels of its pixel per accumulator quad, and the epilogue stores 16 bytes at a time
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[mt][nt], 0, 0, 0);
                }
            // issue order inside the k-step: one LDS read of the next k-step behind each of the first eight matrix
            // instructions (eight reads in a row stall the matrix pipe for the time the LDS takes to accept them)
            if (k + 1 < kKPC) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        NET_STAMP(st_mfma);
        if (step % NCHUNK == NCHUNK - 1) {
            // ---- bias, LeakyReLU(0.2), BatchNorm affine; accumulator column (pixel of the block) = lane & 31, row (channel of the
            // block) = (e & 3) + 8 (e >> 2) + 4 h: per quad e >> 2 four consecutive channels, one 16-byte store ----
            int mtile, ntile;
            tile_of(step / NCHUNK, mtile, ntile);
            const int img0 = H >= 32 ? mtile / T::TILES_PER_IMG : mtile * T::IMGS;
            const int y0 = H >= 32 ? (mtile % T::TILES_PER_IMG) * T::ROWS : 0;
            // (one matrix wave per SIMD runs this alone — the matrix pipe idles meanwhile — so the arithmetic is written four values
            //  wide: four independent adds, multiplies, maxima in a row instead of a dependent chain per value)
            float *dst[2];
            bool live[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int m = wave * 64 + mt * 32 + r;
                const int il = m / (T::ROWS * H), y = y0 + (m / H) % T::ROWS, x = m % H;
                const int img = img0 + il;
                live[mt] = img < p.n_img && mtile < n_mtiles;
                dst[mt] = p.out + ((size_t(img) * H + y) * H + x) * COUT;
            }
            // MaxPool2d(2) of this layer's output in the same registers (POOL: the three encoder levels whose output is both a skip
            // tensor and, pooled, the next level's input). Where the four pixels of a 2 x 2 block sit: H = 32 — a wave's two pixel
            // blocks are rows 2 wave and 2 wave + 1, a lane's pixel is x = r: the other row is the other block of the SAME lane, the
            // other column the lane next door. H = 16 — a block is two rows of 16: lanes r, r ^ 1, r ^ 16. H = 8 — a block is four
            // rows of 8 of one image: lanes r, r ^ 1, r ^ 8. One lane of each four stores the block's 16 bytes per channel quad.
            constexpr int HP2 = H / 2;
            float *pdst[2] = {nullptr, nullptr};
            bool pstore[2] = {false, false};
            if (POOL) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int m = wave * 64 + mt * 32 + r;
                    const int il = m / (T::ROWS * H), y = y0 + (m / H) % T::ROWS, x = m % H;
                    const int img = img0 + il;
                    pdst[mt] = p.pool_out + ((size_t(img) * HP2 + (y >> 1)) * HP2 + (x >> 1)) * COUT;
                    pstore[mt] = live[mt] && (x & 1) == 0 && (y & 1) == 0;   // (H = 32: y is even for block 0, the one that stores)
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = ntile * kBN + nt * 32 + 8 * g + 4 * h;
                    f32x4 v[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (!PAR_LDS) bias = *reinterpret_cast<const f32x4 *>(p.bias + co);
                        f32x4 t = f32x4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
#ifdef NET_DIAG_NO_ARITH   // timing only: the sums consumed as they are
                        v[mt] = t;
#else
                        if (!PAR_LDS) t = t + bias;   // (with the parameters in LDS the sums started at the bias)
                        v[mt] = leaky4(t);   // LeakyReLU(0.2): t for t > 0, 0.2 t below
#endif
                        if (BNORM) {
                            const f32x4 sc = PAR_LDS ? *reinterpret_cast<const f32x4 *>(s_par + COUT + co) : *reinterpret_cast<const f32x4 *>(p.bn_scale + co);
                            const f32x4 sh = PAR_LDS ? *reinterpret_cast<const f32x4 *>(s_par + 2 * COUT + co) : *reinterpret_cast<const f32x4 *>(p.bn_shift + co);
                            v[mt] = v[mt] * sc + sh;
                        }
#ifdef NET_DIAG_KEEP_ONLY   // timing only: the arithmetic kept alive, nothing written
                        asm volatile("" ::"v"(v[mt]));
#else
                        if (live[mt]) *reinterpret_cast<f32x4 *>(dst[mt] + co) = v[mt];
#endif
                    }
                    if (POOL) {
                        if (H >= 32) {
                            f32x4 m4 = max4_raw(v[0], v[1]);
                            m4 = max4_raw(m4, other_lane4<0>(m4));
                            if (pstore[0]) *reinterpret_cast<f32x4 *>(pdst[0] + co) = m4;
                        } else {
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt) {
                                f32x4 m4 = max4_raw(v[mt], other_lane4<0>(v[mt]));
                                m4 = max4_raw(m4, H == 16 ? other_lane4<4>(m4) : other_lane4<3>(m4));
                                if (pstore[mt]) *reinterpret_cast<f32x4 *>(pdst[mt] + co) = m4;
                            }
                        }
                    }
                }
            start_tile(step / NCHUNK + 1);   // (the workgroup's tile after its last: parameters of some n-tile, never used)
        }
        NET_STAMP(st_epi);
        __syncthreads();   // buffer step & 1 may be refilled; buffer (step + 1) & 1 is staged
        NET_STAMP(st_bar);
    }
#ifdef NET_DIAG_STAMPS
    if (blockIdx.x == 17 && tid == 0)
        printf("k_conv3x3<%d,%d,%d> block 17: %d steps, ticks mfma %llu epilogue %llu barrier %llu\n", H, CIN, COUT, n_steps, st_mfma, st_epi, st_bar);
#endif
}

// nn.MaxPool2d(2): [n][2H][2H][C] -> [n][H][H][C]; one thread per output pixel and four channels
template <int H, int C>
__global__ void k_pool2(const float *in, float *out, int n) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(n) * H * H * (C / 4)) return;
    int c4 = int(i % (C / 4));
    size_t px = i / (C / 4);
    int x = int(px % H), y = int((px / H) % H);
    size_t img = px / (H * H);
    const float *q = in + ((img * 2 * H + 2 * y) * 2 * H + 2 * x) * C + c4 * 4;
    f32x4 a = *reinterpret_cast<const f32x4 *>(q), b = *reinterpret_cast<const f32x4 *>(q + C);
    f32x4 c = *reinterpret_cast<const f32x4 *>(q + 2 * H * C), d = *reinterpret_cast<const f32x4 *>(q + 2 * H * C + C);
    *reinterpret_cast<f32x4 *>(out + px * C + c4 * 4) = max4(max4(a, b), max4(c, d));
}

// nn.Upsample(scale_factor=2, mode="bilinear"): [n][H/2][H/2][C] -> [n][H][H][C]
template <int H, int C>
__global__ void k_up2(const float *in, float *out, int n) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(n) * H * H * (C / 4)) return;
    constexpr int HL = H / 2;
    int c4 = int(i % (C / 4));
    size_t px = i / (C / 4);
    int x = int(px % H), y = int((px / H) % H);
    size_t img = px / (H * H);
    int ya, yb, xa, xb;
    float ly, lx;
    up_coord(y, HL, ya, yb, ly);
    up_coord(x, HL, xa, xb, lx);
    const float *q = in + img * HL * HL * C + c4 * 4;
    f32x4 a = *reinterpret_cast<const f32x4 *>(q + (ya * HL + xa) * C), b = *reinterpret_cast<const f32x4 *>(q + (ya * HL + xb) * C);
    f32x4 c = *reinterpret_cast<const f32x4 *>(q + (yb * HL + xa) * C), d = *reinterpret_cast<const f32x4 *>(q + (yb * HL + xb) * C);
    f32x4 top = lerp4(1.f - lx, a, lx, b), bot = lerp4(1.f - lx, c, lx, d);
    *reinterpret_cast<f32x4 *>(out + px * C + c4 * 4) = lerp4(1.f - ly, top, ly, bot);
}

// (n, 7, 32, 32) as the network sees it (read_input, ml/main_stdio_net.py:47-72) -> NHWC with the channels padded to 16
__global__ void k_net_input(const float *in, float *x16, int n) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;   // one thread per pixel
    if (i >= size_t(n) * 1024) return;
    size_t img = i >> 10, px = i & 1023;
    float v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = c < 7 ? in[(img * 7 + c) * 1024 + px] * kDomain : 0.f;   // (activation tensors hold kDomain x the module's values)
    float4 *o = reinterpret_cast<float4 *>(x16 + i * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// test probe: a stored activation tensor / kDomain
__global__ void k_net_unscale(const float *in, float *out, size_t n4) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n4) reinterpret_cast<f32x4 *>(out)[i] = reinterpret_cast<const f32x4 *>(in)[i] * (1.0f / kDomain);
}

// decoder2's Conv2d(K, 3, 1) + ReLU: NHWC 64 channels -> (n, 3, 32, 32). 16 lanes per pixel, 4 channels each.
__global__ void k_net_output(const float *in, const float *w /* [3][64] */, const float *bias, float *out, int n) {
    size_t t = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t pix = t >> 4;
    int q = int(t & 15);
    bool live = pix < size_t(n) * 1024;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (live) v = *reinterpret_cast<const f32x4 *>(in + pix * 64 + q * 4);
    float s[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const float *wo = w + o * 64 + q * 4;
        s[o] = v.x * wo[0] + v.y * wo[1] + v.z * wo[2] + v.w * wo[3];
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) s[o] += __shfl_xor(s[o], d, 16);
    }
    if (live && q < 3) {
        float r = (q == 0 ? s[0] : q == 1 ? s[1] : s[2]) + bias[q];
        size_t img = pix >> 10, px = pix & 1023;
        out[(img * 3 + q) * 1024 + px] = r > 0.f ? r : 0.f;
    }
}

// ---- the two transforms the runner applies around the network, one workgroup per probe ----
__device__ inline double block_sum(double v, double *s_red) {   // 256 threads; every thread gets the sum
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// normalizeMapsDownstream (src/integrators/iisptrenderrunner.cpp:1041-1092 over ImageFilm's operations, src/film/imagefilm.cpp:
// 203-254, 298-379) + read_input's layout (ml/main_stdio_net.py:47-72): the probe's intensity / normals / distance images in
// raster order -> the network's input, NHWC padded to 16 channels, in ImageFilm row order (row h - 1 - y), and the probe's
// channel means for transformMapsUpstream. Double sums for the means, log(1.0 + v) in double, everything else float.
__global__ __launch_bounds__(256) void k_net_normalize(const float *inten, const float *nrm, const float *dist, float *x16, float *chan_mean, int n) {
    __shared__ double s_red[4];
    const int probe = blockIdx.x, t = threadIdx.x;
    const float *ip = inten + size_t(probe) * 3072, *np_ = nrm + size_t(probe) * 3072, *dp = dist + size_t(probe) * 1024;
    double sc[3] = {0, 0, 0}, sd = 0;
    // (a non-finite probe value — the probe pass guards its radiance as SamplerIntegrator::Render does, so none is expected — counts as 0,
    //  the film's own rule for such samples: one inf would otherwise turn the probe's means, and with them the whole prediction, into NaN)
    auto fin = [](float v) { return __builtin_isfinite(v) ? v : 0.f; };
    for (int px = t; px < 1024; px += 256) {
        sc[0] += double(fin(ip[3 * px]));
        sc[1] += double(fin(ip[3 * px + 1]));
        sc[2] += double(fin(ip[3 * px + 2]));
        sd += double(fin(dp[px]));
    }
    const double s0 = block_sum(sc[0], s_red), s1 = block_sum(sc[1], s_red), s2 = block_sum(sc[2], s_red), s3 = block_sum(sd, s_red);
    const float mean = float((s0 + s1 + s2) / 3072.0);
    const float ratio = mean == 0.f ? 0.f : float(1.0 / (10.0 * double(mean)));
    const float z_mean = float(s3 / 1024.0);
    float div = float(10.0 * (double(z_mean) + 1.0));
    if (div == 0.f) div = 1.f;
    const float rdiv = float(1.0 / double(div));
    if (t < 3) chan_mean[size_t(probe) * 3 + t] = float((t == 0 ? s0 : t == 1 ? s1 : s2) / 1024.0);
    for (int px = t; px < 1024; px += 256) {
        const int y = px >> 5, x = px & 31;
        float v[8];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = fin(ip[3 * px + c]) * ratio;
            a = a > 0.f ? a : 0.f;
            v[c] = (float(log(1.0 + double(a))) + -0.1f) * kDomain;
            float b = np_[3 * px + c];
            v[3 + c] = (b < -1.f ? -1.f : (b > 1.f ? 1.f : (b == b ? b : 0.f))) * kDomain;
        }
        float d = (fin(dp[px]) + 1.0f) * rdiv;
        d = d > 0.f ? d : 0.f;
        v[6] = (float(log(1.0 + double(d))) + -0.1f) * kDomain;
        v[7] = 0.f;
        f32x4 *o = reinterpret_cast<f32x4 *>(x16 + ((size_t(probe) * 32 + (31 - y)) * 32 + x) * 16);
        o[0] = f32x4{v[0], v[1], v[2], v[3]};
        o[1] = f32x4{v[4], v[5], v[6], v[7]};
        o[2] = o[3] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// decoder2's Conv2d(K, 3, 1) + ReLU, then transformMapsUpstream (iisptrenderrunner.cpp:1095-1133): exp(v) - 1 in double, every
// channel rescaled so that its mean is the rendered probe's. pred: (n, 32, 32, 3); film_rows != 0: rows as the network emits
// them (ImageFilm order: what iile_iispt_gather reads), else raster order (row h - 1 - y).
__global__ __launch_bounds__(256) void k_net_predict_out(const float *in, const float *w /* [3][64] */, const float *bias, const float *chan_mean,
                                                          float *pred, const int32_t *slot, int film_rows, int n) {
    __shared__ double s_red[4];
    __shared__ float s_w[192];
    const int probe = blockIdx.x, t = threadIdx.x;
    if (t < 192) s_w[t] = w[t];
    __syncthreads();
    float e[4][3];
    double sum[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int px = t + 256 * k;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(in + (size_t(probe) * 1024 + px) * 64);
        float acc[3] = {0.f, 0.f, 0.f};
        for (int q = 0; q < 16; ++q) {
            const f32x4 a = src[q];
#pragma unroll
            for (int o = 0; o < 3; ++o) acc[o] += a.x * s_w[o * 64 + 4 * q] + a.y * s_w[o * 64 + 4 * q + 1] + a.z * s_w[o * 64 + 4 * q + 2] + a.w * s_w[o * 64 + 4 * q + 3];
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float r = acc[o] + bias[o];
            r = r > 0.f ? r : 0.f;                       // the network's final ReLU (and positiveLogInverse's clamp)
            e[k][o] = float(exp(double(r)) - 1.0);
            sum[o] += double(e[k][o]);
        }
    }
    float mul[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const float actual = float(block_sum(sum[o], s_red) / 1024.0);
        mul[o] = actual > 1e-10f ? chan_mean[size_t(probe) * 3 + o] / actual : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int px = t + 256 * k, y = px >> 5, x = px & 31;
        float *o = pred + ((size_t(slot ? slot[probe] : probe) * 32 + (film_rows ? y : 31 - y)) * 32 + x) * 3;
        o[0] = e[k][0] * mul[0];
        o[1] = e[k][1] * mul[1];
        o[2] = e[k][2] * mul[2];
    }
}

// ---- host side ----

struct LayerDef {
    int h, cin, cout, pre, chunk;
    bool bn, deconv;
    int cin_real;   // channels the checkpoint has (7 for the first layer, padded to 16 with zero weights)
};
// forward order (ml/iispt_net.py:27-88): encoder0.0, encoder0.2, encoder1.1, encoder1.4, encoder2.1, encoder2.4, encoder3.1,
// encoder3.4, decoder0.0, decoder0.3, decoder1.0, decoder1.3, decoder2.0, decoder2.2 (decoder2.4 is k_net_output)
const LayerDef kLayers[14] = {
    {32, 16, 64, PRE_NONE, 16, false, false, 7},  {32, 64, 64, PRE_NONE, 16, false, false, 64},
    {16, 64, 128, PRE_NONE, 16, true, false, 64}, {16, 128, 128, PRE_NONE, 16, false, false, 128},
    {8, 128, 256, PRE_NONE, 16, true, false, 128}, {8, 256, 256, PRE_NONE, 16, false, false, 256},
    {4, 256, 512, PRE_NONE, 16, true, false, 256}, {4, 512, 256, PRE_NONE, 16, false, false, 512},
    {8, 512, 256, PRE_CAT, 16, true, true, 512}, {8, 256, 128, PRE_NONE, 16, false, true, 256},
    {16, 256, 128, PRE_CAT, 16, true, true, 256}, {16, 128, 64, PRE_NONE, 16, false, true, 128},
    {32, 128, 64, PRE_CAT, 16, false, true, 128}, {32, 64, 64, PRE_NONE, 16, false, true, 64},
};
const int kBnOfLayer[14] = {-1, -1, 0, -1, 1, -1, 2, -1, 3, -1, 4, -1, -1, -1};

// fp16 halves of a (scaled) weight on the host: round to nearest even through the compiler's _Float16; out-of-range weights saturate
void f16_split(float v, uint16_t *hi, uint16_t *lo) {
    v = std::fmin(std::fmax(v, -kF16Max), kF16Max);
    if (!(v == v)) v = 0.f;
    const _Float16 h = static_cast<_Float16>(v);
    const _Float16 l = static_cast<_Float16>(v - static_cast<float>(h));
    std::memcpy(hi, &h, 2);
    std::memcpy(lo, &l, 2);
}

template <int H, int CIN, int COUT, int PRE, bool BNORM, bool POOL = false>
hipError_t launch_conv(const ConvArgs &a, int n_cus, bool *attr_set, hipStream_t s) {
    using T = Tile<H>;
    constexpr size_t buffers = 2 * (size_t(T::NLP) * kRowB * 2 + kBStep);   // two buffers of {A hi, A lo, B}
    constexpr size_t lds = buffers + 3 * size_t(COUT) * 4 <= 160 * 1024 ? buffers + 3 * size_t(COUT) * 4 : buffers;   // + bias, scale, shift where they fit
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = k_conv3x3<H, CIN, COUT, PRE, BNORM, POOL>;
    if (!*attr_set) {   // once per network object, i.e. per device the object was made on (the attribute is per device)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
        *attr_set = true;
    }
    int n_mtiles = H >= 32 ? a.n_img * T::TILES_PER_IMG : (a.n_img + T::IMGS - 1) / T::IMGS;
    int tiles = ((n_mtiles + 7) / 8) * 8 * (COUT / kBN);
    int blocks = tiles < n_cus ? tiles : n_cus;   // one persistent workgroup per CU (a multiple of 8 either way)
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kThreads), lds, s, a);
    return hipGetLastError();
}

// MaxPool2d(2) behind encoder0 / 1 / 2's second convolution: in that convolution's epilogue (default), or k_pool2 (-DNET_POOL_KERNEL:
// the A/B witness — same bits)
#ifdef NET_POOL_KERNEL
constexpr bool kPoolInEpilogue = false;
#else
constexpr bool kPoolInEpilogue = true;
#endif

hipError_t launch_layer(int layer, const ConvArgs &a, int n_cus, bool *attr_set, hipStream_t s) {
    switch (layer) {
        case 0: return launch_conv<32, 16, 64, PRE_NONE, false>(a, n_cus, attr_set, s);
        case 1: return launch_conv<32, 64, 64, PRE_NONE, false, kPoolInEpilogue>(a, n_cus, attr_set, s);
        case 2: return launch_conv<16, 64, 128, PRE_NONE, true>(a, n_cus, attr_set, s);
        case 3: return launch_conv<16, 128, 128, PRE_NONE, false, kPoolInEpilogue>(a, n_cus, attr_set, s);
        case 4: return launch_conv<8, 128, 256, PRE_NONE, true>(a, n_cus, attr_set, s);
        case 5: return launch_conv<8, 256, 256, PRE_NONE, false, kPoolInEpilogue>(a, n_cus, attr_set, s);
        case 6: return launch_conv<4, 256, 512, PRE_NONE, true>(a, n_cus, attr_set, s);
        case 7: return launch_conv<4, 512, 256, PRE_NONE, false>(a, n_cus, attr_set, s);
        case 8: return launch_conv<8, 512, 256, PRE_CAT, true>(a, n_cus, attr_set, s);
        case 9: return launch_conv<8, 256, 128, PRE_NONE, false>(a, n_cus, attr_set, s);
        case 10: return launch_conv<16, 256, 128, PRE_CAT, true>(a, n_cus, attr_set, s);
        case 11: return launch_conv<16, 128, 64, PRE_NONE, false>(a, n_cus, attr_set, s);
        case 12: return launch_conv<32, 128, 64, PRE_CAT, false>(a, n_cus, attr_set, s);
        case 13: return launch_conv<32, 64, 64, PRE_NONE, false>(a, n_cus, attr_set, s);
    }
    return hipErrorInvalidValue;
}

// floats per probe of each activation tensor; buffer assignment: two ping-pong buffers P and Q, R for the pooled / upsampled
// copies, and the three skip tensors
enum { BUF_P = 0, BUF_Q = 1, BUF_R = 2, BUF_E0 = 3, BUF_E1 = 4, BUF_E2 = 5, N_BUF = 6 };
const size_t kBufFloats[N_BUF] = {65536, 65536, 65536, 65536, 32768, 16384};
// per layer: {in0, in1 (-1: none), out, what runs on `out` afterwards into R: 0 nothing, 1 MaxPool2d(2), 2 Upsample(x2)};
// the network input (padded to 16 channels) sits in P
const int kRoute[14][4] = {
    {BUF_P, -1, BUF_Q, 0},      {BUF_Q, -1, BUF_E0, 1}, {BUF_R, -1, BUF_P, 0},      {BUF_P, -1, BUF_E1, 1}, {BUF_R, -1, BUF_P, 0},
    {BUF_P, -1, BUF_E2, 1},     {BUF_R, -1, BUF_P, 0},  {BUF_P, -1, BUF_Q, 2},      {BUF_R, BUF_E2, BUF_P, 0}, {BUF_P, -1, BUF_Q, 2},
    {BUF_R, BUF_E1, BUF_P, 0},  {BUF_P, -1, BUF_Q, 2},  {BUF_R, BUF_E0, BUF_P, 0},  {BUF_P, -1, BUF_Q, 0},
};

template <int H, int C>
hipError_t launch_pool(const float *in, float *out, int n, hipStream_t s) {   // H: output size
    size_t items = size_t(n) * H * H * (C / 4);
    hipLaunchKernelGGL((k_pool2<H, C>), dim3((items + 255) / 256), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}
template <int H, int C>
hipError_t launch_up(const float *in, float *out, int n, hipStream_t s) {   // H: output size
    size_t items = size_t(n) * H * H * (C / 4);
    hipLaunchKernelGGL((k_up2<H, C>), dim3((items + 255) / 256), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}
// the resampling step behind convolution layer l (kRoute[l][3])
hipError_t launch_resample(int layer, const float *in, float *out, int n, hipStream_t s) {
    switch (layer) {
        case 1: return launch_pool<16, 64>(in, out, n, s);
        case 3: return launch_pool<8, 128>(in, out, n, s);
        case 5: return launch_pool<4, 256>(in, out, n, s);
        case 7: return launch_up<8, 256>(in, out, n, s);
        case 9: return launch_up<16, 128>(in, out, n, s);
        case 11: return launch_up<32, 64>(in, out, n, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

struct iile_iispt_net {
    uint4 *wpack[14] = {};
    float *bias[14] = {};
    float *bn_scale[5] = {}, *bn_shift[5] = {};
    float *w_out = nullptr, *b_out = nullptr;
    float *ws = nullptr;      // activations of the current batch
    size_t ws_floats_per_probe = 0;
    int ws_probes = 0;
    bool attr_set[14] = {};   // the dynamic-LDS attribute of layer l's kernel has been raised on this object's device
    int n_cus = 256;          // persistent grid of the convolution kernels (rounded down to a multiple of 8)
    std::vector<void *> allocs;
};

#define NET_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return iile::api_fail(IILE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {

int net_upload(iile_iispt_net *net, const void *host, size_t bytes, void **dev) {
    void *p = nullptr;
    NET_TRY(hipMalloc(&p, bytes));
    net->allocs.push_back(p);
    NET_TRY(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
    *dev = p;
    return IILE_OK;
}

// B fragments of v_mfma_f32_32x32x16_f16: lane l holds B[k = 8 (l >> 5) + j][column l & 31], j = 0..7; weights times 2^kWShift
std::vector<uint16_t> pack_weights(const LayerDef &L, const float *w) {
    const int sub = L.chunk / 16, nchunk = L.cin / L.chunk, ksteps = nchunk * 9 * sub, ntiles = L.cout / kBN;
    std::vector<uint16_t> out(size_t(ntiles) * ksteps * 256 * 8);
    for (int nt64 = 0; nt64 < ntiles; ++nt64)
        for (int chunk = 0; chunk < nchunk; ++chunk)
            for (int tap = 0; tap < 9; ++tap)
                for (int s = 0; s < sub; ++s) {
                    int ks = (chunk * 9 + tap) * sub + s;
                    int ky = tap / 3, kx = tap % 3;
                    for (int nt = 0; nt < 2; ++nt)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                int ci = chunk * L.chunk + s * 16 + 8 * (lane >> 5) + j;
                                int co = nt64 * kBN + nt * 32 + (lane & 31);
                                float v = 0.f;
                                if (ci < L.cin_real) {
                                    if (!L.deconv)   // Conv2d: weight[co][ci][ky][kx]
                                        v = w[((size_t(co) * L.cin_real + ci) * 3 + ky) * 3 + kx];
                                    else             // ConvTranspose2d: weight[ci][co][ky][kx], mirrored
                                        v = w[((size_t(ci) * L.cout + co) * 3 + (2 - ky)) * 3 + (2 - kx)];
                                }
                                uint16_t hi, lo;
                                f16_split(v * float(1 << kWShift), &hi, &lo);
                                size_t base = ((size_t(nt64) * ksteps + ks) * 4 + nt * 2) * 64 * 8;
                                out[base + size_t(lane) * 8 + j] = hi;
                                out[base + 64 * 8 + size_t(lane) * 8 + j] = lo;
                            }
                }
    return out;
}

// The activation workspace for up to *cap probes per set of launches (1.19 MiB + 16 B each). The batch is a speed knob only (results do
// not depend on it, tests/test_iispt_nn.py): if the device cannot give that much — a smaller card, or one shared with another job — the
// batch is halved until the allocation succeeds (down to 8 probes) and *cap says what was got. IILE_NET_WORKSPACE_MB caps the
// allocation from outside (tests use it to walk this path on a 288 GB device).
constexpr int kDefaultBatch = 8192;   // probes per set of launches when the caller names none: 9.5 GiB of activations; no faster beyond (tools/net_check.py)
int ensure_workspace(iile_iispt_net *net, int *cap) {
    if (*cap <= net->ws_probes) return IILE_OK;
    if (net->ws) (void)hipFree(net->ws);
    net->ws = nullptr;
    net->ws_probes = 0;
    size_t per = 0;
    for (size_t f : kBufFloats) per += f;
    net->ws_floats_per_probe = per;
    size_t limit = ~size_t(0);
    if (const char *e = std::getenv("IILE_NET_WORKSPACE_MB")) limit = size_t(std::max(1.0, atof(e)) * 1048576.0);
    for (int n = *cap;; n = (n + 1) / 2) {
        const size_t bytes = (per + 4) * sizeof(float) * size_t(n);   // + the channel means of a probe
        hipError_t e = bytes > limit ? hipErrorOutOfMemory : hipMalloc(reinterpret_cast<void **>(&net->ws), bytes);
        if (e == hipSuccess) {
            net->ws_probes = n;
            *cap = n;
            return IILE_OK;
        }
        (void)hipGetLastError();
        net->ws = nullptr;
        if (e != hipErrorOutOfMemory || n <= 8)
            return iile::api_fail(IILE_ERR_HIP, std::string("iile_iispt_net: no room for the activations of ") + std::to_string(n) + " probes (" +
                                                    std::to_string(bytes >> 20) + " MiB): " + hipGetErrorString(e));
    }
}

// n probes in sets of at most cap: as many sets as that needs, all of (nearly) the same size — 25 058 probes under a cap of 8 192 run as
// 4 x 6 265, not 3 x 8 192 + 482 (a set of 482 leaves most of the persistent grid idle for 14 layers: measured + 6 % on the frame)
int even_batch(int n, int cap) {
    if (n <= cap) return cap;
    const int sets = (n + cap - 1) / cap;
    return (n + sets - 1) / sets;
}

float *buffer_of(iile_iispt_net *net, int buf, int n_alloc) {
    size_t off = 0;
    for (int i = 0; i < buf; ++i) off += kBufFloats[i] * size_t(n_alloc);
    return net->ws + off;
}

}  // namespace

extern "C" {

int iile_iispt_net_create(const iile_iispt_net_weights *w, iile_iispt_net **out) {
    if (!w || !out) return iile::api_fail(IILE_ERR_ARG, "iile_iispt_net_create: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return iile::api_fail(IILE_ERR_NO_DEVICE, "iile_iispt_net_create: no HIP device (the network has no CPU fallback)");
    for (int i = 0; i < 15; ++i)
        if (!w->conv_weight[i] || !w->conv_bias[i]) return iile::api_fail(IILE_ERR_ARG, "iile_iispt_net_create: missing convolution tensor");
    auto *net = new iile_iispt_net();
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8)
            net->n_cus = prop.multiProcessorCount / 8 * 8;
    }
    auto bail = [&](int rc) {
        iile_iispt_net_destroy(net);
        return rc;
    };
    for (int l = 0; l < 14; ++l) {
        std::vector<uint16_t> pk = pack_weights(kLayers[l], w->conv_weight[l]);
        int rc = net_upload(net, pk.data(), pk.size() * 2, reinterpret_cast<void **>(&net->wpack[l]));
        if (rc) return bail(rc);
        // (activation tensors hold kDomain x the module's values: so do the accumulators' starting values and the BatchNorm shift)
        std::vector<float> b_scaled(w->conv_bias[l], w->conv_bias[l] + kLayers[l].cout);
        for (float &b : b_scaled) b *= kDomain;
        rc = net_upload(net, b_scaled.data(), size_t(kLayers[l].cout) * 4, reinterpret_cast<void **>(&net->bias[l]));
        if (rc) return bail(rc);
        int bn = kBnOfLayer[l];
        if (bn >= 0) {
            // BatchNorm2d in eval mode: y = (x - running_mean) / sqrt(running_var + eps) * weight + bias
            int c = kLayers[l].cout;
            std::vector<float> sc(c), sh(c);
            for (int i = 0; i < c; ++i) {
                float inv = 1.0f / std::sqrt(w->bn_var[bn][i] + w->bn_eps);
                sc[i] = inv * w->bn_weight[bn][i];
                sh[i] = (w->bn_bias[bn][i] - w->bn_mean[bn][i] * sc[i]) * kDomain;
            }
            rc = net_upload(net, sc.data(), size_t(c) * 4, reinterpret_cast<void **>(&net->bn_scale[bn]));
            if (rc) return bail(rc);
            rc = net_upload(net, sh.data(), size_t(c) * 4, reinterpret_cast<void **>(&net->bn_shift[bn]));
            if (rc) return bail(rc);
        }
    }
    std::vector<float> w_out(w->conv_weight[14], w->conv_weight[14] + 3 * 64);
    for (float &v : w_out) v *= 1.0f / kDomain;   // the 1 x 1 convolution reads kDomain x its input: exact either way (a power of two)
    int rc = net_upload(net, w_out.data(), 3 * 64 * 4, reinterpret_cast<void **>(&net->w_out));
    if (rc) return bail(rc);
    rc = net_upload(net, w->conv_bias[14], 3 * 4, reinterpret_cast<void **>(&net->b_out));
    if (rc) return bail(rc);
    *out = net;
    return IILE_OK;
}

// The weights as a flat file (what a host without Python hands over; written by binding.save_net_weights from a state_dict of the
// reference's ml/ training): "IILENET1", BatchNorm2d's eps, then float32 tensors in iile_iispt_net_weights' order — the 15
// convolutions {weight, bias}, the 5 batch norms {weight, bias, running_mean, running_var} — in the checkpoint's own shapes.
int iile_iispt_net_load(const char *path, iile_iispt_net **out) {
    if (!path || !out) return iile::api_fail(IILE_ERR_ARG, "iile_iispt_net_load: null argument");
    FILE *f = std::fopen(path, "rb");
    if (!f) return iile::api_fail(IILE_ERR_ARG, (std::string("iile_iispt_net_load: cannot open ") + path).c_str());
    std::vector<std::vector<float>> t;
    char magic[8];
    float eps = 0.f;
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "IILENET1", 8) == 0 && std::fread(&eps, 4, 1, f) == 1;
    auto tensor = [&](size_t n) {
        t.emplace_back(n);
        ok = ok && std::fread(t.back().data(), 4, n, f) == n;
        return t.size() - 1;
    };
    size_t conv_w[15], conv_b[15], bn[5][4];
    for (int l = 0; l < 15 && ok; ++l) {
        const size_t cout = l < 14 ? size_t(kLayers[l].cout) : 3, cin = l < 14 ? size_t(kLayers[l].cin_real) : 64, k = l < 14 ? 9 : 1;
        conv_w[l] = tensor(cout * cin * k);
        conv_b[l] = tensor(cout);
    }
    for (int l = 0, b = 0; l < 14 && ok; ++l)
        if (kBnOfLayer[l] >= 0) {
            for (int q = 0; q < 4; ++q) bn[b][q] = tensor(size_t(kLayers[l].cout));
            ++b;
        }
    char extra;
    ok = ok && std::fread(&extra, 1, 1, f) == 0;   // exactly these tensors: a file of another architecture is refused
    std::fclose(f);
    if (!ok) return iile::api_fail(IILE_ERR_ARG, (std::string("iile_iispt_net_load: ") + path + " is not an IILENET1 file of IISPTNet's tensors").c_str());
    iile_iispt_net_weights w = {};
    for (int l = 0; l < 15; ++l) w.conv_weight[l] = t[conv_w[l]].data(), w.conv_bias[l] = t[conv_b[l]].data();
    for (int b = 0; b < 5; ++b)
        w.bn_weight[b] = t[bn[b][0]].data(), w.bn_bias[b] = t[bn[b][1]].data(), w.bn_mean[b] = t[bn[b][2]].data(), w.bn_var[b] = t[bn[b][3]].data();
    w.bn_eps = eps;
    return iile_iispt_net_create(&w, out);
}

void iile_iispt_net_destroy(iile_iispt_net *net) {
    if (!net) return;
    for (void *p : net->allocs) (void)hipFree(p);
    if (net->ws) (void)hipFree(net->ws);
    delete net;
}

}  // extern "C"

namespace {
// the 14 convolution layers (and the resampling kernels between them) over the nb probes whose input sits in buffer P
int run_layers(iile_iispt_net *net, int nb, int na, hipStream_t s, float *layer_out_dev, int layer, size_t first) {
    for (int l = 0; l < 14; ++l) {
        ConvArgs a{};
        a.in0 = buffer_of(net, kRoute[l][0], na);
        a.in1 = kRoute[l][1] >= 0 ? buffer_of(net, kRoute[l][1], na) : nullptr;
        a.out = buffer_of(net, kRoute[l][2], na);
        a.wpack = net->wpack[l];
        a.bias = net->bias[l];
        int bn = kBnOfLayer[l];
        a.bn_scale = bn >= 0 ? net->bn_scale[bn] : nullptr;
        a.bn_shift = bn >= 0 ? net->bn_shift[bn] : nullptr;
        a.n_img = nb;
        const bool pooled_here = kPoolInEpilogue && kRoute[l][3] == 1;
        a.pool_out = pooled_here ? buffer_of(net, BUF_R, na) : nullptr;
        NET_TRY(launch_layer(l, a, net->n_cus, &net->attr_set[l], s));
        if (kRoute[l][3] && !pooled_here) NET_TRY(launch_resample(l, a.out, buffer_of(net, BUF_R, na), nb, s));
        if (layer_out_dev && l == layer) {   // test probe: this layer's NHWC activations, as the module has them
            size_t fl = size_t(kLayers[l].h) * kLayers[l].h * kLayers[l].cout;
            hipLaunchKernelGGL(k_net_unscale, dim3(unsigned((fl * size_t(nb) / 4 + 255) / 256)), dim3(256), 0, s, a.out, layer_out_dev + first * fl, fl * size_t(nb) / 4);
            NET_TRY(hipGetLastError());
        }
    }
    return IILE_OK;
}
}  // namespace

extern "C" {

int iile_iispt_net_forward(iile_iispt_net *net, const float *in_dev, float *out_dev, int32_t n, int32_t max_batch, void *stream,
                           float *layer_out_dev, int32_t layer) {
    if (!net || !in_dev || !out_dev || n < 0) return iile::api_fail(IILE_ERR_ARG, "iile_iispt_net_forward: bad argument");
    if (layer_out_dev && (layer < 0 || layer > 13)) return iile::api_fail(IILE_ERR_ARG, "iile_iispt_net_forward: layer out of range");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (max_batch <= 0) max_batch = kDefaultBatch;
    int cap = n < max_batch ? n : max_batch;
    if (n == 0) return IILE_OK;
    int rc = ensure_workspace(net, &cap);
    if (rc) return rc;
    const int na = net->ws_probes;
    cap = even_batch(n, cap);
    for (int first = 0; first < n; first += cap) {
        const int nb = n - first < cap ? n - first : cap;
        hipLaunchKernelGGL(k_net_input, dim3((size_t(nb) * 1024 + 255) / 256), dim3(256), 0, s, in_dev + size_t(first) * 7 * 1024,
                           buffer_of(net, BUF_P, na), nb);
        NET_TRY(hipGetLastError());
        rc = run_layers(net, nb, na, s, layer_out_dev, layer, size_t(first));
        if (rc) return rc;
        hipLaunchKernelGGL(k_net_output, dim3((size_t(nb) * 1024 * 16 + 255) / 256), dim3(256), 0, s, buffer_of(net, BUF_Q, na), net->w_out,
                           net->b_out, out_dev + size_t(first) * 3 * 1024, nb);
        NET_TRY(hipGetLastError());
    }
    return IILE_OK;
}

int iile_iispt_net_predict(iile_iispt_net *net, const float *intensity_dev, const float *normals_dev, const float *distance_dev,
                           float *pred_dev, const int32_t *slot_of_probe_dev, int32_t n, int32_t film_rows, int32_t max_batch, void *stream) {
    if (!net || !intensity_dev || !normals_dev || !distance_dev || !pred_dev || n < 0)
        return iile::api_fail(IILE_ERR_ARG, "iile_iispt_net_predict: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (max_batch <= 0) max_batch = kDefaultBatch;
    int cap = n < max_batch ? n : max_batch;
    if (n == 0) return IILE_OK;
    int rc = ensure_workspace(net, &cap);
    if (rc) return rc;
    const int na = net->ws_probes;
    float *means = net->ws + net->ws_floats_per_probe * size_t(na);   // 3 floats per probe behind the activation buffers
    cap = even_batch(n, cap);
    for (int first = 0; first < n; first += cap) {
        const int nb = n - first < cap ? n - first : cap;
        hipLaunchKernelGGL(k_net_normalize, dim3(nb), dim3(256), 0, s, intensity_dev + size_t(first) * 3072, normals_dev + size_t(first) * 3072,
                           distance_dev + size_t(first) * 1024, buffer_of(net, BUF_P, na), means, nb);
        NET_TRY(hipGetLastError());
        rc = run_layers(net, nb, na, s, nullptr, 0, 0);
        if (rc) return rc;
        hipLaunchKernelGGL(k_net_predict_out, dim3(nb), dim3(256), 0, s, buffer_of(net, BUF_Q, na), net->w_out, net->b_out, means,
                           slot_of_probe_dev ? pred_dev : pred_dev + size_t(first) * 3072, slot_of_probe_dev ? slot_of_probe_dev + first : nullptr,
                           film_rows, nb);
        NET_TRY(hipGetLastError());
    }
    return IILE_OK;
}

}  // extern "C"
