// bvh_build.hip — BVHAccel's HLBVH construction on the device and the re-packing of a flattened
// tree into the traversal kernels' wide records.
//
// iile_bvh_build_hlbvh builds, from the primitives' world bounds, the tree BVHAccel::HLBVHBuild
// (src/accelerators/bvh.cpp:404-472) builds when its treelets are emitted in index order (what one thread does; with
// several threads the reference's `orderedPrimsOffset->fetch_add` makes the leaf order depend on scheduling), already
// flattened as flattenBVHTree (:640-658) lays it out. Stages:
//
//   k_centroid_bounds   bounds of the primitive centroids (:413-416): min / max are exact and order-free
//   k_morton            30-bit Morton codes of the centroids, 10 bits per axis (:418-427, LeftShift3 / EncodeMorton3 :107-130)
//   device_sort_pairs_30  (code, primitive number) pairs, stable, five 6-bit passes over bits 0-29 as RadixSort (:133-181, :430): own kernels
//   k_treelet_flags + scan + k_treelet_starts   runs of equal top 12 bits = treelets (:434-452)
//   k_lbvh_*            emitLBVH (:555-618) for all treelets at once, one thread per sorted position (see below), nodes in
//                       preorder into the treelet's own pool region; leaves take their primitives in sorted order, so
//                       the leaf order IS the sorted order
//   k_upper_split / k_upper_finish   buildUpperSAH (:527-638) over the <= 4096 treelet roots, one launch per level of the
//                       recursion, and the preorder offsets of all subtrees (round 4; a host loop before)
//   k_place_nodes       flattenBVHTree: every treelet node to its depth-first index, second-child offsets rebased
//
// The result must equal the host builder's (csrc/host/bvh_build.cpp, split method "hlbvh") node for node:
// tests/test_gpu_bvh_build.py. Float arithmetic is IEEE (no contraction, correctly rounded division) like the rest.
//
// pack_wide_records: the two-wide and four-wide interior records of DESIGN.md §3 from a flattened tree in HBM — what
// iile_scene_create did in host loops — one thread per node.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../../include/iile_gpu.h"
#include "kernels.h"

namespace iile {
namespace {

#define HIP_TRYB(expr)                                                                            \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return api_fail(IILE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr int kBB = 256;
constexpr float kFltMax = 3.402823466e+38f;

struct Box {
    float mn[3], mx[3];
};

__device__ __forceinline__ uint32_t order_key(float f) {  // monotone float -> u32
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// bvh.cpp:53-56: centroid = .5f * pMin + .5f * pMax
__device__ __forceinline__ void centroid_of(const float *b6, float c[3]) {
    for (int a = 0; a < 3; ++a) c[a] = .5f * b6[a] + .5f * b6[3 + a];
}

__global__ __launch_bounds__(kBB) void k_centroid_bounds(int n, const float *bounds6, uint32_t *keys6) {
    float mn[3] = {kFltMax, kFltMax, kFltMax}, mx[3] = {-kFltMax, -kFltMax, -kFltMax};
    for (int i = blockIdx.x * kBB + threadIdx.x; i < n; i += gridDim.x * kBB) {
        float c[3];
        centroid_of(bounds6 + 6 * size_t(i), c);
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], c[a]);
            mx[a] = fmaxf(mx[a], c[a]);
        }
    }
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    }
    // one set of atomics per block, and few blocks (the launch caps the grid): six words take every block's result, and
    // atomics on one address are served one after another (a set per wavefront of 2 048 blocks: 0.3 ms at 287 k primitives)
    __shared__ uint32_t part[6];
    if (threadIdx.x < 6) part[threadIdx.x] = threadIdx.x < 3 ? order_key(kFltMax) : order_key(-kFltMax);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; ++a) {
            atomicMin(&part[a], order_key(mn[a]));
            atomicMax(&part[3 + a], order_key(mx[a]));
        }
    __syncthreads();
    if (threadIdx.x < 3)
        atomicMin(&keys6[threadIdx.x], part[threadIdx.x]);
    else if (threadIdx.x < 6)
        atomicMax(&keys6[threadIdx.x], part[threadIdx.x]);
}

__device__ __forceinline__ uint32_t left_shift3(uint32_t x) {  // bvh.cpp:107-117
    if (x == (1u << 10)) --x;
    x = (x | (x << 16)) & 0x30000ffu;
    x = (x | (x << 8)) & 0x300f00fu;
    x = (x | (x << 4)) & 0x30c30c3u;
    x = (x | (x << 2)) & 0x9249249u;
    return x;
}

__global__ __launch_bounds__(kBB) void k_morton(int n, const float *bounds6, const uint32_t *keys6, uint32_t *codes, int *numbers) {
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) lo[a] = key_float(keys6[a]), hi[a] = key_float(keys6[3 + a]);
    for (int i = blockIdx.x * kBB + threadIdx.x; i < n; i += gridDim.x * kBB) {
        float c[3];
        centroid_of(bounds6 + 6 * size_t(i), c);
        uint32_t q[3];
        for (int a = 0; a < 3; ++a) {
            float o = c[a] - lo[a];  // Bounds3::Offset, geometry.h:804-810
            if (hi[a] > lo[a]) o /= hi[a] - lo[a];
            q[a] = uint32_t(o * 1024.f);  // mortonScale = 1 << 10
        }
        codes[i] = (left_shift3(q[2]) << 2) | (left_shift3(q[1]) << 1) | left_shift3(q[0]);
        numbers[i] = i;
    }
}

constexpr uint32_t kTreeletMask = 0x3ffc0000u;  // bvh.cpp:437: the top 12 of the 30 bits

__global__ __launch_bounds__(kBB) void k_treelet_flags(int n, const uint32_t *codes, int *flags) {
    for (int i = blockIdx.x * kBB + threadIdx.x; i < n; i += gridDim.x * kBB)
        flags[i] = (i == 0 || ((codes[i] ^ codes[i - 1]) & kTreeletMask) != 0) ? 1 : 0;
}
// incl[i] - 1 = the treelet of sorted position i
__global__ __launch_bounds__(kBB) void k_treelet_starts(int n, const int *flags, const int *incl, int *starts) {
    for (int i = blockIdx.x * kBB + threadIdx.x; i < n; i += gridDim.x * kBB) {
        if (flags[i]) starts[incl[i] - 1] = i;
        if (i == n - 1) starts[incl[i]] = n;
    }
}

__device__ __forceinline__ void box_union(float *mn, float *mx, const float *b6) {
    // Union(Bounds3, Bounds3), geometry.h: std::min / std::max component by component
    for (int a = 0; a < 3; ++a) {
        mn[a] = b6[a] < mn[a] ? b6[a] : mn[a];
        mx[a] = mx[a] < b6[3 + a] ? b6[3 + a] : mx[a];
    }
}

// emitLBVH (bvh.cpp:555-618) without the recursion. Inside a treelet the codes are sorted, so the recursion's node for a
// range [l, r) splits where the highest differing bit changes: with delta(i) = the highest of the low 18 bits in which
// codes[i - 1] and codes[i] differ (-1: equal codes), the split is the one position of (l, r) with the largest delta (it
// is unique: bit k can only rise twice inside a range if a higher bit changed in between), "same bit at both ends, try
// the next bit" (:569-572) is that maximum being below the current bit, and a leaf is a range with fewer than
// maxPrimsInNode primitives or with equal codes throughout (:558). Hence: position i is the split of an interior node
// iff delta(i) >= 0 and its range — from the nearest position on the left to the nearest on the right with a larger
// delta (or the treelet's ends) — holds at least maxPrimsInNode primitives; the ranges between consecutive splits are the
// leaves; a node's parent is the bounding split with the smaller delta. Preorder index of a node = its depth + the
// number of nodes that end at or before its first primitive. All of that is independent per position:
//
//   k_lbvh_ranges   delta, the range by two binary searches over the codes, the interior flag
//   (scan)          F = running count of splits; ipos = the splits compacted
//   k_lbvh_parents  parent split, histogram of range ends
//   (scan)          PE = running count of interior range ends
//   k_lbvh_interior depth by walking the parent chain (<= 18 steps), preorder index, the node minus its box
//   k_lbvh_leaves   one thread per leaf: preorder index, box of its primitives
//   k_lbvh_join     the interior boxes, one launch per split bit from the lowest up (children before parents)
//
// Nodes go to pool[2 * t0 + preorder index] (t0 = the treelet's first sorted position): first child = node + 1, the second
// child's (local) index in `offset`; leaves hold the global sorted position of their first primitive.
struct LbvhArrays {
    const uint32_t *codes;
    const int *numbers, *flags, *incl, *starts;  // flags / incl: treelet starts and their running count
    int *L, *R, *K, *interior, *F, *ipos, *par, *ends, *PE, *pre, *visit;
};
__device__ __forceinline__ int delta_of(uint32_t a, uint32_t b) {
    const uint32_t x = (a ^ b) & 0x3ffffu;
    return x ? 31 - __clz(x) : -1;
}
__global__ __launch_bounds__(kBB) void k_lbvh_ranges(int n, LbvhArrays A, int max_prims) {
    for (int i = blockIdx.x * kBB + threadIdx.x; i < n; i += gridDim.x * kBB) {
        int interior = 0, l = 0, r = 0, k = -1;
        if (!A.flags[i]) {
            const uint32_t ci = A.codes[i];
            k = delta_of(A.codes[i - 1], ci);
            if (k >= 0) {
                const int t = A.incl[i] - 1, t0 = A.starts[t], t1 = A.starts[t + 1];
                const int sh = k + 1;
                // lowest a in [t0, i - 1] whose code agrees with ci above bit k (they form a run ending at i - 1... and
                // going on through i: bit k is 0 on the left of i, 1 from i on)
                int lo = t0, hi = i - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (((A.codes[mid] ^ ci) >> sh) == 0)
                        hi = mid;
                    else
                        lo = mid + 1;
                }
                l = lo;
                lo = i, hi = t1 - 1;  // highest b in [i, t1 - 1] that agrees
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (((A.codes[mid] ^ ci) >> sh) == 0)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                r = lo + 1;
                interior = (r - l) >= max_prims ? 1 : 0;
            }
        }
        A.L[i] = l, A.R[i] = r, A.K[i] = k, A.interior[i] = interior;
    }
}
__global__ __launch_bounds__(kBB) void k_lbvh_parents(int n, LbvhArrays A) {
    for (int i = blockIdx.x * kBB + threadIdx.x; i < n; i += gridDim.x * kBB) {
        A.visit[i] = 0;
        if (!A.interior[i]) continue;
        A.ipos[A.F[i] - 1] = i;
        const int t = A.incl[i] - 1, t0 = A.starts[t], t1 = A.starts[t + 1];
        const int l = A.L[i], r = A.R[i];
        int p = -1;
        if (l != t0) p = l;
        if (r != t1 && (p < 0 || A.K[r] < A.K[p])) p = r;
        A.par[i] = p;
        atomicAdd(&A.ends[r], 1);
    }
}
// nodes of the treelet [t0, t1) that end at or before position p (p: a leaf boundary of the treelet)
__device__ __forceinline__ int ended_before(const LbvhArrays &A, int t0, int t1, int p) {
    const int interior_ends = A.PE[p] - A.PE[t0];
    const int q = p < t1 ? p : t1 - 1;
    const int leaf_ends = (A.F[q] - A.F[t0]) + (p == t1 ? 1 : 0);
    return interior_ends + leaf_ends;
}
__global__ __launch_bounds__(kBB) void k_lbvh_interior(int n, LbvhArrays A, iile_bvh_node *pool) {
    for (int i = blockIdx.x * kBB + threadIdx.x; i < n; i += gridDim.x * kBB) {
        if (!A.interior[i]) continue;
        const int t = A.incl[i] - 1, t0 = A.starts[t], t1 = A.starts[t + 1];
        int depth = 0;
        for (int q = A.par[i]; q >= 0; q = A.par[q]) ++depth;
        const int l = A.L[i];
        const int pre = depth + (l == t0 ? 0 : ended_before(A, t0, t1, l));
        A.pre[i] = pre;
        iile_bvh_node nd;
        for (int c = 0; c < 3; ++c) nd.bmin[c] = 0.f, nd.bmax[c] = 0.f;
        const int inside_left = A.F[i - 1] - A.F[l];  // splits strictly inside (l, i): the first subtree has 2 * that + 1 nodes
        nd.offset = pre + 2 + 2 * inside_left;
        nd.nprims = 0;
        nd.axis = uint8_t(A.K[i] % 3);
        nd.pad = 0;
        pool[2 * size_t(t0) + size_t(pre)] = nd;
    }
}
__global__ __launch_bounds__(kBB) void k_lbvh_leaves(int n, LbvhArrays A, const float *bounds6, iile_bvh_node *pool, int *n_nodes,
                                                     int *error) {
    for (int i = blockIdx.x * kBB + threadIdx.x; i < n; i += gridDim.x * kBB) {
        if (!A.flags[i] && !A.interior[i]) continue;  // leaves start at the treelet's first position and at every split
        const int t = A.incl[i] - 1, t0 = A.starts[t], t1 = A.starts[t + 1];
        const int splits_to_end = A.F[t1 - 1];
        const int e = A.F[i] < splits_to_end ? A.ipos[A.F[i]] : t1;
        int p = -1;
        if (i != t0) p = i;
        if (e != t1 && (p < 0 || A.K[e] < A.K[p])) p = e;
        const int depth_above = p < 0 ? 0 : 1;
        int pre = 0;
        if (p >= 0) {
            int depth = depth_above;
            for (int q = A.par[p]; q >= 0; q = A.par[q]) ++depth;
            pre = depth + (i == t0 ? 0 : ended_before(A, t0, t1, i));
        }
        if (i == t0) n_nodes[t] = 2 * (splits_to_end - A.F[t0]) + 1;
        float mn[3] = {kFltMax, kFltMax, kFltMax}, mx[3] = {-kFltMax, -kFltMax, -kFltMax};
        for (int j = i; j < e; ++j) box_union(mn, mx, bounds6 + 6 * size_t(A.numbers[j]));
        iile_bvh_node *out = pool + 2 * size_t(t0);
        iile_bvh_node nd;
        for (int c = 0; c < 3; ++c) nd.bmin[c] = mn[c], nd.bmax[c] = mx[c];
        nd.offset = i;
        if (e - i > 65535) atomicExch(error, 1);  // LinearBVHNode::nPrimitives is 16 bits
        nd.nprims = uint16_t(e - i);
        nd.axis = 0;
        nd.pad = 0;
        out[pre] = nd;
    }
}
// InitInterior's boxes (bvh.cpp:66-72), bottom-up without any hand-off between running threads: a node that splits at bit k has
// children that split at lower bits or are leaves, so one launch per bit, lowest first, finds every child box final — the
// kernel boundary is the only ordering needed. (Rounds 2-3 let the second child to arrive at a parent join the boxes and climb
// on: two device-scope fences per step, ~3.5 us apiece on gfx950 — 21 ms of a 30 ms build at 4 M primitives.)
__global__ __launch_bounds__(kBB) void k_lbvh_join(int n, LbvhArrays A, iile_bvh_node *pool, int bit) {
    for (int i = blockIdx.x * kBB + threadIdx.x; i < n; i += gridDim.x * kBB) {
        if (!A.interior[i] || A.K[i] != bit) continue;
        iile_bvh_node *out = pool + 2 * size_t(A.starts[A.incl[i] - 1]);
        const int qp = A.pre[i];
        iile_bvh_node me = out[qp];
        const iile_bvh_node a = out[qp + 1], b = out[me.offset];
        for (int c = 0; c < 3; ++c) {
            me.bmin[c] = b.bmin[c] < a.bmin[c] ? b.bmin[c] : a.bmin[c];
            me.bmax[c] = a.bmax[c] < b.bmax[c] ? b.bmax[c] : a.bmax[c];
        }
        out[qp] = me;
    }
}

__global__ __launch_bounds__(kBB) void k_treelet_roots(int n_treelets, const int *starts, const iile_bvh_node *pool, Box *roots) {
    const int t = blockIdx.x * kBB + threadIdx.x;
    if (t >= n_treelets) return;
    const iile_bvh_node &r = pool[2 * size_t(starts[t])];
    for (int a = 0; a < 3; ++a) roots[t].mn[a] = r.bmin[a], roots[t].mx[a] = r.bmax[a];
}

// flattenBVHTree: pool slot i belongs to the treelet of sorted position i / 2
__global__ __launch_bounds__(kBB) void k_place_nodes(int n, const int *incl, const int *starts, const int *n_nodes, const int *base,
                                                     const iile_bvh_node *pool, iile_bvh_node *out) {
    for (int i = blockIdx.x * kBB + threadIdx.x; i < 2 * n; i += gridDim.x * kBB) {
        const int t = incl[i >> 1] - 1;
        const int local = i - 2 * starts[t];
        if (local >= n_nodes[t]) continue;
        iile_bvh_node nd = pool[i];
        if (nd.nprims == 0) nd.offset += base[t];
        out[base[t] + local] = nd;
    }
}
// ---- buildUpperSAH on the device (bvh.cpp:527-638): 12-bucket SAH over the treelet roots ------------------------------
// The reference recurses over a span of treelet roots: bounds and centroid bounds of the span, twelve buckets along the
// widest centroid axis, the cheapest of eleven splits, std::partition, two recursive calls. Everything it computes is a
// function of the SET of roots in the span — min / max unions, counts, the cost expression — so the order std::partition
// leaves the two sides in decides nothing: the tree is the same whichever way each side is ordered, and the layout
// (flattenBVHTree's preorder) follows from the tree. The recursion runs level by level, one launch per level:
//   k_upper_split   every span of the level is split by one wavefront: a pass over its roots for the bounds (wave reductions as
//                   DPP operands), one for the buckets (LDS atomics on order-preserving keys), the eleven costs on eleven
//                   lanes — bucket j in lane j, prefix and suffix scans over the lanes give every candidate's two boxes and
//                   counts, the cost in the reference's float expression —, the first minimum, a stable partition by ballot
//                   ranks into the other of two index arrays. A side of one root is a leaf of the upper tree; a larger side
//                   joins the next level's queue. Upper nodes are numbered level by level in queue order, so a parent's
//                   number is smaller than its children's and the children's numbers are known when they are queued.
//   k_upper_finish  one block: InitInterior's sizes and boxes, deepest level first (Union(c0, c1), operand order kept: std::min /
//                   std::max return their first argument on a tie of -0 and +0), then every upper node and treelet sums its
//                   preorder offset along its path to the root; upper nodes are written to the output array in place, treelets
//                   get the base k_place_nodes moves their blocks to.
// A span whose centroids coincide on the chosen axis (the reference stops with CHECK_NE there) or whose split leaves a side
// empty is cut in the middle, so the recursion always ends. (First version, same round: one block, an LDS queue of spans for its
// sixteen wavefronts — correct, but one CU at idle clocks made 0.55 us per span, 0.9 ms at 1 582 treelets, where the host loop it
// replaced took 0.6; the levels in parallel over the chip take a fifth of that.)
constexpr int kUpMax = 4096;  // treelets are runs of equal top 12 Morton bits: at most 4096
constexpr int kUpBuckets = 12;
constexpr int kUpBatch = 14;  // levels launched before the host looks whether the last one still had spans (IILE_UPPER_BATCH overrides: tests)
struct UpperDev {
    const Box *roots;      // [n_t] treelet root bounds
    const int *n_nodes_t;  // [n_t] nodes per treelet
    int *refs;             // [2][kUpMax] treelet numbers: level L reads array L & 1 and writes the other
    uint32_t *qseg;        // [2][kUpMax] spans of a level: start | end << 13
    int *qlink;            // [2][kUpMax] parent node * 2 + which child, -1 for the root span
    int *qcount;           // [kUpMax + 2] spans per level
    int *level_base;       // [kUpMax + 2] number of the first node of a level
    int *node_link;        // [kUpMax] upper node -> parent * 2 + which, -1 for the root
    int *cref;             // [kUpMax][2] children: >= 0 upper node, < 0 ~treelet
    int *axis;             // [kUpMax]
    int *tparent;          // [kUpMax] treelet -> parent * 2 + which (-1: the treelet is the whole tree)
    int *size;             // [kUpMax] nodes in the subtree of an upper node
    Box *nbox;             // [kUpMax] its box
    int *base;             // [n_t] out: preorder offset of every treelet's block
    iile_bvh_node *out;    // the flattened tree: upper nodes are written here
};
__device__ __forceinline__ float box_area(const float mn[3], const float mx[3]) {  // Bounds3::SurfaceArea, geometry.h:782-785
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2 * (dx * dy + dx * dz + dy * dz);
}
__device__ __forceinline__ float union_min(float a, float b) { return b < a ? b : a; }  // std::min(a, b)
__device__ __forceinline__ float union_max(float a, float b) { return a < b ? b : a; }  // std::max(a, b)
// Cross-lane steps as DPP operands of the min / max / add itself (one instruction per step; __shfl_xor is an LDS round trip):
// row_shr / row_shl move inside a row of 16 lanes, row_bcast:15 / :31 carry a row's last lane into the next row / the upper half
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float ident, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int ident, int v) {
    return __builtin_amdgcn_update_dpp(ident, v, CTRL, 0xf, 0xf, false);
}
constexpr int kRowShr = 0x110, kRowShl = 0x100, kRowBcast15 = 0x142, kRowBcast31 = 0x143;
__device__ __forceinline__ float wave_min_f(float v) {  // all 64 lanes active; the result is uniform
    v = fminf(v, dpp_f<kRowShr + 1>(kFltMax, v)), v = fminf(v, dpp_f<kRowShr + 2>(kFltMax, v));
    v = fminf(v, dpp_f<kRowShr + 4>(kFltMax, v)), v = fminf(v, dpp_f<kRowShr + 8>(kFltMax, v));
    v = fminf(v, dpp_f<kRowBcast15, 0xa>(kFltMax, v)), v = fminf(v, dpp_f<kRowBcast31, 0xc>(kFltMax, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f(float v) {
    v = fmaxf(v, dpp_f<kRowShr + 1>(-kFltMax, v)), v = fmaxf(v, dpp_f<kRowShr + 2>(-kFltMax, v));
    v = fmaxf(v, dpp_f<kRowShr + 4>(-kFltMax, v)), v = fmaxf(v, dpp_f<kRowShr + 8>(-kFltMax, v));
    v = fmaxf(v, dpp_f<kRowBcast15, 0xa>(-kFltMax, v)), v = fmaxf(v, dpp_f<kRowBcast31, 0xc>(-kFltMax, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// inclusive scans over the lanes of a row, towards higher lanes (prefix) and towards lower lanes (suffix)
#define ROW_SCAN(v, ident, OP, DIR)                                                               \
    do {                                                                                          \
        v = OP(v, dpp_f<DIR + 1>(ident, v)), v = OP(v, dpp_f<DIR + 2>(ident, v));                   \
        v = OP(v, dpp_f<DIR + 4>(ident, v)), v = OP(v, dpp_f<DIR + 8>(ident, v));                   \
    } while (0)
__device__ __forceinline__ int row_scan_add(int v, bool prefix) {
    if (prefix) {
        v += dpp_i<kRowShr + 1>(0, v), v += dpp_i<kRowShr + 2>(0, v), v += dpp_i<kRowShr + 4>(0, v), v += dpp_i<kRowShr + 8>(0, v);
    } else {
        v += dpp_i<kRowShl + 1>(0, v), v += dpp_i<kRowShl + 2>(0, v), v += dpp_i<kRowShl + 4>(0, v), v += dpp_i<kRowShl + 8>(0, v);
    }
    return v;
}
__global__ __launch_bounds__(kBB) void k_upper_init(int n_t, UpperDev U) {
    for (int i = blockIdx.x * kBB + threadIdx.x; i < kUpMax + 2; i += gridDim.x * kBB) {
        U.qcount[i] = (i == 0 && n_t > 1) ? 1 : 0;
        U.level_base[i] = 0;
        if (i < kUpMax) {
            U.refs[i] = i;
            U.tparent[i] = -1;
        }
        if (i == 0) {
            U.qseg[0] = 0u | (uint32_t(n_t) << 13);
            U.qlink[0] = -1;
        }
    }
}
__global__ __launch_bounds__(kBB) void k_upper_split(int level, UpperDev U) {
    __shared__ uint32_t bk[kBB / 64][kUpBuckets][8];  // per wavefront: count, 3 min keys, 3 max keys
    const int count = U.qcount[level];
    if (count == 0) return;
    const int lb = U.level_base[level];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (blockIdx.x == 0 && tid == 0) U.level_base[level + 1] = lb + count;  // (read by later launches only)
    const int par = level & 1;
    const int *src = U.refs + par * kUpMax;
    int *dst = U.refs + (par ^ 1) * kUpMax;
    // Everything that steers a wavefront here is wave-uniform BY CONSTRUCTION: values come through readfirstlane, and what one
    // lane would do (queue a span) every lane does with the same operands, the counter bumped by lane 0's operand alone.
    // (In the one-block version `if (lane == 0)` at both ends of a loop body let the compiler thread lane 0's path from the
    // publishing block into the next round and run lanes 1-63 ahead without it — a hang that cost a GPU call to see.)
    for (int sp = blockIdx.x * (kBB / 64) + w; sp < count; sp += gridDim.x * (kBB / 64)) {
        const uint32_t seg = U.qseg[par * kUpMax + sp];
        const int s = int(seg & 8191u), e = int((seg >> 13) & 8191u);
        const int node = lb + sp;
        // bounds of the span and of its centroids (bvh.cpp:537-549)
        float bmn[3] = {kFltMax, kFltMax, kFltMax}, bmx[3] = {-kFltMax, -kFltMax, -kFltMax};
        float cmn[3] = {kFltMax, kFltMax, kFltMax}, cmx[3] = {-kFltMax, -kFltMax, -kFltMax};
        for (int i = s + lane; i < e; i += 64) {
            const Box b = U.roots[src[i]];
            for (int a = 0; a < 3; ++a) {
                bmn[a] = fminf(bmn[a], b.mn[a]), bmx[a] = fmaxf(bmx[a], b.mx[a]);
                const float c = (b.mn[a] + b.mx[a]) * 0.5f;
                cmn[a] = fminf(cmn[a], c), cmx[a] = fmaxf(cmx[a], c);
            }
        }
        for (int a = 0; a < 3; ++a) {
            bmn[a] = wave_min_f(bmn[a]), bmx[a] = wave_max_f(bmx[a]);
            cmn[a] = wave_min_f(cmn[a]), cmx[a] = wave_max_f(cmx[a]);
        }
        int dim;  // Bounds3::MaximumExtent, geometry.h:790-798
        float lo, hi;
        {
            const float dx = cmx[0] - cmn[0], dy = cmx[1] - cmn[1], dz = cmx[2] - cmn[2];
            dim = __builtin_amdgcn_readfirstlane((dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2));
            lo = dim == 0 ? cmn[0] : (dim == 1 ? cmn[1] : cmn[2]);
            hi = dim == 0 ? cmx[0] : (dim == 1 ? cmx[1] : cmx[2]);
            lo = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(lo)));
            hi = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hi)));
        }
        auto bucket_of = [&](const Box &b) {  // bvh.cpp:576-582
            const float centroid = ((dim == 0 ? b.mn[0] : (dim == 1 ? b.mn[1] : b.mn[2])) + (dim == 0 ? b.mx[0] : (dim == 1 ? b.mx[1] : b.mx[2]))) * 0.5f;
            int k = int(kUpBuckets * ((centroid - lo) / (hi - lo)));
            if (k == kUpBuckets) k = kUpBuckets - 1;
            return k;
        };
        const bool flat = !(lo < hi);  // every centroid at one coordinate: no bucket means anything
        int n_left = (e - s) / 2;
        int min_bucket = -1;
        if (!flat) {
            for (int i = lane; i < kUpBuckets * 8; i += 64) {
                const int f = i & 7;
                bk[w][i >> 3][f] = f == 0 ? 0u : (f <= 3 ? order_key(kFltMax) : order_key(-kFltMax));
            }
            __builtin_amdgcn_wave_barrier();
            for (int i = s + lane; i < e; i += 64) {
                const Box b = U.roots[src[i]];
                int k = bucket_of(b);
                k = k < 0 ? 0 : (k > kUpBuckets - 1 ? kUpBuckets - 1 : k);  // (the reference CHECKs 0 <= b < 12)
                atomicAdd(&bk[w][k][0], 1u);
                for (int a = 0; a < 3; ++a) {
                    atomicMin(&bk[w][k][1 + a], order_key(b.mn[a]));
                    atomicMax(&bk[w][k][4 + a], order_key(b.mx[a]));
                }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            // bvh.cpp:588-602. Bucket j sits in lane j (lanes 12-63: empty buckets); the union and count of buckets 0..i are a
            // prefix scan over the lanes, those of buckets i+1..11 a suffix scan read one lane over; lane i < 11 prices split i.
            // (min / max unions are exact in any order; the cost expression is the reference's, operation for operation.)
            float cost;
            int c_prefix;
            {
                const int j = lane < kUpBuckets ? lane : kUpBuckets - 1;
                const bool real = lane < kUpBuckets;
                const int cnt_j = real ? int(bk[w][j][0]) : 0;
                float p_mn[3], p_mx[3], s_mn[3], s_mx[3];
                for (int a = 0; a < 3; ++a) {
                    p_mn[a] = real ? key_float(bk[w][j][1 + a]) : kFltMax;
                    p_mx[a] = real ? key_float(bk[w][j][4 + a]) : -kFltMax;
                    s_mn[a] = p_mn[a], s_mx[a] = p_mx[a];
                    ROW_SCAN(p_mn[a], kFltMax, fminf, kRowShr);
                    ROW_SCAN(p_mx[a], -kFltMax, fmaxf, kRowShr);
                    ROW_SCAN(s_mn[a], kFltMax, fminf, kRowShl);
                    ROW_SCAN(s_mx[a], -kFltMax, fmaxf, kRowShl);
                    s_mn[a] = dpp_f<kRowShl + 1>(kFltMax, s_mn[a]);  // buckets lane+1 .. 11
                    s_mx[a] = dpp_f<kRowShl + 1>(-kFltMax, s_mx[a]);
                }
                c_prefix = row_scan_add(cnt_j, true);
                const int c1 = dpp_i<kRowShl + 1>(0, row_scan_add(cnt_j, false));
                cost = .125f + (float(c_prefix) * box_area(p_mn, p_mx) + float(c1) * box_area(s_mn, s_mx)) / box_area(bmn, bmx);
            }
            // bvh.cpp:605-612: minCost starts at cost[0] and moves to a later cost only if that is smaller — the first of the
            // smallest; a NaN cost (0 x inf: a side without roots) is never smaller, and nothing is smaller than a NaN cost[0]
            {
                const bool cand = lane < kUpBuckets - 1;
                const bool first_is_nan = __builtin_amdgcn_readfirstlane(cost != cost ? 1 : 0) != 0;
                const float c = (cand && cost == cost) ? cost : __builtin_inff();
                const float m = wave_min_f(c);
                const unsigned long long at_min = __ballot(cand && cost == m);
                min_bucket = (first_is_nan || at_min == 0) ? 0 : __builtin_ctzll(at_min);
            }
            const int cnt = __builtin_amdgcn_readlane(c_prefix, min_bucket);  // roots in buckets 0 .. min_bucket
            if (cnt > 0 && cnt < e - s)
                n_left = cnt;
            else
                min_bucket = -1;  // an empty side: the middle cut
        }
        const int mid = s + n_left;
        int first_of[2] = {0, 0};  // the first root of each side (a side of one root is a leaf: no trip through memory for it)
        {  // stable partition into the other array
            int done_l = 0, done_r = 0;
            for (int base = s; base < e; base += 64) {
                const int i = base + lane;
                const bool in = i < e;
                int t = 0;
                bool left = false;
                if (in) {
                    t = src[i];
                    left = min_bucket >= 0 ? bucket_of(U.roots[t]) <= min_bucket : (i - s) < n_left;
                }
                const unsigned long long ml = __ballot(in && left), mr = __ballot(in && !left);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (in) {
                    if (left)
                        dst[s + done_l + __popcll(ml & below)] = t;
                    else
                        dst[mid + done_r + __popcll(mr & below)] = t;
                }
                if (done_l == 0 && ml != 0) first_of[0] = __builtin_amdgcn_readlane(t, __builtin_ctzll(ml));
                if (done_r == 0 && mr != 0) first_of[1] = __builtin_amdgcn_readlane(t, __builtin_ctzll(mr));
                done_l += __popcll(ml), done_r += __popcll(mr);
            }
        }
        U.node_link[node] = U.qlink[par * kUpMax + sp];
        U.axis[node] = dim;
        for (int which = 0; which < 2; ++which) {
            const int cs = which ? mid : s, ce = which ? e : mid;
            if (ce - cs == 1) {  // (uniform)
                const int t = first_of[which];
                U.tparent[t] = node * 2 + which;
                U.cref[node * 2 + which] = ~t;
            } else {
                const int idx = __builtin_amdgcn_readfirstlane(atomicAdd(&U.qcount[level + 1], lane == 0 ? 1 : 0));
                U.qseg[(par ^ 1) * kUpMax + idx] = uint32_t(cs) | (uint32_t(ce) << 13);
                U.qlink[(par ^ 1) * kUpMax + idx] = node * 2 + which;
                U.cref[node * 2 + which] = lb + count + idx;  // the next level's nodes are numbered from lb + count in queue order
            }
        }
    }
}
// One block finishes the tree (at most 4 095 upper nodes): sizes and boxes level by level from the deepest up — InitInterior's
// Union(c0, c1) — with a block barrier between levels, then the preorder offsets: 1 per ancestor plus the first child's subtree
// wherever the path to the root goes through a second child. (A launch per level cost more than the levels' work.)
constexpr int kUpFinishThreads = 1024;
__global__ __launch_bounds__(kUpFinishThreads) void k_upper_finish(int n_t, int depth, UpperDev U) {
    for (int level = depth - 1; level >= 0; --level) {
        const int first = U.level_base[level], last = U.level_base[level + 1];
        for (int node = first + int(threadIdx.x); node < last; node += kUpFinishThreads) {
            Box b[2];
            int sz[2];
            for (int which = 0; which < 2; ++which) {
                const int r = U.cref[node * 2 + which];
                b[which] = r < 0 ? U.roots[~r] : U.nbox[r];
                sz[which] = r < 0 ? U.n_nodes_t[~r] : U.size[r];
            }
            Box box;  // bvh.cpp:66-72
            for (int a = 0; a < 3; ++a) box.mn[a] = union_min(b[0].mn[a], b[1].mn[a]), box.mx[a] = union_max(b[0].mx[a], b[1].mx[a]);
            U.nbox[node] = box;
            U.size[node] = 1 + sz[0] + sz[1];
        }
        __syncthreads();  // (one block, one CU: the level's stores are visible to the next level's loads)
    }
    const int n_upper = n_t - 1;
    for (int i = int(threadIdx.x); i < n_upper + n_t; i += kUpFinishThreads) {
        const bool is_node = i < n_upper;
        int link = is_node ? U.node_link[i] : U.tparent[i - n_upper];
        int off = 0;
        while (link >= 0) {
            const int p = link >> 1;
            if (link & 1) {
                const int r = U.cref[p * 2];
                off += r < 0 ? U.n_nodes_t[~r] : U.size[r];
            }
            off += 1;
            link = U.node_link[p];
        }
        if (is_node) {
            iile_bvh_node nd;
            const Box box = U.nbox[i];
            for (int a = 0; a < 3; ++a) nd.bmin[a] = box.mn[a], nd.bmax[a] = box.mx[a];
            const int r0 = U.cref[i * 2];
            nd.offset = off + 1 + (r0 < 0 ? U.n_nodes_t[~r0] : U.size[r0]);  // secondChildOffset (bvh.cpp:651-656)
            nd.nprims = 0;
            nd.axis = uint8_t(U.axis[i]);
            nd.pad = 0;
            U.out[off] = nd;
        } else {
            U.base[i - n_upper] = off;
        }
    }
}

template <typename T>
struct Dev {
    T *p = nullptr;
    ~Dev() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(n, 1) * sizeof(T)); }
};

int grid_for(int n) { return std::max(1, std::min((n + kBB - 1) / kBB, 256 * 8)); }

// ---- scans and the Morton sort: own kernels (rounds 1-3 called rocPRIM here) ------------------------------------------------
// Prefix sums of int arrays: blocks of 256 threads x 8 items scan their 2048 elements (thread-local sums, a wavefront scan by
// DPP-free shuffles, the four wavefronts' totals through LDS) and leave their totals; the totals are scanned the same way
// (recursively: one more level per factor of 2048) and added back.
constexpr int kScanItems = 8, kScanTile = kBB * kScanItems;
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}
__global__ __launch_bounds__(kBB) void k_scan_tiles(const int *in, int *out, int n, int *tile_sums, int inclusive) {
    __shared__ int wave_tot[kBB / 64];
    const int base = blockIdx.x * kScanTile + int(threadIdx.x) * kScanItems;
    int v[kScanItems], sum = 0;
    for (int j = 0; j < kScanItems; ++j) {
        v[j] = base + j < n ? in[base + j] : 0;
        sum += v[j];
    }
    const int incl = wave_inclusive_scan(sum);
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    int before = incl - sum;  // exclusive over this wavefront's threads
    for (int w = 0; w < int(threadIdx.x >> 6); ++w) before += wave_tot[w];
    int run = before;
    for (int j = 0; j < kScanItems; ++j) {
        const int excl = run;
        run += v[j];
        if (base + j < n) out[base + j] = inclusive ? run : excl;
    }
    if (threadIdx.x == kBB - 1 && tile_sums) tile_sums[blockIdx.x] = run;
}
__global__ __launch_bounds__(kBB) void k_scan_add(int *out, int n, const int *tile_offsets) {
    const int off = tile_offsets[blockIdx.x];
    const int base = blockIdx.x * kScanTile + int(threadIdx.x) * kScanItems;
    for (int j = 0; j < kScanItems; ++j)
        if (base + j < n) out[base + j] += off;
}
// out[i] = in[0] + .. + in[i] (inclusive) or in[0] + .. + in[i - 1] (exclusive); in and out may not overlap
int device_scan(const int *in, int *out, size_t n64, bool inclusive, hipStream_t s) {
    if (n64 == 0) return IILE_OK;
    if (n64 > size_t(0x7fffffff)) return api_fail(IILE_ERR_UNSUPPORTED, "device_scan: more than 2^31 elements");
    const int n = int(n64), tiles = (n + kScanTile - 1) / kScanTile;
    if (tiles == 1) {
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(kBB), 0, s, in, out, n, static_cast<int *>(nullptr), inclusive ? 1 : 0);
        return IILE_OK;
    }
    Dev<int> sums, offs;
    HIP_TRYB(sums.alloc(size_t(tiles)));
    HIP_TRYB(offs.alloc(size_t(tiles)));
    hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(kBB), 0, s, in, out, n, sums.p, inclusive ? 1 : 0);
    const int rc = device_scan(sums.p, offs.p, size_t(tiles), false, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scan_add, dim3(tiles), dim3(kBB), 0, s, out, n, offs.p);
    HIP_TRYB(hipStreamSynchronize(s));  // (the two scratch arrays die with this scope)
    return IILE_OK;
}

// RadixSort (bvh.cpp:133-181): least significant digit first, five passes of six bits over the 30-bit Morton codes, stable — the
// same passes here, each as histogram -> scan -> scatter. A block takes 1024 consecutive pairs; its scatter keeps their order
// within a digit: the four items of a thread are taken in four rounds (round j: element j * 256 + thread), the wavefronts of a
// round one after the other, and inside a wavefront a lane's rank among the lanes with its digit comes from six ballots.
constexpr int kSortBits = 6, kSortDigits = 1 << kSortBits, kSortItems = 4, kSortTile = kBB * kSortItems;
__global__ __launch_bounds__(kBB) void k_sort_histogram(const uint32_t *keys, int n, int shift, int *counts, int n_tiles) {
    __shared__ int hist[kSortDigits];
    if (threadIdx.x < kSortDigits) hist[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * kSortTile;
    for (int j = 0; j < kSortItems; ++j) {
        const int i = base + j * kBB + int(threadIdx.x);
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & (kSortDigits - 1)], 1);
    }
    __syncthreads();
    if (threadIdx.x < kSortDigits) counts[int(threadIdx.x) * n_tiles + blockIdx.x] = hist[threadIdx.x];  // digit-major: one scan gives every offset
}
__global__ __launch_bounds__(kBB) void k_sort_scatter(const uint32_t *keys, const int *vals, uint32_t *keys_out, int *vals_out, int n, int shift,
                                                      const int *offsets, int n_tiles) {
    __shared__ int run[kSortDigits];
    if (threadIdx.x < kSortDigits) run[threadIdx.x] = offsets[int(threadIdx.x) * n_tiles + blockIdx.x];
    __syncthreads();
    const int base = blockIdx.x * kSortTile, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = 0; j < kSortItems; ++j) {
        const int i = base + j * kBB + int(threadIdx.x);
        const bool live = i < n;
        const uint32_t key = live ? keys[i] : 0u;
        const int val = live ? vals[i] : 0;
        const uint32_t digit = (key >> shift) & (kSortDigits - 1);
        for (int w = 0; w < kBB / 64; ++w) {
            if (wave == w) {
                // the lanes of this wavefront that hold the same digit (and an element at all)
                unsigned long long peers = __ballot(live);
                for (int b = 0; b < kSortBits; ++b) {
                    const unsigned long long set = __ballot((digit >> b) & 1u);
                    peers &= ((digit >> b) & 1u) ? set : ~set;
                }
                if (live) {
                    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
                    const int pos = run[digit] + rank;
                    keys_out[pos] = key;
                    vals_out[pos] = val;
                }
                __builtin_amdgcn_wave_barrier();
                if (live && (peers & ((1ull << lane) - 1ull)) == 0) run[digit] += __popcll(peers);  // the digit's first lane moves its cursor on
            }
            __syncthreads();
        }
    }
}
// (keys, vals) sorted by bits [0, 30) of the keys, stable, into (keys_out, vals_out); keys / vals are used as the other buffer
int device_sort_pairs_30(uint32_t *keys, int *vals, uint32_t *keys_out, int *vals_out, int n, hipStream_t s) {
    if (n <= 0) return IILE_OK;
    const int tiles = (n + kSortTile - 1) / kSortTile;
    Dev<int> counts, offsets;
    Dev<uint32_t> keys_tmp;
    Dev<int> vals_tmp;
    HIP_TRYB(counts.alloc(size_t(kSortDigits) * tiles));
    HIP_TRYB(offsets.alloc(size_t(kSortDigits) * tiles));
    HIP_TRYB(keys_tmp.alloc(size_t(n)));
    HIP_TRYB(vals_tmp.alloc(size_t(n)));
    // five passes: in -> tmp -> out -> tmp -> out -> ... ending in `out` (the inputs stay untouched)
    const uint32_t *src_k = keys;
    const int *src_v = vals;
    for (int pass = 0; pass < 5; ++pass) {
        uint32_t *dst_k = (pass & 1) ? keys_tmp.p : keys_out;
        int *dst_v = (pass & 1) ? vals_tmp.p : vals_out;
        if (pass == 4) dst_k = keys_out, dst_v = vals_out;
        if (dst_k == src_k) return api_fail(IILE_ERR_HIP, "device_sort_pairs_30: buffer schedule");
        hipLaunchKernelGGL(k_sort_histogram, dim3(tiles), dim3(kBB), 0, s, src_k, n, pass * kSortBits, counts.p, tiles);
        const int rc = device_scan(counts.p, offsets.p, size_t(kSortDigits) * tiles, false, s);
        if (rc) return rc;
        hipLaunchKernelGGL(k_sort_scatter, dim3(tiles), dim3(kBB), 0, s, src_k, src_v, dst_k, dst_v, n, pass * kSortBits, offsets.p, tiles);
        src_k = dst_k;
        src_v = dst_v;
    }
    HIP_TRYB(hipGetLastError());
    HIP_TRYB(hipStreamSynchronize(s));
    return IILE_OK;
}

// ---- wide records -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBB) void k_interior_flags(int n, const iile_bvh_node *nodes, int *flags) {
    for (int i = blockIdx.x * kBB + threadIdx.x; i < n; i += gridDim.x * kBB) flags[i] = nodes[i].nprims == 0 ? 1 : 0;
}
// record slot of an interior node: its rank among the interior nodes (depth-first), or where `remap` sends that rank
__device__ __forceinline__ int slot_of_rank(const int *remap, int rank) { return remap ? remap[rank] : rank; }
__device__ __forceinline__ int ref_of(const iile_bvh_node *nodes, const int *excl, const int *remap, int node) {
    return nodes[node].nprims == 0 ? slot_of_rank(remap, excl[node]) : ~nodes[node].offset;
}
__global__ __launch_bounds__(kBB) void k_pack_wide(int n, const iile_bvh_node *nodes, const int *excl, const int *remap, float4 *wide,
                                                   float4 *wide4, int *not_nested) {
    const float inf = __builtin_huge_valf();
    for (int i = blockIdx.x * kBB + threadIdx.x; i < n; i += gridDim.x * kBB) {
        const iile_bvh_node nd = nodes[i];
        if (nd.nprims > 0) continue;
        const int child[2] = {i + 1, nd.offset};
        const iile_bvh_node a = nodes[child[0]], b = nodes[child[1]];
        // two-wide record (instrumented kernels): both children's boxes, their refs, the split axis
        const int my_slot = slot_of_rank(remap, excl[i]);
        float4 *w = wide + 4 * size_t(my_slot);
        const int ra = ref_of(nodes, excl, remap, child[0]), rb = ref_of(nodes, excl, remap, child[1]);
        w[0] = make_float4(a.bmin[0], a.bmin[1], a.bmin[2], a.bmax[0]);
        w[1] = make_float4(a.bmax[1], a.bmax[2], b.bmin[0], b.bmin[1]);
        w[2] = make_float4(b.bmin[2], b.bmax[0], b.bmax[1], b.bmax[2]);
        w[3] = make_float4(__int_as_float(ra), __int_as_float(rb), __int_as_float(int(nd.axis)), 0.f);
        // a child's box must lie inside its parent's for the four-wide step to skip the children (dpath.h trav_interior4)
        bool nested = true;
        for (int c = 0; c < 3; ++c)
            nested = nested && a.bmin[c] >= nd.bmin[c] && a.bmax[c] <= nd.bmax[c] && b.bmin[c] >= nd.bmin[c] && b.bmax[c] <= nd.bmax[c];
        if (!nested) atomicExch(not_nested, 1);
        // four-wide record: grandchildren in slots {L's children | L, -} {R's children | R, -}; an unused slot gets an
        // inverted infinite box no ray can enter
        float bx[6][4];
        int refs[4];
        for (int k = 0; k < 4; ++k) {
            for (int c = 0; c < 3; ++c) bx[c][k] = inf, bx[3 + c][k] = -inf;
            refs[k] = 0;
        }
        uint32_t meta = uint32_t(nd.axis) & 3u;
        for (int side = 0; side < 2; ++side) {
            const iile_bvh_node &c = side ? b : a;
            if (c.nprims > 0) {
                for (int k = 0; k < 3; ++k) bx[k][2 * side] = c.bmin[k], bx[3 + k][2 * side] = c.bmax[k];
                refs[2 * side] = ~c.offset;
            } else {
                const int gc[2] = {child[side] + 1, c.offset};
                for (int j = 0; j < 2; ++j) {
                    const iile_bvh_node g = nodes[gc[j]];
                    for (int k = 0; k < 3; ++k) bx[k][2 * side + j] = g.bmin[k], bx[3 + k][2 * side + j] = g.bmax[k];
                    refs[2 * side + j] = g.nprims == 0 ? slot_of_rank(remap, excl[gc[j]]) : ~g.offset;
                }
                meta |= (uint32_t(c.axis) & 3u) << (2 + 2 * side);
            }
        }
        float4 *w4 = wide4 + 8 * size_t(my_slot);
        for (int pl = 0; pl < 6; ++pl) w4[pl] = make_float4(bx[pl][0], bx[pl][1], bx[pl][2], bx[pl][3]);
        if (kRefShift) {  // ref << 2 | axis of {the node, its first child, its second child, -}
            refs[0] = int((uint32_t(refs[0]) << kRefShift) | (meta & 3u));
            refs[1] = int((uint32_t(refs[1]) << kRefShift) | ((meta >> 2) & 3u));
            refs[2] = int((uint32_t(refs[2]) << kRefShift) | ((meta >> 4) & 3u));
            refs[3] = int(uint32_t(refs[3]) << kRefShift);
        }
        w4[6] = make_float4(__int_as_float(refs[0]), __int_as_float(refs[1]), __int_as_float(refs[2]), __int_as_float(refs[3]));
        w4[7] = make_float4(__uint_as_float(meta), 0.f, 0.f, 0.f);
    }
}

}  // namespace

// Validates the child indices and leaf ranges of a flattened tree on the host side of the caller (api.hip) before this.
int pack_wide_records(const iile_bvh_node *d_nodes, int n_nodes, int n_interior, float4 *d_wide, float4 *d_wide4, int *nested_out,
                      const int *d_remap) {
    *nested_out = 1;
    if (n_nodes <= 0) return IILE_OK;
    Dev<int> flags, excl, bad;
    HIP_TRYB(flags.alloc(size_t(n_nodes)));
    HIP_TRYB(excl.alloc(size_t(n_nodes)));
    HIP_TRYB(bad.alloc(1));
    HIP_TRYB(hipMemsetAsync(bad.p, 0, sizeof(int), nullptr));
    hipLaunchKernelGGL(k_interior_flags, dim3(grid_for(n_nodes)), dim3(kBB), 0, nullptr, n_nodes, d_nodes, flags.p);
    {
        const int rc = device_scan(flags.p, excl.p, size_t(n_nodes), false, nullptr);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_pack_wide, dim3(grid_for(n_nodes)), dim3(kBB), 0, nullptr, n_nodes, d_nodes, excl.p, d_remap, d_wide, d_wide4, bad.p);
    HIP_TRYB(hipGetLastError());
    int not_nested = 0;
    HIP_TRYB(hipMemcpy(&not_nested, bad.p, sizeof(int), hipMemcpyDeviceToHost));
    *nested_out = not_nested ? 0 : 1;
    (void)n_interior;
    return IILE_OK;
}

}  // namespace iile

using namespace iile;

extern "C" int iile_bvh_build_hlbvh(int32_t n_prims, const float *bounds6, int32_t max_prims_in_node, iile_bvh_node *nodes_out,
                                    int32_t *n_nodes_out, int32_t *order_out, iile_bvh_build_stats *stats) {
    if (n_prims < 0 || (n_prims > 0 && (!bounds6 || !nodes_out || !order_out)) || !n_nodes_out)
        return api_fail(IILE_ERR_ARG, "iile_bvh_build_hlbvh: null argument");
    int dev_count = 0;
    if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count <= 0)
        return api_fail(IILE_ERR_NO_DEVICE, "no HIP device available: libiile_gpu has no CPU fallback (iile_bvh_build_hlbvh)");
    iile_bvh_build_stats st;
    std::memset(&st, 0, sizeof(st));
    *n_nodes_out = 0;
    if (n_prims == 0) {
        if (stats) *stats = st;
        return IILE_OK;
    }
    if (n_prims > (1 << 30)) return api_fail(IILE_ERR_UNSUPPORTED, "iile_bvh_build_hlbvh: more than 2^30 primitives");
    if (max_prims_in_node < 1) return api_fail(IILE_ERR_ARG, "iile_bvh_build_hlbvh: max_prims_in_node must be at least 1");
    const int n = n_prims;
    const int max_prims = std::min(255, max_prims_in_node);  // BVHAccel's constructor, bvh.cpp:187
    hipEvent_t ev[7] = {};
    struct EvGuard {
        hipEvent_t *e;
        ~EvGuard() {
            for (int i = 0; i < 7; ++i)
                if (e[i]) (void)hipEventDestroy(e[i]);
        }
    } guard{ev};
    for (hipEvent_t &e : ev) HIP_TRYB(hipEventCreate(&e));
    hipStream_t s = nullptr;

    Dev<float> d_bounds;
    Dev<uint32_t> keys6, codes, codes_sorted;
    Dev<int> numbers, numbers_sorted, flags, incl, starts, n_nodes_t, base;
    Dev<iile_bvh_node> pool, out;
    HIP_TRYB(d_bounds.alloc(6 * size_t(n)));
    HIP_TRYB(keys6.alloc(6));
    HIP_TRYB(codes.alloc(size_t(n)));
    HIP_TRYB(codes_sorted.alloc(size_t(n)));
    HIP_TRYB(numbers.alloc(size_t(n)));
    HIP_TRYB(numbers_sorted.alloc(size_t(n)));
    HIP_TRYB(flags.alloc(size_t(n)));
    HIP_TRYB(incl.alloc(size_t(n)));
    HIP_TRYB(starts.alloc(size_t(n) + 1));
    HIP_TRYB(pool.alloc(2 * size_t(n)));
    // (everything is allocated before the first kernel: a hipMalloc between two stages costs more than the upper tree's kernel)
    Dev<Box> d_roots, d_cbox;
    Dev<int> up_ints;  // the upper tree's index arrays, queues and per-node words (UpperDev), one allocation
    HIP_TRYB(d_roots.alloc(size_t(kUpMax)));
    HIP_TRYB(d_cbox.alloc(size_t(kUpMax)));
    HIP_TRYB(up_ints.alloc(14 * size_t(kUpMax) + 8));
    HIP_TRYB(out.alloc(2 * size_t(n)));  // 2 n - 1 nodes at most (one primitive per leaf)
    Dev<int> arena;  // emitLBVH's eleven int arrays of n (+ 1) entries in one allocation
    const size_t stride = (size_t(n) + 1 + 63) & ~size_t(63);
    HIP_TRYB(arena.alloc(11 * stride));
    HIP_TRYB(n_nodes_t.alloc(size_t(kUpMax)));
    HIP_TRYB(base.alloc(size_t(kUpMax)));
    Dev<int> err_flag;
    HIP_TRYB(err_flag.alloc(1));
    HIP_TRYB(hipMemcpyAsync(d_bounds.p, bounds6, 6 * size_t(n) * sizeof(float), hipMemcpyHostToDevice, s));
    const uint32_t key_init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    HIP_TRYB(hipMemcpyAsync(keys6.p, key_init, sizeof(key_init), hipMemcpyHostToDevice, s));

    HIP_TRYB(hipEventRecord(ev[0], s));
    hipLaunchKernelGGL(k_centroid_bounds, dim3(std::min(grid_for(n), 512)), dim3(kBB), 0, s, n, d_bounds.p, keys6.p);
    hipLaunchKernelGGL(k_morton, dim3(grid_for(n)), dim3(kBB), 0, s, n, d_bounds.p, keys6.p, codes.p, numbers.p);
    HIP_TRYB(hipEventRecord(ev[1], s));
    {
        const int rc = device_sort_pairs_30(codes.p, numbers.p, codes_sorted.p, numbers_sorted.p, n, s);
        if (rc) return rc;
    }
    HIP_TRYB(hipEventRecord(ev[2], s));
    hipLaunchKernelGGL(k_treelet_flags, dim3(grid_for(n)), dim3(kBB), 0, s, n, codes_sorted.p, flags.p);
    {
        const int rc = device_scan(flags.p, incl.p, size_t(n), true, s);
        if (rc) return rc;
        HIP_TRYB(hipStreamSynchronize(s));
    }
    hipLaunchKernelGGL(k_treelet_starts, dim3(grid_for(n)), dim3(kBB), 0, s, n, flags.p, incl.p, starts.p);
    int n_treelets = 0;  // runs of equal top 12 bits: at most 4096 (read back with the split count below)
    HIP_TRYB(hipMemsetAsync(err_flag.p, 0, sizeof(int), s));
    int n_splits = 0, h_err = 0;
    {
        int *ap[11];
        for (int k = 0; k < 11; ++k) ap[k] = arena.p + size_t(k) * stride;
        int *const aEnds = ap[7], *const aPE = ap[8], *const aI = ap[3], *const aF = ap[4];
        HIP_TRYB(hipMemsetAsync(aEnds, 0, (size_t(n) + 1) * sizeof(int), s));
        LbvhArrays A{codes_sorted.p, numbers_sorted.p, flags.p, incl.p, starts.p, ap[0], ap[1], ap[2], aI, aF, ap[5],
                     ap[6], aEnds, aPE, ap[9], ap[10]};
        hipLaunchKernelGGL(k_lbvh_ranges, dim3(grid_for(n)), dim3(kBB), 0, s, n, A, max_prims);
        int rc = device_scan(aI, aF, size_t(n), true, s);
        if (rc) return rc;
        hipLaunchKernelGGL(k_lbvh_parents, dim3(grid_for(n)), dim3(kBB), 0, s, n, A);
        rc = device_scan(aEnds, aPE, size_t(n) + 1, true, s);
        if (rc) return rc;
        hipLaunchKernelGGL(k_lbvh_interior, dim3(grid_for(n)), dim3(kBB), 0, s, n, A, pool.p);
        hipLaunchKernelGGL(k_lbvh_leaves, dim3(grid_for(n)), dim3(kBB), 0, s, n, A, d_bounds.p, pool.p, n_nodes_t.p, err_flag.p);
        for (int bit = 0; bit < 18; ++bit) hipLaunchKernelGGL(k_lbvh_join, dim3(grid_for(n)), dim3(kBB), 0, s, n, A, pool.p, bit);
        HIP_TRYB(hipGetLastError());
        HIP_TRYB(hipMemcpyAsync(&n_splits, aF + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRYB(hipMemcpyAsync(&h_err, err_flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRYB(hipMemcpyAsync(&n_treelets, incl.p + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRYB(hipStreamSynchronize(s));
    }
    if (h_err) return api_fail(IILE_ERR_UNSUPPORTED, "iile_bvh_build_hlbvh: a leaf holds more than 65535 primitives (equal Morton codes)");
    if (n_treelets > kUpMax) return api_fail(IILE_ERR_UNSUPPORTED, "iile_bvh_build_hlbvh: more than 4096 treelets");
    HIP_TRYB(hipEventRecord(ev[3], s));
    // buildUpperSAH + the preorder offsets of all subtrees, on the device: every treelet has emitted n_nodes_t nodes, the
    // upper tree adds n_treelets - 1
    const int n_upper = n_treelets - 1;
    const int n_nodes = 2 * n_splits + 2 * n_treelets - 1;
    hipLaunchKernelGGL(k_treelet_roots, dim3((n_treelets + kBB - 1) / kBB), dim3(kBB), 0, s, n_treelets, starts.p, pool.p, d_roots.p);
    {
        int *ui = up_ints.p;
        UpperDev U;
        U.roots = d_roots.p, U.n_nodes_t = n_nodes_t.p;
        U.refs = ui, ui += 2 * kUpMax;
        U.qseg = reinterpret_cast<uint32_t *>(ui), ui += 2 * kUpMax;
        U.qlink = ui, ui += 2 * kUpMax;
        U.qcount = ui, ui += kUpMax + 2;
        U.level_base = ui, ui += kUpMax + 2;
        U.node_link = ui, ui += kUpMax;
        U.cref = ui, ui += 2 * kUpMax;
        U.axis = ui, ui += kUpMax;
        U.tparent = ui, ui += kUpMax;
        U.size = ui, ui += kUpMax;
        U.nbox = d_cbox.p, U.base = base.p, U.out = out.p;
        hipLaunchKernelGGL(k_upper_init, dim3(4), dim3(kBB), 0, s, n_treelets, U);
        // the levels of the recursion, a batch of launches at a time (a launch for a level without spans returns at once); the
        // depth is 11-14 for the scenes at hand and at most n_treelets - 1
        std::vector<int> level_count(size_t(kUpMax) + 2, 0);
        int depth = 0;
        const int split_blocks = std::max(1, std::min((n_treelets / 2 + kBB / 64 - 1) / (kBB / 64), 512));
        int batch = kUpBatch;
        if (const char *e = std::getenv("IILE_UPPER_BATCH")) batch = std::max(1, std::min(atoi(e), kUpMax));
        for (int l0 = 0; l0 < n_upper; l0 += batch) {
            const int l1 = std::min(l0 + batch, n_upper);
            for (int level = l0; level < l1; ++level) hipLaunchKernelGGL(k_upper_split, dim3(split_blocks), dim3(kBB), 0, s, level, U);
            HIP_TRYB(hipMemcpyAsync(level_count.data() + l0, U.qcount + l0, size_t(l1 - l0 + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
            HIP_TRYB(hipStreamSynchronize(s));
            depth = l1;
            bool done = false;
            for (int level = l0; level <= l1 && !done; ++level)
                if (level_count[size_t(level)] == 0) depth = level, done = true;
            if (done) break;
        }
        hipLaunchKernelGGL(k_upper_finish, dim3(1), dim3(kUpFinishThreads), 0, s, n_treelets, depth, U);
    }
    HIP_TRYB(hipEventRecord(ev[4], s));
    hipLaunchKernelGGL(k_place_nodes, dim3(grid_for(2 * n)), dim3(kBB), 0, s, n, incl.p, starts.p, n_nodes_t.p, base.p, pool.p, out.p);
    HIP_TRYB(hipGetLastError());
    HIP_TRYB(hipEventRecord(ev[5], s));
    HIP_TRYB(hipMemcpyAsync(nodes_out, out.p, size_t(n_nodes) * sizeof(iile_bvh_node), hipMemcpyDeviceToHost, s));
    HIP_TRYB(hipMemcpyAsync(order_out, numbers_sorted.p, size_t(n) * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRYB(hipEventRecord(ev[6], s));
    HIP_TRYB(hipStreamSynchronize(s));
    *n_nodes_out = n_nodes;
    auto ms = [&](int a, int b) {
        float v = 0;
        (void)hipEventElapsedTime(&v, ev[a], ev[b]);
        return v;
    };
    st.ms_morton = ms(0, 1), st.ms_sort = ms(1, 2), st.ms_treelets = ms(2, 3), st.ms_upper = ms(3, 4), st.ms_flatten = ms(4, 5);
    st.ms_download = ms(5, 6), st.ms_total = ms(0, 6);
    st.n_treelets = n_treelets;
    st.n_nodes = n_nodes;
    st.n_interior = n_splits + n_upper;
    st.n_leaf = n_splits + n_treelets;
    if (stats) *stats = st;
    return IILE_OK;
}

// Test probe for pack_wide_records: the records of a flattened tree handed over by the host.
extern "C" int32_t iile_wide_ref_shift(void) { return kRefShift; }

extern "C" int iile_bvh_pack_probe(int32_t n_nodes, const iile_bvh_node *nodes, int32_t n_interior, float *wide16, float *wide4_32,
                                   int32_t *nested) {
    if (n_nodes <= 0 || !nodes || !wide16 || !wide4_32 || !nested) return api_fail(IILE_ERR_ARG, "iile_bvh_pack_probe: bad argument");
    // the checks iile_scene_create makes before it packs a tree it was handed: the kernel indexes by these fields, and the
    // caller's buffers are sized by n_interior
    int counted = 0;
    for (int i = 0; i < n_nodes; ++i) {
        const iile_bvh_node &nd = nodes[i];
        if (nd.nprims > 0) {
            if (nd.offset < 0) return api_fail(IILE_ERR_ARG, "iile_bvh_pack_probe: bad leaf range");
            continue;
        }
        if (i + 1 >= n_nodes || nd.offset <= i || nd.offset >= n_nodes) return api_fail(IILE_ERR_ARG, "iile_bvh_pack_probe: bad BVH child index");
        ++counted;
    }
    if (counted != n_interior) return api_fail(IILE_ERR_ARG, "iile_bvh_pack_probe: n_interior does not match the tree");
    int dev_count = 0;
    if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count <= 0)
        return api_fail(IILE_ERR_NO_DEVICE, "no HIP device available: libiile_gpu has no CPU fallback (iile_bvh_pack_probe)");
    Dev<iile_bvh_node> d_nodes;
    Dev<float4> w, w4;
    HIP_TRYB(d_nodes.alloc(size_t(n_nodes)));
    HIP_TRYB(w.alloc(4 * size_t(std::max(n_interior, 1))));
    HIP_TRYB(w4.alloc(8 * size_t(std::max(n_interior, 1))));
    HIP_TRYB(hipMemcpy(d_nodes.p, nodes, size_t(n_nodes) * sizeof(iile_bvh_node), hipMemcpyHostToDevice));
    int nest = 1;
    const int rc = pack_wide_records(d_nodes.p, n_nodes, n_interior, w.p, w4.p, &nest);
    if (rc) return rc;
    HIP_TRYB(hipMemcpy(wide16, w.p, 4 * size_t(n_interior) * sizeof(float4), hipMemcpyDeviceToHost));
    HIP_TRYB(hipMemcpy(wide4_32, w4.p, 8 * size_t(n_interior) * sizeof(float4), hipMemcpyDeviceToHost));
    *nested = nest;
    return IILE_OK;
}
