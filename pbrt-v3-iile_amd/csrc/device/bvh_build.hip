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
//   host: buildUpperSAH (:620-638+) over the <= 4096 treelet roots — a sequential SAH with std::partition over a few
//                       thousand boxes (the reference runs it on one thread too) — and the preorder offsets of the subtrees
//   k_place_nodes / k_place_upper   flattenBVHTree: every node to its depth-first index, second-child offsets rebased
//
// The result must equal the host builder's (csrc/host/bvh_build.cpp, split method "hlbvh") node for node:
// tests/test_gpu_bvh_build.py. Float arithmetic is IEEE (no contraction, correctly rounded division) like the rest.
//
// pack_wide_records: the two-wide and four-wide interior records of DESIGN.md §3 from a flattened tree in HBM — what
// iile_scene_create did in host loops — one thread per node.
#include <hip/hip_runtime.h>

#include <algorithm>
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
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; ++a) {
            atomicMin(&keys6[a], order_key(mn[a]));
            atomicMax(&keys6[3 + a], order_key(mx[a]));
        }
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
//   k_lbvh_leaves   one thread per leaf: box of its primitives, then upwards — the second child to arrive at a parent
//                   (atomic counter, device-scope fences) joins the two boxes and continues
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
        // upwards: the second arrival at a parent owns it
        for (int q = p; q >= 0; q = A.par[q]) {
            __threadfence();
            if (atomicAdd(&A.visit[q], 1) == 0) break;
            __threadfence();
            const int qp = A.pre[q];
            iile_bvh_node me = out[qp];
            const iile_bvh_node a = out[qp + 1], b = out[me.offset];
            for (int c = 0; c < 3; ++c) {
                me.bmin[c] = b.bmin[c] < a.bmin[c] ? b.bmin[c] : a.bmin[c];
                me.bmax[c] = a.bmax[c] < b.bmax[c] ? b.bmax[c] : a.bmax[c];
            }
            out[qp] = me;
        }
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
struct PlacedNode {
    int index;
    iile_bvh_node node;
};
__global__ __launch_bounds__(kBB) void k_place_upper(int n, const PlacedNode *upper, iile_bvh_node *out) {
    const int i = blockIdx.x * kBB + threadIdx.x;
    if (i < n) out[upper[i].index] = upper[i].node;
}

// ---- buildUpperSAH on the host (bvh.cpp:474-553): 12-bucket SAH over the treelet roots -----------------------------
struct HBox {
    float mn[3], mx[3];
    HBox() {  // Bounds3(), geometry.h:752-757
        for (int a = 0; a < 3; ++a) mn[a] = std::numeric_limits<float>::max(), mx[a] = std::numeric_limits<float>::lowest();
    }
    void add(const HBox &b) {
        for (int a = 0; a < 3; ++a) mn[a] = std::min(mn[a], b.mn[a]), mx[a] = std::max(mx[a], b.mx[a]);
    }
    void add_point(const float p[3]) {
        for (int a = 0; a < 3; ++a) mn[a] = std::min(mn[a], p[a]), mx[a] = std::max(mx[a], p[a]);
    }
    float area() const {  // geometry.h:782-785
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return 2 * (dx * dy + dx * dz + dy * dz);
    }
    int max_extent() const {  // geometry.h:790-798
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx > dy && dx > dz) return 0;
        return dy > dz ? 1 : 2;
    }
};
struct UpperNode {
    HBox box;
    int child[2];  // >= 0: upper node, < 0: ~treelet
    int axis;
};
struct UpperBuilder {
    const std::vector<HBox> &roots;  // per treelet
    std::vector<UpperNode> nodes;
    explicit UpperBuilder(const std::vector<HBox> &r) : roots(r) {}
    const HBox &box_of(int ref) const { return ref >= 0 ? nodes[size_t(ref)].box : roots[size_t(~ref)]; }

    int build(std::vector<int> &refs, int start, int end) {
        if (end - start == 1) return refs[size_t(start)];
        nodes.emplace_back();
        const int node = int(nodes.size()) - 1;
        HBox bounds, cb;
        for (int i = start; i < end; ++i) bounds.add(box_of(refs[size_t(i)]));
        for (int i = start; i < end; ++i) {
            const HBox &b = box_of(refs[size_t(i)]);
            const float c[3] = {(b.mn[0] + b.mx[0]) * 0.5f, (b.mn[1] + b.mx[1]) * 0.5f, (b.mn[2] + b.mx[2]) * 0.5f};
            cb.add_point(c);
        }
        const int dim = cb.max_extent();
        constexpr int nBuckets = 12;
        struct Bucket {
            int count = 0;
            HBox bounds;
        } buckets[nBuckets];
        const float lo = cb.mn[dim], hi = cb.mx[dim];
        auto bucket_of = [&](int ref) {
            const HBox &b = box_of(ref);
            const float centroid = (b.mn[dim] + b.mx[dim]) * 0.5f;
            int k = int(nBuckets * ((centroid - lo) / (hi - lo)));
            if (k == nBuckets) k = nBuckets - 1;
            return k;
        };
        for (int i = start; i < end; ++i) {
            const int k = bucket_of(refs[size_t(i)]);
            buckets[k].count++;
            buckets[k].bounds.add(box_of(refs[size_t(i)]));
        }
        float cost[nBuckets - 1];
        for (int i = 0; i < nBuckets - 1; ++i) {
            HBox b0, b1;
            int c0 = 0, c1 = 0;
            for (int j = 0; j <= i; ++j) b0.add(buckets[j].bounds), c0 += buckets[j].count;
            for (int j = i + 1; j < nBuckets; ++j) b1.add(buckets[j].bounds), c1 += buckets[j].count;
            cost[i] = .125f + (c0 * b0.area() + c1 * b1.area()) / bounds.area();
        }
        float min_cost = cost[0];
        int min_bucket = 0;
        for (int i = 1; i < nBuckets - 1; ++i)
            if (cost[i] < min_cost) min_cost = cost[i], min_bucket = i;
        int *pmid = std::partition(&refs[size_t(start)], &refs[size_t(end - 1)] + 1, [&](int r) { return bucket_of(r) <= min_bucket; });
        const int mid = int(pmid - &refs[0]);
        const int c0 = build(refs, start, mid);
        const int c1 = build(refs, mid, end);
        UpperNode &nd = nodes[size_t(node)];
        nd.child[0] = c0, nd.child[1] = c1;
        nd.box = box_of(c0);
        nd.box.add(box_of(c1));
        nd.axis = dim;
        return node;
    }
};

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
        if (kRefShift) {  // IILE_AXES_IN_REFS: ref << 2 | axis of {the node, its first child, its second child, -}
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
    HIP_TRYB(hipMemcpyAsync(d_bounds.p, bounds6, 6 * size_t(n) * sizeof(float), hipMemcpyHostToDevice, s));
    const uint32_t key_init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    HIP_TRYB(hipMemcpyAsync(keys6.p, key_init, sizeof(key_init), hipMemcpyHostToDevice, s));

    HIP_TRYB(hipEventRecord(ev[0], s));
    hipLaunchKernelGGL(k_centroid_bounds, dim3(grid_for(n)), dim3(kBB), 0, s, n, d_bounds.p, keys6.p);
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
    int n_treelets = 0;
    HIP_TRYB(hipMemcpyAsync(&n_treelets, incl.p + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRYB(hipStreamSynchronize(s));
    HIP_TRYB(n_nodes_t.alloc(size_t(n_treelets)));
    HIP_TRYB(base.alloc(size_t(n_treelets)));
    Dev<int> err_flag;
    HIP_TRYB(err_flag.alloc(1));
    HIP_TRYB(hipMemsetAsync(err_flag.p, 0, sizeof(int), s));
    int n_splits = 0;
    {
        Dev<int> arena;  // eleven int arrays of n (+ 1) entries in one allocation
        const size_t stride = (size_t(n) + 1 + 63) & ~size_t(63);
        HIP_TRYB(arena.alloc(11 * stride));
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
        HIP_TRYB(hipGetLastError());
        HIP_TRYB(hipMemcpyAsync(&n_splits, aF + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRYB(hipStreamSynchronize(s));
    }
    HIP_TRYB(hipEventRecord(ev[3], s));
    // the treelet roots come to the host: the upper SAH tree and the preorder offsets of all subtrees
    Dev<Box> d_roots;
    HIP_TRYB(d_roots.alloc(size_t(n_treelets)));
    hipLaunchKernelGGL(k_treelet_roots, dim3((n_treelets + kBB - 1) / kBB), dim3(kBB), 0, s, n_treelets, starts.p, pool.p, d_roots.p);
    std::vector<HBox> roots(static_cast<size_t>(n_treelets));
    std::vector<int> counts(static_cast<size_t>(n_treelets));
    int h_err = 0;
    static_assert(sizeof(HBox) == sizeof(Box), "box layouts");
    HIP_TRYB(hipMemcpyAsync(roots.data(), d_roots.p, size_t(n_treelets) * sizeof(Box), hipMemcpyDeviceToHost, s));
    HIP_TRYB(hipMemcpyAsync(counts.data(), n_nodes_t.p, size_t(n_treelets) * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRYB(hipMemcpyAsync(&h_err, err_flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRYB(hipStreamSynchronize(s));
    if (h_err) return api_fail(IILE_ERR_UNSUPPORTED, "iile_bvh_build_hlbvh: a leaf holds more than 65535 primitives (equal Morton codes)");
    UpperBuilder ub(roots);
    std::vector<int> refs(static_cast<size_t>(n_treelets));
    for (int t = 0; t < n_treelets; ++t) refs[size_t(t)] = ~t;
    ub.nodes.reserve(size_t(n_treelets));
    const int root = ub.build(refs, 0, n_treelets);
    // flattenBVHTree over the upper tree; a treelet reference places its whole preorder block
    std::vector<int> h_base(static_cast<size_t>(n_treelets));
    std::vector<PlacedNode> placed;
    placed.reserve(ub.nodes.size());
    int cursor = 0;
    {
        struct Item {
            int ref, placed_index;  // placed_index >= 0: the second child of that placed node starts here
        };
        // iterative preorder: explicit stack of (ref, owner whose second-child offset this is)
        std::vector<Item> stack;
        stack.push_back({root, -1});
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            if (it.placed_index >= 0) placed[size_t(it.placed_index)].node.offset = cursor;
            if (it.ref < 0) {
                h_base[size_t(~it.ref)] = cursor;
                cursor += counts[size_t(~it.ref)];
                continue;
            }
            const UpperNode &u = ub.nodes[size_t(it.ref)];
            PlacedNode pn;
            std::memset(&pn, 0, sizeof(pn));
            pn.index = cursor++;
            for (int a = 0; a < 3; ++a) pn.node.bmin[a] = u.box.mn[a], pn.node.bmax[a] = u.box.mx[a];
            pn.node.nprims = 0;
            pn.node.axis = uint8_t(u.axis);
            placed.push_back(pn);
            const int me = int(placed.size()) - 1;
            stack.push_back({u.child[1], me});  // visited after the whole first subtree
            stack.push_back({u.child[0], -1});
        }
    }
    const int n_nodes = cursor;
    HIP_TRYB(hipEventRecord(ev[4], s));
    HIP_TRYB(out.alloc(size_t(n_nodes)));
    HIP_TRYB(hipMemcpyAsync(base.p, h_base.data(), size_t(n_treelets) * sizeof(int), hipMemcpyHostToDevice, s));
    Dev<PlacedNode> d_placed;
    HIP_TRYB(d_placed.alloc(placed.size()));
    if (!placed.empty())
        HIP_TRYB(hipMemcpyAsync(d_placed.p, placed.data(), placed.size() * sizeof(PlacedNode), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_place_nodes, dim3(grid_for(2 * n)), dim3(kBB), 0, s, n, incl.p, starts.p, n_nodes_t.p, base.p, pool.p, out.p);
    if (!placed.empty())
        hipLaunchKernelGGL(k_place_upper, dim3((int(placed.size()) + kBB - 1) / kBB), dim3(kBB), 0, s, int(placed.size()), d_placed.p, out.p);
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
    st.n_interior = n_splits + int(ub.nodes.size());
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
