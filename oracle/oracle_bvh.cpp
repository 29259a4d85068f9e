// oracle_bvh.cpp — CPU restatement of BVHAccel::HLBVHBuild for SURVEY.md §8 f4. TEST INFRASTRUCTURE: only tests/ load
// it (through oracle/_build/liboracle.so); nothing of the product includes, links or calls it, and it shares no code with
// the product's builders (pbrt-v3-iile_amd/csrc/host/bvh_build.cpp, csrc/device/bvh_build.hip).
//
// What it follows, operation for operation, as ONE thread runs it (with several threads the reference's leaf order depends
// on which treelet reaches orderedPrimsOffset first):
//   BVHPrimitiveInfo                    src/accelerators/bvh.cpp:50-59    centroid = .5f * pMin + .5f * pMax
//   LeftShift3 / EncodeMorton3          bvh.cpp:107-138
//   RadixSort                           bvh.cpp:140-181                   five stable passes over 6 bits each (bits 0..29)
//   BVHAccel::HLBVHBuild                bvh.cpp:404-472                   centroid bounds, Morton codes, treelets by the top 12 bits
//   BVHAccel::emitLBVH                  bvh.cpp:474-553                   the recursion itself (no closed form), nodes in emission order
//   BVHAccel::buildUpperSAH             bvh.cpp:555-638                   12 buckets, std::partition
//   BVHAccel::flattenBVHTree            bvh.cpp:640-658
//   Bounds3 Union / Offset / SurfaceArea / MaximumExtent   src/core/geometry.h:779-807, 1100-1123
// Parity pin: tests/test_oracle_bvh.py holds a 12-primitive tree worked out by hand from those lines.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/iile_scene.h"

namespace {

struct P3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct B3 {  // Bounds3f(): pMin = max float, pMax = lowest float (geometry.h:716-720)
    P3 mn{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    P3 mx{std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest()};
};
B3 unite(const B3 &a, const B3 &b) {  // geometry.h:1108-1116
    B3 r;
    r.mn = P3{std::min(a.mn.x, b.mn.x), std::min(a.mn.y, b.mn.y), std::min(a.mn.z, b.mn.z)};
    r.mx = P3{std::max(a.mx.x, b.mx.x), std::max(a.mx.y, b.mx.y), std::max(a.mx.z, b.mx.z)};
    return r;
}
B3 unite(const B3 &a, const P3 &p) {  // geometry.h:1100-1106
    B3 r;
    r.mn = P3{std::min(a.mn.x, p.x), std::min(a.mn.y, p.y), std::min(a.mn.z, p.z)};
    r.mx = P3{std::max(a.mx.x, p.x), std::max(a.mx.y, p.y), std::max(a.mx.z, p.z)};
    return r;
}
float surface_area(const B3 &b) {
    const float dx = b.mx.x - b.mn.x, dy = b.mx.y - b.mn.y, dz = b.mx.z - b.mn.z;
    return 2 * (dx * dy + dx * dz + dy * dz);
}
int maximum_extent(const B3 &b) {
    const float dx = b.mx.x - b.mn.x, dy = b.mx.y - b.mn.y, dz = b.mx.z - b.mn.z;
    if (dx > dy && dx > dz) return 0;
    if (dy > dz) return 1;
    return 2;
}

struct PrimInfo {
    int number;
    B3 bounds;
    P3 centroid;
};
struct BuildNode {
    B3 bounds;
    BuildNode *children[2] = {nullptr, nullptr};
    int split_axis = 0, first_prim = 0, n_prims = 0;
};
struct MortonPrim {
    int prim;
    uint32_t code;
};

uint32_t left_shift3(uint32_t x) {
    if (x == (1u << 10)) --x;
    x = (x | (x << 16)) & 0x30000ffu;
    x = (x | (x << 8)) & 0x300f00fu;
    x = (x | (x << 4)) & 0x30c30c3u;
    x = (x | (x << 2)) & 0x9249249u;
    return x;
}
uint32_t encode_morton3(float vx, float vy, float vz) {  // float -> uint32_t as the call LeftShift3(v.z) converts
    return (left_shift3(uint32_t(vz)) << 2) | (left_shift3(uint32_t(vy)) << 1) | left_shift3(uint32_t(vx));
}
void radix_sort(std::vector<MortonPrim> *v) {
    std::vector<MortonPrim> temp(v->size());
    constexpr int bits_per_pass = 6, n_bits = 30, n_passes = n_bits / bits_per_pass;
    for (int pass = 0; pass < n_passes; ++pass) {
        const int low_bit = pass * bits_per_pass;
        std::vector<MortonPrim> &in = (pass & 1) ? temp : *v;
        std::vector<MortonPrim> &out = (pass & 1) ? *v : temp;
        constexpr int n_buckets = 1 << bits_per_pass, bit_mask = n_buckets - 1;
        int bucket_count[n_buckets] = {0};
        for (const MortonPrim &mp : in) ++bucket_count[(mp.code >> low_bit) & bit_mask];
        int out_index[n_buckets];
        out_index[0] = 0;
        for (int i = 1; i < n_buckets; ++i) out_index[i] = out_index[i - 1] + bucket_count[i - 1];
        for (const MortonPrim &mp : in) out[out_index[(mp.code >> low_bit) & bit_mask]++] = mp;
    }
    if (n_passes & 1) std::swap(*v, temp);
}

struct Builder {
    const std::vector<PrimInfo> &info;
    int max_prims_in_node;
    Builder(const std::vector<PrimInfo> &i, int m) : info(i), max_prims_in_node(m) {}
    std::vector<int> ordered;     // orderedPrims: primitive number per final position
    int ordered_offset = 0;       // orderedPrimsOffset (one thread: treelets in index order)
    bool failed = false;          // a CHECK of the reference would have aborted
    std::vector<BuildNode *> upper;  // nodes buildUpperSAH allocates (arena)
    ~Builder() {
        for (BuildNode *n : upper) delete n;
    }

    BuildNode *emit_lbvh(BuildNode *&build_nodes, MortonPrim *mp, int n, int *total, int bit_index) {
        if (bit_index == -1 || n < max_prims_in_node) {
            (*total)++;
            BuildNode *node = build_nodes++;
            B3 bounds;
            const int first = ordered_offset;
            ordered_offset += n;
            for (int i = 0; i < n; ++i) {
                ordered[size_t(first + i)] = mp[i].prim;
                bounds = unite(bounds, info[size_t(mp[i].prim)].bounds);
            }
            node->first_prim = first, node->n_prims = n, node->bounds = bounds;
            node->children[0] = node->children[1] = nullptr;
            return node;
        }
        const uint32_t mask = 1u << bit_index;
        if ((mp[0].code & mask) == (mp[n - 1].code & mask)) return emit_lbvh(build_nodes, mp, n, total, bit_index - 1);
        int search_start = 0, search_end = n - 1;
        while (search_start + 1 != search_end) {
            const int mid = (search_start + search_end) / 2;
            if ((mp[search_start].code & mask) == (mp[mid].code & mask))
                search_start = mid;
            else
                search_end = mid;
        }
        const int split = search_end;
        (*total)++;
        BuildNode *node = build_nodes++;
        BuildNode *c0 = emit_lbvh(build_nodes, mp, split, total, bit_index - 1);
        BuildNode *c1 = emit_lbvh(build_nodes, mp + split, n - split, total, bit_index - 1);
        node->children[0] = c0, node->children[1] = c1;
        node->bounds = unite(c0->bounds, c1->bounds);
        node->split_axis = bit_index % 3;
        node->n_prims = 0;
        return node;
    }

    BuildNode *build_upper_sah(std::vector<BuildNode *> &roots, int start, int end, int *total) {
        const int n_nodes = end - start;
        if (n_nodes == 1) return roots[size_t(start)];
        (*total)++;
        BuildNode *node = new BuildNode;
        upper.push_back(node);
        B3 bounds;
        for (int i = start; i < end; ++i) bounds = unite(bounds, roots[size_t(i)]->bounds);
        B3 cb;
        for (int i = start; i < end; ++i) {
            const B3 &b = roots[size_t(i)]->bounds;
            // (pMin + pMax) * 0.5f
            cb = unite(cb, P3{(b.mn.x + b.mx.x) * 0.5f, (b.mn.y + b.mx.y) * 0.5f, (b.mn.z + b.mx.z) * 0.5f});
        }
        const int dim = maximum_extent(cb);
        if (cb.mx[dim] == cb.mn[dim]) {  // CHECK_NE(centroidBounds.pMax[dim], centroidBounds.pMin[dim])
            failed = true;
            return node;
        }
        constexpr int n_buckets = 12;
        struct Bucket {
            int count = 0;
            B3 bounds;
        } buckets[n_buckets];
        auto bucket_of = [&](const BuildNode *nd) {
            const float centroid = (nd->bounds.mn[dim] + nd->bounds.mx[dim]) * 0.5f;
            int b = int(n_buckets * ((centroid - cb.mn[dim]) / (cb.mx[dim] - cb.mn[dim])));
            if (b == n_buckets) b = n_buckets - 1;
            return b;
        };
        for (int i = start; i < end; ++i) {
            const int b = bucket_of(roots[size_t(i)]);
            buckets[b].count++;
            buckets[b].bounds = unite(buckets[b].bounds, roots[size_t(i)]->bounds);
        }
        float cost[n_buckets - 1];
        for (int i = 0; i < n_buckets - 1; ++i) {
            B3 b0, b1;
            int count0 = 0, count1 = 0;
            for (int j = 0; j <= i; ++j) {
                b0 = unite(b0, buckets[j].bounds);
                count0 += buckets[j].count;
            }
            for (int j = i + 1; j < n_buckets; ++j) {
                b1 = unite(b1, buckets[j].bounds);
                count1 += buckets[j].count;
            }
            cost[i] = .125f + (count0 * surface_area(b0) + count1 * surface_area(b1)) / surface_area(bounds);
        }
        float min_cost = cost[0];
        int min_bucket = 0;
        for (int i = 1; i < n_buckets - 1; ++i)
            if (cost[i] < min_cost) {
                min_cost = cost[i];
                min_bucket = i;
            }
        BuildNode **pmid = std::partition(&roots[size_t(start)], &roots[size_t(end - 1)] + 1,
                                          [&](const BuildNode *nd) { return bucket_of(nd) <= min_bucket; });
        const int mid = int(pmid - &roots[0]);
        if (!(mid > start && mid < end)) {  // CHECK_GT(mid, start); CHECK_LT(mid, end)
            failed = true;
            return node;
        }
        BuildNode *c0 = build_upper_sah(roots, start, mid, total);
        BuildNode *c1 = failed ? c0 : build_upper_sah(roots, mid, end, total);
        node->children[0] = c0, node->children[1] = c1;
        node->bounds = unite(c0->bounds, c1->bounds);
        node->split_axis = dim;
        node->n_prims = 0;
        return node;
    }
};

int flatten(const BuildNode *node, iile_bvh_node *nodes, int *offset) {
    iile_bvh_node *ln = &nodes[*offset];
    std::memset(ln, 0, sizeof(*ln));
    ln->bmin[0] = node->bounds.mn.x, ln->bmin[1] = node->bounds.mn.y, ln->bmin[2] = node->bounds.mn.z;
    ln->bmax[0] = node->bounds.mx.x, ln->bmax[1] = node->bounds.mx.y, ln->bmax[2] = node->bounds.mx.z;
    const int my_offset = (*offset)++;
    if (node->n_prims > 0) {
        ln->offset = node->first_prim;
        ln->nprims = uint16_t(node->n_prims);
    } else {
        ln->axis = uint8_t(node->split_axis);
        ln->nprims = 0;
        flatten(node->children[0], nodes, offset);
        ln->offset = flatten(node->children[1], nodes, offset);
    }
    return my_offset;
}

}  // namespace

extern "C" {

// The tree BVHAccel(prims, maxPrimsInNode, SplitMethod::HLBVH) builds on one thread over primitives with the world
// bounds bounds6[i] = {pMin, pMax}: nodes_out (room for 2 * n_prims) in flattenBVHTree's depth-first order, order_out[p]
// = number of the primitive that ends up at position p of BVHAccel::primitives. Optionally the sorted Morton codes
// (codes_out, n_prims of them: test probes). Returns 0, or 1 where a CHECK of the reference would abort (all treelet
// centroids equal along the split axis), or 2 on bad arguments.
int oracle_bvh_hlbvh(int32_t n_prims, const float *bounds6, int32_t max_prims_in_node, iile_bvh_node *nodes_out, int32_t *n_nodes_out,
                     int32_t *order_out, uint32_t *codes_out) {
    if (n_prims < 0 || !n_nodes_out || (n_prims > 0 && (!bounds6 || !nodes_out || !order_out))) return 2;
    *n_nodes_out = 0;
    if (n_prims == 0) return 0;  // BVHAccel::BVHAccel returns before building (bvh.cpp:189)
    const int max_prims = std::min(255, max_prims_in_node);  // bvh.cpp:186
    std::vector<PrimInfo> info;
    info.resize(size_t(n_prims));
    for (int i = 0; i < n_prims; ++i) {
        const float *b = bounds6 + 6 * size_t(i);
        PrimInfo &pi = info[size_t(i)];
        pi.number = i;
        pi.bounds.mn = P3{b[0], b[1], b[2]};
        pi.bounds.mx = P3{b[3], b[4], b[5]};
        pi.centroid = P3{.5f * b[0] + .5f * b[3], .5f * b[1] + .5f * b[4], .5f * b[2] + .5f * b[5]};
    }
    B3 bounds;
    for (const PrimInfo &pi : info) bounds = unite(bounds, pi.centroid);
    std::vector<MortonPrim> mp;
    mp.resize(size_t(n_prims));
    for (int i = 0; i < n_prims; ++i) {
        constexpr int morton_scale = 1 << 10;
        mp[size_t(i)].prim = info[size_t(i)].number;
        // bounds.Offset(centroid)
        const P3 &c = info[size_t(i)].centroid;
        float ox = c.x - bounds.mn.x, oy = c.y - bounds.mn.y, oz = c.z - bounds.mn.z;
        if (bounds.mx.x > bounds.mn.x) ox /= bounds.mx.x - bounds.mn.x;
        if (bounds.mx.y > bounds.mn.y) oy /= bounds.mx.y - bounds.mn.y;
        if (bounds.mx.z > bounds.mn.z) oz /= bounds.mx.z - bounds.mn.z;
        mp[size_t(i)].code = encode_morton3(ox * morton_scale, oy * morton_scale, oz * morton_scale);
    }
    radix_sort(&mp);
    if (codes_out)
        for (int i = 0; i < n_prims; ++i) codes_out[i] = mp[size_t(i)].code;
    struct Treelet {
        int start, n;
        std::vector<BuildNode> storage;  // arena.Alloc<BVHBuildNode>(2 * nPrimitives)
        BuildNode *root = nullptr;
    };
    std::vector<Treelet> treelets;
    for (int start = 0, end = 1; end <= n_prims; ++end) {
        const uint32_t mask = 0x3ffc0000u;
        if (end == n_prims || ((mp[size_t(start)].code & mask) != (mp[size_t(end)].code & mask))) {
            {
            Treelet tr;
            tr.start = start, tr.n = end - start;
            treelets.push_back(std::move(tr));
            }
            start = end;
        }
    }
    Builder bld(info, max_prims);
    bld.ordered.assign(size_t(n_prims), -1);
    int total = 0;
    for (Treelet &tr : treelets) {
        tr.storage.resize(size_t(2 * tr.n));
        BuildNode *cursor = tr.storage.data();
        int created = 0;
        const int first_bit_index = 29 - 12;
        tr.root = bld.emit_lbvh(cursor, &mp[size_t(tr.start)], tr.n, &created, first_bit_index);
        total += created;
    }
    std::vector<BuildNode *> finished;
    finished.reserve(treelets.size());
    for (Treelet &tr : treelets) finished.push_back(tr.root);
    BuildNode *root = bld.build_upper_sah(finished, 0, int(finished.size()), &total);
    if (bld.failed) return 1;
    int offset = 0;
    flatten(root, nodes_out, &offset);
    if (offset != total) return 1;  // CHECK_EQ(totalNodes, offset), bvh.cpp:232
    *n_nodes_out = total;
    for (int i = 0; i < n_prims; ++i) order_out[i] = bld.ordered[size_t(i)];
    return 0;
}

}  // extern "C"
