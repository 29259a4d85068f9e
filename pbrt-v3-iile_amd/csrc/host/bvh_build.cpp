// bvh_build.cpp — SAH BVH construction and depth-first flattening.
//
// Host-side, once per scene. Restates
// /root/reference/src/accelerators/bvh.cpp:183-402 (recursiveBuild, SAH with
// 12 buckets, leaf when nPrims <= maxPrimsInNode and leafCost <= splitCost) and
// :640-658 (flattenBVHTree: first child at i+1, second at secondChildOffset).
// The node layout and near/far visiting order are what make equal-t ties
// resolve as in the reference (SURVEY.md §7 "Tie-breaking"), so the tree is
// reproduced exactly, including std::partition / std::nth_element for the
// in-range primitive order.
#include <memory>
#include <stdexcept>
#include <string>

#include "host_scene.h"

namespace iile {
namespace {

struct PrimInfo {
    size_t number;
    Bounds3 bounds;
    V3 centroid;
};

struct BuildNode {
    Bounds3 bounds;
    int children[2] = {-1, -1};
    int split_axis = 0, first_prim = 0, n_prims = 0;
};

enum SplitMethod { kSAH, kHLBVH, kMiddle, kEqualCounts };

struct MortonPrim {
    int prim_index;
    uint32_t code;
};

struct Builder {
    int max_prims_in_node;
    int split_method = kSAH;
    std::vector<PrimInfo> info;
    std::vector<BuildNode> nodes;
    std::vector<size_t> ordered;  // creation-order primitive numbers in leaf order
    int n_interior = 0, n_leaf = 0;

    int make_leaf(int node, int start, int end, const Bounds3 &b) {
        int first = int(ordered.size());
        for (int i = start; i < end; ++i) ordered.push_back(info[i].number);
        nodes[node].first_prim = first;
        nodes[node].n_prims = end - start;
        nodes[node].bounds = b;
        ++n_leaf;
        return node;
    }

    int build(int start, int end) {
        nodes.emplace_back();
        const int node = int(nodes.size()) - 1;
        Bounds3 bounds;
        for (int i = start; i < end; ++i) bounds = bunion(bounds, info[i].bounds);
        const int n = end - start;
        if (n == 1) return make_leaf(node, start, end, bounds);
        Bounds3 cb;
        for (int i = start; i < end; ++i) cb = bunion(cb, info[i].centroid);
        const int dim = cb.maximum_extent();
        int mid = (start + end) / 2;
        if (cb.pmax[dim] == cb.pmin[dim]) return make_leaf(node, start, end, bounds);
        bool partitioned = false;
        if (split_method == kMiddle) {  // bvh.cpp:277-291: through the midpoint of the centroids; falls through if one side is empty
            const float pmid = (cb.pmin[dim] + cb.pmax[dim]) / 2;
            PrimInfo *mp = std::partition(&info[start], &info[end - 1] + 1, [dim, pmid](const PrimInfo &pi) { return pi.centroid[dim] < pmid; });
            mid = int(mp - &info[0]);
            partitioned = mid != start && mid != end;
        }
        if (partitioned) {
        } else if (split_method == kMiddle || split_method == kEqualCounts || n <= 2) {  // bvh.cpp:292-303 (and :309-318 for SAH)
            mid = (start + end) / 2;
            std::nth_element(&info[start], &info[mid], &info[end - 1] + 1,
                             [dim](const PrimInfo &a, const PrimInfo &b) {
                                 return a.centroid[dim] < b.centroid[dim];
                             });
        } else {
            constexpr int nBuckets = 12;
            struct Bucket {
                int count = 0;
                Bounds3 bounds;
            } buckets[nBuckets];
            for (int i = start; i < end; ++i) {
                int b = int(nBuckets * cb.offset(info[i].centroid)[dim]);
                if (b == nBuckets) b = nBuckets - 1;
                buckets[b].count++;
                buckets[b].bounds = bunion(buckets[b].bounds, info[i].bounds);
            }
            float cost[nBuckets - 1];
            for (int i = 0; i < nBuckets - 1; ++i) {
                Bounds3 b0, b1;
                int c0 = 0, c1 = 0;
                for (int j = 0; j <= i; ++j) {
                    b0 = bunion(b0, buckets[j].bounds);
                    c0 += buckets[j].count;
                }
                for (int j = i + 1; j < nBuckets; ++j) {
                    b1 = bunion(b1, buckets[j].bounds);
                    c1 += buckets[j].count;
                }
                cost[i] = 1 + (c0 * b0.surface_area() + c1 * b1.surface_area()) / bounds.surface_area();
            }
            float min_cost = cost[0];
            int min_bucket = 0;
            for (int i = 1; i < nBuckets - 1; ++i)
                if (cost[i] < min_cost) {
                    min_cost = cost[i];
                    min_bucket = i;
                }
            float leaf_cost = float(n);
            if (n > max_prims_in_node || min_cost < leaf_cost) {
                PrimInfo *pmid = std::partition(&info[start], &info[end - 1] + 1, [=](const PrimInfo &pi) {
                    int b = int(nBuckets * cb.offset(pi.centroid)[dim]);
                    if (b == nBuckets) b = nBuckets - 1;
                    return b <= min_bucket;
                });
                mid = int(pmid - &info[0]);
            } else
                return make_leaf(node, start, end, bounds);
        }
        int c0 = build(start, mid);
        int c1 = build(mid, end);
        nodes[node].children[0] = c0;
        nodes[node].children[1] = c1;
        nodes[node].bounds = bunion(nodes[c0].bounds, nodes[c1].bounds);
        nodes[node].split_axis = dim;
        nodes[node].n_prims = 0;
        ++n_interior;
        return node;
    }

    // ---- HLBVH, bvh.cpp:107-181, 404-638 (treelets built in index order: what one thread does) ----
    static uint32_t left_shift3(uint32_t x) {
        if (x == (1 << 10)) --x;
        x = (x | (x << 16)) & 0x30000ff;
        x = (x | (x << 8)) & 0x300f00f;
        x = (x | (x << 4)) & 0x30c30c3;
        x = (x | (x << 2)) & 0x9249249;
        return x;
    }
    static uint32_t encode_morton3(V3 v) { return (left_shift3(uint32_t(v.z)) << 2) | (left_shift3(uint32_t(v.y)) << 1) | left_shift3(uint32_t(v.x)); }
    static void radix_sort(std::vector<MortonPrim> *v) {
        std::vector<MortonPrim> temp(v->size());
        const int bits_per_pass = 6, n_bits = 30, n_passes = n_bits / bits_per_pass;
        for (int pass = 0; pass < n_passes; ++pass) {
            const int low_bit = pass * bits_per_pass;
            std::vector<MortonPrim> &in = (pass & 1) ? temp : *v;
            std::vector<MortonPrim> &out = (pass & 1) ? *v : temp;
            const int n_buckets = 1 << bits_per_pass, bit_mask = (1 << bits_per_pass) - 1;
            int count[64] = {0};
            for (const MortonPrim &mp : in) ++count[(mp.code >> low_bit) & bit_mask];
            int out_index[64];
            out_index[0] = 0;
            for (int i = 1; i < n_buckets; ++i) out_index[i] = out_index[i - 1] + count[i - 1];
            for (const MortonPrim &mp : in) out[out_index[(mp.code >> low_bit) & bit_mask]++] = mp;
        }
        if (n_passes & 1) std::swap(*v, temp);
    }
    // `info` stays in creation order here (primitiveInfo[primitiveIndex])
    int emit_lbvh(const MortonPrim *mp, int n, int bit_index) {
        if (bit_index == -1 || n < max_prims_in_node) {
            nodes.emplace_back();
            const int node = int(nodes.size()) - 1;
            Bounds3 bounds;
            const int first = int(ordered.size());
            for (int i = 0; i < n; ++i) {
                ordered.push_back(size_t(mp[i].prim_index));
                bounds = bunion(bounds, info[mp[i].prim_index].bounds);
            }
            nodes[node].first_prim = first;
            nodes[node].n_prims = n;
            nodes[node].bounds = bounds;
            ++n_leaf;
            return node;
        }
        const uint32_t mask = 1u << bit_index;
        if ((mp[0].code & mask) == (mp[n - 1].code & mask)) return emit_lbvh(mp, n, bit_index - 1);
        int search_start = 0, search_end = n - 1;
        while (search_start + 1 != search_end) {
            const int mid = (search_start + search_end) / 2;
            if ((mp[search_start].code & mask) == (mp[mid].code & mask))
                search_start = mid;
            else
                search_end = mid;
        }
        const int split = search_end;
        nodes.emplace_back();
        const int node = int(nodes.size()) - 1;
        const int c0 = emit_lbvh(mp, split, bit_index - 1);
        const int c1 = emit_lbvh(mp + split, n - split, bit_index - 1);
        nodes[node].children[0] = c0;
        nodes[node].children[1] = c1;
        nodes[node].bounds = bunion(nodes[c0].bounds, nodes[c1].bounds);
        nodes[node].split_axis = bit_index % 3;
        nodes[node].n_prims = 0;
        ++n_interior;
        return node;
    }
    int build_upper_sah(std::vector<int> &roots, int start, int end) {
        const int n_nodes = end - start;
        if (n_nodes == 1) return roots[start];
        nodes.emplace_back();
        const int node = int(nodes.size()) - 1;
        Bounds3 bounds, cb;
        for (int i = start; i < end; ++i) bounds = bunion(bounds, nodes[roots[i]].bounds);
        for (int i = start; i < end; ++i) cb = bunion(cb, (nodes[roots[i]].bounds.pmin + nodes[roots[i]].bounds.pmax) * 0.5f);
        const int dim = cb.maximum_extent();
        constexpr int nBuckets = 12;
        struct Bucket {
            int count = 0;
            Bounds3 bounds;
        } buckets[nBuckets];
        const float lo = cb.pmin[dim], hi = cb.pmax[dim];
        auto bucket_of = [&](int r) {
            const float centroid = (nodes[r].bounds.pmin[dim] + nodes[r].bounds.pmax[dim]) * 0.5f;
            int b = int(nBuckets * ((centroid - lo) / (hi - lo)));
            if (b == nBuckets) b = nBuckets - 1;
            return b;
        };
        for (int i = start; i < end; ++i) {
            const int b = bucket_of(roots[i]);
            buckets[b].count++;
            buckets[b].bounds = bunion(buckets[b].bounds, nodes[roots[i]].bounds);
        }
        float cost[nBuckets - 1];
        for (int i = 0; i < nBuckets - 1; ++i) {
            Bounds3 b0, b1;
            int c0 = 0, c1 = 0;
            for (int j = 0; j <= i; ++j) {
                b0 = bunion(b0, buckets[j].bounds);
                c0 += buckets[j].count;
            }
            for (int j = i + 1; j < nBuckets; ++j) {
                b1 = bunion(b1, buckets[j].bounds);
                c1 += buckets[j].count;
            }
            cost[i] = .125f + (c0 * b0.surface_area() + c1 * b1.surface_area()) / bounds.surface_area();
        }
        float min_cost = cost[0];
        int min_bucket = 0;
        for (int i = 1; i < nBuckets - 1; ++i)
            if (cost[i] < min_cost) {
                min_cost = cost[i];
                min_bucket = i;
            }
        int *pmid = std::partition(&roots[start], &roots[end - 1] + 1, [&](int r) { return bucket_of(r) <= min_bucket; });
        const int mid = int(pmid - &roots[0]);
        const int c0 = build_upper_sah(roots, start, mid);
        const int c1 = build_upper_sah(roots, mid, end);
        nodes[node].children[0] = c0;
        nodes[node].children[1] = c1;
        nodes[node].bounds = bunion(nodes[c0].bounds, nodes[c1].bounds);
        nodes[node].split_axis = dim;
        nodes[node].n_prims = 0;
        ++n_interior;
        return node;
    }
    int build_hlbvh() {
        Bounds3 bounds;
        for (const PrimInfo &pi : info) bounds = bunion(bounds, pi.centroid);
        std::vector<MortonPrim> mp(info.size());
        for (size_t i = 0; i < info.size(); ++i) {
            mp[i].prim_index = int(info[i].number);
            const V3 off = bounds.offset(info[i].centroid);
            mp[i].code = encode_morton3(off * float(1 << 10));
        }
        radix_sort(&mp);
        std::vector<int> roots;
        for (int start = 0, end = 1; end <= int(mp.size()); ++end) {
            const uint32_t mask = 0x3ffc0000u;
            if (end == int(mp.size()) || ((mp[start].code & mask) != (mp[end].code & mask))) {
                roots.push_back(emit_lbvh(&mp[start], end - start, 29 - 12));
                start = end;
            }
        }
        return build_upper_sah(roots, 0, int(roots.size()));
    }

    int flatten(int node, std::vector<iile_bvh_node> &out) {
        const int my = int(out.size());
        out.emplace_back();
        const BuildNode bn = nodes[node];
        iile_bvh_node ln;
        std::memset(&ln, 0, sizeof(ln));
        for (int i = 0; i < 3; ++i) {
            ln.bmin[i] = bn.bounds.pmin[i];
            ln.bmax[i] = bn.bounds.pmax[i];
        }
        if (bn.n_prims > 0) {
            ln.offset = bn.first_prim;
            ln.nprims = uint16_t(bn.n_prims);
            out[my] = ln;
        } else {
            ln.axis = uint8_t(bn.split_axis);
            ln.nprims = 0;
            flatten(bn.children[0], out);
            ln.offset = flatten(bn.children[1], out);
            out[my] = ln;
        }
        return my;
    }
};

}  // namespace

void build_bvh(HostScene *scene) {
    Builder b;
    b.max_prims_in_node = std::min(255, scene->max_node_prims);
    // CreateBVHAccelerator, bvh.cpp:740-760
    b.split_method = scene->accel_split == "hlbvh" ? kHLBVH : (scene->accel_split == "middle" ? kMiddle : (scene->accel_split == "equal" ? kEqualCounts : kSAH));
    const size_t n = scene->prims.size();
    b.info.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const Bounds3 &wb = scene->prims[i].world_bound;
        b.info[i].number = i;
        b.info[i].bounds = wb;
        b.info[i].centroid = .5f * wb.pmin + .5f * wb.pmax;  // bvh.cpp:53-56
    }
    scene->nodes.clear();
    if (n == 0) return;
    if (b.split_method == kHLBVH && scene->bvh_hook) {
        // the plugged-in builder (the device build of libiile_gpu) hands back the flattened tree and the leaf order
        std::vector<float> bounds6(6 * n);
        for (size_t i = 0; i < n; ++i)
            for (int c = 0; c < 3; ++c) {
                bounds6[6 * i + c] = b.info[i].bounds.pmin[c];
                bounds6[6 * i + 3 + c] = b.info[i].bounds.pmax[c];
            }
        std::vector<int32_t> order(n);
        scene->nodes.resize(2 * n);
        int32_t n_nodes = 0;
        const int rc = scene->bvh_hook(int32_t(n), bounds6.data(), scene->max_node_prims, scene->nodes.data(), &n_nodes, order.data(), nullptr);
        if (rc != 0 || n_nodes <= 0 || size_t(n_nodes) > 2 * n) throw std::runtime_error("the bvh_build hook failed (code " + std::to_string(rc) + ")");
        scene->nodes.resize(size_t(n_nodes));
        b.ordered.assign(order.begin(), order.end());
        std::vector<char> seen(n, 0);
        for (size_t v : b.ordered) {
            if (v >= n || seen[v]) throw std::runtime_error("the bvh_build hook returned an order that is no permutation");
            seen[v] = 1;
        }
        for (const iile_bvh_node &nd : scene->nodes) (nd.nprims ? b.n_leaf : b.n_interior)++;
    } else {
        b.nodes.reserve(2 * n);
        b.ordered.reserve(n);
        int root = b.split_method == kHLBVH ? b.build_hlbvh() : b.build(0, int(n));
        scene->nodes.reserve(b.nodes.size());
        b.flatten(root, scene->nodes);
    }
    scene->n_interior = b.n_interior;
    scene->n_leaf = b.n_leaf;

    scene->o_flags.resize(n);
    scene->o_material.resize(n);
    scene->o_light.resize(n);
    scene->o_shape.resize(n);
    scene->o_tri_p.assign(n * 9, 0.f);
    scene->o_tri_n.assign(n * 9, 0.f);
    scene->o_tri_uv.assign(n * 6, 0.f);
    scene->o_alpha.clear();
    for (const HostPrim &p : scene->prims)
        if (p.flags & IILE_PRIM_HAS_ALPHA) {
            scene->o_alpha.assign(n * 2, IILE_ALPHA_NONE);
            break;
        }
    for (size_t i = 0; i < n; ++i) {
        const HostPrim &p = scene->prims[b.ordered[i]];
        scene->o_flags[i] = p.flags;
        scene->o_material[i] = p.material;
        scene->o_light[i] = p.light;
        scene->o_shape[i] = p.shape;
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < 3; ++c) {
                scene->o_tri_p[9 * i + 3 * k + c] = p.p[k][c];
                scene->o_tri_n[9 * i + 3 * k + c] = p.n[k][c];
            }
        for (int c = 0; c < 6; ++c) scene->o_tri_uv[6 * i + c] = p.uv[c];
        if (!scene->o_alpha.empty()) {
            scene->o_alpha[2 * i] = p.alpha;
            scene->o_alpha[2 * i + 1] = p.shadow_alpha;
        }
    }
}

}  // namespace iile
