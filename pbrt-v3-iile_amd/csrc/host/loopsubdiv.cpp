// loopsubdiv.cpp — Loop subdivision tessellator for `Shape "loopsubdiv"`.
//
// Host-side scene preparation. Restates
// /root/reference/src/shapes/loopsubdiv.cpp:149-400 with index-based
// connectivity instead of pointer graphs; the floating-point evaluation order
// (one-ring walk order, weight formulas, double-precision cross product) is
// kept so that the emitted limit positions and normals are bit-identical.
// The reference orders edge endpoints by pointer value; that only decides the
// order of commutative additions, so index order is equivalent.
#include <map>
#include <utility>

#include "host_scene.h"

namespace iile {
namespace {

struct SVert {
    V3 p;
    int start_face = -1;
    int child = -1;
    bool regular = false, boundary = false;
};
struct SFace {
    int v[3] = {-1, -1, -1};
    int f[3] = {-1, -1, -1};
    int children[4] = {-1, -1, -1, -1};
};

inline int nxt(int i) { return (i + 1) % 3; }
inline int prv(int i) { return (i + 2) % 3; }

struct Mesh {
    std::vector<SVert> V;
    std::vector<SFace> F;

    int vnum(int face, int vert) const {
        for (int i = 0; i < 3; ++i)
            if (F[face].v[i] == vert) return i;
        return -1;
    }
    int next_face(int face, int vert) const { return F[face].f[vnum(face, vert)]; }
    int prev_face(int face, int vert) const { return F[face].f[prv(vnum(face, vert))]; }
    int next_vert(int face, int vert) const { return F[face].v[nxt(vnum(face, vert))]; }
    int prev_vert(int face, int vert) const { return F[face].v[prv(vnum(face, vert))]; }
    int other_vert(int face, int v0, int v1) const {
        for (int i = 0; i < 3; ++i)
            if (F[face].v[i] != v0 && F[face].v[i] != v1) return F[face].v[i];
        return -1;
    }
    // loopsubdiv.cpp:120-136
    int valence(int vert) const {
        int f = V[vert].start_face;
        if (!V[vert].boundary) {
            int nf = 1;
            while ((f = next_face(f, vert)) != V[vert].start_face) ++nf;
            return nf;
        }
        int nf = 1;
        while ((f = next_face(f, vert)) != -1) ++nf;
        f = V[vert].start_face;
        while ((f = prev_face(f, vert)) != -1) ++nf;
        return nf + 1;
    }
    // loopsubdiv.cpp:436-455
    void one_ring(int vert, V3 *p) const {
        if (!V[vert].boundary) {
            int face = V[vert].start_face;
            do {
                *p++ = V[next_vert(face, vert)].p;
                face = next_face(face, vert);
            } while (face != V[vert].start_face);
        } else {
            int face = V[vert].start_face, f2;
            while ((f2 = next_face(face, vert)) != -1) face = f2;
            *p++ = V[next_vert(face, vert)].p;
            do {
                *p++ = V[prev_vert(face, vert)].p;
                face = prev_face(face, vert);
            } while (face != -1);
        }
    }
    // loopsubdiv.cpp:426-434
    V3 weight_one_ring(int vert, float beta) const {
        int val = valence(vert);
        std::vector<V3> ring(val);
        one_ring(vert, ring.data());
        V3 p = (1 - val * beta) * V[vert].p;
        for (int i = 0; i < val; ++i) p = p + beta * ring[i];
        return p;
    }
    // loopsubdiv.cpp:457-466
    V3 weight_boundary(int vert, float beta) const {
        int val = valence(vert);
        std::vector<V3> ring(val);
        one_ring(vert, ring.data());
        V3 p = (1 - 2 * beta) * V[vert].p;
        p = p + beta * ring[0];
        p = p + beta * ring[val - 1];
        return p;
    }
};

// loopsubdiv.cpp:138-147
inline float beta_of(int valence) { return valence == 3 ? 3.f / 16.f : 3.f / (8.f * valence); }
inline float loop_gamma(int valence) { return 1.f / (valence + 3.f / (8.f * beta_of(valence))); }

typedef std::pair<int, int> EdgeKey;
inline EdgeKey edge_key(int a, int b) { return EdgeKey(std::min(a, b), std::max(a, b)); }

}  // namespace

void loop_subdivide(int n_levels, const std::vector<int> &indices, const std::vector<V3> &P,
                    std::vector<int> *out_indices, std::vector<V3> *out_P, std::vector<V3> *out_N) {
    Mesh M;
    const int n_verts = int(P.size());
    const int n_faces = int(indices.size() / 3);
    M.V.resize(n_verts);
    M.F.resize(n_faces);
    std::vector<int> v(n_verts), f(n_faces);
    for (int i = 0; i < n_verts; ++i) {
        M.V[i].p = P[i];
        v[i] = i;
    }
    // face -> vertex links; a vertex's startFace is the LAST face naming it (cpp:170-178)
    for (int i = 0; i < n_faces; ++i) {
        f[i] = i;
        for (int j = 0; j < 3; ++j) {
            int vi = indices[3 * i + j];
            M.F[i].v[j] = vi;
            M.V[vi].start_face = i;
        }
    }
    // neighbour links through an edge set (cpp:180-200)
    {
        std::map<EdgeKey, std::pair<int, int>> edges;  // key -> (face, edgeNum)
        for (int i = 0; i < n_faces; ++i) {
            for (int e = 0; e < 3; ++e) {
                EdgeKey k = edge_key(M.F[i].v[e], M.F[i].v[nxt(e)]);
                auto it = edges.find(k);
                if (it == edges.end())
                    edges[k] = std::make_pair(i, e);
                else {
                    M.F[it->second.first].f[it->second.second] = i;
                    M.F[i].f[e] = it->second.first;
                    edges.erase(it);
                }
            }
        }
    }
    // boundary / regular classification (cpp:202-216)
    for (int i = 0; i < n_verts; ++i) {
        int face = M.V[i].start_face;
        do {
            face = M.next_face(face, i);
        } while (face != -1 && face != M.V[i].start_face);
        M.V[i].boundary = (face == -1);
        if (!M.V[i].boundary && M.valence(i) == 6)
            M.V[i].regular = true;
        else if (M.V[i].boundary && M.valence(i) == 4)
            M.V[i].regular = true;
        else
            M.V[i].regular = false;
    }

    for (int level = 0; level < n_levels; ++level) {
        std::vector<int> new_faces, new_verts;
        // children allocation (cpp:227-239)
        for (int vert : v) {
            SVert c;
            c.regular = M.V[vert].regular;
            c.boundary = M.V[vert].boundary;
            M.V.push_back(c);
            M.V[vert].child = int(M.V.size()) - 1;
            new_verts.push_back(M.V[vert].child);
        }
        for (int face : f)
            for (int k = 0; k < 4; ++k) {
                M.F.push_back(SFace());
                M.F[face].children[k] = int(M.F.size()) - 1;
                new_faces.push_back(M.F[face].children[k]);
            }
        // even vertices (cpp:243-257)
        for (int vert : v) {
            V3 np;
            if (!M.V[vert].boundary) {
                if (M.V[vert].regular)
                    np = M.weight_one_ring(vert, 1.f / 16.f);
                else
                    np = M.weight_one_ring(vert, beta_of(M.valence(vert)));
            } else
                np = M.weight_boundary(vert, 1.f / 8.f);
            M.V[M.V[vert].child].p = np;
        }
        // odd (edge) vertices (cpp:259-292)
        std::map<EdgeKey, int> edge_verts;
        for (int face : f) {
            for (int k = 0; k < 3; ++k) {
                int a = M.F[face].v[k], b = M.F[face].v[nxt(k)];
                EdgeKey key = edge_key(a, b);
                if (edge_verts.count(key)) continue;
                SVert nv;
                nv.regular = true;
                nv.boundary = (M.F[face].f[k] == -1);
                nv.start_face = M.F[face].children[3];
                int e0 = key.first, e1 = key.second;
                if (nv.boundary) {
                    nv.p = 0.5f * M.V[e0].p;
                    nv.p = nv.p + 0.5f * M.V[e1].p;
                } else {
                    nv.p = (3.f / 8.f) * M.V[e0].p;
                    nv.p = nv.p + (3.f / 8.f) * M.V[e1].p;
                    nv.p = nv.p + (1.f / 8.f) * M.V[M.other_vert(face, e0, e1)].p;
                    nv.p = nv.p + (1.f / 8.f) * M.V[M.other_vert(M.F[face].f[k], e0, e1)].p;
                }
                M.V.push_back(nv);
                int id = int(M.V.size()) - 1;
                new_verts.push_back(id);
                edge_verts[key] = id;
            }
        }
        // topology of the refined mesh (cpp:296-336)
        for (int vert : v) {
            int vn = M.vnum(M.V[vert].start_face, vert);
            M.V[M.V[vert].child].start_face = M.F[M.V[vert].start_face].children[vn];
        }
        for (int face : f) {
            for (int j = 0; j < 3; ++j) {
                const int *ch = M.F[face].children;
                M.F[ch[3]].f[j] = ch[nxt(j)];
                M.F[ch[j]].f[nxt(j)] = ch[3];
                int f2 = M.F[face].f[j];
                M.F[ch[j]].f[j] = (f2 != -1) ? M.F[f2].children[M.vnum(f2, M.F[face].v[j])] : -1;
                f2 = M.F[face].f[prv(j)];
                M.F[ch[j]].f[prv(j)] = (f2 != -1) ? M.F[f2].children[M.vnum(f2, M.F[face].v[j])] : -1;
            }
        }
        for (int face : f) {
            for (int j = 0; j < 3; ++j) {
                const int *ch = M.F[face].children;
                M.F[ch[j]].v[j] = M.V[M.F[face].v[j]].child;
                int ev = edge_verts[edge_key(M.F[face].v[j], M.F[face].v[nxt(j)])];
                M.F[ch[j]].v[nxt(j)] = ev;
                M.F[ch[nxt(j)]].v[j] = ev;
                M.F[ch[3]].v[j] = ev;
            }
        }
        f.swap(new_faces);
        v.swap(new_verts);
    }

    // limit surface positions (cpp:343-351)
    std::vector<V3> plimit(v.size());
    for (size_t i = 0; i < v.size(); ++i) {
        if (M.V[v[i]].boundary)
            plimit[i] = M.weight_boundary(v[i], 1.f / 5.f);
        else
            plimit[i] = M.weight_one_ring(v[i], loop_gamma(M.valence(v[i])));
    }
    for (size_t i = 0; i < v.size(); ++i) M.V[v[i]].p = plimit[i];

    // limit surface tangents -> normals (cpp:353-390)
    out_N->clear();
    out_N->reserve(v.size());
    std::vector<V3> ring(16);
    for (int vert : v) {
        V3 S(0, 0, 0), T(0, 0, 0);
        int val = M.valence(vert);
        if (val > int(ring.size())) ring.resize(val);
        M.one_ring(vert, ring.data());
        const V3 vp = M.V[vert].p;
        if (!M.V[vert].boundary) {
            for (int j = 0; j < val; ++j) {
                S = S + std::cos(2 * kPi * j / val) * ring[j];
                T = T + std::sin(2 * kPi * j / val) * ring[j];
            }
        } else {
            S = ring[val - 1] - ring[0];
            if (val == 2)
                T = ring[0] + ring[1] - 2.f * vp;
            else if (val == 3)
                T = ring[1] - vp;
            else if (val == 4)
                T = -1.f * ring[0] + 2.f * ring[1] + 2.f * ring[2] + -1.f * ring[3] + -2.f * vp;
            else {
                float theta = kPi / float(val - 1);
                T = std::sin(theta) * (ring[0] + ring[val - 1]);
                for (int k = 1; k < val - 1; ++k) {
                    float wt = (2 * std::cos(theta) - 2) * std::sin(k * theta);
                    T = T + wt * ring[k];
                }
                T = -T;
            }
        }
        out_N->push_back(cross(S, T));
    }

    // triangle list (cpp:392-409)
    std::map<int, int> used;
    for (size_t i = 0; i < v.size(); ++i) used[v[i]] = int(i);
    out_indices->clear();
    out_indices->reserve(3 * f.size());
    for (int face : f)
        for (int j = 0; j < 3; ++j) out_indices->push_back(used[M.F[face].v[j]]);
    *out_P = plimit;
}

}  // namespace iile
