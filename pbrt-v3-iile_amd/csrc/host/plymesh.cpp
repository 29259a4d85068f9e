// plymesh.cpp — PLY reader for `Shape "plymesh"` (host side).
//
// What the reference takes from a PLY file (src/shapes/plymesh.cpp:149-300, through the rply
// library): vertex x/y/z, optional nx/ny/nz, optional texture coordinates named (u, v), (s, t),
// (texture_u, texture_v) or (texture_s, texture_t); faces from the list property
// "vertex_indices" with three or four entries, a quad (a, b, c, d) becoming the triangles
// (a, b, c) and (d, a, c) (plymesh.cpp:121-140); other face sizes are ignored. All values are
// cast to float / int. Formats: ascii, binary_little_endian, binary_big_endian.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "host_scene.h"

namespace iile {
namespace {

enum PlyType { T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64, T_BAD };
PlyType ply_type(const std::string &s) {
    if (s == "char" || s == "int8") return T_I8;
    if (s == "uchar" || s == "uint8") return T_U8;
    if (s == "short" || s == "int16") return T_I16;
    if (s == "ushort" || s == "uint16") return T_U16;
    if (s == "int" || s == "int32") return T_I32;
    if (s == "uint" || s == "uint32") return T_U32;
    if (s == "float" || s == "float32") return T_F32;
    if (s == "double" || s == "float64") return T_F64;
    return T_BAD;
}
int type_size(PlyType t) {
    static const int sz[] = {1, 1, 2, 2, 4, 4, 4, 8, 0};
    return sz[t];
}
struct Prop {
    std::string name;
    bool is_list = false;
    PlyType count_type = T_BAD, type = T_BAD;
};
struct Element {
    std::string name;
    long count = 0;
    std::vector<Prop> props;
};

struct Reader {
    const std::string &buf;
    size_t pos;
    int format;  // 0 ascii, 1 little endian, 2 big endian
    bool ok = true;
    double read(PlyType t) {
        if (format == 0) {
            while (pos < buf.size() && (buf[pos] == ' ' || buf[pos] == '\n' || buf[pos] == '\r' || buf[pos] == '\t')) ++pos;
            size_t e = pos;
            while (e < buf.size() && !(buf[e] == ' ' || buf[e] == '\n' || buf[e] == '\r' || buf[e] == '\t')) ++e;
            if (e == pos) {
                ok = false;
                return 0;
            }
            const std::string tok = buf.substr(pos, e - pos);
            pos = e;
            char *end = nullptr;
            const double v = std::strtod(tok.c_str(), &end);
            if (end == tok.c_str()) ok = false;
            return v;
        }
        const int n = type_size(t);
        if (pos + size_t(n) > buf.size()) {
            ok = false;
            return 0;
        }
        unsigned char b[8];
        for (int i = 0; i < n; ++i) b[i] = static_cast<unsigned char>(buf[pos + (format == 1 ? i : n - 1 - i)]);
        pos += size_t(n);
        switch (t) {
        case T_I8: { int8_t v; std::memcpy(&v, b, 1); return v; }
        case T_U8: return b[0];
        case T_I16: { int16_t v; std::memcpy(&v, b, 2); return v; }
        case T_U16: { uint16_t v; std::memcpy(&v, b, 2); return v; }
        case T_I32: { int32_t v; std::memcpy(&v, b, 4); return v; }
        case T_U32: { uint32_t v; std::memcpy(&v, b, 4); return v; }
        case T_F32: { float v; std::memcpy(&v, b, 4); return v; }
        case T_F64: { double v; std::memcpy(&v, b, 8); return v; }
        default: ok = false; return 0;
        }
    }
};

}  // namespace

bool load_ply(const std::string &path, std::vector<V3> *P, std::vector<V3> *N, std::vector<float> *uv,
              std::vector<int> *indices, std::string *err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        *err = "Couldn't open PLY file \"" + path + "\"";
        return false;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string buf = ss.str();
    // ---- header
    size_t pos = 0;
    auto line = [&](std::string *out) -> bool {
        if (pos >= buf.size()) return false;
        size_t e = buf.find('\n', pos);
        if (e == std::string::npos) e = buf.size();
        *out = buf.substr(pos, e - pos);
        if (!out->empty() && out->back() == '\r') out->pop_back();
        pos = e + 1;
        return true;
    };
    std::string l;
    if (!line(&l) || l != "ply") {
        *err = "Unable to read the header of PLY file \"" + path + "\"";
        return false;
    }
    int format = -1;
    std::vector<Element> elems;
    bool ended = false;
    while (line(&l)) {
        std::istringstream ls(l);
        std::string kw;
        ls >> kw;
        if (kw == "format") {
            std::string fmt;
            ls >> fmt;
            format = fmt == "ascii" ? 0 : (fmt == "binary_little_endian" ? 1 : (fmt == "binary_big_endian" ? 2 : -1));
        } else if (kw == "element") {
            Element e;
            ls >> e.name >> e.count;
            elems.push_back(e);
        } else if (kw == "property") {
            if (elems.empty()) break;
            Prop p;
            std::string t;
            ls >> t;
            if (t == "list") {
                std::string ct, vt;
                ls >> ct >> vt >> p.name;
                p.is_list = true;
                p.count_type = ply_type(ct);
                p.type = ply_type(vt);
                if (p.count_type == T_BAD) p.type = T_BAD;
            } else {
                p.type = ply_type(t);
                ls >> p.name;
            }
            if (p.type == T_BAD) {
                *err = path + ": unsupported PLY property type in \"" + l + "\"";
                return false;
            }
            elems.back().props.push_back(p);
        } else if (kw == "end_header") {
            ended = true;
            break;
        }  // comment, obj_info: skipped
    }
    if (!ended || format < 0) {
        *err = "Unable to read the header of PLY file \"" + path + "\"";
        return false;
    }
    long n_vert = 0, n_face = 0;
    int seen_vert = 0, seen_face = 0;
    const long remaining = long(buf.size()) - long(pos);
    for (const Element &e : elems) {
        // an element of n records takes at least n bytes of body (ascii: one character per value): a header that
        // promises more than the file holds, or a negative count, is rejected before anything is sized from it
        if (e.count < 0 || (!e.props.empty() && e.count > remaining)) {
            *err = path + ": PLY element \"" + e.name + "\" has an impossible count";
            return false;
        }
        if (e.name == "vertex") n_vert = e.count, ++seen_vert;
        if (e.name == "face") n_face = e.count, ++seen_face;
    }
    if (seen_vert > 1 || seen_face > 1) {
        *err = path + ": PLY file declares a vertex / face element twice";
        return false;
    }
    if (n_vert == 0 || n_face == 0) {
        *err = path + ": PLY file is invalid! No face/vertex elements found!";
        return false;
    }
    // ---- body
    Reader rd{buf, pos, format};
    P->assign(size_t(n_vert), V3(0, 0, 0));
    bool have_xyz[3] = {false, false, false}, have_n[3] = {false, false, false}, have_uv[2] = {false, false};
    std::vector<V3> nrm(size_t(n_vert), V3(0, 0, 0));
    std::vector<float> tex(2 * size_t(n_vert), 0.f);
    indices->clear();
    for (const Element &e : elems) {
        for (long i = 0; i < e.count; ++i) {
            for (const Prop &p : e.props) {
                if (!p.is_list) {
                    const float v = float(rd.read(p.type));
                    if (e.name != "vertex") continue;
                    V3 &pp = (*P)[size_t(i)];
                    V3 &nn = nrm[size_t(i)];
                    if (p.name == "x") pp.x = v, have_xyz[0] = true;
                    else if (p.name == "y") pp.y = v, have_xyz[1] = true;
                    else if (p.name == "z") pp.z = v, have_xyz[2] = true;
                    else if (p.name == "nx") nn.x = v, have_n[0] = true;
                    else if (p.name == "ny") nn.y = v, have_n[1] = true;
                    else if (p.name == "nz") nn.z = v, have_n[2] = true;
                    else if (p.name == "u" || p.name == "s" || p.name == "texture_u" || p.name == "texture_s")
                        tex[2 * size_t(i)] = v, have_uv[0] = true;
                    else if (p.name == "v" || p.name == "t" || p.name == "texture_v" || p.name == "texture_t")
                        tex[2 * size_t(i) + 1] = v, have_uv[1] = true;
                } else {
                    const long len = long(rd.read(p.count_type));
                    int face[4] = {0, 0, 0, 0};
                    for (long k = 0; k < len; ++k) {
                        const int v = int(rd.read(p.type));
                        if (k < 4) face[k] = v;
                    }
                    if (e.name != "face" || !(p.name == "vertex_indices" || p.name == "vertex_index")) continue;
                    if (len != 3 && len != 4) continue;  // "Ignoring face with %i vertices"
                    for (int k = 0; k < len; ++k)
                        if (face[k] < 0 || face[k] >= n_vert) {
                            *err = "plymesh: Vertex reference " + std::to_string(face[k]) + " is out of bounds!";
                            return false;
                        }
                    indices->push_back(face[0]);
                    indices->push_back(face[1]);
                    indices->push_back(face[2]);
                    if (len == 4) {  // plymesh.cpp:135-140
                        indices->push_back(face[3]);
                        indices->push_back(face[0]);
                        indices->push_back(face[2]);
                    }
                }
                if (!rd.ok) {
                    *err = path + ": unable to read the contents of PLY file";
                    return false;
                }
            }
        }
    }
    if (!(have_xyz[0] && have_xyz[1] && have_xyz[2])) {
        *err = path + ": Vertex coordinate property not found!";
        return false;
    }
    N->clear();
    uv->clear();
    if (have_n[0] && have_n[1] && have_n[2]) *N = nrm;
    if (have_uv[0] && have_uv[1]) *uv = tex;
    return true;
}

}  // namespace iile
