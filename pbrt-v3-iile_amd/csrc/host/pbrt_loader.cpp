// pbrt_loader.cpp — minimal .pbrt scene-description front end.
//
// The reference's parser/API layer (src/core/parser.cpp, api.cpp, paramset.cpp)
// is OUT OF SCOPE as a re-implementation; the GPU box
// only receives this repository, so the path needs its own loader for the
// directives killeroo-class scenes use. What IS in scope is bit-equal scene
// preparation: number parsing through strtol/strtof
// (parser.cpp:258-300), CTM post-multiplication (api.cpp:934-995),
// `Camera` storing Inverse(CTM) (api.cpp:1118-1123), one primitive per
// triangle with the area light attached per shape (api.cpp:1371-1430).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>

#include "host_scene.h"

namespace iile {
namespace {

struct Token {
    enum Kind { Word, String, LBracket, RBracket, End, Error } kind = End;
    std::string text;  // Error: the tokenizer's message ("premature EOF", "unterminated string")
};

class Lexer {
  public:
    bool open(const std::string &path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) return false;
        std::stringstream ss;
        ss << f.rdbuf();
        buf_ = ss.str();
        pos_ = 0;
        return true;
    }
    // parser.cpp:150-255: whitespace, '#' comments, quoted strings, brackets,
    // everything else runs to the next delimiter.
    Token next() {
        Token t;
        while (pos_ < buf_.size()) {
            char ch = buf_[pos_];
            if (ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r') {
                ++pos_;
            } else if (ch == '#') {
                while (pos_ < buf_.size() && buf_[pos_] != '\n' && buf_[pos_] != '\r') ++pos_;
            } else
                break;
        }
        if (pos_ >= buf_.size()) return t;
        char ch = buf_[pos_];
        if (ch == '"') {
            // scan to the closing quote (parser.cpp:196-232): a newline ends the string in error,
            // a backslash takes the next character with it (decodeEscaped, parser.cpp:67-93)
            t.kind = Token::String;
            ++pos_;
            for (;;) {
                if (pos_ >= buf_.size()) {
                    t.kind = Token::Error;
                    t.text = "premature EOF";
                    return t;
                }
                char c = buf_[pos_++];
                if (c == '"') break;
                if (c == '\n') {
                    t.kind = Token::Error;
                    t.text = "unterminated string";
                    return t;
                }
                if (c == '\\') {
                    if (pos_ >= buf_.size()) {
                        t.kind = Token::Error;
                        t.text = "premature EOF";
                        return t;
                    }
                    const char e = buf_[pos_++];
                    switch (e) {
                    case 'b': c = '\b'; break;
                    case 'f': c = '\f'; break;
                    case 'n': c = '\n'; break;
                    case 'r': c = '\r'; break;
                    case 't': c = '\t'; break;
                    case '\\': c = '\\'; break;
                    case '\'': c = '\''; break;
                    case '"': c = '"'; break;
                    default:
                        t.kind = Token::Error;
                        t.text = std::string("unexpected escaped character \"") + e + "\"";
                        return t;
                    }
                }
                t.text.push_back(c);
            }
        } else if (ch == '[') {
            t.kind = Token::LBracket;
            ++pos_;
        } else if (ch == ']') {
            t.kind = Token::RBracket;
            ++pos_;
        } else {
            size_t s = pos_;
            while (pos_ < buf_.size()) {
                char c = buf_[pos_];
                if (c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '"' || c == '[' || c == ']')
                    break;
                ++pos_;
            }
            t.kind = Token::Word;
            t.text = buf_.substr(s, pos_ - s);
        }
        return t;
    }

  private:
    std::string buf_;
    size_t pos_ = 0;
};

// parser.cpp:258-300: all-digit tokens go through strtol, everything else
// through strtof; the value travels as double and is cast at the use site.
bool parse_number(const std::string &s, double *out) {
    if (s.size() == 1) {
        if (!(s[0] >= '0' && s[0] <= '9')) return false;
        *out = s[0] - '0';
        return true;
    }
    bool is_int = true;
    for (char c : s)
        if (!(c >= '0' && c <= '9')) is_int = false;
    char *end = nullptr;
    double v;
    if (is_int)
        v = double(strtol(s.c_str(), &end, 10));
    else
        v = strtof(s.c_str(), &end);
    if (v == 0 && end == s.c_str()) return false;
    *out = v;
    return true;
}

struct Param {
    std::string type, name;
    std::vector<double> nums;
    std::vector<std::string> strs;
};
struct ParamSet {
    std::vector<Param> params;
    const Param *find(const std::string &name) const {
        for (const Param &p : params)
            if (p.name == name) return &p;
        return nullptr;
    }
    float one_float(const std::string &name, float def) const {
        const Param *p = find(name);
        return (p && p->type == "float" && p->nums.size() == 1) ? float(p->nums[0]) : def;
    }
    int one_int(const std::string &name, int def) const {
        const Param *p = find(name);
        return (p && p->type == "integer" && p->nums.size() == 1) ? int(p->nums[0]) : def;
    }
    bool one_bool(const std::string &name, bool def) const {
        const Param *p = find(name);
        if (!p || p->type != "bool" || p->strs.size() != 1) return def;
        return p->strs[0] == "true";
    }
    std::string one_string(const std::string &name, const std::string &def) const {
        const Param *p = find(name);
        return (p && p->type == "string" && p->strs.size() == 1) ? p->strs[0] : def;
    }
    bool rgb(const std::string &name, float out[3]) const {
        const Param *p = find(name);
        if (!p || (p->type != "color" && p->type != "rgb") || p->nums.size() != 3) return false;
        for (int i = 0; i < 3; ++i) out[i] = float(p->nums[i]);
        return true;
    }
};

// A named texture, folded to its value: only textures that are constant over the surface are
// supported ("constant", and "scale" / "mix" of such), so evaluating one at a hit is the same as
// evaluating it once here (textures/constant.h, scale.h, mix.h).
struct ConstTexture {
    bool is_float = true;
    float v[3] = {0, 0, 0};
    int image = -1;  // an "imagemap" spectrum texture: index into HostScene::textures (not constant: kept by reference)
    bool scaled = false;  // a "scale" of that image with a constant: v is the constant factor
};
struct GraphicsState {
    int material = -1;  // index into scene->materials, -1 = default matte
    bool has_area_light = false;
    ParamSet area_light_params;
    bool reverse_orientation = false;
    std::map<std::string, ConstTexture> textures;  // graphicsState.floatTextures / spectrumTextures, api.cpp:1190-1260
};

class Loader {
  public:
    Loader(HostScene *s, std::string *err) : scene_(s), err_(err) {}

    bool run(const std::string &path) {
        size_t slash = path.find_last_of('/');
        search_dir_ = (slash == std::string::npos) ? "." : path.substr(0, slash);
        return parse_file(path);
    }

  private:
    HostScene *scene_;
    std::string *err_;
    std::string search_dir_;
    Xform ctm_;
    std::map<std::string, Xform> named_cs_;
    GraphicsState gs_;
    std::vector<GraphicsState> gs_stack_;
    std::vector<Xform> ctm_stack_;
    bool in_world_ = false;
    int default_material_ = -1;

    bool fail(const std::string &m) {
        // a tokenizer error ends the token stream; whatever the parser then complains about,
        // the tokenizer's message is the cause (parser.cpp:199-212)
        if (err_) *err_ = lex_error_.empty() ? m : lex_error_;
        return false;
    }
    std::string lex_error_;
    std::map<std::string, int> named_materials_;  // MakeNamedMaterial, api.cpp:1286-1316

    bool parse_file(const std::string &path) {
        Lexer lex;
        if (!lex.open(path)) return fail("cannot open scene file " + path);
        Token pending;
        bool have_pending = false;
        auto next = [&]() {
            if (have_pending) {
                have_pending = false;
                return pending;
            }
            Token t = lex.next();
            if (t.kind == Token::Error) {
                if (lex_error_.empty()) lex_error_ = t.text;
                t.kind = Token::End;
            }
            return t;
        };
        auto unget = [&](const Token &t) {
            pending = t;
            have_pending = true;
        };
        auto numbers = [&](int n, float *out) -> bool {
            for (int i = 0; i < n; ++i) {
                Token t = next();
                double v;
                if (t.kind != Token::Word || !parse_number(t.text, &v)) return false;
                out[i] = float(v);
            }
            return true;
        };
        // parser.cpp:549-700 (parseParams): `"type name" value | [ values ]`
        auto params = [&](ParamSet *ps) -> bool {
            while (true) {
                Token t = next();
                if (t.kind != Token::String) {
                    unget(t);
                    return true;
                }
                Param p;
                std::istringstream decl(t.text);
                decl >> p.type >> p.name;
                if (p.type.empty() || p.name.empty()) return fail("bad parameter declaration \"" + t.text + "\"");
                auto add_value = [&](const Token &v) -> bool {
                    if (v.kind == Token::String) {
                        p.strs.push_back(v.text);
                        return true;
                    }
                    if (v.kind != Token::Word) return false;
                    if (p.type == "bool") {
                        p.strs.push_back(v.text);
                        return true;
                    }
                    double d;
                    if (!parse_number(v.text, &d)) return false;
                    p.nums.push_back(d);
                    return true;
                };
                Token v = next();
                if (v.kind == Token::LBracket) {
                    while (true) {
                        v = next();
                        if (v.kind == Token::RBracket) break;
                        if (v.kind == Token::End) return fail("premature EOF in parameter list");
                        if (!add_value(v)) return fail("bad value for parameter " + p.name);
                    }
                } else if (!add_value(v))
                    return fail("bad value for parameter " + p.name);
                if (p.type == "point3") p.type = "point";
                if (p.type == "vector3") p.type = "vector";
                if (p.type == "normal3") p.type = "normal";
                ps->params.push_back(std::move(p));
            }
        };

        while (true) {
            Token t = next();
            if (t.kind == Token::End) {
                if (!lex_error_.empty()) return fail(lex_error_);
                break;
            }
            if (t.kind != Token::Word) return fail("unexpected token \"" + t.text + "\"");
            const std::string &d = t.text;
            float f[16];
            if (d == "AttributeBegin") {
                gs_stack_.push_back(gs_);
                ctm_stack_.push_back(ctm_);
            } else if (d == "AttributeEnd") {
                if (gs_stack_.empty()) return fail("unmatched AttributeEnd");
                gs_ = gs_stack_.back();
                gs_stack_.pop_back();
                ctm_ = ctm_stack_.back();
                ctm_stack_.pop_back();
            } else if (d == "TransformBegin") {
                ctm_stack_.push_back(ctm_);
            } else if (d == "TransformEnd") {
                if (ctm_stack_.empty()) return fail("unmatched TransformEnd");
                ctm_ = ctm_stack_.back();
                ctm_stack_.pop_back();
            } else if (d == "Identity") {
                ctm_ = Xform();
            } else if (d == "Translate") {  // api.cpp:934-942
                if (!numbers(3, f)) return fail("Translate: expected 3 numbers");
                ctm_ = ctm_ * xf_translate(V3(f[0], f[1], f[2]));
            } else if (d == "Scale") {  // api.cpp:984-991
                if (!numbers(3, f)) return fail("Scale: expected 3 numbers");
                ctm_ = ctm_ * xf_scale(f[0], f[1], f[2]);
            } else if (d == "Rotate") {  // api.cpp:973-982
                if (!numbers(4, f)) return fail("Rotate: expected 4 numbers");
                ctm_ = ctm_ * xf_rotate(f[0], V3(f[1], f[2], f[3]));
            } else if (d == "LookAt") {  // api.cpp:993-1007
                if (!numbers(9, f)) return fail("LookAt: expected 9 numbers");
                Xform la;
                xf_lookat(V3(f[0], f[1], f[2]), V3(f[3], f[4], f[5]), V3(f[6], f[7], f[8]), &la);
                ctm_ = ctm_ * la;
            } else if (d == "Transform" || d == "ConcatTransform") {  // api.cpp:944-971
                Token b = next();
                if (b.kind != Token::LBracket) return fail(d + ": expected [");
                if (!numbers(16, f)) return fail(d + ": expected 16 numbers");
                b = next();
                if (b.kind != Token::RBracket) return fail(d + ": expected ]");
                Xform x(Mat4(f[0], f[4], f[8], f[12], f[1], f[5], f[9], f[13], f[2], f[6], f[10], f[14],
                             f[3], f[7], f[11], f[15]));
                ctm_ = (d == "Transform") ? x : ctm_ * x;
            } else if (d == "CoordinateSystem" || d == "CoordSysTransform") {
                Token n = next();
                if (n.kind != Token::String) return fail(d + ": expected name");
                if (d == "CoordinateSystem")
                    named_cs_[n.text] = ctm_;
                else if (named_cs_.count(n.text))
                    ctm_ = named_cs_[n.text];
            } else if (d == "ReverseOrientation") {
                gs_.reverse_orientation = !gs_.reverse_orientation;
            } else if (d == "WorldBegin") {  // api.cpp:1160-1168
                in_world_ = true;
                ctm_ = Xform();
                named_cs_["world"] = ctm_;
            } else if (d == "WorldEnd") {
                in_world_ = false;
            } else if (d == "Include") {
                Token n = next();
                if (n.kind != Token::String) return fail("Include: expected filename");
                std::string p = n.text;
                if (p.empty() || p[0] != '/') p = search_dir_ + "/" + p;
                if (!parse_file(p)) return false;
            } else if (d == "Texture") {  // pbrtTexture, api.cpp:1190-1260: "name" "float|spectrum|color" "class" params
                Token n1 = next(), n2 = next(), n3 = next();
                if (n1.kind != Token::String || n2.kind != Token::String || n3.kind != Token::String)
                    return fail("Texture: expected \"name\" \"type\" \"class\"");
                ParamSet ps;
                if (!params(&ps)) return false;
                if (!make_texture(n1.text, n2.text, n3.text, ps)) return false;
            } else if (d == "MakeNamedMaterial") {  // api.cpp:1286-1316
                Token n = next();
                if (n.kind != Token::String) return fail("MakeNamedMaterial: expected quoted name");
                ParamSet ps;
                if (!params(&ps)) return false;
                const std::string type = ps.one_string("type", "");
                if (type.empty()) return fail("MakeNamedMaterial: the \"string type\" parameter is required");
                const int idx = make_material(type, ps);
                if (idx < 0) return false;
                named_materials_[n.text] = idx;
            } else if (d == "NamedMaterial") {  // api.cpp:1318-1336
                Token n = next();
                if (n.kind != Token::String) return fail("NamedMaterial: expected quoted name");
                auto it = named_materials_.find(n.text);
                if (it == named_materials_.end()) return fail("NamedMaterial \"" + n.text + "\" unknown.");
                gs_.material = it->second;
            } else if (d == "Camera" || d == "Film" || d == "Sampler" || d == "Integrator" ||
                       d == "PixelFilter" || d == "Accelerator" || d == "Material" ||
                       d == "AreaLightSource" || d == "Shape" || d == "LightSource") {
                Token n = next();
                if (n.kind != Token::String) return fail(d + ": expected quoted name");
                ParamSet ps;
                if (!params(&ps)) return false;
                if (!directive(d, n.text, ps)) return false;
            } else
                return fail("unsupported directive \"" + d + "\"");
        }
        return true;
    }

    bool directive(const std::string &d, const std::string &name, const ParamSet &ps) {
        HostScene &s = *scene_;
        if (d == "Camera") {  // api.cpp:1118-1123, cameras/perspective.cpp:283-330
            if (name != "perspective") return fail("only the perspective camera is supported, got " + name);
            s.camera_name = name;
            s.camera_to_world = inverse(ctm_);
            named_cs_["camera"] = s.camera_to_world;
            s.shutter_open = ps.one_float("shutteropen", 0.f);
            s.shutter_close = ps.one_float("shutterclose", 1.f);
            if (s.shutter_close < s.shutter_open) std::swap(s.shutter_close, s.shutter_open);
            s.lens_radius = ps.one_float("lensradius", 0.f);
            s.focal_distance = ps.one_float("focaldistance", 1e6);
            s.frame_aspect = ps.one_float("frameaspectratio", -1.f);
            const Param *sw = ps.find("screenwindow");
            if (sw && sw->nums.size() == 4) {
                s.has_screen_window = true;
                for (int i = 0; i < 4; ++i) s.screen_window[i] = float(sw->nums[i]);
            }
            s.fov = ps.one_float("fov", 90.);
            float half = ps.one_float("halffov", -1.f);
            if (half > 0.f) s.fov = 2.f * half;
        } else if (d == "Film") {  // film.cpp:259-304
            if (name != "image") return fail("only Film \"image\" is supported");
            s.xres = ps.one_int("xresolution", 1280);
            s.yres = ps.one_int("yresolution", 720);
            s.film_filename = ps.one_string("filename", "pbrt.exr");
            const Param *cw = ps.find("cropwindow");
            if (cw && cw->nums.size() == 4) {
                float c[4];
                for (int i = 0; i < 4; ++i) c[i] = float(cw->nums[i]);
                s.crop[0] = clampT(std::min(c[0], c[1]), 0.f, 1.f);
                s.crop[1] = clampT(std::max(c[0], c[1]), 0.f, 1.f);
                s.crop[2] = clampT(std::min(c[2], c[3]), 0.f, 1.f);
                s.crop[3] = clampT(std::max(c[2], c[3]), 0.f, 1.f);
            }
            s.film_scale = ps.one_float("scale", 1.);
            s.film_diagonal = ps.one_float("diagonal", 35.);
            s.max_sample_luminance =
                ps.one_float("maxsampleluminance", std::numeric_limits<float>::infinity());
        } else if (d == "Sampler") {  // samplers/halton.cpp:129-135, samplers/sobol.cpp:65-71
            if (name != "halton" && name != "sobol") return fail("only Sampler \"halton\" and \"sobol\" are supported, got " + name);
            s.sampler_name = name;
            s.spp = ps.one_int("pixelsamples", 16);
            s.sample_at_pixel_center = name == "halton" && ps.one_bool("samplepixelcenter", false);
        } else if (d == "PixelFilter") {  // MakeFilter, api.cpp:855-874; Create*Filter in src/filters/*.cpp
            s.filter_name = name;
            if (name == "box") {
                s.filter_rx = ps.one_float("xwidth", 0.5f);
                s.filter_ry = ps.one_float("ywidth", 0.5f);
            } else if (name == "gaussian") {
                s.filter_rx = ps.one_float("xwidth", 2.f);
                s.filter_ry = ps.one_float("ywidth", 2.f);
                s.filter_p0 = ps.one_float("alpha", 2.f);
            } else if (name == "mitchell") {
                s.filter_rx = ps.one_float("xwidth", 2.f);
                s.filter_ry = ps.one_float("ywidth", 2.f);
                s.filter_p0 = ps.one_float("B", 1.f / 3.f);
                s.filter_p1 = ps.one_float("C", 1.f / 3.f);
            } else if (name == "sinc") {
                s.filter_rx = ps.one_float("xwidth", 4.f);
                s.filter_ry = ps.one_float("ywidth", 4.f);
                s.filter_p0 = ps.one_float("tau", 3.f);
            } else if (name == "triangle") {
                s.filter_rx = ps.one_float("xwidth", 2.f);
                s.filter_ry = ps.one_float("ywidth", 2.f);
            } else
                return fail("Filter \"" + name + "\" unknown.");
        } else if (d == "Integrator") {  // integrators/path.cpp:214-231
            // "iispt" (CreateIISPTIntegrator, src/integrators/iispt.cpp:790-820) reads the same parameters; which of the two renders
            // the frame is the host's choice (iile_host_scene_info::integrator)
            if (name != "path" && name != "iispt") return fail("only Integrator \"path\" and \"iispt\" are supported, got " + name);
            s.integrator_iispt = name == "iispt";
            s.max_depth = ps.one_int("maxdepth", 5);
            s.rr_threshold = ps.one_float("rrthreshold", 1.);
            s.light_strategy = ps.one_string("lightsamplestrategy", "spatial");
            if (s.light_strategy != "spatial" && s.light_strategy != "uniform" && s.light_strategy != "power") {
                // CreateLightSampleDistribution, lightdistrib.cpp:58-63
                std::fprintf(stderr, "Error: Light sample distribution type \"%s\" unknown. Using \"spatial\".\n", s.light_strategy.c_str());
                s.light_strategy = "spatial";
            }
            if (const Param *pb = ps.find("pixelbounds")) {  // path.cpp:216-229
                // (CreateIISPTIntegrator reads it too, iispt.cpp:797-811, but nothing of IISPTIntegrator::Render looks at the member: the
                //  runners get film->GetSampleBounds(), iispt.cpp:395-409 — finalize leaves the sample bounds in place for "iispt")
                if (pb->type != "integer" || pb->nums.size() != 4)
                    std::fprintf(stderr, "Error: Expected four values for \"pixelbounds\" parameter. Got %d.\n", int(pb->nums.size()));
                else {
                    s.has_pixel_bounds = true;
                    for (int i = 0; i < 4; ++i) s.pixel_bounds_given[i] = int(pb->nums[size_t(i)]);
                }
            }
        } else if (d == "Accelerator") {  // accelerators/bvh.cpp:740-760
            if (name != "bvh") return fail("only Accelerator \"bvh\" is supported");
            s.accel_split = ps.one_string("splitmethod", "sah");
            if (s.accel_split != "sah" && s.accel_split != "hlbvh" && s.accel_split != "middle" && s.accel_split != "equal") {
                std::fprintf(stderr, "Warning: BVH split method \"%s\" unknown.  Using \"sah\".\n", s.accel_split.c_str());  // bvh.cpp:753-756
                s.accel_split = "sah";
            }
            s.max_node_prims = ps.one_int("maxnodeprims", 4);
        } else if (d == "Material") {
            int idx = make_material(name, ps);
            if (idx < 0) return false;
            gs_.material = idx;
        } else if (d == "AreaLightSource") {  // api.cpp:1360-1369
            if (name != "area" && name != "diffuse") return fail("unknown area light " + name);
            gs_.has_area_light = true;
            gs_.area_light_params = ps;
        } else if (d == "LightSource") {  // api.cpp:1344-1358 (pbrtLightSource), MakeLight api.cpp:770-806
            if (name != "point" && name != "spot" && name != "distant" && name != "infinite" && name != "exinfinite")
                return fail("LightSource \"" + name + "\" is not supported (point, spot, distant, infinite; area lights)");
            float I[3] = {1, 1, 1}, sc[3] = {1, 1, 1};
            ps.rgb((name == "distant" || name == "infinite" || name == "exinfinite") ? "L" : "I", I);
            ps.rgb("scale", sc);
            auto point_param = [&](const char *pname, V3 def, V3 *out) -> bool {
                *out = def;
                if (const Param *pp = ps.find(pname)) {
                    if (pp->type != "point" || pp->nums.size() != 3) return false;
                    *out = V3(float(pp->nums[0]), float(pp->nums[1]), float(pp->nums[2]));
                }
                return true;
            };
            V3 from, to;
            if (!point_param("from", V3(0, 0, 0), &from) || !point_param("to", V3(0, 0, 1), &to))
                return fail(name + " light: bad \"from\" / \"to\"");
            iile_light lt;
            std::memset(&lt, 0, sizeof(lt));
            for (int i = 0; i < 3; ++i) lt.lemit[i] = I[i] * sc[i];
            lt.sphere = -1;
            if (name == "point") {  // CreatePointLight, lights/point.cpp:80-88
                const Xform l2w = xf_translate(from) * ctm_;
                const V3 pl = l2w.point(V3(0, 0, 0));
                lt.type = IILE_LIGHT_POINT;
                lt.pos[0] = pl.x;
                lt.pos[1] = pl.y;
                lt.pos[2] = pl.z;
            } else if (name == "spot") {  // CreateSpotLight, lights/spot.cpp:104-124
                const float coneangle = ps.one_float("coneangle", 30.f), conedelta = ps.one_float("conedeltaangle", 5.f);
                const V3 dir = normalize(to - from);
                V3 du, dv;  // CoordinateSystem, geometry.h:1020-1028
                if (std::abs(dir.x) > std::abs(dir.y))
                    du = div(V3(-dir.z, 0, dir.x), std::sqrt(dir.x * dir.x + dir.z * dir.z));
                else
                    du = div(V3(0, dir.z, -dir.y), std::sqrt(dir.y * dir.y + dir.z * dir.z));
                dv = cross(dir, du);
                Mat4 m;
                const float rows[4][4] = {{du.x, du.y, du.z, 0}, {dv.x, dv.y, dv.z, 0}, {dir.x, dir.y, dir.z, 0}, {0, 0, 0, 1}};
                for (int r = 0; r < 4; ++r)
                    for (int c = 0; c < 4; ++c) m.m[r][c] = rows[r][c];
                const Xform dir_to_z(m);
                const Xform l2w = ctm_ * xf_translate(from) * inverse(dir_to_z);
                const V3 pl = l2w.point(V3(0, 0, 0));
                lt.type = IILE_LIGHT_SPOT;
                lt.pos[0] = pl.x;
                lt.pos[1] = pl.y;
                lt.pos[2] = pl.z;
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) lt.w2l[3 * r + c] = l2w.inv.m[r][c];  // WorldToLight = Inverse(LightToWorld)
                lt.cos_total_width = std::cos(radians(coneangle));                   // SpotLight ctor, spot.cpp:43-51
                lt.cos_falloff_start = std::cos(radians(coneangle - conedelta));
            } else if (name == "infinite" || name == "exinfinite") {  // CreateInfiniteLight, lights/infinite.cpp:176-186
                lt.type = IILE_LIGHT_INFINITE;
                lt.n_samples = ps.one_int("samples", ps.one_int("nsamples", 1));   // infinite.cpp:181-182
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) {
                        lt.l2w[3 * r + c] = ctm_.m.m[r][c];
                        lt.w2l[3 * r + c] = ctm_.inv.m[r][c];
                    }
                // InfiniteAreaLight ctor (infinite.cpp:42-84): Lmap = the environment map times L (one texel of L
                // without a map or when it cannot be read), not flipped; then the sampling distribution
                std::vector<float> rgb;
                int w = 0, h = 0;
                std::string mapname = ps.one_string("mapname", "");
                if (!mapname.empty()) {
                    if (mapname[0] != '/') mapname = search_dir_ + "/" + mapname;
                    std::string why;
                    if (read_image(mapname, &rgb, &w, &h, &why)) {
                        for (size_t i = 0; i < rgb.size(); ++i) rgb[i] *= lt.lemit[i % 3];
                    } else {
                        std::fprintf(stderr, "Warning: %s\n", why.c_str());
                        rgb.clear();
                    }
                }
                if (rgb.empty()) {
                    rgb.assign(lt.lemit, lt.lemit + 3);
                    w = h = 1;
                }
                HostTexture ht;
                std::string why;
                if (!build_environment_light(rgb, w, h, &ht, &scene_->env_dist, &lt.dist_w, &lt.dist_h, &lt.dist_offset, &why))
                    return fail("infinite light: " + why);
                scene_->textures.push_back(std::move(ht));
                lt.env_tex = int(scene_->textures.size()) - 1;
                // world_radius is set once the scene bounds are known (finalize_scene)
            } else {  // CreateDistantLight, lights/distant.cpp:94-102; ctor :43-48
                const V3 w = normalize(ctm_.vector(from - to));
                lt.type = IILE_LIGHT_DISTANT;
                lt.pos[0] = w.x;
                lt.pos[1] = w.y;
                lt.pos[2] = w.z;
                // world_radius is set once the scene bounds are known (finalize_scene)
            }
            scene_->lights.push_back(lt);
        } else if (d == "Shape") {
            return make_shape(name, ps);
        }
        return true;
    }

    // materials/matte.cpp:64-71, plastic.cpp:72-84, microfacet.h:123-128
    // value of a float / spectrum parameter that may be given directly or as a reference to a named
    // texture (TextureParams::GetFloatTexture / GetSpectrumTexture, paramset.cpp)
    bool tex_value(const ParamSet &ps, const std::string &name, bool want_float, const float def[3], float out[3]) {
        for (int i = 0; i < 3; ++i) out[i] = def[i];
        const Param *p = ps.find(name);
        if (!p) return true;
        if (p->type == "texture") {
            if (p->strs.size() != 1) return fail("bad texture reference for \"" + name + "\"");
            auto it = gs_.textures.find(p->strs[0]);
            if (it == gs_.textures.end() || it->second.is_float != want_float)
                return fail(std::string("Couldn't find ") + (want_float ? "float" : "spectrum") + " texture named \"" +
                            p->strs[0] + "\" for parameter \"" + name + "\"");
            if (it->second.image >= 0)
                return fail("texture \"" + p->strs[0] + "\": scale / mix of image textures is not supported");
            for (int i = 0; i < 3; ++i) out[i] = it->second.v[i];
            return true;
        }
        if (want_float) {
            if (p->type == "float" && p->nums.size() == 1) out[0] = out[1] = out[2] = float(p->nums[0]);
        } else if ((p->type == "color" || p->type == "rgb") && p->nums.size() == 3) {
            for (int i = 0; i < 3; ++i) out[i] = float(p->nums[i]);
        }
        return true;
    }
    // "scale" with exactly one plain spectrum image texture among tex1 / tex2 and a constant for the other
    bool scale_of_image(const ParamSet &ps, ConstTexture *t) {
        const char *names[2] = {"tex1", "tex2"};
        int which = -1;
        for (int k = 0; k < 2; ++k) {
            const Param *p = ps.find(names[k]);
            if (!p || p->type != "texture" || p->strs.size() != 1) continue;
            auto it = gs_.textures.find(p->strs[0]);
            if (it == gs_.textures.end() || it->second.image < 0 || it->second.is_float || it->second.scaled) continue;
            if (which >= 0) return false;  // two images: not folded
            which = k;
            t->image = it->second.image;
        }
        if (which < 0) return false;
        const float one[3] = {1, 1, 1};
        ParamSet rest;
        for (const Param &p : ps.params)
            if (p.name != names[which]) rest.params.push_back(p);
        float c[3];
        if (!tex_value(rest, names[1 - which], false, one, c)) {
            t->image = -1;
            return false;
        }
        for (int i = 0; i < 3; ++i) t->v[i] = c[i];
        t->scaled = true;
        return true;
    }
    bool make_texture(const std::string &name, const std::string &type, const std::string &cls, const ParamSet &ps) {
        const bool is_float = type == "float";
        if (!is_float && type != "spectrum" && type != "color") return fail("Texture type \"" + type + "\" unknown.");
        const float zero[3] = {0, 0, 0}, one[3] = {1, 1, 1}, half[3] = {.5f, .5f, .5f};
        ConstTexture t;
        t.is_float = is_float;
        if (cls == "constant") {  // CreateConstant*Texture, textures/constant.cpp: "value" default 1
            if (!tex_value(ps, "value", is_float, one, t.v)) return false;
        } else if (cls == "scale" && !is_float && scale_of_image(ps, &t)) {
            // an image texture times a constant: kept as (image, factor); the product is taken at the hit
        } else if (cls == "scale") {  // ScaleTexture::Evaluate = tex1 * tex2, textures/scale.h:56-58 (defaults 1, 1)
            float a[3], b[3];
            if (!tex_value(ps, "tex1", is_float, one, a) || !tex_value(ps, "tex2", is_float, one, b)) return false;
            for (int i = 0; i < 3; ++i) t.v[i] = a[i] * b[i];
        } else if (cls == "mix") {  // MixTexture::Evaluate = (1 - amt) * t1 + amt * t2, textures/mix.h:57-61 (0, 1, 0.5)
            float a[3], b[3], amt[3];
            if (!tex_value(ps, "tex1", is_float, zero, a) || !tex_value(ps, "tex2", is_float, one, b) ||
                !tex_value(ps, "amount", true, half, amt))
                return false;
            for (int i = 0; i < 3; ++i) t.v[i] = (1 - amt[0]) * a[i] + amt[0] * b[i];
        } else if (cls == "imagemap") {  // CreateImageSpectrumTexture, textures/imagemap.cpp:148-187
            const std::string mapping = ps.one_string("mapping", "uv");
            if (mapping != "uv") return fail("2D texture mapping \"" + mapping + "\" is not supported (uv only)");
            HostTexture ht;
            std::memset(&ht.t, 0, sizeof(ht.t));
            ht.t.su = ps.one_float("uscale", 1.f);
            ht.t.sv = ps.one_float("vscale", 1.f);
            ht.t.du = ps.one_float("udelta", 0.f);
            ht.t.dv = ps.one_float("vdelta", 0.f);
            ht.t.max_aniso = ps.one_float("maxanisotropy", 8.f);
            ht.t.trilinear = ps.one_bool("trilinear", false) ? 1 : 0;
            const std::string wrap = ps.one_string("wrap", "repeat");
            ht.t.wrap = wrap == "black" ? IILE_WRAP_BLACK : (wrap == "clamp" ? IILE_WRAP_CLAMP : IILE_WRAP_REPEAT);
            const float scale = ps.one_float("scale", 1.f);
            std::string filename = ps.one_string("filename", "");
            if (!filename.empty() && filename[0] != '/') filename = search_dir_ + "/" + filename;  // ParamSet::FindOneFilename
            const bool gamma = ps.one_bool("gamma", image_is_8bit(filename));
            std::vector<float> rgb;
            int w = 0, h = 0;
            std::string why;
            if (!read_image(filename, &rgb, &w, &h, &why)) {
                // imagemap.cpp:63-70: Warning + a constant grey 1 x 1 texture
                std::fprintf(stderr, "Warning: %s\nWarning: Creating a constant grey texture to replace \"%s\".\n", why.c_str(),
                             filename.c_str());
                rgb.assign(3, 0.5f);
                w = h = 1;
            }
            if (!build_image_texture(rgb, w, h, scale, gamma, is_float, &ht, &why)) return fail("Texture \"" + name + "\": " + why);
            scene_->textures.push_back(std::move(ht));
            t.image = int(scene_->textures.size()) - 1;
        } else
            return fail("Texture class \"" + cls + "\" is not supported (constant, scale, mix of constants; imagemap)");
        gs_.textures[name] = t;
        return true;
    }
    // replaces references to named textures among a material's parameters by their values
    // (an image texture stays a reference: `image_of` gets parameter name -> texture index, and the
    // parameter a placeholder colour that the material's constant-value checks see as non-black)
    bool resolve_textures(const ParamSet &in, ParamSet *out, std::map<std::string, int> *image_of) {
        *out = in;
        for (Param &p : out->params) {
            if (p.type != "texture") continue;
            if (p.strs.size() != 1) return fail("bad texture reference for \"" + p.name + "\"");
            auto it = gs_.textures.find(p.strs[0]);
            if (it == gs_.textures.end()) return fail("Couldn't find texture named \"" + p.strs[0] + "\" for parameter \"" + p.name + "\"");
            if (it->second.image >= 0 && it->second.is_float &&
                (p.name == "bumpmap" || p.name == "roughness" || p.name == "uroughness" || p.name == "vroughness" || p.name == "sigma")) {
                (*image_of)[p.name] = it->second.image;
                p.strs.clear();
                p.type = p.name == "bumpmap" ? "bumpimage" : (p.name == "sigma" ? "sigmaimage" : "roughimage");  // consumed below; no material reads a parameter of these types
                continue;
            }
            if (it->second.image >= 0) {
                if (it->second.is_float || (p.name != "Kd" && p.name != "Ks" && p.name != "Kr" && p.name != "Kt" && p.name != "opacity"))
                    return fail("image texture \"" + p.strs[0] + "\" on parameter \"" + p.name + "\" is not supported (Kd, Ks, Kr, Kt, opacity)");
                (*image_of)[p.name] = it->second.image;
                p.strs.clear();
                p.type = "color";  // the constant the image's value is multiplied with at the hit (1 unless "scale"d)
                if (it->second.scaled)
                    p.nums = {double(it->second.v[0]), double(it->second.v[1]), double(it->second.v[2])};
                else
                    p.nums = {1.0, 1.0, 1.0};
                continue;
            }
            p.strs.clear();
            if (it->second.is_float) {
                p.type = "float";
                p.nums = {double(it->second.v[0])};
            } else {
                p.type = "color";
                p.nums = {double(it->second.v[0]), double(it->second.v[1]), double(it->second.v[2])};
            }
        }
        return true;
    }
    // TrowbridgeReitzDistribution::RoughnessToAlpha, microfacet.h:123-128
    static float roughness_to_alpha(float roughness) {
        roughness = std::max(roughness, 1e-3f);
        const float x = std::log(roughness);
        return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
    }
    int make_material(const std::string &name, const ParamSet &ps_in) {
        ParamSet ps;
        std::map<std::string, int> image_of;
        if (!resolve_textures(ps_in, &ps, &image_of)) return -1;
        iile_material m;
        std::memset(&m, 0, sizeof(m));
        m.kd_tex = m.ks_tex = m.kr_tex = m.kt_tex = m.bump_tex = m.rough_tex = m.sigma_tex = m.opacity_tex = -1;
        m.rough_tex_v = -2;   // (alpha_v follows alpha: every material but an uber / glass with "vroughness")
        m.opacity[0] = m.opacity[1] = m.opacity[2] = 1.f;
        auto image = [&](const char *param) {
            auto it = image_of.find(param);
            return it == image_of.end() ? -1 : it->second;
        };
        if (name == "matte") {
            m.type = IILE_MAT_MATTE;
            float kd[3] = {0.5f, 0.5f, 0.5f};
            ps.rgb("Kd", kd);
            for (int i = 0; i < 3; ++i) m.kd[i] = kd[i];
            m.sigma = clampT(ps.one_float("sigma", 0.f), 0.f, 90.f);  // matte.cpp:56
            if (m.sigma != 0.f) {  // OrenNayar ctor, reflection.h:414-420
                const float sg = radians(m.sigma);
                const float sigma2 = sg * sg;
                m.on_a = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                m.on_b = 0.45f * sigma2 / (sigma2 + 0.09f);
            }
        } else if (name == "plastic" || name == "uber") {
            const bool uber = name == "uber";
            m.type = uber ? IILE_MAT_UBER : IILE_MAT_PLASTIC;
            float kd[3] = {0.25f, 0.25f, 0.25f}, ks[3] = {0.25f, 0.25f, 0.25f};
            ps.rgb("Kd", kd);
            ps.rgb("Ks", ks);
            for (int i = 0; i < 3; ++i) {
                m.kd[i] = kd[i];
                m.ks[i] = ks[i];
            }
            m.roughness = ps.one_float("roughness", .1f);
            if (uber) {  // CreateUberMaterial, uber.cpp:102-127
                float kr[3] = {0, 0, 0}, kt[3] = {0, 0, 0}, op[3] = {1, 1, 1};
                ps.rgb("Kr", kr);
                ps.rgb("Kt", kt);
                ps.rgb("opacity", op);
                for (int i = 0; i < 3; ++i) m.kr[i] = kr[i], m.kt[i] = kt[i], m.opacity[i] = op[i];
                // uber.cpp:73-86: roughu = uroughness or roughness, roughv = vroughness or roughu (numbers here; images: below)
                const float ur = ps.one_float("uroughness", m.roughness), vr = ps.one_float("vroughness", ur);
                m.roughness = ur;
                m.roughness_v = vr;
                if (const Param *pv = ps.find("vroughness")) m.rough_tex_v = pv->type == "roughimage" ? image("vroughness") : -1;
                m.eta = ps.find("eta") ? ps.one_float("eta", 1.5f) : ps.one_float("index", 1.5f);
            }
            if (!uber) m.roughness_v = m.roughness;
            m.remap_roughness = ps.one_bool("remaproughness", true) ? 1 : 0;
            // The reference re-evaluates this per hit (plastic.cpp:61-63, uber.cpp:79-84); it is a per-material constant, so it is
            // evaluated once here.
            m.alpha = m.remap_roughness ? roughness_to_alpha(m.roughness) : m.roughness;
            m.alpha_v = m.remap_roughness ? roughness_to_alpha(m.roughness_v) : m.roughness_v;
        } else if (name == "glass") {  // CreateGlassMaterial, glass.cpp:94-113
            m.type = IILE_MAT_GLASS;
            float kr[3] = {1, 1, 1}, kt[3] = {1, 1, 1};
            ps.rgb("Kr", kr);
            ps.rgb("Kt", kt);
            for (int i = 0; i < 3; ++i) {
                m.kr[i] = kr[i];
                m.kt[i] = kt[i];
            }
            m.eta = ps.find("eta") ? ps.one_float("eta", 1.5f) : ps.one_float("index", 1.5f);
            // glass.cpp:52-73: urough == vrough == 0 is the smooth dielectric; otherwise MicrofacetReflection + MicrofacetTransmission over
            // one TrowbridgeReitzDistribution(RoughnessToAlpha(urough), ...(vrough))
            const float ur = ps.one_float("uroughness", 0.f), vr = ps.one_float("vroughness", 0.f);
            m.roughness = ur;
            m.roughness_v = vr;
            m.rough_tex_v = -1;
            m.remap_roughness = ps.one_bool("remaproughness", true) ? 1 : 0;
            if (ur != 0.f || vr != 0.f) {   // `bool isSpecular = urough == 0 && vrough == 0`, glass.cpp:63
                m.alpha = m.remap_roughness ? roughness_to_alpha(ur) : ur;
                m.alpha_v = m.remap_roughness ? roughness_to_alpha(vr) : vr;
            }
        } else if (name == "mirror") {  // CreateMirrorMaterial, mirror.cpp:57-63
            m.type = IILE_MAT_MIRROR;
            float kr[3] = {0.9f, 0.9f, 0.9f};
            ps.rgb("Kr", kr);
            for (int i = 0; i < 3; ++i) m.kr[i] = kr[i];
        } else {
            fail("Material \"" + name + "\" is not supported (matte, plastic, uber, mirror, glass)");
            return -1;
        }
        // which parameters each material looks up (an image given for one it does not have is ignored, as a
        // constant would be)
        if (m.type == IILE_MAT_MATTE || m.type == IILE_MAT_PLASTIC || m.type == IILE_MAT_UBER) m.kd_tex = image("Kd");
        if (m.type == IILE_MAT_PLASTIC || m.type == IILE_MAT_UBER) m.ks_tex = image("Ks");
        if (m.type == IILE_MAT_UBER || m.type == IILE_MAT_MIRROR || m.type == IILE_MAT_GLASS) m.kr_tex = image("Kr");
        if (m.type == IILE_MAT_GLASS || m.type == IILE_MAT_UBER) m.kt_tex = image("Kt");
        if (m.type == IILE_MAT_UBER) m.opacity_tex = image("opacity");   // GetSpectrumTexture("opacity", 1.f), uber.cpp:117
        if (const Param *sp = ps.find("sigma"))
            if (sp->type == "sigmaimage") {
                if (m.type != IILE_MAT_MATTE) {
                    fail("sigma: a float \"imagemap\" texture is supported on matte only");
                    return -1;
                }
                m.sigma_tex = image("sigma");
            }
        {   // float images for the roughness parameters: "roughness" (plastic, uber), "uroughness" / "vroughness" (uber)
            const Param *rp = ps.find("roughness"), *up = ps.find("uroughness"), *vp = ps.find("vroughness");
            const bool r_img = rp && rp->type == "roughimage", u_img = up && up->type == "roughimage", v_img = vp && vp->type == "roughimage";
            if ((r_img && m.type != IILE_MAT_PLASTIC && m.type != IILE_MAT_UBER) || ((u_img || v_img) && m.type != IILE_MAT_UBER)) {
                fail("roughness: a float \"imagemap\" texture is supported on plastic (\"roughness\") and uber (\"roughness\", \"uroughness\", \"vroughness\") only");
                return -1;
            }
            // roughu = roughnessu ? roughnessu : roughness (uber.cpp:79-82): "roughness" is not looked at when "uroughness" is there
            if (m.type == IILE_MAT_UBER && up)
                m.rough_tex = u_img ? image("uroughness") : -1;
            else if (r_img)
                m.rough_tex = image("roughness");
        }
        if (const Param *bp = ps.find("bumpmap")) {  // GetFloatTextureOrNull("bumpmap") of every material's Create*
            if (bp->type != "bumpimage" || m.type == IILE_MAT_GLASS) {
                fail("bumpmap: only a float \"imagemap\" texture on matte / plastic / uber / mirror is supported");
                return -1;
            }
            m.bump_tex = image("bumpmap");
        }
        scene_->materials.push_back(m);
        return int(scene_->materials.size()) - 1;
    }

    int current_material() {
        if (gs_.material >= 0) return gs_.material;
        if (default_material_ < 0) default_material_ = make_material("matte", ParamSet());
        return default_material_;
    }

    bool make_shape(const std::string &name, const ParamSet &ps) {
        HostScene &s = *scene_;
        const Xform o2w = ctm_;
        const bool flip = gs_.reverse_orientation ^ o2w.swaps_handedness();
        int mat = current_material();
        if (mat < 0) return false;
        if (name == "sphere") {  // shapes/sphere.cpp:318-327, sphere.h:50-60
            float radius = ps.one_float("radius", 1.f);
            float zmin = ps.one_float("zmin", -radius);
            float zmax = ps.one_float("zmax", radius);
            float phimax = ps.one_float("phimax", 360.f);
            iile_sphere sp;
            std::memset(&sp, 0, sizeof(sp));
            std::memcpy(sp.o2w, o2w.m.m, sizeof(sp.o2w));
            std::memcpy(sp.o2w_inv, o2w.inv.m, sizeof(sp.o2w_inv));
            sp.radius = radius;
            sp.zmin = clampT(std::min(zmin, zmax), -radius, radius);
            sp.zmax = clampT(std::max(zmin, zmax), -radius, radius);
            sp.theta_min = std::acos(clampT(std::min(zmin, zmax) / radius, -1.f, 1.f));
            sp.theta_max = std::acos(clampT(std::max(zmin, zmax) / radius, -1.f, 1.f));
            sp.phi_max = radians(clampT(phimax, 0.f, 360.f));
            sp.reverse_orientation = gs_.reverse_orientation;
            sp.swaps_handedness = o2w.swaps_handedness();
            s.spheres.push_back(sp);
            HostPrim pr;
            pr.flags = IILE_PRIM_SPHERE | (flip ? IILE_PRIM_FLIP : 0);
            pr.material = mat;
            pr.shape = int(s.spheres.size()) - 1;
            // Shape::WorldBound = ObjectToWorld(ObjectBound()), shape.cpp:54, sphere.cpp:43-46
            pr.world_bound =
                o2w.bounds(Bounds3(V3(-radius, -radius, sp.zmin), V3(radius, radius, sp.zmax)));
            if (gs_.has_area_light) {  // api.cpp:808-826 (MakeAreaLight), lights/diffuse.cpp:125-146
                float L[3] = {1, 1, 1}, sc[3] = {1, 1, 1};
                gs_.area_light_params.rgb("L", L);
                gs_.area_light_params.rgb("scale", sc);
                iile_light lt;
                std::memset(&lt, 0, sizeof(lt));
                for (int i = 0; i < 3; ++i) lt.lemit[i] = L[i] * sc[i];
                lt.two_sided = gs_.area_light_params.one_bool("twosided", false);
                lt.n_samples = gs_.area_light_params.one_int("samples", gs_.area_light_params.one_int("nsamples", 1));
                lt.sphere = pr.shape;
                s.lights.push_back(lt);
                pr.light = int(s.lights.size()) - 1;
            }
            s.prims.push_back(pr);
            return true;
        }
        std::vector<int> indices;
        std::vector<V3> P, N;
        std::vector<float> uv;
        if (name == "plymesh") {  // shapes/plymesh.cpp:149-300
            std::string fn = ps.one_string("filename", "");
            if (fn.empty()) return fail("plymesh: \"filename\" is required");
            if (fn[0] != '/') fn = search_dir_ + "/" + fn;
            std::string perr;
            if (!load_ply(fn, &P, &N, &uv, &indices, &perr)) return fail(perr);
        } else {
            const Param *pi = ps.find("indices");
            const Param *pp = ps.find("P");
            if (!pi || !pp) return fail("Shape " + name + ": \"indices\" and \"P\" are required");
            for (double v : pi->nums) indices.push_back(int(v));
            for (size_t i = 0; i + 2 < pp->nums.size(); i += 3)
                P.push_back(V3(float(pp->nums[i]), float(pp->nums[i + 1]), float(pp->nums[i + 2])));
        }
        if (name == "plymesh") {
        } else if (name == "trianglemesh") {  // shapes/triangle.cpp:616-714
            const Param *pu = ps.find("uv");
            if (!pu) pu = ps.find("st");
            if (pu) {
                for (double v : pu->nums) uv.push_back(float(v));
                if (uv.size() / 2 < P.size()) uv.clear();
            }
            const Param *pn = ps.find("N");
            if (pn && pn->nums.size() == 3 * P.size())
                for (size_t i = 0; i + 2 < pn->nums.size(); i += 3)
                    N.push_back(V3(float(pn->nums[i]), float(pn->nums[i + 1]), float(pn->nums[i + 2])));
            if (ps.find("S")) return fail("trianglemesh \"S\" tangents are not supported");
        } else if (name == "loopsubdiv") {  // shapes/loopsubdiv.cpp:402-424
            int levels = ps.one_int("levels", ps.one_int("nlevels", 3));
            std::vector<int> oi;
            std::vector<V3> oP, oN;
            loop_subdivide(levels, indices, P, &oi, &oP, &oN);
            indices.swap(oi);
            P.swap(oP);
            N.swap(oN);
        } else
            return fail("Shape \"" + name + "\" is not supported (sphere, trianglemesh, plymesh, loopsubdiv)");
        // "alpha" / "shadowalpha": a float texture by name, or a float that masks everything when it is 0
        // (CreateTriangleMeshShape, triangle.cpp:689-710; CreatePLYMesh, plymesh.cpp:259-285)
        int alpha_mask[2] = {IILE_ALPHA_NONE, IILE_ALPHA_NONE};
        if (name == "trianglemesh" || name == "plymesh") {
            const char *pnames[2] = {"alpha", "shadowalpha"};
            for (int k = 0; k < 2; ++k) {
                const Param *pa = ps.find(pnames[k]);
                if (!pa) continue;
                if (pa->type == "texture") {
                    if (pa->strs.size() != 1) return fail(std::string("bad texture reference for \"") + pnames[k] + "\"");
                    auto it = gs_.textures.find(pa->strs[0]);
                    if (it == gs_.textures.end() || !it->second.is_float)
                        return fail("Couldn't find float texture \"" + pa->strs[0] + "\" for \"" + pnames[k] + "\" parameter");
                    if (it->second.image >= 0)
                        alpha_mask[k] = it->second.image;
                    else if (it->second.v[0] == 0.f)
                        alpha_mask[k] = IILE_ALPHA_ZERO;
                } else if (pa->type == "float" && pa->nums.size() == 1 && float(pa->nums[0]) == 0.f) {
                    alpha_mask[k] = IILE_ALPHA_ZERO;
                }
            }
        }
        for (int idx : indices)
            if (idx < 0 || idx >= int(P.size())) return fail("trianglemesh has out-of-bounds vertex index");
        // TriangleMesh ctor, shapes/triangle.cpp:54-93: vertices and normals to world space
        std::vector<V3> Pw(P.size()), Nw(N.size());
        for (size_t i = 0; i < P.size(); ++i) Pw[i] = o2w.point(P[i]);
        for (size_t i = 0; i < N.size(); ++i) Nw[i] = o2w.normal(N[i]);
        int mesh_id = s.n_meshes++;
        size_t ntris = indices.size() / 3;
        s.prims.reserve(s.prims.size() + ntris);
        for (size_t t = 0; t < ntris; ++t) {
            HostPrim pr;
            pr.flags = (N.empty() ? 0 : IILE_PRIM_HAS_NORMALS) | (uv.empty() ? 0 : IILE_PRIM_HAS_UV) |
                       (flip ? IILE_PRIM_FLIP : 0) |
                       ((alpha_mask[0] != IILE_ALPHA_NONE || alpha_mask[1] != IILE_ALPHA_NONE) ? IILE_PRIM_HAS_ALPHA : 0);
            pr.alpha = alpha_mask[0];
            pr.shadow_alpha = alpha_mask[1];
            pr.material = mat;
            pr.shape = mesh_id;
            for (int k = 0; k < 3; ++k) {
                int vi = indices[3 * t + k];
                pr.p[k] = Pw[vi];
                if (!N.empty()) pr.n[k] = Nw[vi];
                if (!uv.empty()) {
                    pr.uv[2 * k] = uv[2 * vi];
                    pr.uv[2 * k + 1] = uv[2 * vi + 1];
                }
            }
            // Triangle::WorldBound, shapes/triangle.cpp:180-186
            pr.world_bound = bunion(Bounds3(pr.p[0], pr.p[1]), pr.p[2]);
            if (gs_.has_area_light) {  // pbrtShape: one DiffuseAreaLight per shape, i.e. per triangle (api.cpp:1403-1419)
                float L[3] = {1, 1, 1}, sc[3] = {1, 1, 1};
                gs_.area_light_params.rgb("L", L);
                gs_.area_light_params.rgb("scale", sc);
                iile_light lt;
                std::memset(&lt, 0, sizeof(lt));
                for (int i = 0; i < 3; ++i) lt.lemit[i] = L[i] * sc[i];
                lt.two_sided = gs_.area_light_params.one_bool("twosided", false);
                lt.n_samples = gs_.area_light_params.one_int("samples", gs_.area_light_params.one_int("nsamples", 1));
                lt.sphere = -1;
                lt.type = IILE_LIGHT_AREA_TRIANGLE;
                lt.prim = -1;  // set once the primitives are in BVH order (finalize_scene)
                s.lights.push_back(lt);
                pr.light = int(s.lights.size()) - 1;
            }
            s.prims.push_back(pr);
        }
        return true;
    }
};

}  // namespace

bool load_pbrt_file(const std::string &path, HostScene *scene, std::string *err) {
    Loader l(scene, err);
    return l.run(path);
}

}  // namespace iile
