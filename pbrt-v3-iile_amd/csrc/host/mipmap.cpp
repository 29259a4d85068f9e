// mipmap.cpp — MIPMap<RGBSpectrum>'s constructor (/root/reference/src/core/mipmap.h:111-208) and the
// texel preparation of ImageTexture::GetTexture (src/textures/imagemap.cpp:53-101) on the host: the
// pyramid is scene data like the BVH, built once, in the reference's float arithmetic and evaluation
// order, and handed to the device (and to the test oracle) as plain arrays.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "host_scene.h"

namespace iile {
namespace {

struct Rgb3 {
    float c[3];
};
inline Rgb3 operator*(float w, Rgb3 a) { return Rgb3{{w * a.c[0], w * a.c[1], w * a.c[2]}}; }
inline Rgb3 operator+(Rgb3 a, Rgb3 b) { return Rgb3{{a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2]}}; }
inline Rgb3 clamp0(Rgb3 a) {  // Spectrum::Clamp(0, Infinity)
    const float inf = std::numeric_limits<float>::infinity();
    return Rgb3{{clampT(a.c[0], 0.f, inf), clampT(a.c[1], 0.f, inf), clampT(a.c[2], 0.f, inf)}};
}

inline int mod_i(int a, int b) {  // pbrt.h:310-314
    int r = a - (a / b) * b;
    return r < 0 ? r + b : r;
}
inline bool is_pow2(int v) { return v && !(v & (v - 1)); }
inline int round_up_pow2(int v) {  // pbrt.h:341-349
    v--;
    v |= v >> 1;
    v |= v >> 2;
    v |= v >> 4;
    v |= v >> 8;
    v |= v >> 16;
    return v + 1;
}
inline int log2_int(uint32_t v) { return 31 - __builtin_clz(v); }


// texture.cpp:254-262 with tau = 2 (texture.h:148)
float lanczos(float x, float tau = 2.f) {
    x = std::abs(x);
    if (x < 1e-5f) return 1;
    if (x > 1.f) return 0;
    x *= kPi;
    float s = std::sin(x * tau) / (x * tau);
    float l = std::sin(x) / x;
    return s * l;
}

struct ResampleWeight {
    int first_texel;
    float weight[4];
};
// mipmap.h:78-97
std::vector<ResampleWeight> resample_weights(int old_res, int new_res) {
    std::vector<ResampleWeight> wt(new_res);
    const float filterwidth = 2.f;
    for (int i = 0; i < new_res; ++i) {
        float center = (i + .5f) * old_res / new_res;
        wt[i].first_texel = int(std::floor((center - filterwidth) + 0.5f));
        for (int j = 0; j < 4; ++j) {
            float pos = wt[i].first_texel + j + .5f;
            wt[i].weight[j] = lanczos((pos - center) / filterwidth);
        }
        float inv_sum = 1 / (wt[i].weight[0] + wt[i].weight[1] + wt[i].weight[2] + wt[i].weight[3]);
        for (int j = 0; j < 4; ++j) wt[i].weight[j] *= inv_sum;
    }
    return wt;
}

// MIPMap::Texel, mipmap.h:210-231
Rgb3 texel(const Rgb3 *lvl, int w, int h, int wrap, int s, int t) {
    switch (wrap) {
    case IILE_WRAP_REPEAT:
        s = mod_i(s, w);
        t = mod_i(t, h);
        break;
    case IILE_WRAP_CLAMP:
        s = clampT(s, 0, w - 1);
        t = clampT(t, 0, h - 1);
        break;
    default:
        if (s < 0 || s >= w || t < 0 || t >= h) return Rgb3{{0, 0, 0}};
    }
    return lvl[size_t(t) * w + s];
}

}  // namespace

// pbrt.h:295-298
float inverse_gamma_correct(float value) {
    if (value <= 0.04045f) return value * 1.f / 12.92f;
    return std::pow((value + 0.055f) * 1.f / 1.055f, 2.4f);
}

void ewa_weight_lut(float *lut) {  // mipmap.h:199-205
    for (int i = 0; i < IILE_EWA_LUT_SIZE; ++i) {
        float alpha = 2;
        float r2 = float(i) / float(IILE_EWA_LUT_SIZE - 1);
        lut[i] = std::exp(-alpha * r2) - std::exp(-alpha);
    }
}

// `rgb`: the image as the readers return it (row 0 = top scanline). Does GetTexture's y flip and convertIn
// (scale, inverse gamma), then MIPMap's constructor.
// `as_float`: ImageTexture<Float, Float> — each texel is the luminance of the image's texel (convertIn to Float,
// imagemap.h:101-104), kept in all three channels: MIPMap<Float> does per texel what MIPMap<RGBSpectrum> does
// per channel, so the pyramid and the lookups are shared.
bool build_image_texture(const std::vector<float> &rgb, int width, int height, float scale, bool gamma, bool as_float,
                         HostTexture *out, std::string *err) {
    std::vector<Rgb3> img(size_t(width) * height);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const float *src = &rgb[(size_t(height - 1 - y) * width + x) * 3];  // imagemap.cpp:67-74
            Rgb3 &d = img[size_t(y) * width + x];
            if (as_float) {
                const float y = 0.212671f * src[0] + 0.715160f * src[1] + 0.072169f * src[2];  // RGBSpectrum::y()
                d.c[0] = d.c[1] = d.c[2] = scale * (gamma ? inverse_gamma_correct(y) : y);
            } else
                for (int c = 0; c < 3; ++c) d.c[c] = scale * (gamma ? inverse_gamma_correct(src[c]) : src[c]);  // imagemap.h:96-100
        }
    std::vector<float> flat(img.size() * 3);
    std::memcpy(flat.data(), img.data(), flat.size() * sizeof(float));
    return build_mip_pyramid(flat, width, height, out, err);
}

bool build_mip_pyramid(const std::vector<float> &rgb, int width, int height, HostTexture *out, std::string *err) {
    std::vector<Rgb3> img(size_t(width) * height);
    std::memcpy(img.data(), rgb.data(), img.size() * sizeof(Rgb3));
    const int wrap = out->t.wrap;
    int res[2] = {width, height};
    if (!is_pow2(res[0]) || !is_pow2(res[1])) {
        const int rp[2] = {round_up_pow2(res[0]), round_up_pow2(res[1])};
        // zoom in s, mipmap.h:130-150
        std::vector<ResampleWeight> sw = resample_weights(res[0], rp[0]);
        std::vector<Rgb3> resampled(size_t(rp[0]) * rp[1], Rgb3{{0, 0, 0}});
        for (int t = 0; t < res[1]; ++t)
            for (int s = 0; s < rp[0]; ++s) {
                Rgb3 acc{{0.f, 0.f, 0.f}};
                for (int j = 0; j < 4; ++j) {
                    int orig = sw[s].first_texel + j;
                    if (wrap == IILE_WRAP_REPEAT)
                        orig = mod_i(orig, res[0]);
                    else if (wrap == IILE_WRAP_CLAMP)
                        orig = clampT(orig, 0, res[0] - 1);
                    if (orig >= 0 && orig < res[0]) acc = acc + sw[s].weight[j] * img[size_t(t) * res[0] + orig];
                }
                resampled[size_t(t) * rp[0] + s] = acc;
            }
        // zoom in t, mipmap.h:152-177
        std::vector<ResampleWeight> tw = resample_weights(res[1], rp[1]);
        std::vector<Rgb3> work(rp[1]);
        for (int s = 0; s < rp[0]; ++s) {
            for (int t = 0; t < rp[1]; ++t) {
                Rgb3 acc{{0.f, 0.f, 0.f}};
                for (int j = 0; j < 4; ++j) {
                    int off = tw[t].first_texel + j;
                    if (wrap == IILE_WRAP_REPEAT)
                        off = mod_i(off, res[1]);
                    else if (wrap == IILE_WRAP_CLAMP)
                        off = clampT(off, 0, res[1] - 1);
                    if (off >= 0 && off < res[1]) acc = acc + tw[t].weight[j] * resampled[size_t(off) * rp[0] + s];
                }
                work[t] = acc;
            }
            for (int t = 0; t < rp[1]; ++t) resampled[size_t(t) * rp[0] + s] = clamp0(work[t]);
        }
        img.swap(resampled);
        res[0] = rp[0];
        res[1] = rp[1];
    }
    const int n_levels = 1 + log2_int(uint32_t(std::max(res[0], res[1])));
    if (n_levels > IILE_MAX_TEX_LEVELS) {
        *err = "texture larger than 32768 texels on a side";
        return false;
    }
    out->t.n_levels = n_levels;
    std::vector<Rgb3> all(img);
    int64_t off = 0;
    int w = res[0], h = res[1];
    for (int l = 0; l < n_levels; ++l) {
        out->t.level_w[l] = w;
        out->t.level_h[l] = h;
        out->t.level_offset[l] = off;
        if (l + 1 == n_levels) break;
        const int sw = std::max(1, w / 2), th = std::max(1, h / 2);
        const int64_t next = off + int64_t(w) * h;
        all.resize(size_t(next + int64_t(sw) * th));
        for (int t = 0; t < th; ++t)
            for (int s = 0; s < sw; ++s) {  // mipmap.h:189-196
                const Rgb3 *fine = &all[size_t(off)];
                all[size_t(next) + size_t(t) * sw + s] =
                    .25f * (((texel(fine, w, h, wrap, 2 * s, 2 * t) + texel(fine, w, h, wrap, 2 * s + 1, 2 * t)) +
                             texel(fine, w, h, wrap, 2 * s, 2 * t + 1)) +
                            texel(fine, w, h, wrap, 2 * s + 1, 2 * t + 1));
            }
        off = next;
        w = sw;
        h = th;
    }
    out->texels.resize(all.size() * 3);
    std::memcpy(out->texels.data(), all.data(), all.size() * sizeof(Rgb3));
    return true;
}

namespace {
// MIPMap::triangle and Lookup(st, width) (mipmap.h:233-262) on a built pyramid; Log2 through libm as in the
// reference (this runs on the host there too)
Rgb3 pyr_texel(const HostTexture &t, int level, int s, int tt) {
    return texel(reinterpret_cast<const Rgb3 *>(t.texels.data()) + t.t.level_offset[level], t.t.level_w[level], t.t.level_h[level],
                 t.t.wrap, s, tt);
}
Rgb3 pyr_triangle(const HostTexture &t, int level, float s_, float t_) {
    level = clampT(level, 0, t.t.n_levels - 1);
    float s = s_ * t.t.level_w[level] - 0.5f;
    float tt = t_ * t.t.level_h[level] - 0.5f;
    int s0 = int(std::floor(s)), t0 = int(std::floor(tt));
    float ds = s - s0, dt = tt - t0;
    return (((1 - ds) * (1 - dt)) * pyr_texel(t, level, s0, t0) + ((1 - ds) * dt) * pyr_texel(t, level, s0, t0 + 1)) +
           (ds * (1 - dt)) * pyr_texel(t, level, s0 + 1, t0) + (ds * dt) * pyr_texel(t, level, s0 + 1, t0 + 1);
}
Rgb3 pyr_lookup(const HostTexture &t, float s, float tt, float width) {
    const float inv_log2 = 1.442695040888963387004650940071f;
    float level = t.t.n_levels - 1 + std::log(std::max(width, 1e-8f)) * inv_log2;
    if (level < 0) return pyr_triangle(t, 0, s, tt);
    if (level >= t.t.n_levels - 1) return pyr_texel(t, t.t.n_levels - 1, 0, 0);
    int il = int(std::floor(level));
    float delta = level - il;
    return (1 - delta) * pyr_triangle(t, il, s, tt) + delta * pyr_triangle(t, il + 1, s, tt);
}
// Distribution1D, sampling.h:57-69: {func[n], cdf[n + 1], funcInt}
void dist1d(const float *f, int n, float *out) {
    float *func = out, *cdf = out + n, *func_int = out + 2 * n + 1;
    for (int i = 0; i < n; ++i) func[i] = f[i];
    cdf[0] = 0;
    for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / n;
    *func_int = cdf[n];
    if (*func_int == 0)
        for (int i = 1; i < n + 1; ++i) cdf[i] = float(i) / float(n);
    else
        for (int i = 1; i < n + 1; ++i) cdf[i] /= *func_int;
}
}  // namespace

void mip_lookup_width(const HostTexture &t, float s, float tt, float width, float rgb[3]) {
    const Rgb3 c = pyr_lookup(t, s, tt, width);
    for (int i = 0; i < 3; ++i) rgb[i] = c.c[i];
}

bool build_environment_light(const std::vector<float> &rgb, int width, int height, HostTexture *tex, std::vector<float> *dist,
                             int *dist_w, int *dist_h, int64_t *dist_offset, std::string *err) {
    std::memset(&tex->t, 0, sizeof(tex->t));  // MIPMap defaults: EWA, maxAniso 8, repeat (mipmap.h:66-67)
    tex->t.wrap = IILE_WRAP_REPEAT;
    tex->t.max_aniso = 8.f;
    tex->t.su = tex->t.sv = 1.f;
    if (!build_mip_pyramid(rgb, width, height, tex, err)) return false;
    // scalar image of filtered luminance * sin(theta), infinite.cpp:65-80
    const int w = 2 * tex->t.level_w[0], h = 2 * tex->t.level_h[0];
    std::vector<float> img(size_t(w) * h);
    const float fwidth = 0.5f / std::min(w, h);
    for (int v = 0; v < h; ++v) {
        const float vp = (v + .5f) / float(h);
        const float sin_theta = std::sin(kPi * (v + .5f) / h);
        for (int u = 0; u < w; ++u) {
            const float up = (u + .5f) / float(w);
            const Rgb3 c = pyr_lookup(*tex, up, vp, fwidth);
            img[u + size_t(v) * w] = 0.212671f * c.c[0] + 0.715160f * c.c[1] + 0.072169f * c.c[2];  // RGBSpectrum::y
            img[u + size_t(v) * w] *= sin_theta;
        }
    }
    // Distribution2D, sampling.cpp:159-174
    *dist_w = w;
    *dist_h = h;
    *dist_offset = int64_t(dist->size());
    const size_t row = size_t(2 * w + 2);
    dist->resize(dist->size() + row * h + size_t(2 * h + 2));
    float *base = dist->data() + *dist_offset;
    std::vector<float> marg(h);
    for (int v = 0; v < h; ++v) {
        dist1d(&img[size_t(v) * w], w, base + row * v);
        marg[v] = base[row * v + 2 * w + 1];
    }
    dist1d(marg.data(), h, base + row * h);
    return true;
}

}  // namespace iile
