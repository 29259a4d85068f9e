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

const float kPi = 3.14159265358979323846f;
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

}  // namespace iile
