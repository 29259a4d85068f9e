// imageio.cpp — the image readers behind ImageTexture: ReadImage of
// /root/reference/src/core/imageio.cpp:60-82 for the formats that need no library the image lacks:
// PFM (imageio.cpp:350-436), TGA (imageio.cpp:216-256 over ext/targa) and PNG (imageio.cpp:258-287
// over ext/lodepng; here: the chunk walk and un-filtering over zlib's inflate). OpenEXR scan-line files: exr.cpp.
//
// All three return what the reference's readers return: RGB floats, row 0 = top scanline, 8-bit samples
// as c / 255.f.
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_scene.h"

namespace iile {
namespace {

bool has_extension(const std::string &name, const char *ext) {  // fileutil.h HasExtension: case-insensitive suffix
    const size_t n = std::strlen(ext);
    if (name.size() < n) return false;
    for (size_t i = 0; i < n; ++i)
        if (std::tolower((unsigned char)name[name.size() - n + i]) != std::tolower((unsigned char)ext[i])) return false;
    return true;
}

bool read_file(const std::string &path, std::vector<uint8_t> *out) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out->resize(n > 0 ? size_t(n) : 0);
    const bool ok = n >= 0 && std::fread(out->data(), 1, out->size(), f) == out->size();
    std::fclose(f);
    return ok;
}

// ---- PFM -----------------------------------------------------------------------------------------
// header words are separated by single whitespace characters (readWord, imageio.cpp:328-348)
bool pfm_word(const std::vector<uint8_t> &d, size_t *pos, std::string *w) {
    w->clear();
    while (*pos < d.size()) {
        const uint8_t c = d[(*pos)++];
        if (c == ' ' || c == '\n' || c == '\t') return true;
        w->push_back(char(c));
        if (w->size() >= 80) return false;
    }
    return true;
}

bool read_pfm(const std::string &path, std::vector<float> *rgb, int *w, int *h, std::string *err) {
    std::vector<uint8_t> d;
    auto fail = [&]() {
        *err = "Error reading PFM file \"" + path + "\"";
        return false;
    };
    if (!read_file(path, &d)) return fail();
    size_t pos = 0;
    std::string word;
    if (!pfm_word(d, &pos, &word)) return fail();
    int nc;
    if (word == "Pf")
        nc = 1;
    else if (word == "PF")
        nc = 3;
    else
        return fail();
    if (!pfm_word(d, &pos, &word)) return fail();
    const int width = std::atoi(word.c_str());
    if (!pfm_word(d, &pos, &word)) return fail();
    const int height = std::atoi(word.c_str());
    if (!pfm_word(d, &pos, &word)) return fail();
    float scale = 0;
    std::sscanf(word.c_str(), "%f", &scale);
    if (width <= 0 || height <= 0) return fail();
    const size_t n = size_t(nc) * width * height;
    if (d.size() - pos < n * 4) return fail();
    std::vector<float> data(n);
    // the file's first row is the bottom scanline
    for (int y = height - 1; y >= 0; --y) {
        std::memcpy(&data[size_t(y) * nc * width], &d[pos], size_t(nc) * width * 4);
        pos += size_t(nc) * width * 4;
    }
    const bool file_little = scale < 0.f;
    if (!file_little)  // this host is little-endian
        for (size_t i = 0; i < n; ++i) {
            uint8_t b[4];
            std::memcpy(b, &data[i], 4);
            std::swap(b[0], b[3]);
            std::swap(b[1], b[2]);
            std::memcpy(&data[i], b, 4);
        }
    if (std::abs(scale) != 1.f)
        for (size_t i = 0; i < n; ++i) data[i] *= std::abs(scale);
    rgb->resize(size_t(3) * width * height);
    for (size_t i = 0; i < size_t(width) * height; ++i)
        for (int c = 0; c < 3; ++c) (*rgb)[3 * i + c] = nc == 1 ? data[i] : data[3 * i + c];
    *w = width;
    *h = height;
    return true;
}

// ---- TGA -----------------------------------------------------------------------------------------
// Truevision TGA: uncompressed / run-length encoded true-colour (24/32 bit), grey (8 bit) and
// colour-mapped (8 bit indices into a 24/32 bit map) images; the origin bits of the descriptor say where
// row 0 / column 0 are. Output follows ReadImageTGA: top-to-bottom, left-to-right, BGR(A) -> RGB.
bool read_tga(const std::string &path, std::vector<float> *rgb, int *w, int *h, std::string *err) {
    std::vector<uint8_t> d;
    auto fail = [&](const char *why) {
        *err = "Unable to read from TGA file \"" + path + "\" (" + why + ")";
        return false;
    };
    if (!read_file(path, &d)) return fail("cannot open");
    if (d.size() < 18) return fail("truncated header");
    const int id_len = d[0], map_type = d[1], img_type = d[2];
    const int map_origin = d[3] | (d[4] << 8), map_len = d[5] | (d[6] << 8), map_depth = d[7];
    const int width = d[12] | (d[13] << 8), height = d[14] | (d[15] << 8), depth = d[16], desc = d[17];
    const bool rle = (img_type & 8) != 0;
    const int kind = img_type & 7;  // 1 colour-mapped, 2 true colour, 3 grey
    if (kind < 1 || kind > 3 || img_type > 11) return fail("unsupported image type");
    if (width <= 0 || height <= 0) return fail("bad size");
    if (kind == 1 && (map_type != 1 || depth != 8 || (map_depth != 24 && map_depth != 32))) return fail("unsupported colour map");
    // ReadImageTGA reads three bytes B, G, R at each pixel: only 24 / 32 bit true colour makes sense
    if (kind == 2 && depth != 24 && depth != 32) return fail("unsupported pixel depth");
    if (kind == 3 && depth != 8) return fail("unsupported grey depth");
    size_t pos = 18 + size_t(id_len);
    const int map_bpp = (map_depth + 7) / 8;
    std::vector<uint8_t> cmap;
    if (map_type == 1) {
        const size_t nb = size_t(map_len) * map_bpp;
        if (d.size() < pos + nb) return fail("truncated colour map");
        cmap.assign(d.begin() + pos, d.begin() + pos + nb);
        pos += nb;
    }
    const int bpp = (depth + 7) / 8;
    const size_t npx = size_t(width) * height;
    std::vector<uint8_t> px(npx * bpp);
    if (!rle) {
        if (d.size() < pos + px.size()) return fail("truncated image data");
        std::memcpy(px.data(), &d[pos], px.size());
    } else {
        size_t i = 0;
        while (i < npx) {
            if (pos >= d.size()) return fail("truncated run-length data");
            const int hdr = d[pos++];
            const size_t count = size_t(hdr & 0x7f) + 1;
            if (i + count > npx) return fail("run crosses the end of the image");
            if (hdr & 0x80) {
                if (d.size() < pos + bpp) return fail("truncated run-length data");
                for (size_t k = 0; k < count; ++k) std::memcpy(&px[(i + k) * bpp], &d[pos], bpp);
                pos += bpp;
            } else {
                if (d.size() < pos + count * bpp) return fail("truncated run-length data");
                std::memcpy(&px[i * bpp], &d[pos], count * bpp);
                pos += count * bpp;
            }
            i += count;
        }
    }
    const bool right_to_left = (desc & 0x10) != 0, top_to_bottom = (desc & 0x20) != 0;
    rgb->resize(3 * npx);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const int sx = right_to_left ? width - 1 - x : x, sy = top_to_bottom ? y : height - 1 - y;
            const uint8_t *s = &px[(size_t(sy) * width + sx) * bpp];
            uint8_t b, g, r;
            if (kind == 3) {
                b = g = r = s[0];
            } else if (kind == 1) {
                const int idx = int(s[0]) - map_origin;
                if (idx < 0 || idx >= map_len) return fail("colour index out of range");
                b = cmap[size_t(idx) * map_bpp], g = cmap[size_t(idx) * map_bpp + 1], r = cmap[size_t(idx) * map_bpp + 2];
            } else {
                b = s[0], g = s[1], r = s[2];
            }
            float *o = &(*rgb)[(size_t(y) * width + x) * 3];
            o[0] = r / 255.f;
            o[1] = g / 255.f;
            o[2] = b / 255.f;
        }
    *w = width;
    *h = height;
    return true;
}

// ---- PNG -----------------------------------------------------------------------------------------
uint32_t be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// lodepng_decode24_file: any PNG colour type converted to 8-bit RGB (alpha dropped, grey replicated,
// palette looked up, 16-bit samples reduced to their high byte). Non-interlaced images only.
bool read_png(const std::string &path, std::vector<float> *rgb, int *w, int *h, std::string *err) {
    std::vector<uint8_t> d;
    auto fail = [&](const char *why) {
        *err = "Error reading PNG \"" + path + "\": " + why;
        return false;
    };
    if (!read_file(path, &d)) return fail("failed to open file for reading");
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (d.size() < 8 + 25 || std::memcmp(d.data(), sig, 8) != 0) return fail("incorrect PNG signature");
    size_t pos = 8;
    uint32_t width = 0, height = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool seen_end = false;
    while (!seen_end && pos + 12 <= d.size()) {
        const uint32_t len = be32(&d[pos]);
        const uint8_t *tag = &d[pos + 4];
        if (pos + 12 + size_t(len) > d.size()) return fail("chunk length larger than the file");
        const uint8_t *body = &d[pos + 8];
        if (crc32(crc32(0, Z_NULL, 0), tag, 4 + len) != be32(body + len)) return fail("invalid CRC");
        if (!std::memcmp(tag, "IHDR", 4)) {
            if (len != 13) return fail("invalid IHDR");
            width = be32(body), height = be32(body + 4);
            depth = body[8], ctype = body[9], interlace = body[12];
            if (body[10] != 0 || body[11] != 0) return fail("invalid compression / filter method");
        } else if (!std::memcmp(tag, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!std::memcmp(tag, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(tag, "IEND", 4)) {
            seen_end = true;
        }
        pos += 12 + size_t(len);
    }
    if (ctype < 0 || width == 0 || height == 0) return fail("no IHDR chunk");
    if (interlace != 0) return fail("interlaced PNGs are not supported");
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return fail("illegal colour type");
    }
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!depth_ok) return fail("illegal bit depth for this colour type");
    const size_t bits_pp = size_t(channels) * depth;
    const size_t stride = (size_t(width) * bits_pp + 7) / 8;
    const size_t bpp = std::max<size_t>(1, bits_pp / 8);  // filter distance in bytes
    std::vector<uint8_t> raw((stride + 1) * size_t(height));
    uLongf raw_len = raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), idat.size()) != Z_OK || raw_len != raw.size())
        return fail("zlib stream does not decode to the image size");
    // un-filter in place (PNG spec section 9)
    std::vector<uint8_t> img(stride * size_t(height));
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t *in = &raw[(stride + 1) * y];
        const int ft = in[0];
        ++in;
        uint8_t *out = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? out[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int v = in[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: return fail("illegal scanline filter");
            }
            out[i] = uint8_t(v);
        }
    }
    rgb->resize(size_t(3) * width * height);
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t *row = &img[stride * y];
        for (uint32_t x = 0; x < width; ++x) {
            uint8_t c[3];
            if (ctype == 0 || ctype == 3) {
                int v;
                if (depth == 16)
                    v = row[2 * x];
                else if (depth == 8)
                    v = row[x];
                else {
                    const size_t bit = size_t(x) * depth;
                    v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                }
                if (ctype == 3) {
                    if (size_t(v) * 3 + 3 > plte.size())
                        c[0] = c[1] = c[2] = 0;  // index past the palette decodes as black
                    else
                        c[0] = plte[3 * v], c[1] = plte[3 * v + 1], c[2] = plte[3 * v + 2];
                } else {
                    if (depth < 8) v = (v * 255) / ((1 << depth) - 1);
                    c[0] = c[1] = c[2] = uint8_t(v);
                }
            } else {
                const size_t bytes = depth / 8;
                const uint8_t *p = row + size_t(x) * channels * bytes;
                if (ctype == 4)
                    c[0] = c[1] = c[2] = p[0];
                else
                    c[0] = p[0], c[1] = p[bytes], c[2] = p[2 * bytes];
            }
            float *o = &(*rgb)[(size_t(y) * width + x) * 3];
            for (int k = 0; k < 3; ++k) o[k] = c[k] / 255.f;
        }
    }
    *w = int(width);
    *h = int(height);
    return true;
}

}  // namespace

bool read_image(const std::string &path, std::vector<float> *rgb, int *w, int *h, std::string *err) {
    if (has_extension(path, ".exr")) return read_exr(path, rgb, w, h, err);  // exr.cpp: scan-line files, ZIP / ZIPS / none
    if (has_extension(path, ".tga")) return read_tga(path, rgb, w, h, err);
    if (has_extension(path, ".png")) return read_png(path, rgb, w, h, err);
    if (has_extension(path, ".pfm")) return read_pfm(path, rgb, w, h, err);
    const size_t dot = path.find_last_of('.');
    *err = "Unable to load image stored in format \"" + (dot == std::string::npos ? std::string("(unknown)") : path.substr(dot + 1)) +
           "\" for filename \"" + path + "\".";
    return false;
}

bool image_is_8bit(const std::string &path) { return has_extension(path, ".tga") || has_extension(path, ".png"); }

}  // namespace iile
