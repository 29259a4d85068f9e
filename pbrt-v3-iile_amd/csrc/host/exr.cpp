// exr.cpp — OpenEXR scan-line images without the OpenEXR library (absent from the image): what ReadImageEXR /
// WriteImageEXR (src/core/imageio.cpp:138-214) exchange with the rest of pbrt.
//
// ReadImageEXR goes through Imf::RgbaInputFile: every channel arrives as a 16-bit `half` whatever its stored type (FLOAT
// channels are rounded to half on the way, UINT converted), R / G / B taken by name, a luminance-only file (Y) spread
// over the three, the image is the file's DATA window. WriteImageEXR writes R, G, B as half (Imf::WRITE_RGB), data
// window = the cropped pixel bounds inside a display window of the full resolution.
//
// Supported here: single-part scan-line files, compression NONE, RLE, ZIPS (one line per block) and ZIP (16 lines), channel
// types HALF / FLOAT / UINT, both line orders. Tiles, deep data, multi-part files, sub-sampled channels and the PIZ /
// PXR24 / B44 / DWA coders are refused with a message naming what was found (Imf's default coder for RGBA
// files is PIZ: such files have to be re-saved as ZIP). The writer emits ZIP.
//
// File layout (OpenEXR "Technical Introduction" / ImfHeader, ImfZip): magic 0x01312f76, version 2, attributes
// `name\0type\0size value`, a zero byte, one u64 offset per chunk, chunks `y, byteCount, data`. A block's bytes before
// compression: line by line, per line the channels in alphabetical order, each w values. ZIP: bytes de-interleaved
// (even positions first), a delta predictor (d = b[i] - b[i-1] + 384 mod 256), zlib; stored raw when that is no shorter.
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "host_scene.h"

namespace iile {
namespace {

// half <-> float as Imath's `half` does it: round to nearest even, overflow to infinity, denormals kept
uint16_t float_to_half(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t e = int32_t((x >> 23) & 0xffu) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (e <= 0) {
        if (e < -10) return uint16_t(sign);  // below half the smallest denormal: +-0
        m |= 0x800000u;
        const int t = 14 - e;                                  // shift that lands the value on the denormal grid
        const uint32_t a = (1u << (t - 1)) - 1, b = (m >> t) & 1u;  // round to nearest even
        return uint16_t(sign | ((m + a + b) >> t));
    }
    if (e == 0xff - 127 + 15) {
        if (m == 0) return uint16_t(sign | 0x7c00u);  // infinity
        m >>= 13;
        return uint16_t(sign | 0x7c00u | m | (m == 0));  // NaN, payload kept non-zero
    }
    m = m + 0xfffu + ((m >> 13) & 1u);  // round to nearest even
    int32_t ee = e;
    if (m & 0x800000u) {
        m = 0;
        ee += 1;
    }
    if (ee > 30) return uint16_t(sign | 0x7c00u);  // overflow
    return uint16_t(sign | (uint32_t(ee) << 10) | (m >> 13));
}
float half_to_float(uint16_t h) {
    const uint32_t sign = uint32_t(h & 0x8000u) << 16;
    int32_t e = (h >> 10) & 0x1f;
    uint32_t m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else {  // denormal: normalise
            while (!(m & 0x400u)) {
                m <<= 1;
                --e;
            }
            ++e;
            m &= 0x3ffu;
            x = sign | (uint32_t(e + 127 - 15) << 23) | (m << 13);
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | (uint32_t(e + 127 - 15) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

struct Reader {
    const std::vector<uint8_t> &d;
    size_t pos = 0;
    bool ok = true;
    explicit Reader(const std::vector<uint8_t> &data) : d(data) {}
    bool need(size_t n) {  // (no addition that could wrap: pos <= d.size() is the invariant)
        if (pos > d.size() || n > d.size() - pos) ok = false;
        return ok;
    }
    uint8_t u8() { return need(1) ? d[pos++] : 0; }
    int32_t i32() {
        if (!need(4)) return 0;
        uint32_t v = uint32_t(d[pos]) | uint32_t(d[pos + 1]) << 8 | uint32_t(d[pos + 2]) << 16 | uint32_t(d[pos + 3]) << 24;
        pos += 4;
        return int32_t(v);
    }
    uint64_t u64() {
        const uint64_t lo = uint32_t(i32()), hi = uint32_t(i32());
        return lo | hi << 32;
    }
    std::string cstr() {
        std::string s;
        while (need(1) && d[pos] != 0 && s.size() < 256) s.push_back(char(d[pos++]));
        if (ok) ++pos;
        return s;
    }
};

bool read_all(const std::string &path, std::vector<uint8_t> *out) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out->resize(n > 0 ? size_t(n) : 0);
    const size_t got = n > 0 ? std::fread(out->data(), 1, size_t(n), f) : 0;
    std::fclose(f);
    return got == out->size();
}

struct Channel {
    std::string name;
    int type;  // 0 UINT, 1 HALF, 2 FLOAT
};

}  // namespace

bool read_exr(const std::string &path, std::vector<float> *rgb, int *w, int *h, std::string *err) {
    auto fail = [&](const std::string &why) {
        *err = "Unable to read image file \"" + path + "\": " + why;
        return false;
    };
    std::vector<uint8_t> d;
    if (!read_all(path, &d)) return fail("cannot open");
    Reader r(d);
    if (uint32_t(r.i32()) != 0x01312f76u) return fail("not an OpenEXR file");
    const uint32_t version = uint32_t(r.i32());
    if ((version & 0xffu) != 2) return fail("unsupported OpenEXR version");
    if (version & 0x200u) return fail("tiled OpenEXR files are not supported");
    if (version & 0x800u) return fail("deep OpenEXR files are not supported");
    if (version & 0x1000u) return fail("multi-part OpenEXR files are not supported");
    std::vector<Channel> channels;
    int compression = -1, line_order = 0;
    int dw[4] = {0, 0, -1, -1};
    bool have_dw = false;
    for (;;) {
        const std::string name = r.cstr();
        if (!r.ok) return fail("truncated header");
        if (name.empty()) break;
        const std::string type = r.cstr();
        const int32_t size = r.i32();
        if (!r.ok || size < 0 || !r.need(size_t(size))) return fail("truncated header");
        const size_t end = r.pos + size_t(size);
        if (name == "channels") {
            while (r.pos < end) {
                Channel c;
                c.name = r.cstr();
                if (c.name.empty()) break;
                c.type = r.i32();
                r.u8();  // pLinear
                r.u8(), r.u8(), r.u8();
                const int xs = r.i32(), ys = r.i32();
                if (!r.ok || c.type < 0 || c.type > 2) return fail("bad channel list");
                if (xs != 1 || ys != 1) return fail("sub-sampled channels are not supported");
                channels.push_back(c);
            }
        } else if (name == "compression") {
            compression = r.u8();
        } else if (name == "dataWindow") {
            for (int k = 0; k < 4; ++k) dw[k] = r.i32();
            have_dw = true;
        } else if (name == "lineOrder") {
            line_order = r.u8();
        }
        r.pos = end;
    }
    if (!have_dw || channels.empty() || compression < 0) return fail("header lacks channels, compression or dataWindow");
    static const char *const kCoder[] = {"NONE", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB"};
    if (compression < 0 || compression > 3)
        return fail(std::string("compression ") + (compression < 10 ? kCoder[compression] : "?") +
                    " is not supported (re-save the file with ZIP, ZIPS, RLE or no compression)");
    const int64_t width = int64_t(dw[2]) - dw[0] + 1, height = int64_t(dw[3]) - dw[1] + 1;
    if (width <= 0 || height <= 0 || width > 65536 || height > 65536) return fail("bad data window");
    (void)line_order;  // every chunk names its first line
    // channels are stored in alphabetical order; the header lists them that way
    size_t line_bytes = 0;
    std::vector<size_t> chan_off(channels.size());
    for (size_t c = 0; c < channels.size(); ++c) {
        chan_off[c] = line_bytes;
        line_bytes += size_t(width) * (channels[c].type == 1 ? 2 : 4);
    }
    const int lines_per_block = compression == 3 ? 16 : 1;
    const int64_t n_blocks = (height + lines_per_block - 1) / lines_per_block;
    if (uint64_t(n_blocks) * 8 > d.size() || width * height > (int64_t(1) << 28)) return fail("data window larger than the file can hold");
    std::vector<uint64_t> offsets(static_cast<size_t>(n_blocks));
    for (int64_t b = 0; b < n_blocks; ++b) offsets[size_t(b)] = r.u64();
    if (!r.ok) return fail("truncated offset table");
    int ci[3] = {-1, -1, -1}, cy = -1;
    for (size_t c = 0; c < channels.size(); ++c) {
        if (channels[c].name == "R") ci[0] = int(c);
        if (channels[c].name == "G") ci[1] = int(c);
        if (channels[c].name == "B") ci[2] = int(c);
        if (channels[c].name == "Y") cy = int(c);
    }
    if (ci[0] < 0 && ci[1] < 0 && ci[2] < 0 && cy < 0) return fail("no R, G, B or Y channel");
    const bool luminance = ci[0] < 0 && ci[1] < 0 && ci[2] < 0;
    // a header of a few KB must not make the reader zero-fill gigabytes before it has seen a chunk: no coder here expands
    // more than zlib's ~1032 : 1 (RLE: 128 : 1; none: 1 : 1)
    const uint64_t decoded = uint64_t(line_bytes) * uint64_t(height);
    const uint64_t max_ratio = compression == 0 ? 1 : (compression == 1 ? 128 : 1040);
    if (decoded / max_ratio > d.size()) return fail("data window larger than the file can hold");
    rgb->assign(size_t(width) * size_t(height) * 3, 0.f);
    std::vector<uint8_t> raw, tmp;
    for (int64_t b = 0; b < n_blocks; ++b) {
        const uint64_t off = offsets[size_t(b)];
        if (off > d.size() || d.size() - off < 8) return fail("chunk offset beyond the file");  // off + 8 would wrap for offsets near 2^64
        Reader c(d);
        c.pos = size_t(off);
        const int32_t y0 = c.i32(), nbytes = c.i32();
        if (nbytes < 0 || !c.need(size_t(nbytes))) return fail("truncated chunk");
        const int64_t first = int64_t(y0) - dw[1];
        if (first < 0 || first >= height || first % lines_per_block != 0) return fail("chunk with a bad line number");
        const int64_t n_lines = std::min<int64_t>(lines_per_block, height - first);
        const size_t want = size_t(n_lines) * line_bytes;
        raw.resize(want);
        if (compression == 0 || size_t(nbytes) == want) {
            if (size_t(nbytes) != want) return fail("chunk size does not match the data window");
            std::memcpy(raw.data(), d.data() + c.pos, want);
        } else {
            tmp.resize(want);
            if (compression == 1) {  // ImfRle rleUncompress: a negative count n copies -n bytes, a count n >= 0 repeats the next byte n + 1 times
                const uint8_t *in = d.data() + c.pos;
                size_t left = size_t(nbytes), at = 0;
                while (left > 0) {
                    const int cnt = int(int8_t(*in++));
                    if (cnt < 0) {
                        const size_t k = size_t(-cnt);
                        if (left < k + 1 || at + k > want) return fail("run-length data does not decode to the block size");
                        std::memcpy(tmp.data() + at, in, k);
                        in += k, at += k, left -= k + 1;
                    } else {
                        const size_t k = size_t(cnt) + 1;
                        if (left < 2 || at + k > want) return fail("run-length data does not decode to the block size");
                        std::memset(tmp.data() + at, *in++, k);
                        at += k, left -= 2;
                    }
                }
                if (at != want) return fail("run-length data does not decode to the block size");
            } else {
                uLongf got = uLongf(want);
                if (uncompress(tmp.data(), &got, d.data() + c.pos, uLong(nbytes)) != Z_OK || got != want)
                    return fail("zlib stream does not decode to the block size");
            }
            for (size_t i = 1; i < want; ++i) tmp[i] = uint8_t(tmp[i - 1] + tmp[i] - 128);  // predictor
            const size_t half = (want + 1) / 2;
            for (size_t i = 0; i < want; ++i) raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];  // re-interleave
        }
        for (int64_t ln = 0; ln < n_lines; ++ln) {
            const uint8_t *line = raw.data() + size_t(ln) * line_bytes;
            float *out = rgb->data() + size_t(first + ln) * size_t(width) * 3;
            auto value = [&](int chan, int64_t x) -> float {  // as RgbaInputFile delivers it: a half
                const uint8_t *p = line + chan_off[size_t(chan)];
                if (channels[size_t(chan)].type == 1) {
                    const uint16_t hv = uint16_t(p[2 * x] | p[2 * x + 1] << 8);
                    return half_to_float(hv);
                }
                uint32_t v;
                std::memcpy(&v, p + 4 * x, 4);
                if (channels[size_t(chan)].type == 0) return half_to_float(float_to_half(float(v)));
                float f;
                std::memcpy(&f, &v, 4);
                return half_to_float(float_to_half(f));
            };
            for (int64_t x = 0; x < width; ++x)
                for (int k = 0; k < 3; ++k) {
                    const int chan = luminance ? cy : ci[k];
                    out[3 * x + k] = chan >= 0 ? value(chan, x) : 0.f;
                }
        }
    }
    *w = int(width);
    *h = int(height);
    return true;
}

// WriteImageEXR: rgb is (y1 - y0) x (x1 - x0) pixels, row 0 the top scanline, placed at (x0, y0) of a total_w x total_h image
bool write_exr(const std::string &path, const float *rgb, int x0, int y0, int x1, int y1, int total_w, int total_h, std::string *err) {
    const int width = x1 - x0, height = y1 - y0;
    if (width <= 0 || height <= 0 || total_w <= 0 || total_h <= 0) {
        *err = "write_exr: empty image";
        return false;
    }
    std::vector<uint8_t> out;
    auto put_u8 = [&](uint8_t v) { out.push_back(v); };
    auto put_i32 = [&](int32_t v) {
        for (int k = 0; k < 4; ++k) out.push_back(uint8_t(uint32_t(v) >> (8 * k)));
    };
    auto put_f32 = [&](float f) {
        int32_t v;
        std::memcpy(&v, &f, 4);
        put_i32(v);
    };
    auto put_str = [&](const char *s) {
        while (*s) out.push_back(uint8_t(*s++));
        out.push_back(0);
    };
    auto attr = [&](const char *name, const char *type, int32_t size) {
        put_str(name);
        put_str(type);
        put_i32(size);
    };
    put_i32(0x01312f76);
    put_i32(2);
    attr("channels", "chlist", 3 * 18 + 1);
    for (const char *c : {"B", "G", "R"}) {
        put_str(c);
        put_i32(1);  // HALF
        put_u8(0), put_u8(0), put_u8(0), put_u8(0);
        put_i32(1), put_i32(1);
    }
    put_u8(0);
    attr("compression", "compression", 1);
    put_u8(3);  // ZIP
    attr("dataWindow", "box2i", 16);
    put_i32(x0), put_i32(y0), put_i32(x1 - 1), put_i32(y1 - 1);
    attr("displayWindow", "box2i", 16);
    put_i32(0), put_i32(0), put_i32(total_w - 1), put_i32(total_h - 1);
    attr("lineOrder", "lineOrder", 1);
    put_u8(0);
    attr("pixelAspectRatio", "float", 4);
    put_f32(1.f);
    attr("screenWindowCenter", "v2f", 8);
    put_f32(0.f), put_f32(0.f);
    attr("screenWindowWidth", "float", 4);
    put_f32(1.f);
    put_u8(0);
    const int n_blocks = (height + 15) / 16;
    const size_t table_at = out.size();
    out.resize(out.size() + size_t(n_blocks) * 8);
    const size_t line_bytes = size_t(width) * 6;
    std::vector<uint8_t> raw, tmp, packed;
    for (int b = 0; b < n_blocks; ++b) {
        const int first = 16 * b, n_lines = std::min(16, height - first);
        const size_t n = size_t(n_lines) * line_bytes;
        raw.resize(n);
        for (int ln = 0; ln < n_lines; ++ln)
            for (int c = 0; c < 3; ++c) {  // B, G, R
                uint8_t *dst = raw.data() + size_t(ln) * line_bytes + size_t(c) * size_t(width) * 2;
                const float *src = rgb + size_t(first + ln) * size_t(width) * 3 + (2 - c);
                for (int x = 0; x < width; ++x) {
                    const uint16_t hv = float_to_half(src[3 * size_t(x)]);
                    dst[2 * x] = uint8_t(hv);
                    dst[2 * x + 1] = uint8_t(hv >> 8);
                }
            }
        tmp.resize(n);
        const size_t half = (n + 1) / 2;
        for (size_t i = 0; i < n; ++i) tmp[(i & 1) ? half + i / 2 : i / 2] = raw[i];
        int p = tmp[0];
        for (size_t i = 1; i < n; ++i) {
            const int dlt = int(tmp[i]) - p + (128 + 256);
            p = tmp[i];
            tmp[i] = uint8_t(dlt);
        }
        uLongf clen = compressBound(uLong(n));
        packed.resize(clen);
        const bool zipped = compress(packed.data(), &clen, tmp.data(), uLong(n)) == Z_OK && clen < n;
        const uint64_t at = out.size();
        for (int k = 0; k < 8; ++k) out[table_at + size_t(b) * 8 + size_t(k)] = uint8_t(at >> (8 * k));
        put_i32(y0 + first);
        put_i32(int32_t(zipped ? clen : n));
        const uint8_t *src = zipped ? packed.data() : raw.data();
        out.insert(out.end(), src, src + (zipped ? size_t(clen) : n));
    }
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f || std::fwrite(out.data(), 1, out.size(), f) != out.size()) {
        if (f) std::fclose(f);
        *err = "Unable to write image file \"" + path + "\"";
        return false;
    }
    std::fclose(f);
    return true;
}

}  // namespace iile
