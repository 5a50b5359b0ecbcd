// Image files to linear float pixels for bitmap textures and environment maps: what Bitmap(EAuto, stream) followed by
// Bitmap::convert(ELuminance | ERGB, EFloat32, gamma 1) produces in the reference (src/libcore/bitmap.cpp: readPNG
// :2452-2559, readPFM :3776-3814, readOpenEXR :2707-2967; src/libcore/fmtconv.cpp:1093-1160 for the transfer curve).
// Decoders are written against the file formats themselves; zlib does the inflation.
#include "../../include/mtsamd.h"
#include "host_scene.h"

#include <sys/stat.h>
#include <zlib.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

namespace mtsamd {

void setLastError(const std::string &msg);     // mtsamd.hip
// Largest image the readers accept: what a header claims is checked against this before anything is allocated.
constexpr int64_t kMaxImageSide = 65536, kMaxImagePixels = 1ll << 28;

namespace {

struct Raw {
    int w = 0, h = 0, channels = 0;     // channels as stored: 1 luminance, 2 luminance + alpha, 3 RGB, 4 RGBA
    float gamma = 1.0f;                 // Bitmap::m_gamma: -1 = sRGB curve
    bool integer = false;               // values are n / maxValue of an integer format
    std::vector<float> v;               // integer formats: already divided by the format's maximum, curve still applied
};

bool readFile(const char *path, std::vector<uint8_t> &out) {
    struct stat info;
    if (!path || stat(path, &info) != 0 || !S_ISREG(info.st_mode)) return false;     // a directory opens, and reports a size of 2^63 - 1
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t) n : 0);
    const bool ok = n >= 0 && fread(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}

uint32_t be32(const uint8_t *p) { return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]; }
uint32_t le32(const uint8_t *p) { return ((uint32_t) p[3] << 24) | ((uint32_t) p[2] << 16) | ((uint32_t) p[1] << 8) | p[0]; }

bool inflateAll(const uint8_t *src, size_t n, std::vector<uint8_t> &dst) {
    uLongf len = (uLongf) dst.size();
    return uncompress(dst.data(), &len, src, (uLong) n) == Z_OK && len == dst.size();
}

// ---- PNG ---------------------------------------------------------------------------------------------------------------
std::string decodePng(const std::vector<uint8_t> &f, Raw &img) {
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (f.size() < 8 || memcmp(f.data(), sig, 8)) return "not a PNG file";
    uint32_t width = 0, height = 0;
    int depth = 0, colour = -1, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool srgbChunk = false; double gama = 0;
    size_t p = 8;
    while (p + 12 <= f.size()) {
        const uint32_t len = be32(&f[p]);
        const char *type = (const char *) &f[p + 4];
        if (p + 12 + (size_t) len > f.size()) return "truncated PNG chunk";
        const uint8_t *d = &f[p + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) { width = be32(d); height = be32(d + 4); depth = d[8]; colour = d[9]; interlace = d[12]; }
        else if (!memcmp(type, "PLTE", 4)) palette.assign(d, d + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(d, d + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!memcmp(type, "sRGB", 4)) srgbChunk = true;
        else if (!memcmp(type, "gAMA", 4) && len >= 4) gama = be32(d) / 100000.0;
        else if (!memcmp(type, "IEND", 4)) break;
        p += 12 + (size_t) len;
    }
    if (width == 0 || height == 0 || colour < 0) return "PNG without a header";
    if ((int64_t) width > kMaxImageSide || (int64_t) height > kMaxImageSide || (int64_t) width * (int64_t) height > kMaxImagePixels) return "PNG size out of range";
    if (interlace) return "interlaced PNG files are not supported";
    const int samples = colour == 0 ? 1 : colour == 2 ? 3 : colour == 3 ? 1 : colour == 4 ? 2 : colour == 6 ? 4 : 0;
    if (!samples || (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16)) return "unsupported PNG colour type or bit depth";
    if (colour == 3 && palette.empty()) return "PNG palette missing";
    const size_t bpp = std::max<size_t>(1, (size_t) samples * depth / 8), stride = ((size_t) width * samples * depth + 7) / 8;
    std::vector<uint8_t> raw((stride + 1) * height);
    if (!inflateAll(idat.data(), idat.size(), raw)) return "PNG image data could not be inflated";
    std::vector<uint8_t> cur(stride), prev(stride, 0);
    // colour types as the reference asks libpng to deliver them: palette -> RGB, tRNS -> alpha, grey 2/4 bit -> 8 bit
    const bool pal = colour == 3;
    const bool palAlpha = pal && !trns.empty(), keyAlpha = !pal && !trns.empty() && (colour == 0 || colour == 2);
    img.w = (int) width; img.h = (int) height;
    img.channels = pal ? (palAlpha ? 4 : 3) : samples + (keyAlpha ? 1 : 0);
    img.integer = true;
    img.v.resize((size_t) width * height * img.channels);
    const int outDepth = pal ? 8 : (depth == 2 || depth == 4) ? 8 : depth;     // 1-bit grey stays a bit mask (0 / 1)
    const float maxv = depth == 16 ? 65535.0f : outDepth == 1 ? 1.0f : 255.0f;
    const float recip = 1.0f / maxv;
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t *line = &raw[(stride + 1) * y];
        const int filter = line[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            switch (filter) {
                case 0: break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) / 2; break;
                case 4: { const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
                          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: return "invalid PNG filter type";
            }
            cur[i] = (uint8_t) (line[1 + i] + pred);
        }
        float *o = &img.v[(size_t) y * width * img.channels];
        auto sample = [&](size_t idx) -> uint32_t {        // idx-th sample of the row
            if (depth == 8) return cur[idx];
            if (depth == 16) return ((uint32_t) cur[2 * idx] << 8) | cur[2 * idx + 1];
            const size_t bit = idx * depth;
            return (cur[bit / 8] >> (8 - depth - (bit % 8))) & ((1u << depth) - 1u);
        };
        for (uint32_t x = 0; x < width; ++x) {
            if (pal) {
                const uint32_t k = sample(x);
                if ((size_t) k * 3 + 2 >= palette.size()) return "PNG palette index out of range";
                for (int c = 0; c < 3; ++c) *o++ = (float) palette[3 * k + c] * recip;
                if (palAlpha) *o++ = (float) (k < trns.size() ? trns[k] : 255) * recip;
            } else {
                uint32_t s[4] = { 0, 0, 0, 0 };
                for (int c = 0; c < samples; ++c) s[c] = sample((size_t) x * samples + c);
                for (int c = 0; c < samples; ++c) {
                    uint32_t val = s[c];
                    if (depth == 2) val *= 85u; else if (depth == 4) val *= 17u;    // png_set_expand_gray_1_2_4_to_8
                    *o++ = (float) val * recip;
                }
                if (keyAlpha) {
                    bool match = true;
                    for (int c = 0; c < samples && match; ++c)
                        match = trns.size() >= (size_t) 2 * (c + 1) && s[c] == (((uint32_t) trns[2 * c] << 8) | trns[2 * c + 1]);
                    *o++ = match ? 0.0f : 1.0f;
                }
            }
        }
        prev.swap(cur);
    }
    // bitmap.cpp:2534-2541
    if (srgbChunk) img.gamma = -1.0f;
    else if (gama > 0) img.gamma = 1.0f / (float) gama;
    else img.gamma = -1.0f;
    return "";
}

// ---- PFM (bitmap.cpp:3776-3814) ----------------------------------------------------------------------------------------
std::string decodePfm(const std::vector<uint8_t> &f, Raw &img) {
    size_t p = 0;
    auto token = [&]() {
        while (p < f.size() && isspace(f[p])) ++p;
        std::string t;
        while (p < f.size() && !isspace(f[p])) t.push_back((char) f[p++]);
        return t;
    };
    const std::string magic = token();
    if (magic != "PF" && magic != "Pf") return "not a PFM file";
    img.channels = magic == "PF" ? 3 : 1;
    img.w = atoi(token().c_str()); img.h = atoi(token().c_str());
    const float scaleAndOrder = (float) atof(token().c_str());
    if (img.w <= 0 || img.h <= 0 || scaleAndOrder == 0) return "invalid PFM header";
    if ((int64_t) img.w > kMaxImageSide || (int64_t) img.h > kMaxImageSide || (int64_t) img.w * (int64_t) img.h > kMaxImagePixels) return "PFM size out of range";
    ++p;    // the single white-space byte after the header
    const size_t n = (size_t) img.w * img.h * img.channels;
    if (p > f.size() || n > (f.size() - p) / 4) return "truncated PFM file";
    img.v.resize(n);
    const bool little = scaleAndOrder <= 0.0f;
    const float scale = std::fabs(scaleAndOrder);
    for (int y = 0; y < img.h; ++y)     // the file's rows run bottom to top
        for (size_t i = 0; i < (size_t) img.w * img.channels; ++i) {
            const uint8_t *s = &f[p + (((size_t) (img.h - 1 - y) * img.w * img.channels) + i) * 4];
            const uint32_t bits = little ? le32(s) : be32(s);
            float val;
            memcpy(&val, &bits, 4);
            if (scale != 1) val *= scale;
            img.v[(size_t) y * img.w * img.channels + i] = val;
        }
    img.gamma = 1.0f;
    return "";
}

// ---- OpenEXR, scan-line files without compression or with ZIPS / ZIP / PIZ ------------------------------------------------------------
float halfBitsToFloat(uint16_t h) {
    const uint32_t sign = ((uint32_t) h & 0x8000u) << 16;
    int e = (h >> 10) & 0x1f;
    uint32_t m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { while (!(m & 0x400u)) { m <<= 1; --e; } ++e; m &= ~0x400u; x = sign | ((uint32_t) (e + 112) << 23) | (m << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((uint32_t) (e + 112) << 23) | (m << 13);
    float out;
    memcpy(&out, &x, 4);
    return out;
}

// ---- PIZ: the compression Mitsuba's own OpenEXR files and most others use. A block of 32 scan lines is stored as the
// range of 16-bit values that occur (a bitmap), then the Huffman-coded Haar-like wavelet transform of the values'
// ranks, channel after channel (the OpenEXR file format: ImfPizCompressor, ImfHuf, ImfWav).
struct BitReader {
    const uint8_t *p, *end;
    uint64_t c = 0;
    int lc = 0;
    bool ok = true;
    uint32_t bits(int n) {
        while (lc < n) {
            if (p >= end) { ok = false; c <<= 8; } else c = (c << 8) | *p++;
            lc += 8;
        }
        lc -= n;
        return (uint32_t) ((c >> lc) & ((1ull << n) - 1ull));
    }
};

bool pizHuffman(const uint8_t *src, size_t n, std::vector<uint16_t> &out) {
    constexpr int kEncSize = (1 << 16) + 1, kShortRun = 59, kLongRun = 63, kShortestLongRun = 2 + kLongRun - kShortRun;
    if (n == 0) return out.empty();
    if (n < 20) return false;
    const uint32_t im = le32(src), iM = le32(src + 4), nBits = le32(src + 12);
    if (im >= (uint32_t) kEncSize || iM >= (uint32_t) kEncSize || im > iM) return false;
    // code lengths, six bits each, runs of zeros packed
    std::vector<uint8_t> length(kEncSize, 0);
    BitReader r{ src + 20, src + n };
    for (uint32_t i = im; i <= iM; ++i) {
        const uint32_t l = r.bits(6);
        if (!r.ok) return false;
        if (l == (uint32_t) kLongRun) {
            uint32_t run = r.bits(8) + kShortestLongRun;
            if (i + run > iM + 1) return false;
            i += run - 1;
        } else if (l >= (uint32_t) kShortRun) {
            uint32_t run = l - kShortRun + 2;
            if (i + run > iM + 1) return false;
            i += run - 1;
        } else length[i] = (uint8_t) l;
    }
    const uint8_t *data = r.p;      // the coded values start at the next byte
    if ((uint64_t) nBits > 8ull * (uint64_t) (src + n - data)) return false;
    // canonical codes: within a length in order of the symbol, the longest lengths first
    uint64_t first[60] = { 0 }, count[60] = { 0 };
    for (int i = 0; i < kEncSize; ++i) ++count[length[i]];
    {
        uint64_t c = 0;
        for (int l = 58; l > 0; --l) { const uint64_t nc = (c + count[l]) >> 1; first[l] = c; c = nc; }
    }
    uint32_t offset[60] = { 0 };
    for (int l = 1, acc = 0; l <= 58; ++l) { offset[l] = (uint32_t) acc; acc += (int) count[l]; }
    std::vector<uint32_t> symbols(kEncSize), fill(offset, offset + 60);
    for (int i = 0; i < kEncSize; ++i) if (length[i]) symbols[fill[length[i]]++] = (uint32_t) i;
    // a table for the codes of up to 12 bits; longer ones are walked bit by bit
    constexpr int kFast = 12;
    std::vector<uint32_t> fast(1u << kFast, 0u);    // symbol << 8 | length, 0 = not a short code
    for (int l = 1; l <= kFast; ++l)
        for (uint64_t k = 0; k < count[l]; ++k) {
            const uint64_t code = first[l] + k;
            if (code >> l) return false;
            const uint32_t entry = (symbols[offset[l] + k] << 8) | (uint32_t) l;
            for (uint32_t pad = 0; pad < (1u << (kFast - l)); ++pad) fast[(code << (kFast - l)) | pad] = entry;
        }
    BitReader d{ data, src + n };
    uint64_t used = 0;
    size_t o = 0;
    while (o < out.size()) {
        uint32_t symbol = 0;
        int len = 0;
        // peek
        while (d.lc < kFast) { d.c = (d.c << 8) | (d.p < d.end ? *d.p++ : 0); d.lc += 8; }
        const uint32_t entry = fast[(d.c >> (d.lc - kFast)) & ((1u << kFast) - 1u)];
        if (entry) { len = (int) (entry & 0xffu); symbol = entry >> 8; d.lc -= len; }
        else {
            uint64_t code = 0;
            bool found = false;
            while (len < 58) {
                code = (code << 1) | d.bits(1);
                ++len;
                if (count[len] && code >= first[len] && code - first[len] < count[len]) { symbol = symbols[offset[len] + (code - first[len])]; found = true; break; }
            }
            if (!found) return false;
        }
        used += (uint64_t) len;
        if (used > nBits) return false;
        if (symbol == iM) {         // run-length code: repeat the last value
            const uint32_t run = d.bits(8);
            used += 8;
            if (used > nBits || o == 0 || o + run > out.size()) return false;
            const uint16_t v = out[o - 1];
            for (uint32_t k = 0; k < run; ++k) out[o++] = v;
        } else out[o++] = (uint16_t) symbol;
    }
    return true;
}

inline void wdec14(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    const int ls = (int16_t) l, hi = (int16_t) h;
    const int ai = ls + (hi & 1) + (hi >> 1);
    a = (uint16_t) (int16_t) ai;
    b = (uint16_t) (int16_t) (ai - hi);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xffff;
    const int aa = (d + bb - 0x8000) & 0xffff;
    b = (uint16_t) bb; a = (uint16_t) aa;
}
void pizWaveletDecode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1, p2;
    while (p <= n) p <<= 1;
    p >>= 1; p2 = p; p >>= 1;
    while (p >= 1) {
        uint16_t *py = in;
        uint16_t *const ey = in + (ptrdiff_t) oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t) oy * p, oy2 = (ptrdiff_t) oy * p2, ox1 = (ptrdiff_t) ox * p, ox2 = (ptrdiff_t) ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t *px = py;
            uint16_t *const ex = py + (ptrdiff_t) ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                if (w14) { wdec14(*px, *p10, i00, i10); wdec14(*p01, *p11, i01, i11); wdec14(i00, i01, *px, *p01); wdec14(i10, i11, *p10, *p11); }
                else { wdec16(*px, *p10, i00, i10); wdec16(*p01, *p11, i01, i11); wdec16(i00, i01, *px, *p01); wdec16(i10, i11, *p10, *p11); }
            }
            if (nx & p) {
                uint16_t *p10 = px + oy1;
                if (w14) wdec14(*px, *p10, i00, *p10); else wdec16(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {
            uint16_t *px = py;
            uint16_t *const ex = py + (ptrdiff_t) ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1;
                if (w14) wdec14(*px, *p01, i00, *p01); else wdec16(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p; p >>= 1;
    }
}

// One PIZ block -> the bytes of its scan lines as an uncompressed file stores them. sizes[c] = 16-bit words per sample of
// channel c (1 for half, 2 for float / uint), all channels `w` samples wide.
bool pizDecodeBlock(const uint8_t *src, size_t n, const std::vector<int> &sizes, int w, int lines, std::vector<uint8_t> &out) {
    constexpr int kBitmapSize = 8192;
    if (n < 4) return false;
    const uint32_t minNonZero = src[0] | (src[1] << 8), maxNonZero = src[2] | (src[3] << 8);
    if (maxNonZero >= (uint32_t) kBitmapSize) return false;
    std::vector<uint8_t> bitmap(kBitmapSize, 0);
    size_t p = 4;
    if (minNonZero <= maxNonZero) {
        const size_t k = maxNonZero - minNonZero + 1;
        if (p + k > n) return false;
        memcpy(&bitmap[minNonZero], src + p, k);
        p += k;
    }
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for (uint32_t i = 0; i < 65536; ++i)
        if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = (uint16_t) i;
    const uint16_t maxValue = (uint16_t) (k - 1);
    if (p + 4 > n) return false;
    const uint32_t length = le32(src + p);
    p += 4;
    if (p + length > n) return false;
    size_t total = 0;
    for (int sz : sizes) total += (size_t) w * lines * sz;
    std::vector<uint16_t> tmp(total);
    if (!pizHuffman(src + p, length, tmp)) return false;
    size_t start = 0;
    std::vector<size_t> starts;
    for (int sz : sizes) {
        starts.push_back(start);
        for (int j = 0; j < sz; ++j) pizWaveletDecode(&tmp[start + j], w, sz, lines, w * sz, maxValue);
        start += (size_t) w * lines * sz;
    }
    for (uint16_t &v : tmp) v = lut[v];
    out.resize(total * 2);
    size_t o = 0;
    std::vector<size_t> cursor = starts;
    for (int y = 0; y < lines; ++y)
        for (size_t c = 0; c < sizes.size(); ++c)
            for (int x = 0; x < w * sizes[c]; ++x) {
                const uint16_t v = tmp[cursor[c]++];
                out[o++] = (uint8_t) (v & 0xffu); out[o++] = (uint8_t) (v >> 8);
            }
    return true;
}

std::string decodeExr(const std::vector<uint8_t> &f, Raw &img) {
    if (f.size() < 8 || le32(&f[0]) != 20000630u) return "not an OpenEXR file";
    const uint32_t version = le32(&f[4]);
    if (version & 0x1a00u) return "tiled, deep and multi-part OpenEXR files are not supported";
    struct Channel { std::string name; int type; };
    std::vector<Channel> channels;
    int compression = -1, x0 = 0, y0 = 0, x1 = -1, y1 = -1;
    size_t p = 8;
    auto cstr = [&](std::string &out) { out.clear(); while (p < f.size() && f[p]) out.push_back((char) f[p++]); ++p; return p <= f.size(); };
    while (p < f.size() && f[p]) {
        std::string name, type;
        if (!cstr(name) || !cstr(type) || p + 4 > f.size()) return "truncated OpenEXR header";
        const uint32_t size = le32(&f[p]); p += 4;
        if (p + size > f.size()) return "truncated OpenEXR header";
        const uint8_t *d = &f[p];
        if (name == "channels") {
            size_t q = 0;
            while (q < size && d[q]) {
                Channel c;
                while (q < size && d[q]) c.name.push_back((char) d[q++]);
                ++q;
                if (q + 16 > size) return "malformed OpenEXR channel list";
                c.type = (int) le32(d + q);
                if (le32(d + q + 8) != 1 || le32(d + q + 12) != 1) return "subsampled OpenEXR channels are not supported";
                q += 16;
                channels.push_back(c);
            }
        } else if (name == "compression") { if (size < 1) return "malformed OpenEXR compression attribute"; compression = d[0]; }
        else if (name == "dataWindow" && size >= 16) { x0 = (int) le32(d); y0 = (int) le32(d + 4); x1 = (int) le32(d + 8); y1 = (int) le32(d + 12); }
        p += size;
    }
    ++p;
    if (channels.empty() || x1 < x0 || y1 < y0) return "OpenEXR header without channels or data window";
    if (compression != 0 && compression != 2 && compression != 3 && compression != 4)
        return "OpenEXR compression other than none, ZIPS, ZIP and PIZ is not supported";
    // a damaged header must not turn into a gigantic allocation
    const int64_t w64 = (int64_t) x1 - x0 + 1, h64 = (int64_t) y1 - y0 + 1;
    if (w64 > kMaxImageSide || h64 > kMaxImageSide || w64 * h64 > kMaxImagePixels) return "OpenEXR data window out of range";
    const int w = (int) w64, h = (int) h64, linesPerBlock = compression == 3 ? 16 : compression == 4 ? 32 : 1;
    if (p + (size_t) ((h + linesPerBlock - 1) / linesPerBlock) * 8 > f.size()) return "truncated OpenEXR offset table";
    auto find = [&](const char *n) { for (size_t i = 0; i < channels.size(); ++i) if (channels[i].name == n) return (int) i; return -1; };
    int src[3] = { find("R"), find("G"), find("B") };
    img.channels = 3;
    if (src[0] < 0 || src[1] < 0 || src[2] < 0) {
        src[0] = find("Y");
        if (src[0] < 0) return "OpenEXR file has neither R, G, B nor Y channels";
        img.channels = 1;
    }
    std::vector<size_t> chanOffset(channels.size());    // byte offset of a channel inside one scan line of a block
    size_t lineBytes = 0;
    for (size_t i = 0; i < channels.size(); ++i) {
        if (channels[i].type < 0 || channels[i].type > 2) return "unknown OpenEXR pixel type";
        chanOffset[i] = lineBytes;
        lineBytes += (size_t) w * (channels[i].type == 1 ? 2 : 4);
    }
    img.w = w; img.h = h; img.gamma = 1.0f;
    img.v.assign((size_t) w * h * img.channels, 0.0f);
    const int blocks = (h + linesPerBlock - 1) / linesPerBlock;
    if (p + (size_t) blocks * 8 > f.size()) return "truncated OpenEXR offset table";
    std::vector<uint8_t> buf, tmp;
    for (int b = 0; b < blocks; ++b) {
        const uint64_t off = (uint64_t) le32(&f[p + 8 * b]) | ((uint64_t) le32(&f[p + 8 * b + 4]) << 32);
        if (off + 8 > f.size()) return "OpenEXR block offset out of range";
        const int by = (int) le32(&f[off]) - y0;
        const uint32_t dataSize = le32(&f[off + 4]);
        if (by < 0 || by >= h || off + 8 + dataSize > f.size()) return "malformed OpenEXR block";
        const int lines = std::min(linesPerBlock, h - by);
        const size_t rawSize = lineBytes * lines;
        buf.resize(rawSize);
        if (compression == 0 || dataSize == rawSize) memcpy(buf.data(), &f[off + 8], std::min<size_t>(dataSize, rawSize));
        else if (compression == 4) {
            std::vector<int> sizes;
            for (const Channel &c : channels) sizes.push_back(c.type == 1 ? 1 : 2);
            if (!pizDecodeBlock(&f[off + 8], dataSize, sizes, w, lines, buf) || buf.size() != rawSize) return "OpenEXR PIZ block could not be decoded";
        } else {
            tmp.resize(rawSize);
            if (!inflateAll(&f[off + 8], dataSize, tmp)) return "OpenEXR block could not be inflated";
            for (size_t i = 1; i < rawSize; ++i) tmp[i] = (uint8_t) (tmp[i - 1] + tmp[i] - 128);    // predictor
            const size_t half = (rawSize + 1) / 2;                                                  // de-interleave
            for (size_t i = 0; i < rawSize; ++i) buf[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
        }
        for (int l = 0; l < lines; ++l)
            for (int c = 0; c < img.channels; ++c) {
                const Channel &ch = channels[src[c]];
                const uint8_t *s = &buf[(size_t) l * lineBytes + chanOffset[src[c]]];
                float *o = &img.v[((size_t) (by + l) * w) * img.channels + c];
                for (int x = 0; x < w; ++x, o += img.channels) {
                    if (ch.type == 1) *o = halfBitsToFloat((uint16_t) (s[2 * x] | (s[2 * x + 1] << 8)));
                    else if (ch.type == 2) { const uint32_t bits = le32(s + 4 * x); memcpy(o, &bits, 4); }
                    else *o = (float) le32(s + 4 * x);
                }
            }
    }
    return "";
}

// FormatConverter::undoGamma (fmtconv.cpp:1093-1102)
float undoGamma(float value, float gamma) {
    if (gamma == -1) {
        if (value <= 0.04045f) return value * (float) (1.0 / 12.92);
        return std::pow((float) ((value + 0.055f) * (float) (1.0 / 1.055)), 2.4f);
    }
    return std::pow(value, gamma);
}

}  // namespace

std::string loadImage(const char *path, float gammaOverride, int &w, int &h, int &channels, std::vector<float> &pixels) {
    std::vector<uint8_t> file;
    if (!path || !readFile(path, file)) return std::string("could not read the image file \"") + (path ? path : "") + "\"";
    Raw img;
    std::string err;
    if (file.size() >= 8 && file[0] == 0x89 && file[1] == 'P') err = decodePng(file, img);
    else if (file.size() >= 2 && file[0] == 'P' && (file[1] == 'F' || file[1] == 'f')) err = decodePfm(file, img);
    else if (file.size() >= 4 && le32(&file[0]) == 20000630u) err = decodeExr(file, img);
    else if (file.size() >= 2 && file[0] == 0xff && file[1] == 0xd8) err = "JPEG files are not supported (PNG, PFM and OpenEXR are)";
    else err = "unknown image file format (PNG, PFM and OpenEXR are supported)";
    if (!err.empty()) return std::string(path) + ": " + err;
    const float gamma = gammaOverride != 0 ? gammaOverride : img.gamma;    // Bitmap::setGamma (bitmap.cpp:251-252 of the texture plugin)
    w = img.w; h = img.h;
    channels = img.channels >= 3 ? 3 : 1;
    pixels.resize((size_t) w * h * channels);
    // alpha is dropped (fmtconv.cpp: ERGBA -> ERGB, ELuminanceAlpha -> ELuminance); the curve is undone per value
    for (size_t i = 0; i < (size_t) w * h; ++i)
        for (int c = 0; c < channels; ++c) {
            float v = img.v[i * img.channels + c];
            if (gamma != 1) v = undoGamma(v, gamma);
            pixels[i * channels + c] = v;
        }
    return "";
}

}  // namespace mtsamd

extern "C" int mtsamd_image_load(const char *path, float gamma, mtsamd_image *out) {
    if (!out) { mtsamd::setLastError("mtsamd_image_load: null output record"); return MTSAMD_EINVAL; }
    out->width = out->height = out->channels = 0; out->pixels = nullptr;
    std::vector<float> px;
    int w, h, c;
    std::string err;
    try { err = mtsamd::loadImage(path, gamma, w, h, c, px); }
    catch (const std::exception &e) { err = std::string(path ? path : "") + ": " + e.what(); }       // nothing is thrown across the C boundary
    if (!err.empty()) { mtsamd::setLastError(err); return MTSAMD_EINVAL; }
    out->pixels = (float *) malloc(px.size() * sizeof(float));
    if (!out->pixels) { mtsamd::setLastError("mtsamd_image_load: out of memory"); return MTSAMD_EINVAL; }
    memcpy(out->pixels, px.data(), px.size() * sizeof(float));
    out->width = w; out->height = h; out->channels = c;
    return MTSAMD_OK;
}

extern "C" void mtsamd_image_free(mtsamd_image *image) {
    if (image && image->pixels) { free(image->pixels); image->pixels = nullptr; }
}
