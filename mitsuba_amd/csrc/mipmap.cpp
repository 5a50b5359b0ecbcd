// MIP pyramid of a bitmap texture, built on the host and uploaded once: what TMIPMap's constructor produces
// (include/mitsuba/render/mipmap.h:155-303). Level 0 is the image itself; every further level halves the previous one
// (rounding up) with Bitmap::resample and the two-lobed Lanczos filter, clamped to [0, maxValue]
// (src/libcore/bitmap.cpp:2230-2329, include/mitsuba/core/rfilter.h:123-198, src/rfilters/lanczos.cpp:44-56); texels are
// stored in half precision (src/textures/bitmap.cpp:169-177), four halves per texel so that one 8-byte load fetches one.
#include "mipmap.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace mtsamd {

// float -> half, round to nearest even (what OpenEXR's half(float) does)
uint16_t toHalf(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u, mant = x & 0x007fffffu;
    const int exp = (int) ((x >> 23) & 0xffu);
    if (exp == 0xff) return (uint16_t) (sign | 0x7c00u | (mant ? (0x200u | (mant >> 13)) : 0u));
    const int e = exp - 112;                    // half exponent
    if (e >= 31) return (uint16_t) (sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t) sign;
        const uint32_t m = mant | 0x00800000u;
        const int shift = 14 - e;
        const uint32_t halfway = 1u << (shift - 1), rest = m & ((1u << shift) - 1u);
        uint32_t r = m >> shift;
        if (rest > halfway || (rest == halfway && (r & 1u))) ++r;
        return (uint16_t) (sign | r);
    }
    uint32_t r = ((uint32_t) e << 10) | (mant >> 13);
    const uint32_t rest = mant & 0x1fffu;
    if (rest > 0x1000u || (rest == 0x1000u && (r & 1u))) ++r;      // may carry into the exponent, up to infinity: as intended
    return (uint16_t) (sign | r);
}

float fromHalf(uint16_t h) {
    const uint32_t sign = ((uint32_t) h & 0x8000u) << 16, mant = h & 0x3ffu;
    const int exp = (h >> 10) & 0x1f;
    uint32_t x;
    if (exp == 0) {
        if (mant == 0) x = sign;
        else {      // denormal: value = mant * 2^-24
            float f = (float) mant * 0x1p-24f;
            memcpy(&x, &f, 4);
            x |= sign;
        }
    } else if (exp == 31) x = sign | 0x7f800000u | (mant << 13);
    else x = sign | ((uint32_t) (exp + 112) << 23) | (mant << 13);
    float out;
    memcpy(&out, &x, 4);
    return out;
}

namespace {

inline int wrapIndex(int pos, int n, int bc, int &constant) {
    constant = -1;
    if (pos >= 0 && pos < n) return pos;
    switch (bc) {
        case MTSAMD_WRAP_CLAMP: return std::min(n - 1, std::max(0, pos));
        case MTSAMD_WRAP_REPEAT: { const int r = pos % n; return r < 0 ? r + n : r; }
        case MTSAMD_WRAP_MIRROR: { int r = pos % (2 * n); if (r < 0) r += 2 * n; return r >= n ? 2 * n - r - 1 : r; }
        case MTSAMD_WRAP_ZERO: constant = 0; return 0;
        default: constant = 1; return 0;
    }
}

float lanczos2(float x) {
    x = std::fabs(x);
    if (x < 1e-4f) return 1.0f;
    if (x > 2.0f) return 0.0f;
    const float x1 = 3.14159265358979323846f * x, x2 = x1 / 2.0f;
    return (std::sin(x1) * std::sin(x2)) / (x1 * x2);
}

// Filter taps of one axis: output sample i is the sum over j < taps of source[first[i] + j] * weight[i * taps + j]
struct Taps {
    int taps = 0;
    std::vector<int> first;
    std::vector<float> weight;
    Taps(int sourceRes, int targetRes) {
        float radius = 2.0f, invScale = 1.0f;
        if (targetRes < sourceRes) {
            const float scale = (float) sourceRes / (float) targetRes;
            invScale = 1 / scale;
            radius *= scale;
        }
        taps = (int) std::ceil(radius * 2);
        first.resize(targetRes);
        weight.resize((size_t) taps * targetRes);
        for (int i = 0; i < targetRes; ++i) {
            const float center = (i + 0.5f) / targetRes * sourceRes;
            first[i] = (int) std::floor(center - radius + 0.5f);
            float sum = 0;
            float *w = &weight[(size_t) i * taps];
            for (int j = 0; j < taps; ++j) {
                const float pos = first[i] + j + 0.5f - center;
                w[j] = lanczos2(pos * invScale);
                sum += w[j];
            }
            const float normalization = 1.0f / sum;
            for (int j = 0; j < taps; ++j) w[j] = w[j] * normalization;
        }
    }
};

// One separable pass along an axis of `count` lines: line l starts at src + l * srcLine / dst + l * dstLine, consecutive
// samples are srcStep / dstStep floats apart, three channels each.
void resampleAxis(const float *src, int sourceRes, size_t srcStep, size_t srcLine, float *dst, int targetRes, size_t dstStep, size_t dstLine,
                  int count, int bc, float hi) {
    const Taps t(sourceRes, targetRes);
    for (int l = 0; l < count; ++l) {
        const float *s = src + (size_t) l * srcLine;
        float *d = dst + (size_t) l * dstLine;
        for (int i = 0; i < targetRes; ++i) {
            const float *w = &t.weight[(size_t) i * t.taps];
            float acc[3] = { 0, 0, 0 };
            for (int j = 0; j < t.taps; ++j) {
                int constant;
                const int pos = wrapIndex(t.first[i] + j, sourceRes, bc, constant);
                for (int c = 0; c < 3; ++c) acc[c] += (constant >= 0 ? (float) constant : s[(size_t) pos * srcStep + c]) * w[j];
            }
            for (int c = 0; c < 3; ++c) d[(size_t) i * dstStep + c] = std::min(hi, std::max(0.0f, acc[c]));
        }
    }
}

}  // namespace

std::string buildMipPyramid(const mtsamd_texture &t, float maxValue, MipPyramid &out) {
    if (!t.pixels || t.width <= 0 || t.height <= 0 || (t.channels != 1 && t.channels != 3)) return "bitmap texture: invalid image";
    if (t.filter_type < MTSAMD_MIP_NEAREST || t.filter_type > MTSAMD_MIP_EWA) return "bitmap texture: unknown filter type";
    if (t.wrap_u < 0 || t.wrap_u > MTSAMD_WRAP_ONE || t.wrap_v < 0 || t.wrap_v > MTSAMD_WRAP_ONE) return "bitmap texture: unknown wrap mode";
    if ((uint64_t) t.width * (uint64_t) t.height > (1ull << 28)) return "bitmap texture: image too large";
    const int w = t.width, h = t.height;
    std::vector<float> cur((size_t) w * h * 3);
    bool negative = false;
    for (size_t i = 0; i < (size_t) w * h; ++i)
        for (int c = 0; c < 3; ++c) {
            const float v = t.pixels[i * t.channels + (t.channels == 3 ? c : 0)];
            negative |= v < 0;
            cur[i * 3 + c] = v;
        }
    if (negative) for (float &v : cur) v = std::max(v, 0.0f);     // mipmap.h:232-240
    // statistics of level 0 in the reference's accumulation order (include/mitsuba/core/barray.h:308-334)
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY }, sum[3] = { 0, 0, 0 };
    for (size_t i = 0; i < (size_t) w * h; ++i)
        for (int c = 0; c < 3; ++c) { const float v = cur[i * 3 + c]; mn[c] = std::min(mn[c], v); mx[c] = std::max(mx[c], v); sum[c] += v; }
    const float recip = 1.0f / (float) (w * h);
    for (int c = 0; c < 3; ++c) { out.minimum[c] = mn[c]; out.maximum[c] = mx[c]; out.average[c] = sum[c] * recip; }

    out.levels.clear(); out.texels.clear();
    auto store = [&](const std::vector<float> &data, int lw, int lh) {
        DevMipLevel L;
        L.offset = (uint32_t) (out.texels.size() / 4);
        L.w = lw; L.h = lh;
        L.ratioX = out.levels.empty() ? 1.0f : (float) lw / (float) w;
        L.ratioY = out.levels.empty() ? 1.0f : (float) lh / (float) h;
        L.pad[0] = L.pad[1] = L.pad[2] = 0;
        out.levels.push_back(L);
        const size_t base = out.texels.size();
        out.texels.resize(base + (size_t) lw * lh * 4);
        for (size_t i = 0; i < (size_t) lw * lh; ++i) {
            for (int c = 0; c < 3; ++c) out.texels[base + i * 4 + c] = toHalf(data[i * 3 + c]);
            out.texels[base + i * 4 + 3] = 0;
        }
    };
    store(cur, w, h);
    if (t.filter_type == MTSAMD_MIP_TRILINEAR || t.filter_type == MTSAMD_MIP_EWA) {
        int sx = w, sy = h;
        std::vector<float> tmp, next;
        while (sx > 1 || sy > 1) {
            const int nx = std::max(1, (sx + 1) / 2), ny = std::max(1, (sy + 1) / 2);
            const float *source = cur.data();
            if (nx != sx) {     // rows first
                std::vector<float> &target = ny == sy ? next : tmp;
                target.resize((size_t) nx * sy * 3);
                resampleAxis(cur.data(), sx, 3, (size_t) sx * 3, target.data(), nx, 3, (size_t) nx * 3, sy, t.wrap_u, maxValue);
                source = target.data();
            }
            if (ny != sy) {     // then columns
                next.resize((size_t) nx * ny * 3);
                resampleAxis(source, sy, (size_t) nx * 3, 3, next.data(), ny, (size_t) nx * 3, 3, nx, t.wrap_v, maxValue);
            }
            store(next, nx, ny);
            cur.swap(next);
            sx = nx; sy = ny;
        }
    }
    return "";
}

void ewaWeightTable(float lut[64]) {    // mipmap.h:296-302, math::fastexp = exp in double rounded to float
    for (int i = 0; i < 64; ++i) {
        const float r2 = (float) i / (float) (64 - 1);
        lut[i] = (float) std::exp((double) (-2.0f * r2)) - (float) std::exp((double) -2.0f);
    }
}

}  // namespace mtsamd
