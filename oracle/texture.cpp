// TEST INFRASTRUCTURE ONLY -- see texture.h. Restates include/mitsuba/render/mipmap.h (TMIPMap), the Resampler of
// include/mitsuba/core/rfilter.h and Bitmap::resample (src/libcore/bitmap.cpp:2230-2329) in single precision, in the
// reference's operation order.
#include "texture.h"
#include "bsdf.h"
#include <cstring>

namespace orc {

// ---- half (OpenEXR half.h / half.cpp: half::half(float), half::convert: round to nearest even) -----------------------
uint16_t floatToHalf(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    int e = (int) ((x >> 23) & 0xffu) - (127 - 15);
    uint32_t m = x & 0x007fffffu;
    if (e <= 0) {
        if (e < -10) return (uint16_t) sign;            // too small: signed zero
        m |= 0x00800000u;                               // denormalised half: shift the mantissa, round to nearest even
        const int t = 14 - e;
        const uint32_t a = (1u << (t - 1)) - 1u, b = (m >> t) & 1u;
        m = (m + a + b) >> t;
        return (uint16_t) (sign | m);
    } else if (e == 0xff - (127 - 15)) {
        if (m == 0) return (uint16_t) (sign | 0x7c00u); // infinity
        m >>= 13;
        return (uint16_t) (sign | 0x7c00u | m | (m == 0));      // NaN, mantissa kept non-zero
    }
    m = m + 0x00000fffu + ((m >> 13) & 1u);             // normalised: round to nearest even
    if (m & 0x00800000u) { m = 0; e += 1; }
    if (e > 30) return (uint16_t) (sign | 0x7c00u);     // overflow
    return (uint16_t) (sign | ((uint32_t) e << 10) | (m >> 13));
}

float halfToFloat(uint16_t h) {
    const uint32_t sign = ((uint32_t) h & 0x8000u) << 16;
    int e = (h >> 10) & 0x1f;
    uint32_t m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            while (!(m & 0x400u)) { m <<= 1; e -= 1; }
            e += 1; m &= ~0x400u;
            x = sign | ((uint32_t) (e + (127 - 15)) << 23) | (m << 13);
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((uint32_t) (e + (127 - 15)) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

static int modulo(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }
static int floorToInt(float v) { return (int) std::floor(v); }
static int ceilToInt(float v) { return (int) std::ceil(v); }

// ---- Resampler<float> (rfilter.h:107-468) ----------------------------------------------------------------------------
namespace {

// LanczosSincFilter::eval with two lobes (src/rfilters/lanczos.cpp:44-56)
float lanczos2(float x) {
    const float radius = 2.0f;
    x = std::abs(x);
    if (x < Epsilon) return 1.0f;
    else if (x > radius) return 0.0f;
    const float x1 = M_PI_F * x, x2 = x1 / radius;
    return (std::sin(x1) * std::sin(x2)) / (x1 * x2);
}

struct Resampler {
    int bc, sourceRes, targetRes, taps, halfTaps;
    std::vector<int> start;
    std::vector<float> weights;
    Resampler(int bc_, int sourceRes_, int targetRes_) : bc(bc_), sourceRes(sourceRes_), targetRes(targetRes_) {
        float filterRadius = 2.0f, scale = 1.0f, invScale = 1.0f;
        if (targetRes < sourceRes) {    // low-pass: the filter widens when downsampling
            scale = (float) sourceRes / (float) targetRes;
            invScale = 1 / scale;
            filterRadius *= scale;
        }
        taps = ceilToInt(filterRadius * 2);
        if (sourceRes == targetRes && (taps % 2) != 1) --taps;
        halfTaps = taps / 2;
        if (sourceRes != targetRes) {   // resampling mode
            start.resize(targetRes);
            weights.resize((size_t) taps * targetRes);
            for (int i = 0; i < targetRes; i++) {
                const float center = (i + 0.5f) / targetRes * sourceRes;
                start[i] = floorToInt(center - filterRadius + 0.5f);
                float sum = 0;
                for (int j = 0; j < taps; j++) {
                    const float pos = start[i] + j + 0.5f - center;
                    const float weight = lanczos2(pos * invScale);
                    weights[(size_t) i * taps + j] = weight;
                    sum += weight;
                }
                const float normalization = 1.0f / sum;
                for (int j = 0; j < taps; j++) weights[(size_t) i * taps + j] = weights[(size_t) i * taps + j] * normalization;
            }
        } else {                        // filtering mode
            weights.resize(taps);
            float sum = 0;
            for (int i = 0; i < taps; i++) { weights[i] = lanczos2((float) (i - halfTaps)); sum += weights[i]; }
            const float normalization = 1.0f / sum;
            for (int i = 0; i < taps; i++) weights[i] = weights[i] * normalization;
        }
    }
    float lookup(const float *source, int pos, size_t stride, int offset) const {
        if (pos < 0 || pos >= sourceRes) {
            switch (bc) {
                case BC_CLAMP: pos = std::min(sourceRes - 1, std::max(0, pos)); break;
                case BC_REPEAT: pos = modulo(pos, sourceRes); break;
                case BC_MIRROR: pos = modulo(pos, 2 * sourceRes); if (pos >= sourceRes) pos = 2 * sourceRes - pos - 1; break;
                case BC_ZERO: return 0.0f;
                case BC_ONE: return 1.0f;
            }
        }
        return source[stride * pos + offset];
    }
    // resampleAndClamp / resample: the border and the main loops of the reference compute the same sums (the border ones
    // through lookup()), so one loop with lookup() everywhere gives the same values
    void run(const float *source, size_t sourceStride, float *target, size_t targetStride, int channels, bool clamp, float lo, float hi) const {
        sourceStride *= channels;
        for (int i = 0; i < targetRes; ++i) {
            const int first = start.empty() ? i - halfTaps : start[i];
            const float *w = start.empty() ? weights.data() : &weights[(size_t) i * taps];
            for (int ch = 0; ch < channels; ++ch) {
                float result = 0;
                for (int j = 0; j < taps; ++j) result += lookup(source, first + j, sourceStride, ch) * w[j];
                target[(size_t) i * targetStride * channels + ch] = clamp ? std::min(hi, std::max(lo, result)) : result;
            }
        }
    }
};

}  // namespace

void resampleLanczos2(const std::vector<float> &src, int sw, int sh, int ch, int bcu, int bcv, std::vector<float> &dst, int dw, int dh,
                      float minValue, float maxValue) {
    const bool clamp = minValue != -std::numeric_limits<float>::infinity() || maxValue != std::numeric_limits<float>::infinity();
    dst.assign((size_t) dw * dh * ch, 0.0f);
    if (sw == dw && sh == dh) { dst = src; return; }
    const std::vector<float> *source = &src;
    std::vector<float> temp;
    int curW = sw;
    if (sw != dw) {         // along x, row by row
        Resampler r(bcu, sw, dw);
        std::vector<float> &out = sh == dh ? dst : temp;
        out.assign((size_t) dw * sh * ch, 0.0f);
        for (int y = 0; y < sh; ++y) r.run(&src[(size_t) y * sw * ch], 1, &out[(size_t) y * dw * ch], 1, ch, clamp, minValue, maxValue);
        source = &out; curW = dw;
    }
    if (sh != dh) {         // along y, column by column
        Resampler r(bcv, sh, dh);
        for (int x = 0; x < curW; ++x) r.run(&(*source)[(size_t) x * ch], curW, &dst[(size_t) x * ch], dw, ch, clamp, minValue, maxValue);
    }
}

// ---- TMIPMap (mipmap.h:155-303) --------------------------------------------------------------------------------------
MIPMap::MIPMap(const float *pixels, int w, int h, int channels, int bcu_, int bcv_, int filterType_, float maxAnisotropy_, float maxValue)
    : bcu(bcu_), bcv(bcv_), filterType(filterType_), maxAnisotropy(maxAnisotropy_) {
    std::vector<float> bitmap((size_t) w * h * 3);
    for (size_t i = 0; i < (size_t) w * h; ++i)
        for (int c = 0; c < 3; ++c) bitmap[i * 3 + c] = pixels[i * channels + (channels == 3 ? c : 0)];
    auto store = [&](const std::vector<float> &data, int lw, int lh, bool stats) {
        Level L; L.w = lw; L.h = lh; L.texels.resize((size_t) lw * lh * 3);
        float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY }, avg[3] = { 0, 0, 0 };
        for (size_t i = 0; i < (size_t) lw * lh; ++i)
            for (int c = 0; c < 3; ++c) {
                const float v = data[i * 3 + c];
                mn[c] = std::min(mn[c], v); mx[c] = std::max(mx[c], v); avg[c] += v;
                L.texels[i * 3 + c] = floatToHalf(v);
            }
        if (stats) {        // LinearArray::init with statistics (barray.h:308-334)
            minimum = Spectrum(mn[0], mn[1], mn[2]); maximum = Spectrum(mx[0], mx[1], mx[2]);
            average = Spectrum(avg[0], avg[1], avg[2]) / (float) (lw * lh);
        }
        L.sizeRatio = Vec2{ (float) lw / (float) w, (float) lh / (float) h };
        if (levels.empty()) L.sizeRatio = Vec2{ 1, 1 };
        levels.push_back(std::move(L));
    };
    store(bitmap, w, h, true);
    if (std::min(std::min(minimum.x, minimum.y), minimum.z) < 0) {      // negative values are clamped (mipmap.h:232-240)
        for (float &v : bitmap) v = std::max(v, 0.0f);
        levels.clear();
        store(bitmap, w, h, true);
    }
    if (filterType != MIP_NEAREST && filterType != MIP_BILINEAR) {
        int sx = w, sy = h;
        while (sx > 1 || sy > 1) {
            const int nx = std::max(1, (sx + 1) / 2), ny = std::max(1, (sy + 1) / 2);
            std::vector<float> next;
            resampleLanczos2(bitmap, sx, sy, 3, bcu, bcv, next, nx, ny, 0.0f, maxValue);
            store(next, nx, ny, false);
            bitmap.swap(next);
            sx = nx; sy = ny;
        }
    }
    for (int i = 0; i < 64; ++i) {      // mipmap.h:296-302
        const float r2 = (float) i / (float) (64 - 1);
        weightLut[i] = fastexp(-2.0f * r2) - fastexp(-2.0f);
    }
}

Spectrum MIPMap::evalTexel(int level, int x, int y) const {      // mipmap.h:503-563
    const Level &L = levels[level];
    if (x < 0 || x >= L.w) {
        switch (bcu) {
            case BC_REPEAT: x = modulo(x, L.w); break;
            case BC_CLAMP: x = std::min(L.w - 1, std::max(0, x)); break;
            case BC_MIRROR: x = modulo(x, 2 * L.w); if (x >= L.w) x = 2 * L.w - x - 1; break;
            case BC_ZERO: return Spectrum(0.0f);
            case BC_ONE: return Spectrum(1.0f);
        }
    }
    if (y < 0 || y >= L.h) {
        switch (bcv) {
            case BC_REPEAT: y = modulo(y, L.h); break;
            case BC_CLAMP: y = std::min(L.h - 1, std::max(0, y)); break;
            case BC_MIRROR: y = modulo(y, 2 * L.h); if (y >= L.h) y = 2 * L.h - y - 1; break;
            case BC_ZERO: return Spectrum(0.0f);
            case BC_ONE: return Spectrum(1.0f);
        }
    }
    const uint16_t *t = &L.texels[((size_t) y * L.w + x) * 3];
    return Spectrum(halfToFloat(t[0]), halfToFloat(t[1]), halfToFloat(t[2]));
}

Spectrum MIPMap::evalBox(int level, const Vec2 &uv) const {      // mipmap.h:566-569
    const Level &L = levels[level];
    return evalTexel(level, floorToInt(uv.x * L.w), floorToInt(uv.y * L.h));
}

Spectrum MIPMap::evalBilinear(int level, const Vec2 &uv) const { // mipmap.h:575-596
    if (!std::isfinite(uv.x) || !std::isfinite(uv.y)) return Spectrum(0.0f);
    else if (level >= (int) levels.size()) return evalBox((int) levels.size() - 1, uv);
    const Level &L = levels[level];
    const float u = uv.x * L.w - 0.5f, v = uv.y * L.h - 0.5f;
    const int xPos = floorToInt(u), yPos = floorToInt(v);
    const float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
    return evalTexel(level, xPos, yPos) * dx2 * dy2 + evalTexel(level, xPos, yPos + 1) * dx2 * dy1
         + evalTexel(level, xPos + 1, yPos) * dx1 * dy2 + evalTexel(level, xPos + 1, yPos + 1) * dx1 * dy1;
}

void MIPMap::evalGradientBilinear(int level, const Vec2 &uv, Spectrum *gradient) const {     // mipmap.h:601-626
    if (!std::isfinite(uv.x) || !std::isfinite(uv.y)) { gradient[0] = gradient[1] = Spectrum(0.0f); return; }
    else if (level >= (int) levels.size()) { evalGradientBilinear((int) levels.size() - 1, uv, gradient); return; }
    const Level &L = levels[level];
    const float u = uv.x * L.w - 0.5f, v = uv.y * L.h - 0.5f;
    const int xPos = floorToInt(u), yPos = floorToInt(v);
    const float dx = u - xPos, dy = v - yPos;
    const Spectrum p00 = evalTexel(level, xPos, yPos), p10 = evalTexel(level, xPos + 1, yPos),
                   p01 = evalTexel(level, xPos, yPos + 1), p11 = evalTexel(level, xPos + 1, yPos + 1);
    const Spectrum tmp = p01 + p10 - p11;
    gradient[0] = (p10 + p00 * (dy - 1) - tmp * dy) * (float) L.w;
    gradient[1] = (p01 + p00 * (dx - 1) - tmp * dx) * (float) L.h;
}

static float log2Ref(float value) {     // math::log2 (src/libcore/math.cpp:103-106)
    const float invLn2 = 1.0f / std::log(2.0f);
    return fastlog(value) * invLn2;
}

Spectrum MIPMap::eval(const Vec2 &uv, const Vec2 &d0, const Vec2 &d1) const {     // mipmap.h:629-715
    if (filterType == MIP_NEAREST) return evalBox(0, uv);
    else if (filterType == MIP_BILINEAR) return evalBilinear(0, uv);
    const Level &L0 = levels[0];
    const float du0 = d0.x * L0.w, dv0 = d0.y * L0.h, du1 = d1.x * L0.w, dv1 = d1.y * L0.h;
    float A = dv0 * dv0 + dv1 * dv1, B = -2.0f * (du0 * dv0 + du1 * dv1), C = du0 * du0 + du1 * du1, F = A * C - B * B * 0.25f;
    const float root = hypot2(A - C, B), Aprime = 0.5f * (A + C - root), Cprime = 0.5f * (A + C + root);
    float majorRadius = Aprime != 0 ? std::sqrt(F / Aprime) : 0, minorRadius = Cprime != 0 ? std::sqrt(F / Cprime) : 0;
    if (filterType == MIP_TRILINEAR || !(minorRadius > 0) || !(majorRadius > 0) || F < 0) {
        const float level = log2Ref(std::max(majorRadius, Epsilon));
        const int ilevel = floorToInt(level);
        if (ilevel < 0) return evalBilinear(0, uv);
        const float a = level - ilevel;
        return evalBilinear(ilevel, uv) * (1.0f - a) + evalBilinear(ilevel + 1, uv) * a;
    } else {
        if (minorRadius * maxAnisotropy < majorRadius) {
            minorRadius = majorRadius / maxAnisotropy;
            const float theta = 0.5f * std::atan(B / (A - C));
            float sinTheta, cosTheta;
            ::sincosf(theta, &sinTheta, &cosTheta);
            const float a2 = majorRadius * majorRadius, b2 = minorRadius * minorRadius, sinTheta2 = sinTheta * sinTheta,
                        cosTheta2 = cosTheta * cosTheta, sin2Theta = 2 * sinTheta * cosTheta;
            A = a2 * cosTheta2 + b2 * sinTheta2;
            B = (a2 - b2) * sin2Theta;
            C = a2 * sinTheta2 + b2 * cosTheta2;
            F = a2 * b2;
        }
        const float scale = 1.0f / F;
        A *= scale; B *= scale; C *= scale;
        const float level = std::max(0.0f, log2Ref(minorRadius));
        const int ilevel = (int) level;
        const float a = level - ilevel;
        if (majorRadius < 1 || !(A > 0 && C > 0)) return evalBilinear(ilevel, uv);
        return evalEWA(ilevel, uv, A, B, C) * (1.0f - a) + evalEWA(ilevel + 1, uv, A, B, C) * a;
    }
}

Spectrum MIPMap::evalEWA(int level, const Vec2 &uv, float A, float B, float C) const {        // mipmap.h:764-834
    if (!std::isfinite(A + B + C + uv.x + uv.y)) return Spectrum(0.0f);
    else if (level >= (int) levels.size()) return evalBox((int) levels.size() - 1, uv);
    const Level &L = levels[level];
    const float u = uv.x * L.w - 0.5f, v = uv.y * L.h - 0.5f;
    const Vec2 &ratio = L.sizeRatio;
    A /= ratio.x * ratio.x; B /= ratio.x * ratio.y; C /= ratio.y * ratio.y;
    const float invDet = 1.0f / (-B * B + 4.0f * A * C), deltaU = 2.0f * std::sqrt(C * invDet), deltaV = 2.0f * std::sqrt(A * invDet);
    const int u0 = ceilToInt(u - deltaU), u1 = floorToInt(u + deltaU), v0 = ceilToInt(v - deltaV), v1 = floorToInt(v + deltaV);
    const float As = A * 64, Bs = B * 64, Cs = C * 64;
    Spectrum result(0.0f);
    float denominator = 0.0f;
    const float ddq = 2 * As, uu0 = (float) u0 - u;
    for (int vt = v0; vt <= v1; ++vt) {
        const float vv = (float) vt - v;
        float q = As * uu0 * uu0 + (Bs * uu0 + Cs * vv) * vv;
        float dq = As * (2 * uu0 + 1) + Bs * vv;
        for (int ut = u0; ut <= u1; ++ut) {
            if (q < 64.0f) {
                const uint32_t qi = (uint32_t) q;
                if (qi < 64) {
                    const float weight = weightLut[(int) q];
                    result += evalTexel(level, ut, vt) * weight;
                    denominator += weight;
                }
            }
            q += dq;
            dq += ddq;
        }
    }
    if (denominator == 0) return evalBilinear(level, uv);
    return result / denominator;
}

}  // namespace orc
