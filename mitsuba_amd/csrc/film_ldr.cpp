// LDRFilm::develop (src/films/ldrfilm.cpp:300-321) on the host: weight division, tone mapping, gamma, 8-bit quantisation.
// The film has already been downloaded by the caller (2 M pixels at most: a few milliseconds of host work per image).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/mtsamd.h"

namespace mtsamd_host {

namespace {

// FormatConverterImpl::applyGamma (src/libcore/fmtconv.cpp:1104-1111)
inline float applyGamma(float value, float invGamma) {
    if (invGamma == -1.0f)
        return value <= 0.0031308f ? 12.92f * value : 1.055f * std::pow(value, (float) (1.0 / 2.4)) - 0.055f;
    return std::pow(value, invGamma);
}

// convertScalar<uint8_t>(float) (fmtconv.cpp:1136-1160): round to nearest, clamp; NaN ends up as 0
inline uint8_t quantise(float value, float multiplier, float invGamma) {
    value *= multiplier;
    if (invGamma != 1.0f) value = applyGamma(value, invGamma);
    const float scaled = std::max(0.0f, value * 255.0f + 0.5f);
    return (uint8_t) std::min(255.0f, scaled);
}

inline float luminance(const float *c) { return c[0] * 0.212671f + c[1] * 0.715160f + c[2] * 0.072169f; }

// tonemapReinhard (src/libcore/bitmap.cpp:1707-1854) on a float image of `channels` per pixel, colour or monochrome
void reinhard(float *data, size_t pixels, int channels, bool colour, float key, float burn) {
    float maxLuminance = 0, logAvgLuminance = 0;
    for (size_t i = 0; i < pixels; ++i) {
        const float *p = data + i * channels;
        const float lum = colour ? luminance(p) : p[0];
        if (lum == 1024) maxLuminance = 0.0f;       // bitmap.cpp:1745-1746
        maxLuminance = std::max(maxLuminance, lum);
        logAvgLuminance += std::log(1e-3f + lum);
    }
    logAvgLuminance = std::exp(logAvgLuminance / pixels);
    if (maxLuminance == 0) return;      // black image
    burn = std::min(1.0f, std::max(1e-8f, 1 - burn));
    const float scale = key / logAvgLuminance, Lwhite = maxLuminance * scale;
    const float invWp2 = 1 / (Lwhite * Lwhite * std::pow(burn, 4.0f));
    for (size_t i = 0; i < pixels; ++i) {
        float *p = data + i * channels;
        if (colour) {
            float X = p[0] * 0.412453f + p[1] * 0.357580f + p[2] * 0.180423f;
            float Y = p[0] * 0.212671f + p[1] * 0.715160f + p[2] * 0.072169f;
            float Z = p[0] * 0.019334f + p[1] * 0.119193f + p[2] * 0.950227f;
            const float normalization = 1 / (X + Y + Z), x = X * normalization, y = Y * normalization, Lp = Y * scale;
            Y = Lp * (1.0f + Lp * invWp2) / (1.0f + Lp);
            const float ratio = Y / y;
            X = ratio * x;
            Z = ratio * (1.0f - x - y);
            p[0] = 3.240479f * X + -1.537150f * Y + -0.498535f * Z;
            p[1] = -0.969256f * X + 1.875991f * Y + 0.041556f * Z;
            p[2] = 0.055648f * X + -0.204043f * Y + 1.057311f * Z;
        } else {
            const float Lp = p[0] * scale;
            p[0] = Lp * (1.0f + Lp * invWp2) / (1.0f + Lp);
        }
    }
}

}  // namespace

const char *developLdr(const float *film, size_t pixels, const mtsamd_ldr_opts &o, uint8_t *out) {
    if (o.pixel_format < MTSAMD_PIXEL_LUMINANCE || o.pixel_format > MTSAMD_PIXEL_RGBA) return "unknown pixel format";
    if (o.tonemap != MTSAMD_TONEMAP_GAMMA && o.tonemap != MTSAMD_TONEMAP_REINHARD) return "the tone mapping method must be gamma or reinhard";
    const bool colour = o.pixel_format >= MTSAMD_PIXEL_RGB;
    const bool alpha = o.pixel_format == MTSAMD_PIXEL_LUMINANCE_ALPHA || o.pixel_format == MTSAMD_PIXEL_RGBA;
    const int channels = (colour ? 3 : 1) + (alpha ? 1 : 0);
    // Bitmap::convert ESpectrumAlphaWeight -> [luminance | RGB][alpha], float (fmtconv.cpp:955-1005)
    std::vector<float> img(pixels * (size_t) channels);
    for (size_t i = 0; i < pixels; ++i) {
        const float *s = film + 5 * i;
        const float w = s[4], inv = w != 0 ? 1 / w : w;
        float *d = &img[i * channels];
        if (colour) { d[0] = s[0] * inv; d[1] = s[1] * inv; d[2] = s[2] * inv; }
        else d[0] = luminance(s) * inv;
        if (alpha) d[channels - 1] = s[3] * inv;
    }
    float multiplier = 1.0f;
    if (o.tonemap == MTSAMD_TONEMAP_REINHARD) reinhard(img.data(), pixels, channels, colour, o.key, o.burn);
    else multiplier = std::pow(2.0f, o.exposure);
    const float invGamma = 1.0f / o.gamma;
    const int nColour = colour ? 3 : 1;
    for (size_t i = 0; i < pixels; ++i) {
        const float *s = &img[i * channels];
        uint8_t *d = out + i * channels;
        for (int c = 0; c < nColour; ++c) d[c] = quantise(s[c], multiplier, invGamma);
        if (alpha) d[channels - 1] = quantise(s[channels - 1], 1.0f, 1.0f);
    }
    return "";
}

}  // namespace mtsamd_host
