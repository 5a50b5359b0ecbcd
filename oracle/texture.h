// TEST INFRASTRUCTURE ONLY -- bitmap textures of the CPU oracle: the MIP map of the reference restated
// (include/mitsuba/render/mipmap.h: TMIPMap<Color3, Color3h>), with the resampler that builds its levels
// (include/mitsuba/core/rfilter.h:107-468, src/libcore/bitmap.cpp:2230-2329) and the uv partials a camera ray gives
// an intersection (src/librender/intersection.cpp:5-76).
#pragma once
#include "vec.h"
#include <vector>
#include <cstdint>

namespace orc {

// OpenEXR's half: round to nearest even, overflow to infinity (the type the reference stores texels in, bitmap.cpp:169-177)
uint16_t floatToHalf(float f);
float halfToFloat(uint16_t h);

// ReconstructionFilter::EBoundaryCondition (rfilter.h:53-64), EMIPFilterType (mipmap.h:52-61)
enum { BC_CLAMP = 0, BC_REPEAT = 1, BC_MIRROR = 2, BC_ZERO = 3, BC_ONE = 4 };
enum { MIP_NEAREST = 0, MIP_BILINEAR = 1, MIP_TRILINEAR = 2, MIP_EWA = 3 };

struct MIPMap {
    struct Level { int w, h; std::vector<uint16_t> texels; Vec2 sizeRatio; };     // 3 halves per texel, row-major from the top
    std::vector<Level> levels;
    int bcu = BC_REPEAT, bcv = BC_REPEAT, filterType = MIP_EWA;
    float maxAnisotropy = 20.0f;
    Spectrum minimum, maximum, average;
    float weightLut[64];

    // pixels: w * h * channels linear floats (channels 1 or 3; a luminance image is kept as three equal channels, which
    // gives every channel the values TMIPMap<Color1, Color1h> would hold); maxValue clamps the resampled levels
    MIPMap(const float *pixels, int w, int h, int channels, int bcu, int bcv, int filterType, float maxAnisotropy, float maxValue = 1.0f);

    Spectrum evalTexel(int level, int x, int y) const;
    Spectrum evalBox(int level, const Vec2 &uv) const;
    Spectrum evalBilinear(int level, const Vec2 &uv) const;
    void evalGradientBilinear(int level, const Vec2 &uv, Spectrum *gradient) const;
    Spectrum eval(const Vec2 &uv, const Vec2 &d0, const Vec2 &d1) const;
    Spectrum evalEWA(int level, const Vec2 &uv, float A, float B, float C) const;
};

// Resampler<float> + Bitmap::resample: src (sw x sh x ch) -> dst (dw x dh x ch) with the two-lobed Lanczos filter
void resampleLanczos2(const std::vector<float> &src, int sw, int sh, int ch, int bcu, int bcv, std::vector<float> &dst, int dw, int dh,
                      float minValue, float maxValue);

}  // namespace orc
