// Host side of bitmap textures: the MIP pyramid as it is uploaded (see mipmap.cpp) and the image file reader (image_io.cpp).
#pragma once
#include "../../include/mtsamd.h"
#include "device_types.h"
#include <string>
#include <vector>

namespace mtsamd {

struct MipPyramid {
    std::vector<DevMipLevel> levels;    // offsets in texels, relative to the start of this pyramid
    std::vector<uint16_t> texels;       // four halves per texel (R, G, B, unused), levels one after the other
    float minimum[3], maximum[3], average[3];       // of level 0, before the conversion to half (mipmap.h:224-240)
};

// maxValue: upper clamp of the resampled levels (1 for BitmapTexture, mipmap.h:165,261). Returns "" on success.
std::string buildMipPyramid(const mtsamd_texture &t, float maxValue, MipPyramid &out);
void ewaWeightTable(float lut[64]);
uint16_t toHalf(float f);
float fromHalf(uint16_t h);

std::string loadImage(const char *path, float gammaOverride, int &w, int &h, int &channels, std::vector<float> &pixels);

}  // namespace mtsamd
