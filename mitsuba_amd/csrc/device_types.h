// Plain-data views of the uploaded scene and of the wavefront queues, shared by the host driver and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mtsamd {

// BSDF::EBSDFType bits (include/mitsuba/render/bsdf.h:230-284)
enum : unsigned {
    ENull = 0x1, EDiffuseReflection = 0x2, EDiffuseTransmission = 0x4, EGlossyReflection = 0x8,
    EGlossyTransmission = 0x10, EDeltaReflection = 0x20, EDeltaTransmission = 0x40,
    EAnisotropic = 0x1000, ENonSymmetric = 0x4000, EFrontSide = 0x8000, EBackSide = 0x10000,
    ETransmission = EDiffuseTransmission | EDeltaTransmission | EGlossyTransmission | ENull,
    ESmooth = EDiffuseReflection | EDiffuseTransmission | EGlossyReflection | EGlossyTransmission,
    EDelta = ENull | EDeltaReflection | EDeltaTransmission
};

struct alignas(16) DevMaterial {
    int32_t type, distribution, sampleVisible, variant;
    uint32_t typeFlags;
    float alphaU, alphaV, eta;
    float invEta, invEta2, specSamplingWeight, fdrInt;      // fdrInt = internal evalDiffuse(alpha)
    float diffuse[3]; uint32_t rtransOffset;
    float specular[3]; uint32_t rtransN;
    float transmittance[3];
    int32_t twoSided;           // nested in a twosided adapter (src/bsdfs/twosided.cpp); typeFlags is then the adapter's
    float condEta[3];
    int32_t backMaterial;       // twosided: record that answers when the surface is seen from behind (may be this one)
    float condK[3];
    uint32_t ownFlags;          // BSDF type bits of this record's own components
};

// Two-colour procedural texture (src/textures/checkerboard.cpp, gridtexture.cpp) with the colours as the BSDF sees
// them (scale textures and the energy-conservation factor applied)
// Bitmap texture (type 3, src/textures/bitmap.cpp): color0 is the factor on the looked-up value (scale textures and
// the energy-conservation factor), color1 holds, as raw bits, the first entry of the texture's levels in
// DevScene::mipLevels, their number, and filter type | wrapU << 8 | wrapV << 16; lineWidth is the anisotropy bound.
// `scale`: a factor still to be applied to the looked-up value (the textures on their own keep their scale texture's
// factor apart, so that a gradient is scaled after the difference is taken, like ScaleTexture::evalGradient); 1 where
// the colours already carry every factor.
struct alignas(16) DevTexture {
    float color0[3]; int32_t type;
    float color1[3]; float lineWidth;
    float uOffset, vOffset, uScale, vScale;
    float scale, pad[3];
};

// One level of a MIP pyramid: where its texels start (in texels of four halves: R, G, B, unused), its size and
// TMIPMap::m_sizeRatio (include/mitsuba/render/mipmap.h:264-266)
struct alignas(16) DevMipLevel {
    uint32_t offset; int32_t w, h; float ratioX;
    float ratioY; uint32_t pad[3];
};

// Where on a surface a BSDF's textures are looked up: Intersection::uv and, when the intersection has them
// (Intersection::hasUVPartials, set for the first vertex of a camera path whose BSDF filters its textures), the
// partials dudx, dvdx, dudy, dvdy (include/mitsuba/render/shape.h:150-165)
struct SurfUV {
    float u, v;
    float dudx, dvdx, dudy, dvdy;
    uint32_t filtered;
};
__host__ __device__ inline SurfUV surfUV(float u, float v) { return SurfUV{ u, v, 0.0f, 0.0f, 0.0f, 0.0f, 0u }; }

struct DevShape {
    uint32_t firstTri, nTris;
    int32_t material, emitter;
    uint32_t flags, pad[3];
};

// Limits of the emitter tables the shading kernel keeps in workgroup memory
static constexpr uint32_t kLdsEmitters = 8, kLdsEmitterTris = 16;

struct alignas(16) DevEmitter {
    float radiance[3];
    float pdfDiscrete;      // samplingWeight * normalization (scene.h:848-850)
    uint32_t shape, firstTri, nTris, cdfOffset;
    float invArea;
    uint32_t kind;          // 0 area light, 1 constant environment, 2 point, 3 spot, 4 directional
    uint32_t delta;         // kinds 2..4: index of the record in DevScene::deltaEmitters
    float pad;
};

// Emitter without a surface (src/emitters/point.cpp, spot.cpp, directional.cpp)
struct alignas(16) DevDeltaEmitter {
    float position[3];      // point, spot; directional: centre of the disk the light starts from (directional.cpp:161-162)
    float cosCutoff;
    float direction[3];     // directional: trafo(Vector(0, 0, 1))
    float cosBeam;
    float toLocalZ[3];      // spot: third row of the inverse transform's linear part (Frame::cosTheta of the local direction)
    float cutoffAngle;
    float invTransitionWidth, pad[3];
};

struct DevScene {
    const float4 *nodes;        // 5 x float4 per wide node, nodeStride float4 apart
    const float4 *tris;         // 3 x float4 per triangle record (BVH order), triStride float4 apart
    uint32_t nodeStride, triStride;
    // Shading record of a triangle, global order, 8 x float4 (one 128-byte line): three vertex positions whose w components carry
    // material | flags << 24, emitter (int, -1 = none) and shape; three vertex normals (zero without normals); the three
    // texture coordinates (uv0, uv1 | uv2, 0, 0; zero for meshes without).
    const float4 *shadeTris;
    const DevShape *shapes;
    const DevMaterial *materials;
    const DevEmitter *emitters;
    const DevDeltaEmitter *deltaEmitters;
    const float *emitterCdf;    // nEmitters + 1
    const float *areaCdf;       // concatenated per-emitter triangle CDFs (nTris + 1 each)
    const float *rtrans;        // concatenated reduced transmittance tables
    // Only when some material has a texture: per material the 1-based texture of its diffuse, specular and
    // transmittance colour (0 = the constant in the record), and the textures. w: bit 0 = the BSDF the shape holds wants
    // uv partials at the first vertex of a path (BSDF::usesRayDifferentials); bits 1-2 = the record sits in a bumpmap (1) or
    // normalmap (2) adapter whose texture is entry (w >> 8) - 1 of plainTextures; bit 3 = the record's roughness is textured
    const uint4 *materialTextures;
    // Only when some roughness is textured: per material the 1-based entries of plainTextures that give alphaU and alphaV
    // (0 = the record's constant)
    const uint2 *materialAlphaTextures;
    const DevTexture *textures;
    const DevTexture *plainTextures;    // the description's textures on their own (scale factor applied), in its order
    // Only with bitmap textures: the levels of all MIP pyramids, their texels, and the 64-entry weight table of the
    // elliptically weighted average (mipmap.h:296-302)
    const DevMipLevel *mipLevels;
    const uint2 *texels;
    const float *ewaLut;
    uint32_t nEmitters, nTris, nShapes, nMaterials;
    // Scenes with a handful of area lights (and no other emitter kind): the shading kernel copies their tables --
    // choice distribution, records, triangle distributions, triangles -- into workgroup memory once per workgroup, so that
    // sampling an emitter costs no chain of dependent global loads. emitterTris = their triangles in all (0 = no copy),
    // areaCdfLen = entries of areaCdf.
    uint32_t emitterTris, areaCdfLen;
    float aabbMin[3], aabbMax[3];       // enlarged scene bounds (gkdtree.h:1213-1220)
    float sampleToCamera[16];
    float camToWorld[12];
    float nearClip, farClip, invResX, invResY;
    int32_t cropW, cropH;
    float filterRadius;
    float filterValues[32];
    int32_t maxDepth, rrDepth, strictNormals, hideEmitters;
    // constant environment emitter (src/emitters/constant.cpp): its index in `emitters` or -1, radiance, the sphere
    // it sits on (1.5 x the bounding sphere of scene and camera) and its share of the emitter choice
    int32_t envEmitter;
    float envRadiance[3], envCenter[3], envRadius, envPdfDiscrete;
    // environment map (src/emitters/envmap.cpp) in place of the constant radiance: its MIP pyramid (entries of mipLevels),
    // the marginal / conditional distributions over luminance * sin(theta) (envmap.cpp:260-329), the emitter's frame
    int32_t envIsMap;
    uint32_t envFirstLevel, envLevels, envW, envH;
    float envNormalization, envScale, envPixelX, envPixelY;
    float envToWorld[9], envToLocal[9];
    const float *envCdfRows;    // envH + 1
    const float *envCdfCols;    // envH rows of envW + 1
    const float *envRowWeights; // envH
    // This record in device memory: what the device functions that are real calls receive instead of the address of the
    // kernel argument (taking that address makes every lane keep a private copy of the whole record).
    const DevScene *self;
};

// Wavefront queue of live paths (structure of arrays, one float4/uint4 stream each)
struct PathQueue {
    float4 *rayO;       // origin.xyz, mint
    float4 *rayD;       // direction.xyz, maxt
    float4 *thr;        // throughput.rgb, eta
    uint4 *misc;        // sample slot, depth | flags << 16, rng state lo, rng state hi
    float *bsdfPdf;     // pdf of the BSDF sample that produced this ray
    // Only with an environment emitter: cosine between the ray and the reference normal of the vertex it left
    // (DirectSamplingRecord::refN, records.inl:160-164), -2 where that normal is zero; what
    // ConstantBackgroundEmitter::pdfDirect needs when the ray escapes
    float *refCos;
};

struct ShadowQueue {
    float4 *rayO, *rayD;
    float4 *contrib;    // rgb, sample slot (bits)
};

enum : unsigned { PF_PREV_DELTA = 1u, PF_REFN_OK = 2u, PF_SCATTERED = 4u };

struct PassParams {
    int32_t x0, y0, w, h;       // pixel rectangle of this pass
    int32_t spp;                // samples per pixel in this pass
    int32_t sppTotal, sampleBase;
    uint32_t seed;
    uint32_t sppMagic, wMagic;  // ceil(2^32 / spp), ceil(2^32 / w) (0 for a divisor of 1): see divideBy
    // Interleaved bands (the share of one GPU in a tile partition): the h rows of the pass are rows
    // y0 + (r / bandRows) * bandPeriod + r % bandRows of the frame, r = 0 .. h-1. bandRows == 0: h consecutive rows.
    int32_t bandRows, bandPeriod;
    uint32_t bandMagic;         // ceil(2^32 / bandRows)
};

// n / d for 32-bit n through one multiply-high: with m = ceil(2^32 / d) the product's high word is the quotient or
// one more, which the remainder test corrects. The GPU has no integer divide instruction; a generic n / d costs ~35.
__host__ __device__ inline uint32_t divisionMagic(uint32_t d) { return d <= 1 ? 0u : (uint32_t) ((0x100000000ull + d - 1) / d); }
__device__ inline uint32_t divideBy(uint32_t n, uint32_t d, uint32_t magic) {
    if (magic == 0) return n;
    uint32_t q = __umulhi(n, magic);
    if (q * d > n) --q;
    return q;
}

// counters[] layout in device memory: four 32-bit queue counters (next-queue length, shadow-queue length, and the
// ray hand-out counters of the closest-hit and the any-hit traversal launches), then 64-bit traversal statistics
// (extend nodes, extend triangles, shadow nodes, shadow triangles) addressed as uint64 index 2..5, and the two ray
// counts of the kernel that finishes a pass (uint64 index 8, 9)
enum { CNT_NEXT = 0, CNT_SHADOW = 1, CNT_EXT_FETCH = 2, CNT_SHD_FETCH = 3, CNT_QUEUE_WORDS = 4,
       CNT64_EXT_NODES = 2, CNT64_EXT_TRIS = 3, CNT64_SHD_NODES = 4, CNT64_SHD_TRIS = 5, CNT64_EXT_EMPTY = 6, CNT64_EXT_STALE = 7,
       CNT64_TAIL_NORMAL = 8, CNT64_TAIL_SHADOW = 9, CNT_WORDS = 20 };

}  // namespace mtsamd
