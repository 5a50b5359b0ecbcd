// CPU-side preparation of a scene description for upload: validation, material configuration
// (what the BSDF plugins' configure() methods derive), sampling tables, camera matrices and the BVH.
#pragma once
#include "../../include/mtsamd.h"
#include "bvh.h"
#include "device_types.h"
#include <string>
#include <vector>

namespace mtsamd {

struct HostScene {
    std::vector<float> positions, normals, texcoords;
    std::vector<uint32_t> indices, triShape;
    std::vector<float> shadeTris;               // packed per-triangle shading records (see DevScene)
    std::vector<DevShape> shapes;
    std::vector<DevMaterial> materials;
    std::vector<DevEmitter> emitters;
    std::vector<float> emitterCdf, areaCdf, rtrans;
    std::vector<DevTexture> textures;           // one per (material, colour) that is textured
    std::vector<uint32_t> materialTextures;     // 4 per material (see DevScene), empty without textures
    std::vector<uint32_t> materialAlphaTextures;    // 2 per material (see DevScene), empty without textured roughness
    std::vector<DevTexture> plainTextures;      // the description's textures on their own, in its order (mtsamd_texture_eval)
    std::vector<DevMipLevel> mipLevels;         // levels of all bitmap textures' pyramids
    std::vector<uint16_t> texels;               // their texels, four halves each
    std::vector<float> envCdfRows, envCdfCols, envRowWeights;   // environment map sampling tables (see DevScene)
    BVH bvh;
    DevScene params;        // scalar members filled; pointers left null until upload
    mtsamd_scene_desc desc; // copy of the scalar parts of the description
    bool hasNormals = false, hasTexcoords = false;
    int32_t envEmitter = -1;    // index of the constant environment emitter in `emitters`, if any
    std::vector<DevDeltaEmitter> deltaEmitters;     // point, spot and directional emitters
    bool mixture = false;       // some material is a mixturebsdf or textured (selects the kernels that know every feature)
    bool blend = false;         // some material is a mixturebsdf (the kernels with the mixture code)
    bool extEmitters = false;   // some emitter is not an area light (selects the kernels that know the other kinds)
};

// Returns an empty string on success.
std::string prepareScene(const mtsamd_scene_desc *desc, HostScene &out);

}  // namespace mtsamd
