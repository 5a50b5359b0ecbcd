// Host-side scene ingest in C++ for the subset of the Mitsuba 0.5 scene format that the `path` hot path needs:
// the XML description (src/librender/scenehandler.cpp:306-693) restricted to scene / integrator / sensor / sampler /
// film / rfilter / bsdf / texture (checkerboard, gridtexture, bitmap, scale) / shape / emitter / ref / transform /
// include / default / alias and the scalar property tags
// with "$parameter" substitution; triangle meshes from `.serialized` containers (src/librender/trimesh.cpp:175-294),
// Wavefront OBJ files (src/shapes/obj.cpp: one mesh per group / material change, materials from the MTL library or from
// named <bsdf> children), Stanford PLY files (src/shapes/ply.cpp) and the
// `cube` plugin (src/shapes/cube.cpp). Everything else is rejected with a message -- nothing is
// silently substituted. The result owns flat arrays and hands out a mtsamd_scene_desc for the C ABI.
#pragma once
#include "../../../include/mtsamd.h"

#include <map>
#include <string>
#include <vector>

namespace mtsamd_host {

struct LoadOptions {
    int width = 0, height = 0;          // override the film size when both are positive
    std::string dataDir;                // directory that holds microfacet/<distribution>.dat (roughplastic)
    int maxDepth = 0;                   // non-zero: override the integrator's maxDepth
    std::map<std::string, std::string> params;   // -D name=value: values for "$name" in the scene file
};

struct LoadedScene {
    std::vector<float> positions, normals, texcoords;
    std::vector<uint32_t> indices;
    std::vector<mtsamd_shape> shapes;
    std::vector<mtsamd_material> materials;
    std::vector<std::vector<float>> rtransTables;   // storage behind mtsamd_material::rtrans
    std::vector<mtsamd_emitter> emitters;
    std::vector<std::vector<float>> emitterPixels;  // storage behind mtsamd_emitter::pixels (environment maps), indexed like `emitters`
    std::vector<mtsamd_texture> textures;           // what the materials' *_texture fields index (1-based)
    std::vector<std::vector<float>> texturePixels;  // storage behind mtsamd_texture::pixels, one entry per texture (empty if not a bitmap)
    std::map<std::string, uint32_t> namedTextures;  // <texture id="..."> declared at the top level of the scene
    std::string sceneDir;                           // where relative file names of the scene file are looked up
    mtsamd_camera camera;
    mtsamd_film film;
    mtsamd_integrator integrator;
    bool anyNormals = false, anyTexcoords = false;
    int sampleCount = 4;
    std::string integratorType;
    std::string filmType;               // "hdrfilm", "ldrfilm", ... as the scene file names it
    int hdrPixelFormat = MTSAMD_PIXEL_RGB;      // HDRFilm parameters (src/films/hdrfilm.cpp:213-260): channels ...
    bool hdrHalf = true;                        // ... and component format of the OpenEXR output (float16 is HDRFilm's default)
    // Render-time annotations of hdrfilm / ldrfilm (src/films/annotations.h:32-155): the film's metadata['key'] properties.
    // kind 0: a string that may refer to $scene['...'], $film['...'], $sampler['...'], $sensor['...'], $integrator['...'];
    // 1: an integer, 2: a float (copied as they are). `elements` holds the properties of those four scene elements as
    // Properties::getAsString prints them (src/libcore/properties.cpp:184-201, 298-309), "type" = the plugin name.
    struct Annotation { std::string key; int kind = 0; std::string text; long i = 0; double f = 0; };
    std::vector<Annotation> annotations;
    std::map<std::string, std::map<std::string, std::string>> elements;
    std::string sourceFile;
    int blockSize = 32;                         // Scene's blockSize property (src/librender/scene.cpp), only reported
    mtsamd_ldr_opts ldr = { MTSAMD_PIXEL_RGB, MTSAMD_TONEMAP_GAMMA, -1.0f, 0.0f, 0.18f, 0.0f };     // LDRFilm parameters (src/films/ldrfilm.cpp:131-183)

    // The returned description points into this object; keep it alive until mtsamd_scene_create has returned.
    mtsamd_scene_desc desc() const;
};

struct Mesh {
    std::string name;
    std::vector<float> positions, normals, texcoords;   // normals / texcoords may be empty
    std::vector<uint32_t> indices;
    bool faceNormals = false;
};

// One mesh of a Wavefront OBJ file with the name of the material that was current for it, and a material of an MTL file
struct ObjPart { Mesh mesh; std::string material; };
struct MtlMaterial {
    float kd[3] = { 0, 0, 0 }, ks[3] = { 0, 0, 0 }, ns = 0;      // sRGB values as written in the file
    int illum = 0;
    std::string mapKd, mapKs, bump;
    bool hasMapKd = false, masked = false;
};

// Each loader returns an empty string on success or an error message.
std::string loadSerialized(const std::string &path, int shapeIndex, Mesh &out);
std::string loadObj(const std::string &path, bool flipTexCoords, Mesh &out);
std::string loadObjParts(const std::string &path, bool flipTexCoords, bool collapse, std::vector<ObjPart> &parts, std::string &materialLibrary);
std::string loadMtl(const std::string &path, std::map<std::string, MtlMaterial> &out);
std::string loadPly(const std::string &path, Mesh &out);
void makeCube(Mesh &out);
void makeRectangle(Mesh &out);
std::string loadSceneXml(const std::string &path, const LoadOptions &opts, LoadedScene &out);

// Metadata of an output image (Bitmap::getMetadata): kind 0 string, 1 integer, 2 float
struct Metadata { std::string key; int kind = 0; std::string s; int32_t i = 0; float f = 0; };
// The scene's annotations with their references resolved (annotations.h:76-141), plus "generatedBy" unless the scene sets it
std::string resolveAnnotations(const LoadedScene &scene, const std::string &destFile, double renderSeconds, int coreCount, std::vector<Metadata> &out);

// Portable float map, the format HDRFilm writes for ".pfm" (src/films/hdrfilm.cpp:226-227): RGB, bottom row first.
std::string writePfm(const std::string &path, const float *rgba, int width, int height);
// OpenEXR (HDRFilm's default container): luminance / luminance-alpha / RGB / RGBA channels (MTSAMD_PIXEL_*), 16- or
// 32-bit float, uncompressed scan lines, top row first.
// Metadata becomes header attributes of type string / int / float (Bitmap::writeOpenEXR, src/libcore/bitmap.cpp:3191-3228).
std::string writeExr(const std::string &path, const float *rgba, int width, int height, int pixelFormat, bool half, const std::vector<Metadata> *metadata = nullptr);
// PNG, LDRFilm's default container (src/films/ldrfilm.cpp:141-142): 8 bits per channel, 1-4 channels, top row first.
// Metadata becomes tEXt chunks (Bitmap::writePNG, src/libcore/bitmap.cpp:2609-2628).
std::string writePng(const std::string &path, const uint8_t *pixels, int width, int height, int channels, const std::vector<Metadata> *metadata = nullptr);

}  // namespace mtsamd_host
