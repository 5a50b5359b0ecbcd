// TEST INFRASTRUCTURE ONLY -- internal scene representation of the CPU oracle.
#pragma once
#include "vec.h"
#include "oracle.h"
#include "texture.h"
#include <memory>
#include <vector>
#include <atomic>
#include <string>

namespace orc {

// include/mitsuba/render/triaccel.h:37-52
struct TriAccel {
    uint32_t k;
    float n_u, n_v, n_d;
    float a_u, a_v, b_nu, b_nv;
    float c_nu, c_nv;
    uint32_t shapeIndex, primIndex;
    int load(const Vec3 &A, const Vec3 &B, const Vec3 &C);
    inline bool rayIntersect(const Ray &ray, float mint, float maxt, float &u, float &v, float &t) const;
};

// 8-byte node like include/mitsuba/render/gkdtree.h:452-601: children are adjacent
// (left = index, right = index+1); leaves hold a [start,end) range into the index list.
struct KDNode {
    uint32_t combined;  // inner: (leftIndex << 2) | axis ; leaf: (primStart << 2) | 3
    union { float split; uint32_t primEnd; };
    bool isLeaf() const { return (combined & 3) == 3; }
    int axis() const { return combined & 3; }
    uint32_t left() const { return combined >> 2; }
    uint32_t primStart() const { return combined >> 2; }
};

// include/mitsuba/core/pmf.h:40-188
struct DiscreteDistribution {
    std::vector<float> cdf;
    float sum = 0, normalization = 0;
    DiscreteDistribution() { cdf.push_back(0.0f); }
    void append(float v) { cdf.push_back(cdf.back() + v); }
    size_t size() const { return cdf.size() - 1; }
    float operator[](size_t i) const { return cdf[i + 1] - cdf[i]; }
    float normalize();
    size_t sample(float v) const;
    size_t sampleReuse(float &v) const;
    size_t sampleReuse(float &v, float &pdf) const;
};

// A colour parameter of a BSDF as the reference holds it: a Texture. ConstantSpectrumTexture (type 0), Checkerboard
// (src/textures/checkerboard.cpp), GridTexture (src/textures/gridtexture.cpp) under Texture2D's uv transform
// (src/librender/texture.cpp:81-121), inside any number of ScaleTextures (src/textures/scale.cpp; innermost first).
// Type 3: BitmapTexture (src/textures/bitmap.cpp) over a MIP map.
struct Texture {
    int type = 0;
    Spectrum color0, color1;
    float lineWidth = 0.01f;
    Vec2 uvOffset{ 0, 0 }, uvScale{ 1, 1 };
    std::vector<float> scales;
    std::shared_ptr<MIPMap> mip;
    // Texture2D::eval(its, filter = true) (texture.cpp:112-121) with the intersection of the query in progress
    Spectrum eval(const Vec2 &uv) const;
    // Texture2D::evalGradient(its, gradient) (texture.cpp:123-141; bitmap.cpp:456-480)
    void evalGradient(const Vec2 &uv, Spectrum *gradient) const;
    Spectrum getMaximum() const;
    Spectrum getAverage() const;
    bool usesRayDifferentials() const { return type == 3; }     // bitmap.cpp:541-543; false for the procedural ones
private:
    Spectrum evalAt(const Vec2 &uv, bool filtered, const Vec2 &d0, const Vec2 &d1) const;    // before the ScaleTextures
};
// bRec.its.uv of the BSDF query in progress on this thread (the reference hands the intersection to the BSDF; here the
// integrator sets it before it calls eval / pdf / sample), and its uv partials if it has any (Intersection::hasUVPartials,
// dudx, dvdx, dudy, dvdy: set by Intersection::computePartials for the first vertex of a camera path)
extern thread_local Vec2 g_textureUV;
struct UVPartials { bool has = false; float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0; };
extern thread_local UVPartials g_texturePartials;
const char *buildTexture(const orc_texture &t, Texture &out);

struct Material {
    orc_material m;
    // derived at configure() time
    Texture diffuseTex, specularTex, transmittanceTex;   // after ensureEnergyConservation
    Spectrum D() const { return diffuseTex.eval(g_textureUV); }
    Spectrum S() const { return specularTex.eval(g_textureUV); }
    Spectrum T() const { return transmittanceTex.eval(g_textureUV); }
    // roughness of roughconductor / roughdielectric: m_alphaU->eval(bRec.its).average() where it is textured
    // (roughconductor.cpp:264-268, roughdielectric.cpp:303-308; Spectrum::average sums, then multiplies by 1 / 3)
    bool alphaUTextured = false, alphaVTextured = false;
    Texture alphaUTex, alphaVTex;
    static float average(const Spectrum &v) { float r = 0; r += v.x; r += v.y; r += v.z; return r * (1.0f / 3.0f); }
    float AU() const { return alphaUTextured ? average(alphaUTex.eval(g_textureUV)) : m.alpha_u; }
    float AV() const { return alphaVTextured ? average(alphaVTex.eval(g_textureUV)) : m.alpha_v; }
    Spectrum condEta, condK;                     // divided by extEta
    float eta = 1, invEta = 1, invEta2 = 1;
    float specularSamplingWeight = 0;
    std::vector<float> rtrans;
    float fdrInt = 0;                            // plastic: fresnelDiffuseReflectance(1/eta) (plastic.cpp:195)
    unsigned own_flags = 0;                      // BSDF::EBSDFType combined over this BRDF's own components
    unsigned type_flags = 0;                     // what the integrator sees: own_flags, or the twosided adapter's
    const Material *back = nullptr;              // twosided: BRDF seen from behind (may be this one)
    // MixtureBSDF (src/bsdfs/mixturebsdf.cpp:115-172): nested BSDFs, weights after the energy rescale, normalised choice
    std::vector<const Material *> mixChildren;
    std::vector<float> mixWeights;
    DiscreteDistribution mixPdf;
    bool usesRayDifferentials = false;           // BSDF::usesRayDifferentials(): some texture of it filters its lookups
    bool twoSidedRayDifferentials = false;       // the same of the twosided adapter around it (either side's)
    // BumpMap / NormalMap adapter around this BSDF (src/bsdfs/bumpmap.cpp, normalmap.cpp): 0 none, 1 bump, 2 normal
    int frameKind = 0;
    Texture frameTex;
    void configure(const std::vector<Texture> &textures);
};

// BSDF::EBSDFType bits (include/mitsuba/render/bsdf.h:230-284)
enum {
    ENull = 0x1, EDiffuseReflection = 0x2, EDiffuseTransmission = 0x4, EGlossyReflection = 0x8,
    EGlossyTransmission = 0x10, EDeltaReflection = 0x20, EDeltaTransmission = 0x40,
    EAnisotropic = 0x1000, ENonSymmetric = 0x4000, EFrontSide = 0x8000, EBackSide = 0x10000,
    EReflection = EDiffuseReflection | EDeltaReflection | EGlossyReflection,
    ETransmission = EDiffuseTransmission | EDeltaTransmission | EGlossyTransmission | ENull,
    EDiffuse = EDiffuseReflection | EDiffuseTransmission,
    EGlossy = EGlossyReflection | EGlossyTransmission,
    ESmooth = EDiffuse | EGlossy,
    EDelta = ENull | EDeltaReflection | EDeltaTransmission
};

struct Intersection {
    float t = std::numeric_limits<float>::infinity();
    float u = 0, v = 0;
    uint32_t shape = 0xFFFFFFFFu, prim = 0xFFFFFFFFu;
    Vec3 p, dpdu, dpdv, wi;
    Vec2 uv;
    Frame geoFrame, shFrame;
    bool isValid() const { return t != std::numeric_limits<float>::infinity(); }
    Vec3 toLocal(const Vec3 &v_) const { return shFrame.toLocal(v_); }
    Vec3 toWorld(const Vec3 &v_) const { return shFrame.toWorld(v_); }
};

// bRec.its of the BSDF query in progress on this thread, for the adapters that perturb the shading frame; null = a
// query without an intersection (the batch entry points): the local frame is then the canonical one
extern thread_local const Intersection *g_bsdfIts;

struct TraversalCounters { uint64_t steps = 0, tests = 0; };

struct Scene {
    orc_scene_desc desc;
    std::vector<Vec3> positions, normals;
    std::vector<Vec2> texcoords;
    std::vector<uint32_t> indices;
    std::vector<orc_shape> shapes;
    std::vector<Material> materials;
    std::vector<Texture> textures;             // the description's textures on their own (orc_texture_eval)
    bool anyRayDifferentials = false;          // some BSDF filters its texture lookups: camera rays carry their offset rays
    std::vector<orc_emitter> emitters;
    std::vector<uint32_t> triShape;            // triangle -> shape
    std::vector<Vec3> uvDpdu, uvDpdv;          // per-triangle UV tangents (meshes with texcoords)
    std::vector<uint8_t> uvDpduValid;
    std::vector<DiscreteDistribution> areaDistr;   // per shape (emitters only)
    std::vector<float> invSurfaceArea;
    DiscreteDistribution emitterPDF;
    // constant environment emitter (src/emitters/constant.cpp): index in `emitters` or -1, and the bounding sphere
    // it is placed on (1.5 x the sphere around the scene bounds and the camera position, constant.cpp:68-72)
    int envEmitter = -1;
    Vec3 envCenter;
    float envRadius = 0;
    // EnvironmentMap (src/emitters/envmap.cpp:99-676) when the environment emitter is of type 4: the MIP map of the image
    // (repeat along u, clamp along v, elliptical filter with anisotropy 10, no upper clamp), the marginal and conditional
    // distributions over luminance times sin(theta) (:260-329) and the emitter's frame
    struct EnvMap {
        std::shared_ptr<MIPMap> mip;
        std::vector<float> cdfRows, cdfCols, rowWeights;
        int w = 0, h = 0;
        float normalization = 0, scale = 1;
        Vec2 pixelSize{ 0, 0 };
        float toWorld[9], toLocal[9];       // linear parts of the emitter-to-world transform and of its inverse
        Vec3 world(const Vec3 &v) const { return Vec3(toWorld[0] * v.x + toWorld[1] * v.y + toWorld[2] * v.z, toWorld[3] * v.x + toWorld[4] * v.y + toWorld[5] * v.z, toWorld[6] * v.x + toWorld[7] * v.y + toWorld[8] * v.z); }
        Vec3 local(const Vec3 &v) const { return Vec3(toLocal[0] * v.x + toLocal[1] * v.y + toLocal[2] * v.z, toLocal[3] * v.x + toLocal[4] * v.y + toLocal[5] * v.z, toLocal[6] * v.x + toLocal[7] * v.y + toLocal[8] * v.z); }
    };
    std::shared_ptr<EnvMap> envMap;
    // Emitters without a surface (point.cpp, spot.cpp, directional.cpp), indexed like `emitters`
    struct DeltaEmitter {
        Vec3 position;              // point, spot: trafo.transformAffine(Point(0))
        Vec3 direction;             // directional: trafo(Vector(0, 0, 1))
        float toLocal[9];           // spot: linear part of trafo.inverse(), row-major
        float cosCutoff = 0, cosBeam = 0, cutoffAngle = 0, invTransitionWidth = 0;    // SpotEmitter::configure (spot.cpp:89-94)
    };
    std::vector<DeltaEmitter> deltaEmitters;
    // sphere a directional emitter starts from: 1.1 x the bounding sphere of the kd-tree bounds (directional.cpp:88-94)
    Vec3 dirCenter;
    float dirRadius = 0;

    // kd-tree
    std::vector<TriAccel> triAccel;
    std::vector<KDNode> nodes;
    std::vector<uint32_t> kdIndices;
    AABB aabb, tightAABB;
    int kdMaxDepth = 0;

    // camera (src/sensors/perspective.cpp:126-180)
    float sampleToCamera[16];
    float camToWorld[16];
    float aspect = 1;
    Vec2 invResolution;

    std::string build(const orc_scene_desc *d);
    const char *buildEnvMap(const orc_emitter &em);
    void buildKDTree();

    // src/librender/skdtree.cpp:112-142 / 207-226
    bool rayIntersect(const Ray &ray, Intersection &its, TraversalCounters *c = nullptr) const;
    bool rayIntersectShadow(const Ray &ray, TraversalCounters *c = nullptr) const;
    bool rayIntersectBrute(const Ray &ray, Intersection &its, bool shadow) const;
    void fillIntersectionRecord(const Ray &ray, Intersection &its) const;
    template <bool shadowRay>
    bool rayIntersectHavran(const Ray &ray, float mint, float maxt, float &t,
                            Intersection *its, TraversalCounters *c) const;
};

AABB triangleClippedAABB(const Vec3 &p0, const Vec3 &p1, const Vec3 &p2, const AABB &box);

}  // namespace orc
