// TEST INFRASTRUCTURE ONLY -- CPU oracle: microfacet distribution, Fresnel terms, warps and the
// BSDF models on the `path` hot path (plus smooth conductor / plastic and the twosided adapter), restated from the reference files cited per function.
#pragma once
#include "scene.h"

namespace orc {

// ---- src/libcore/warp.cpp ------------------------------------------------------------------
Vec2 squareToUniformDiskConcentric(const Vec2 &sample);   // warp.cpp:89-110
Vec3 squareToCosineHemisphere(const Vec2 &sample);        // warp.cpp:51-60
inline float squareToCosineHemispherePdf(const Vec3 &d) { return INV_PI_F * Frame::cosTheta(d); }  // warp.h:56
Vec2 squareToUniformTriangle(const Vec2 &sample);         // warp.cpp:84-87
Vec3 squareToUniformSphere(const Vec2 &sample);           // warp.cpp:25-31

// ---- src/libcore/math.cpp:25-86 --------------------------------------------------------------
float erfinv(float x);
float erf(float x);
float hypot2(float a, float b);

// ---- src/libcore/util.cpp:651-681, 739-761 ---------------------------------------------------
float fresnelDielectricExt(float cosThetaI, float &cosThetaT, float eta);
inline float fresnelDielectricExt(float cosThetaI, float eta) { float c; return fresnelDielectricExt(cosThetaI, c, eta); }
Spectrum fresnelConductorExact(float cosThetaI, const Spectrum &eta, const Spectrum &k);

// ---- src/libcore/spline.cpp:23-60 ------------------------------------------------------------
float evalCubicInterp1D(float x, const float *values, size_t size, float min, float max);
float evalCubicInterp2D(float px, float py, const float *values, size_t sx, size_t sy);
float evalCubicInterp3D(float px, float py, float pz, const float *values, size_t sx, size_t sy, size_t sz);

// ---- src/bsdfs/microfacet.h ------------------------------------------------------------------
struct MicrofacetDistribution {
    int type;          // ORC_DISTR_*
    float alphaU, alphaV;
    bool sampleVisible;
    MicrofacetDistribution(int type_, float au, float av, bool sv)
        : type(type_), alphaU(std::max(au, 1e-4f)), alphaV(std::max(av, 1e-4f)), sampleVisible(sv) {}
    bool isIsotropic() const { return alphaU == alphaV; }
    void scaleAlpha(float value) { alphaU *= value; alphaV *= value; }   // :178-183
    float getAlpha() const { return alphaU; }
    float eval(const Vec3 &m) const;                                    // :191-234
    Vec3 sample(const Vec3 &wi, const Vec2 &sample, float &pdf) const;  // :240-249
    Vec3 sample(const Vec3 &wi, const Vec2 &sample) const;              // :255-264
    float pdf(const Vec3 &wi, const Vec3 &m) const;                     // :270-275
    Vec3 sampleAll(const Vec2 &sample, float &pdf) const;               // :287-395
    float pdfAll(const Vec3 &m) const { return eval(m) * Frame::cosTheta(m); }
    Vec3 sampleVisible_(const Vec3 &wi, const Vec2 &sample) const;      // :421-459
    float pdfVisible(const Vec3 &wi, const Vec3 &m) const;              // :462-466
    float smithG1(const Vec3 &v, const Vec3 &m) const;                  // :477-514
    float G(const Vec3 &wi, const Vec3 &wo, const Vec3 &m) const { return smithG1(wi, m) * smithG1(wo, m); }
    float projectRoughness(const Vec3 &v) const;                        // :541-551
    Vec2 sampleVisible11(float thetaI, Vec2 sample) const;              // :573-697
};

struct BSDFSample {
    Vec3 wo;
    float eta = 1.0f;
    unsigned sampledType = 0;
};

// eval returns f * cos(theta_o); measure is solid angle for all smooth lobes
Spectrum bsdfEval(const Material &m, const Vec3 &wi, const Vec3 &wo);
float bsdfPdf(const Material &m, const Vec3 &wi, const Vec3 &wo);
// `sampler1D` stands for bRec.sampler->next1D(): BSDFs flagged EUsesSampler (roughdielectric) draw one more number
// from the sample stream, after the 2-D sample; it is called at most once and only by them.
struct Sampler1D { virtual float next1D() = 0; virtual ~Sampler1D() {} };
Spectrum bsdfSample(const Material &m, const Vec3 &wi, const Vec2 &sample, BSDFSample &out, float &pdf, Sampler1D *sampler1D = nullptr);

// TwoSidedBRDF::configure over the whole material list; returns an error message or nullptr
const char *linkMixtures(std::vector<Material> &materials);
const char *linkTwoSided(std::vector<Material> &materials);

}  // namespace orc
