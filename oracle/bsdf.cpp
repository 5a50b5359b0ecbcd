// TEST INFRASTRUCTURE ONLY -- CPU oracle: BSDF models of the `path` hot path.
#include "bsdf.h"
#include <stdexcept>

namespace orc {

static inline void sincos_(float theta, float *s, float *c) { ::sincosf(theta, s, c); }

// ---- warps (src/libcore/warp.cpp) -------------------------------------------------------------
Vec2 squareToUniformDiskConcentric(const Vec2 &sample) {
    float r1 = 2.0f * sample.x - 1.0f;
    float r2 = 2.0f * sample.y - 1.0f;
    float phi, r;
    if (r1 == 0 && r2 == 0) {
        r = phi = 0;
    } else if (r1 * r1 > r2 * r2) {
        r = r1;
        phi = (M_PI_F / 4.0f) * (r2 / r1);
    } else {
        r = r2;
        phi = (M_PI_F / 2.0f) - (r1 / r2) * (M_PI_F / 4.0f);
    }
    float cosPhi, sinPhi;
    sincos_(phi, &sinPhi, &cosPhi);
    return Vec2{ r * cosPhi, r * sinPhi };
}

Vec3 squareToCosineHemisphere(const Vec2 &sample) {
    Vec2 p = squareToUniformDiskConcentric(sample);
    float z = safe_sqrt(1.0f - p.x * p.x - p.y * p.y);
    if (z == 0) z = 1e-10f;
    return Vec3(p.x, p.y, z);
}

Vec3 squareToUniformSphere(const Vec2 &sample) {
    float z = 1.0f - 2.0f * sample.y;
    float r = safe_sqrt(1.0f - z * z);
    float sinPhi, cosPhi;
    sincos_(2.0f * M_PI_F * sample.x, &sinPhi, &cosPhi);
    return Vec3(r * cosPhi, r * sinPhi, z);
}

Vec2 squareToUniformTriangle(const Vec2 &sample) {
    float a = safe_sqrt(1.0f - sample.x);
    return Vec2{ 1 - a, a * sample.y };
}

// ---- src/libcore/math.cpp ----------------------------------------------------------------------
float erfinv(float x) {
    float w = -fastlog((1.0f - x) * (1.0f + x));
    float p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = 3.43273939e-07f + p * w;
        p = -3.5233877e-06f + p * w;
        p = -4.39150654e-06f + p * w;
        p = 0.00021858087f + p * w;
        p = -0.00125372503f + p * w;
        p = -0.00417768164f + p * w;
        p = 0.246640727f + p * w;
        p = 1.50140941f + p * w;
    } else {
        w = std::sqrt(w) - 3.0f;
        p = -0.000200214257f;
        p = 0.000100950558f + p * w;
        p = 0.00134934322f + p * w;
        p = -0.00367342844f + p * w;
        p = 0.00573950773f + p * w;
        p = -0.0076224613f + p * w;
        p = 0.00943887047f + p * w;
        p = 1.00167406f + p * w;
        p = 2.83297682f + p * w;
    }
    return p * x;
}

float erf(float x) {
    float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f;
    float p = 0.3275911f;
    float sign = signum(x);
    x = std::abs(x);
    float t = 1.0f / (1.0f + p * x);
    float y = 1.0f - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * fastexp(-x * x);
    return sign * y;
}

float hypot2(float a, float b) {
    float r;
    if (std::abs(a) > std::abs(b)) {
        r = b / a;
        r = std::abs(a) * std::sqrt(1.0f + r * r);
    } else if (b != 0.0f) {
        r = a / b;
        r = std::abs(b) * std::sqrt(1.0f + r * r);
    } else {
        r = 0.0f;
    }
    return r;
}

// ---- src/libcore/util.cpp ----------------------------------------------------------------------
float fresnelDielectricExt(float cosThetaI_, float &cosThetaT_, float eta) {
    if (eta == 1) {
        cosThetaT_ = -cosThetaI_;
        return 0.0f;
    }
    float scale = (cosThetaI_ > 0) ? 1 / eta : eta,
          cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) {
        cosThetaT_ = 0.0f;
        return 1.0f;
    }
    float cosThetaI = std::abs(cosThetaI_);
    float cosThetaT = std::sqrt(cosThetaTSqr);
    float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}

static inline Spectrum specSafeSqrt(const Spectrum &s) { return Spectrum(safe_sqrt(s.x), safe_sqrt(s.y), safe_sqrt(s.z)); }

Spectrum fresnelConductorExact(float cosThetaI, const Spectrum &eta, const Spectrum &k) {
    float cosThetaI2 = cosThetaI * cosThetaI, sinThetaI2 = 1 - cosThetaI2, sinThetaI4 = sinThetaI2 * sinThetaI2;
    Spectrum temp1 = eta * eta - k * k - Spectrum(sinThetaI2),
             a2pb2 = specSafeSqrt(temp1 * temp1 + k * k * eta * eta * 4.0f),
             a = specSafeSqrt((a2pb2 + temp1) * 0.5f);
    Spectrum term1 = a2pb2 + Spectrum(cosThetaI2), term2 = a * (2 * cosThetaI);
    Spectrum Rs2 = (term1 - term2) / (term1 + term2);
    Spectrum term3 = a2pb2 * cosThetaI2 + Spectrum(sinThetaI4), term4 = term2 * sinThetaI2;
    Spectrum Rp2 = Rs2 * (term3 - term4) / (term3 + term4);
    return 0.5f * (Rp2 + Rs2);
}

// ---- src/libcore/spline.cpp --------------------------------------------------------------------
float evalCubicInterp1D(float x, const float *values, size_t size, float min, float max) {
    if (!(x >= min && x <= max)) return 0.0f;
    float t = ((x - min) * (size - 1)) / (max - min);
    size_t k = std::max((size_t) 0, std::min((size_t) t, size - 2));
    float f0 = values[k], f1 = values[k + 1], d0, d1;
    if (k > 0) d0 = 0.5f * (values[k + 1] - values[k - 1]);
    else d0 = values[k + 1] - values[k];
    if (k + 2 < size) d1 = 0.5f * (values[k + 2] - values[k]);
    else d1 = values[k + 1] - values[k];
    t = t - (float) k;
    float t2 = t * t, t3 = t2 * t;
    return (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
}

static bool cubicWeights(float p, size_t size, float *weights, size_t &knot) {
    // one dimension of evalCubicInterp2D/3D (spline.cpp:242-287) with min=0, max=1
    if (!(p >= 0.0f && p <= 1.0f)) return false;
    float t = ((p - 0.0f) * (size - 1)) / (1.0f - 0.0f);
    knot = std::min((size_t) t, size - 2);
    t = t - (float) knot;
    float t2 = t * t, t3 = t2 * t;
    weights[0] = 0.0f;
    weights[1] = 2 * t3 - 3 * t2 + 1;
    weights[2] = -2 * t3 + 3 * t2;
    weights[3] = 0.0f;
    float d0 = t3 - 2 * t2 + t, d1 = t3 - t2;
    if (knot > 0) { weights[2] += 0.5f * d0; weights[0] -= 0.5f * d0; }
    else { weights[2] += d0; weights[1] -= d0; }
    if (knot + 2 < size) { weights[3] += 0.5f * d1; weights[1] -= 0.5f * d1; }
    else { weights[2] += d1; weights[1] -= d1; }
    return true;
}

float evalCubicInterp2D(float px, float py, const float *values, size_t sx, size_t sy) {
    float w[2][4];
    size_t knot[2];
    if (!cubicWeights(px, sx, w[0], knot[0])) return 0.0f;
    if (!cubicWeights(py, sy, w[1], knot[1])) return 0.0f;
    float result = 0.0f;
    for (int y = -1; y <= 2; ++y) {
        float wy = w[1][y + 1];
        for (int x = -1; x <= 2; ++x) {
            float wxy = w[0][x + 1] * wy;
            if (wxy == 0) continue;
            size_t pos = (knot[1] + y) * sx + knot[0] + x;
            result += values[pos] * wxy;
        }
    }
    return result;
}

float evalCubicInterp3D(float px, float py, float pz, const float *values, size_t sx, size_t sy, size_t sz) {
    float w[3][4];
    size_t knot[3];
    if (!cubicWeights(px, sx, w[0], knot[0])) return 0.0f;
    if (!cubicWeights(py, sy, w[1], knot[1])) return 0.0f;
    if (!cubicWeights(pz, sz, w[2], knot[2])) return 0.0f;
    float result = 0.0f;
    for (int z = -1; z <= 2; ++z) {
        float wz = w[2][z + 1];
        for (int y = -1; y <= 2; ++y) {
            float wyz = w[1][y + 1] * wz;
            for (int x = -1; x <= 2; ++x) {
                float wxyz = w[0][x + 1] * wyz;
                if (wxyz == 0) continue;
                size_t pos = ((knot[2] + z) * sy + (knot[1] + y)) * sx + knot[0] + x;
                result += values[pos] * wxyz;
            }
        }
    }
    return result;
}

// ---- src/bsdfs/microfacet.h --------------------------------------------------------------------
float MicrofacetDistribution::eval(const Vec3 &m) const {
    if (Frame::cosTheta(m) <= 0) return 0.0f;
    float cosTheta2 = Frame::cosTheta2(m);
    float beckmannExponent = ((m.x * m.x) / (alphaU * alphaU) + (m.y * m.y) / (alphaV * alphaV)) / cosTheta2;
    float result;
    if (type == ORC_DISTR_BECKMANN) {
        result = fastexp(-beckmannExponent) / (M_PI_F * alphaU * alphaV * cosTheta2 * cosTheta2);
    } else {
        float root = (1.0f + beckmannExponent) * cosTheta2;
        result = 1.0f / (M_PI_F * alphaU * alphaV * root * root);
    }
    if (result * Frame::cosTheta(m) < 1e-20f) result = 0;
    return result;
}

Vec3 MicrofacetDistribution::sample(const Vec3 &wi, const Vec2 &s, float &pdf_) const {
    Vec3 m;
    if (sampleVisible) {
        m = sampleVisible_(wi, s);
        pdf_ = pdfVisible(wi, m);
    } else {
        m = sampleAll(s, pdf_);
    }
    return m;
}

Vec3 MicrofacetDistribution::sample(const Vec3 &wi, const Vec2 &s) const {
    if (sampleVisible) return sampleVisible_(wi, s);
    float pdf_;
    return sampleAll(s, pdf_);
}

float MicrofacetDistribution::pdf(const Vec3 &wi, const Vec3 &m) const {
    return sampleVisible ? pdfVisible(wi, m) : pdfAll(m);
}

Vec3 MicrofacetDistribution::sampleAll(const Vec2 &sample, float &pdf_) const {
    float cosThetaM = 0.0f, sinPhiM, cosPhiM, alphaSqr;
    if (isIsotropic()) {
        sincos_((2.0f * M_PI_F) * sample.y, &sinPhiM, &cosPhiM);
        alphaSqr = alphaU * alphaU;
    } else {
        float phiM = std::atan(alphaV / alphaU * std::tan(M_PI_F + 2 * M_PI_F * sample.y))
                     + M_PI_F * std::floor(2 * sample.y + 0.5f);
        sincos_(phiM, &sinPhiM, &cosPhiM);
        float cosSc = cosPhiM / alphaU, sinSc = sinPhiM / alphaV;
        alphaSqr = 1.0f / (cosSc * cosSc + sinSc * sinSc);
    }
    if (type == ORC_DISTR_BECKMANN) {
        float tanThetaMSqr = alphaSqr * -fastlog(1.0f - sample.x);
        cosThetaM = 1.0f / std::sqrt(1.0f + tanThetaMSqr);
        pdf_ = (1.0f - sample.x) / (M_PI_F * alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM);
    } else {
        float tanThetaMSqr = alphaSqr * sample.x / (1.0f - sample.x);
        cosThetaM = 1.0f / std::sqrt(1.0f + tanThetaMSqr);
        float temp = 1 + tanThetaMSqr / alphaSqr;
        pdf_ = INV_PI_F / (alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM * temp * temp);
    }
    if (pdf_ < 1e-20f) pdf_ = 0;
    float sinThetaM = std::sqrt(std::max(0.0f, 1 - cosThetaM * cosThetaM));
    return Vec3(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
}

Vec3 MicrofacetDistribution::sampleVisible_(const Vec3 &_wi, const Vec2 &sample) const {
    Vec3 wi = normalize(Vec3(alphaU * _wi.x, alphaV * _wi.y, _wi.z));
    float theta = 0, phi = 0;
    if (wi.z < 0.99999f) {
        theta = std::acos(wi.z);
        phi = std::atan2(wi.y, wi.x);
    }
    float sinPhi, cosPhi;
    sincos_(phi, &sinPhi, &cosPhi);
    Vec2 slope = sampleVisible11(theta, sample);
    slope = Vec2{ cosPhi * slope.x - sinPhi * slope.y, sinPhi * slope.x + cosPhi * slope.y };
    slope.x *= alphaU;
    slope.y *= alphaV;
    float normalization = 1.0f / std::sqrt(slope.x * slope.x + slope.y * slope.y + 1.0f);
    return Vec3(-slope.x * normalization, -slope.y * normalization, normalization);
}

float MicrofacetDistribution::pdfVisible(const Vec3 &wi, const Vec3 &m) const {
    if (Frame::cosTheta(wi) == 0) return 0.0f;
    return smithG1(wi, m) * absDot(wi, m) * eval(m) / std::abs(Frame::cosTheta(wi));
}

float MicrofacetDistribution::projectRoughness(const Vec3 &v) const {
    float invSinTheta2 = 1 / Frame::sinTheta2(v);
    if (isIsotropic() || invSinTheta2 <= 0) return alphaU;
    float cosPhi2 = v.x * v.x * invSinTheta2;
    float sinPhi2 = v.y * v.y * invSinTheta2;
    return std::sqrt(cosPhi2 * alphaU * alphaU + sinPhi2 * alphaV * alphaV);
}

float MicrofacetDistribution::smithG1(const Vec3 &v, const Vec3 &m) const {
    if (dot(v, m) * Frame::cosTheta(v) <= 0) return 0.0f;
    float tanTheta = std::abs(Frame::tanTheta(v));
    if (tanTheta == 0.0f) return 1.0f;
    float alpha = projectRoughness(v);
    if (type == ORC_DISTR_BECKMANN) {
        float a = 1.0f / (alpha * tanTheta);
        if (a >= 1.6f) return 1.0f;
        float aSqr = a * a;
        return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
    } else {
        float root = alpha * tanTheta;
        return 2.0f / (1.0f + hypot2(1.0f, root));
    }
}

Vec2 MicrofacetDistribution::sampleVisible11(float thetaI, Vec2 sample) const {
    const float SQRT_PI_INV = 1 / std::sqrt(M_PI_F);
    Vec2 slope;
    if (type == ORC_DISTR_BECKMANN) {
        if (thetaI < 1e-4f) {
            float sinPhi, cosPhi;
            float r = std::sqrt(-fastlog(1.0f - sample.x));
            sincos_(2 * M_PI_F * sample.y, &sinPhi, &cosPhi);
            return Vec2{ r * cosPhi, r * sinPhi };
        }
        float tanThetaI = std::tan(thetaI);
        float cotThetaI = 1 / tanThetaI;
        float a = -1, c = erf(cotThetaI);
        float sample_x = std::max(sample.x, 1e-6f);
        float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
        float b = c - (1 + c) * std::pow(1 - sample_x, fit);
        float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * std::exp(-cotThetaI * cotThetaI));
        int it = 0;
        while (++it < 10) {
            if (!(b >= a && b <= c)) b = 0.5f * (a + c);
            float invErf = erfinv(b);
            float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * std::exp(-invErf * invErf)) - sample_x;
            float derivative = normalization * (1 - invErf * tanThetaI);
            if (std::abs(value) < 1e-5f) break;
            if (value > 0) c = b; else a = b;
            b -= value / derivative;
        }
        slope.x = erfinv(b);
        slope.y = erfinv(2.0f * std::max(sample.y, 1e-6f) - 1.0f);
    } else {
        if (thetaI < 1e-4f) {
            float sinPhi, cosPhi;
            float r = safe_sqrt(sample.x / (1 - sample.x));
            sincos_(2 * M_PI_F * sample.y, &sinPhi, &cosPhi);
            return Vec2{ r * cosPhi, r * sinPhi };
        }
        float tanThetaI = std::tan(thetaI);
        float a = 1 / tanThetaI;
        float G1 = 2.0f / (1.0f + safe_sqrt(1.0f + 1.0f / (a * a)));
        float A = 2.0f * sample.x / G1 - 1.0f;
        if (std::abs(A) == 1) A -= signum(A) * Epsilon;
        float tmp = 1.0f / (A * A - 1.0f);
        float B = tanThetaI;
        float D = safe_sqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
        float slope_x_1 = B * tmp - D;
        float slope_x_2 = B * tmp + D;
        slope.x = (A < 0.0f || slope_x_2 > 1.0f / tanThetaI) ? slope_x_1 : slope_x_2;
        float S;
        if (sample.y > 0.5f) { S = 1.0f; sample.y = 2.0f * (sample.y - 0.5f); }
        else { S = -1.0f; sample.y = 2.0f * (0.5f - sample.y); }
        float z = (sample.y * (sample.y * (sample.y * (-0.365728915865723f) + 0.790235037209296f) - 0.424965825137544f) + 0.000152998850436920f) /
                  (sample.y * (sample.y * (sample.y * (sample.y * 0.169507819808272f - 0.397203533833404f) - 0.232500544458471f) + 1.0f) - 0.539825872510702f);
        slope.y = S * z * std::sqrt(1.0f + slope.x * slope.x);
    }
    return slope;
}

// ---- Material configuration ----------------------------------------------------------------------
static Spectrum toSpec(const float *v) { return Spectrum(v[0], v[1], v[2]); }

thread_local Vec2 g_textureUV{ 0, 0 };
thread_local UVPartials g_texturePartials;
thread_local const Intersection *g_bsdfIts = nullptr;

static int modulo(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }     // math::modulo

Spectrum Texture::evalAt(const Vec2 &uv, bool filtered, const Vec2 &d0, const Vec2 &d1) const {
    if (type == 1) {        // checkerboard.cpp:66-75 (its filtered variant is the same function)
        const int x = 2 * modulo((int) (uv.x * 2), 2) - 1, y = 2 * modulo((int) (uv.y * 2), 2) - 1;
        return x * y == 1 ? color0 : color1;
    } else if (type == 2) { // gridtexture.cpp:75-88
        float x = uv.x - (float) (int) std::floor(uv.x), y = uv.y - (float) (int) std::floor(uv.y);
        if (x > .5) x -= 1;
        if (y > .5) y -= 1;
        return (std::abs(x) < lineWidth || std::abs(y) < lineWidth) ? color1 : color0;
    } else if (type == 3) { // bitmap.cpp:431-454 (no partials: full-resolution level), :486-499 (filtered)
        if (filtered) return mip->eval(uv, d0, d1);
        return mip->filterType != MIP_NEAREST ? mip->evalBilinear(0, uv) : mip->evalBox(0, uv);
    }
    return color0;
}

Spectrum Texture::eval(const Vec2 &its_uv) const {
    Spectrum value = color0;
    if (type != 0) {
        // Texture2D::eval(its, filter) (texture.cpp:112-121)
        const Vec2 uv{ its_uv.x * uvScale.x + uvOffset.x, its_uv.y * uvScale.y + uvOffset.y };
        const UVPartials &pt = g_texturePartials;
        value = evalAt(uv, pt.has, Vec2{ pt.dudx * uvScale.x, pt.dvdx * uvScale.y }, Vec2{ pt.dudy * uvScale.x, pt.dvdy * uvScale.y });
    }
    for (float sc : scales) value *= sc;      // ScaleTexture::eval
    return value;
}

void Texture::evalGradient(const Vec2 &its_uv, Spectrum *gradient) const {
    gradient[0] = gradient[1] = Spectrum(0.0f);
    if (type == 0) return;  // ConstantSpectrumTexture: Texture::evalGradient leaves zero (texture.cpp:40-42)
    const Vec2 uv{ its_uv.x * uvScale.x + uvOffset.x, its_uv.y * uvScale.y + uvOffset.y };
    if (type == 3) {        // bitmap.cpp:456-480
        if (mip->filterType != MIP_NEAREST) mip->evalGradientBilinear(0, uv, gradient);
    } else {                // Texture2D::evalGradient(uv, gradient): forward differences (texture.cpp:132-141)
        const float eps = Epsilon;
        const Vec2 none{ 0, 0 };
        const Spectrum value = evalAt(uv, false, none, none), valueU = evalAt(Vec2{ uv.x + eps, uv.y }, false, none, none),
                       valueV = evalAt(Vec2{ uv.x, uv.y + eps }, false, none, none);
        gradient[0] = (valueU - value) * (1 / eps);
        gradient[1] = (valueV - value) * (1 / eps);
    }
    gradient[0] *= uvScale.x; gradient[1] *= uvScale.y;
    for (float sc : scales) { gradient[0] *= sc; gradient[1] *= sc; }     // ScaleTexture::evalGradient (scale.cpp)
}

Spectrum Texture::getMaximum() const {
    Spectrum m = color0;
    if (type == 3) m = mip->maximum;
    else if (type != 0) m = Spectrum(std::max(color0.x, color1.x), std::max(color0.y, color1.y), std::max(color0.z, color1.z));
    for (float sc : scales) m *= sc;
    return m;
}

Spectrum Texture::getAverage() const {
    Spectrum a = color0;
    if (type == 1) a = (color0 + color1) * 0.5f;
    else if (type == 2) {
        const float interiorWidth = std::max(0.0f, 1 - 2 * lineWidth), interiorArea = interiorWidth * interiorWidth, lineArea = 1 - interiorArea;
        a = color1 * lineArea + color0 * interiorArea;
    } else if (type == 3) a = mip->average;
    for (float sc : scales) a *= sc;
    return a;
}

// A texture of the scene description on its own
const char *buildTexture(const orc_texture &t, Texture &out) {
    out = Texture();
    if (t.type < 1 || t.type > 3) return "unsupported texture type";
    out.type = t.type;
    out.color0 = toSpec(t.color0); out.color1 = toSpec(t.color1);
    out.lineWidth = t.line_width;
    out.uvOffset = Vec2{ t.u_offset, t.v_offset }; out.uvScale = Vec2{ t.u_scale, t.v_scale };
    if (t.scale != 1.0f) out.scales.push_back(t.scale);
    if (t.type == 3) {
        if (!t.pixels || t.width <= 0 || t.height <= 0 || (t.channels != 1 && t.channels != 3)) return "bitmap texture: invalid image";
        if (t.filter_type < 0 || t.filter_type > 3 || t.wrap_u < 0 || t.wrap_u > 4 || t.wrap_v < 0 || t.wrap_v > 4) return "bitmap texture: invalid filter or wrap mode";
        // bitmap.cpp:232-235: only the EWA filter keeps its anisotropy bound
        out.mip = std::make_shared<MIPMap>(t.pixels, t.width, t.height, t.channels, t.wrap_u, t.wrap_v, t.filter_type,
                                           t.filter_type == MIP_EWA ? t.max_anisotropy : 1.0f);
    }
    return nullptr;
}

static const char *makeTexture(const float *constant, int32_t index, const std::vector<Texture> &textures, Texture &out) {
    out = Texture();
    out.color0 = out.color1 = toSpec(constant);
    if (index == 0) return nullptr;
    if (index < 0 || (size_t) index > textures.size()) return "material texture index out of range";
    out = textures[index - 1];
    return nullptr;
}

// BSDF::ensureEnergyConservation (src/librender/bsdf.cpp:88-146): a ScaleTexture around the texture
static void ensureEnergy(bool enabled, Texture &a) {
    if (!enabled) return;
    float actualMax = a.getMaximum().max();
    if (actualMax > 1.0f) a.scales.push_back(0.99f * (1.0f / actualMax));
}
static void ensureEnergy2(bool enabled, Texture &a, Texture &b) {
    if (!enabled) return;
    float actualMax = (a.getMaximum() + b.getMaximum()).max();
    if (actualMax > 1.0f) { float scale = 0.99f * (1.0f / actualMax); a.scales.push_back(scale); b.scales.push_back(scale); }
}

// fresnelDiffuseReflectance(eta, fast = false) (src/libcore/util.cpp:806-859): the integral over [0,1] of
// fresnelDielectricExt(sqrt(xi), eta). The reference runs an adaptive Gauss-Lobatto rule to a relative error of
// 1e-5; a midpoint rule with 2^20 panels in double is far inside that.
static float fresnelDiffuseReflectance(float eta) {
    const int n = 1 << 20;
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        const double xi = (i + 0.5) / n;
        sum += (double) fresnelDielectricExt((float) std::sqrt(xi), eta);
    }
    return (float) (sum / n);
}

void Material::configure(const std::vector<Texture> &textures) {
    bool ee = m.ensure_energy_conservation != 0;
    Texture &diffuse = diffuseTex, &specular = specularTex, &transmittance = transmittanceTex;
    for (const char *err : { makeTexture(m.diffuse_reflectance, m.diffuse_texture, textures, diffuse),
                             makeTexture(m.specular_reflectance, m.specular_texture, textures, specular),
                             makeTexture(m.specular_transmittance, m.transmittance_texture, textures, transmittance) })
        if (err) throw std::runtime_error(err);
    alphaUTextured = m.alpha_u_texture != 0; alphaVTextured = m.alpha_v_texture != 0;
    if (alphaUTextured || alphaVTextured) {
        if (m.type != ORC_BSDF_ROUGHCONDUCTOR && m.type != ORC_BSDF_ROUGHDIELECTRIC && m.type != ORC_BSDF_ROUGHPLASTIC)
            throw std::runtime_error("a textured roughness is supported for roughconductor, roughdielectric and roughplastic only");
        if (m.type == ORC_BSDF_ROUGHPLASTIC && m.alpha_u_texture != m.alpha_v_texture)
            throw std::runtime_error("The 'roughplastic' plugin currently does not support anisotropic microfacet distributions!");
        for (int32_t t : { m.alpha_u_texture, m.alpha_v_texture })
            if (t < 0 || (size_t) t > textures.size()) throw std::runtime_error("roughness texture index out of range");
        if (alphaUTextured) alphaUTex = textures[m.alpha_u_texture - 1];
        if (alphaVTextured) alphaVTex = textures[m.alpha_v_texture - 1];
    }
    frameKind = m.frame_kind;
    if (frameKind < 0 || frameKind > 2) throw std::runtime_error("unknown frame adapter");
    if (frameKind) {
        if (m.type == ORC_BSDF_MIXTURE) throw std::runtime_error("bumpmap / normalmap around a mixturebsdf is not supported");
        if (m.frame_texture < 1 || (size_t) m.frame_texture > textures.size()) throw std::runtime_error("bumpmap / normalmap: texture index out of range");
        frameTex = textures[m.frame_texture - 1];
    }
    {   // m_usesRayDifferentials of the plugins' configure() (e.g. diffuse.cpp:101, roughplastic.cpp:301-304, dielectric.cpp:197-199)
        const bool d = diffuse.usesRayDifferentials(), sp = specular.usesRayDifferentials(), tr = transmittance.usesRayDifferentials();
        switch (m.type) {
            case ORC_BSDF_DIFFUSE: usesRayDifferentials = d; break;
            case ORC_BSDF_ROUGHCONDUCTOR: case ORC_BSDF_CONDUCTOR: usesRayDifferentials = sp; break;      // + the roughness textures, below
            case ORC_BSDF_ROUGHPLASTIC: case ORC_BSDF_PLASTIC: case ORC_BSDF_WARD: usesRayDifferentials = d || sp; break;
            case ORC_BSDF_DIELECTRIC: case ORC_BSDF_ROUGHDIELECTRIC: usesRayDifferentials = sp || tr; break;
            default: usesRayDifferentials = false;
        }
    }
    // roughconductor.cpp:240-244, roughdielectric.cpp:264-268: the roughness textures count as well
    if ((alphaUTextured && alphaUTex.usesRayDifferentials()) || (alphaVTextured && alphaVTex.usesRayDifferentials())) usesRayDifferentials = true;
    unsigned aniso = (m.alpha_u != m.alpha_v || m.alpha_u_texture != m.alpha_v_texture) ? EAnisotropic : 0;
    switch (m.type) {
        case ORC_BSDF_DIFFUSE:   // src/bsdfs/diffuse.cpp:93-104
            ensureEnergy(ee, diffuse);
            type_flags = diffuse.getMaximum().max() > 0 ? (EDiffuseReflection | EFrontSide) : 0;
            break;
        case ORC_BSDF_ROUGHCONDUCTOR:   // src/bsdfs/roughconductor.cpp:187-190,228-250
            ensureEnergy(ee, specular);
            condEta = toSpec(m.eta) / m.ext_eta;
            condK = toSpec(m.k) / m.ext_eta;
            type_flags = EGlossyReflection | EFrontSide | aniso;
            break;
        case ORC_BSDF_ROUGHPLASTIC: {   // src/bsdfs/roughplastic.cpp:213,255-307
            ensureEnergy(ee, specular);
            ensureEnergy(ee, diffuse);
            eta = m.int_ior / m.ext_ior;
            invEta2 = 1.0f / (eta * eta);
            float dAvg = luminance(diffuse.getAverage()), sAvg = luminance(specular.getAverage());
            specularSamplingWeight = sAvg / (dAvg + sAvg);
            if (m.rtrans && m.rtrans_n) rtrans.assign(m.rtrans, m.rtrans + m.rtrans_n);
            if (alphaUTextured) {
                // roughplastic.cpp:281-299 with a non-constant alpha: the exterior table stays two-dimensional; the block is
                // thetaSamples, alphaSamples, alphaMin, alphaMax, T[alpha][theta], diffuse exterior [alpha], diffuse interior [alpha]
                if (rtrans.size() < 4 || !(rtrans[0] >= 2 && rtrans[1] >= 2 && rtrans[3] > rtrans[2]) ||
                    rtrans.size() != 4 + (size_t) rtrans[0] * (size_t) rtrans[1] + 2 * (size_t) rtrans[1])
                    throw std::runtime_error("roughplastic with a textured roughness: malformed two-dimensional transmittance table");
            }
            type_flags = EGlossyReflection | EDiffuseReflection | EFrontSide;
            break;
        }
        case ORC_BSDF_DIELECTRIC:   // src/bsdfs/dielectric.cpp:158-159,184-202
            ensureEnergy(ee, specular);
            ensureEnergy(ee, transmittance);
            eta = m.int_ior / m.ext_ior;
            invEta = 1 / eta;
            type_flags = EDeltaReflection | EDeltaTransmission | EFrontSide | EBackSide | ENonSymmetric;
            break;
        case ORC_BSDF_WARD: {   // src/bsdfs/ward.cpp:139-171
            ensureEnergy2(ee, specular, diffuse);
            float dAvg = luminance(diffuse.getAverage()), sAvg = luminance(specular.getAverage());
            specularSamplingWeight = sAvg / (dAvg + sAvg);
            type_flags = EGlossyReflection | EDiffuseReflection | EFrontSide | aniso;
            break;
        }
        case ORC_BSDF_CONDUCTOR:   // src/bsdfs/conductor.cpp:172-175,187-200
            ensureEnergy(ee, specular);
            condEta = toSpec(m.eta) / m.ext_eta;
            condK = toSpec(m.k) / m.ext_eta;
            type_flags = EDeltaReflection | EFrontSide;
            break;
        case ORC_BSDF_ROUGHDIELECTRIC:   // src/bsdfs/roughdielectric.cpp:183-211,245-274
            ensureEnergy(ee, specular);
            ensureEnergy(ee, transmittance);
            eta = m.int_ior / m.ext_ior;
            invEta = 1 / eta;
            type_flags = EGlossyReflection | EGlossyTransmission | EFrontSide | EBackSide | ENonSymmetric | aniso;
            break;
        case ORC_BSDF_PLASTIC: {   // src/bsdfs/plastic.cpp:156,187-216
            ensureEnergy(ee, specular);
            ensureEnergy(ee, diffuse);
            eta = m.int_ior / m.ext_ior;
            fdrInt = fresnelDiffuseReflectance(1 / eta);
            float dAvg = luminance(diffuse.getAverage()), sAvg = luminance(specular.getAverage());
            specularSamplingWeight = sAvg / (dAvg + sAvg);
            invEta2 = 1 / (eta * eta);
            type_flags = EDeltaReflection | EDiffuseReflection | EFrontSide;
            break;
        }
    }
    if (frameKind) {        // bumpmap.cpp:92-96, normalmap.cpp:64-68: every component becomes anisotropic, partials are wanted
        if (type_flags) type_flags |= EAnisotropic;
        usesRayDifferentials = true;
    }
    own_flags = type_flags;
    back = nullptr;
}

// MixtureBSDF::configure (src/bsdfs/mixturebsdf.cpp:115-172): weights rescaled when they sum to more than one, the
// discrete choice proportional to them, the type the union of the nested BSDFs' components. Nested BSDFs are
// restricted to one-sided models other than mixtures.
const char *linkMixtures(std::vector<Material> &materials) {
    for (Material &mat : materials) {
        if (mat.m.type != ORC_BSDF_MIXTURE) continue;
        const int n = mat.m.mix_count;
        if (n < 1 || n > 4) return "mixturebsdf: between one and four nested BSDFs are supported";
        float total = 0;
        for (int i = 0; i < n; ++i) {
            if (mat.m.mix_weight[i] < 0) return "Invalid BSDF weight!";
            total += mat.m.mix_weight[i];
        }
        if (total <= 0) return "The weights must sum to a value greater than zero!";
        const float scale = (mat.m.ensure_energy_conservation && total > 1) ? 1.0f / total : 1.0f;
        mat.mixChildren.clear(); mat.mixWeights.clear();
        mat.mixPdf = DiscreteDistribution();
        mat.own_flags = 0;
        for (int i = 0; i < n; ++i) {
            const int c = mat.m.mix_material[i];
            if (c < 0 || c >= (int) materials.size()) return "mixturebsdf: nested material index out of range";
            const Material &child = materials[c];
            const int t = child.m.type;
            (void) t;
            if (child.m.two_sided || child.m.type == ORC_BSDF_MIXTURE)
                return "mixturebsdf: twosided and mixturebsdf cannot be nested";
            if (child.frameKind) return "mixturebsdf: bumpmap and normalmap cannot be nested";
            mat.mixChildren.push_back(&child);
            mat.mixWeights.push_back(scale != 1.0f ? mat.m.mix_weight[i] * scale : mat.m.mix_weight[i]);
            mat.mixPdf.append(mat.mixWeights.back());
            mat.own_flags |= child.own_flags;
            mat.usesRayDifferentials |= child.usesRayDifferentials;     // mixturebsdf.cpp:167
        }
        mat.mixPdf.normalize();
        mat.type_flags = mat.own_flags;
    }
    return nullptr;
}

// TwoSidedBRDF::configure (src/bsdfs/twosided.cpp:84-106): the adapter's components are the front BRDF's on the
// front side and the back BRDF's on the back side; nesting a transmissive BSDF is an error.
const char *linkTwoSided(std::vector<Material> &materials) {
    for (Material &mat : materials) {
        if (!mat.m.two_sided) continue;
        const int b = mat.m.back_material;
        if (b >= (int) materials.size()) return "twosided: back material index out of range";
        mat.back = b < 0 ? &mat : &materials[b];
        const unsigned f = mat.own_flags, k = mat.back->own_flags;
        if ((f | k) & ETransmission) return "Only materials without a transmission component can be nested!";
        mat.type_flags = (f ? ((f & ~EBackSide) | EFrontSide) : 0u) | (k ? ((k & ~EFrontSide) | EBackSide) : 0u);
    }
    for (Material &mat : materials)     // twosided.cpp:91-92 (after every record's own flag is final)
        if (mat.m.two_sided && mat.back) mat.twoSidedRayDifferentials = mat.usesRayDifferentials || mat.back->usesRayDifferentials;
    return nullptr;
}

// ---- RoughTransmittance::eval with the index of refraction fixed: fully reduced (src/bsdfs/rtrans.h:183-194,233) or,
//      for a textured roughness, still a function of alpha (rtrans.h:195-205) -----------------------------------------
static float rtransWarpedAlpha(const Material &mat, float alpha) {
    return std::pow((alpha - mat.rtrans[2]) / (mat.rtrans[3] - mat.rtrans[2]), 0.25f);
}
static float rtransEval(const Material &mat, float cosTheta, float alpha) {
    float warpedCosTheta = std::pow(std::abs(cosTheta), 0.25f);
    if (!(cosTheta >= 0)) return 0.f;
    float result;
    if (mat.alphaUTextured)
        result = evalCubicInterp2D(warpedCosTheta, rtransWarpedAlpha(mat, alpha), mat.rtrans.data() + 4, (size_t) mat.rtrans[0], (size_t) mat.rtrans[1]);
    else
        result = evalCubicInterp1D(warpedCosTheta, mat.rtrans.data(), mat.rtrans.size(), 0.0f, 1.0f);
    return std::min(1.0f, std::max(0.0f, result));
}
// m_internalRoughTransmittance->evalDiffuse(alpha) (rtrans.h:249-259)
static float rtransDiffuseInterior(const Material &mat, float alpha) {
    if (!mat.alphaUTextured) return mat.m.rtrans_diffuse_int;
    const size_t nTheta = (size_t) mat.rtrans[0], nAlpha = (size_t) mat.rtrans[1];
    float result = evalCubicInterp1D(rtransWarpedAlpha(mat, alpha), mat.rtrans.data() + 4 + nTheta * nAlpha + nAlpha, nAlpha, 0.0f, 1.0f);
    return std::min(1.0f, std::max(0.0f, result));
}

static inline Vec3 reflect(const Vec3 &wi, const Vec3 &m) { return 2 * dot(wi, m) * m - wi; }

// src/libcore/util.cpp sphericalDirection
static Vec3 sphericalDirection(float theta, float phi) {
    float sinTheta, cosTheta, sinPhi, cosPhi;
    sincos_(theta, &sinTheta, &cosTheta);
    sincos_(phi, &sinPhi, &cosPhi);
    return Vec3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}

// ---- eval -----------------------------------------------------------------------------------------
static Spectrum evalOneSided(const Material &mat, const Vec3 &wi, const Vec3 &wo) {
    const orc_material &m = mat.m;
    switch (m.type) {
        case ORC_BSDF_DIFFUSE: {   // diffuse.cpp:110-118
            if (!(mat.own_flags & EDiffuseReflection)) return Spectrum(0.0f);
            if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0) return Spectrum(0.0f);
            return mat.D() * (INV_PI_F * Frame::cosTheta(wo));
        }
        case ORC_BSDF_ROUGHCONDUCTOR: {   // roughconductor.cpp:257-294
            if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0) return Spectrum(0.0f);
            Vec3 H = normalize(wo + wi);
            MicrofacetDistribution distr(m.distribution, mat.AU(), mat.AV(), m.sample_visible != 0);
            const float D = distr.eval(H);
            if (D == 0) return Spectrum(0.0f);
            const Spectrum F = fresnelConductorExact(dot(wi, H), mat.condEta, mat.condK) * mat.S();
            const float G = distr.G(wi, wo, H);
            float model = D * G / (4.0f * Frame::cosTheta(wi));
            return F * model;
        }
        case ORC_BSDF_ROUGHPLASTIC: {   // roughplastic.cpp:326-382
            if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0) return Spectrum(0.0f);
            MicrofacetDistribution distr(m.distribution, mat.AU(), mat.AU(), m.sample_visible != 0);
            Spectrum result(0.0f);
            {
                const Vec3 H = normalize(wo + wi);
                const float D = distr.eval(H);
                const float F = fresnelDielectricExt(dot(wi, H), mat.eta);
                const float G = distr.G(wi, wo, H);
                float value = F * D * G / (4.0f * Frame::cosTheta(wi));
                result += mat.S() * value;
            }
            {
                Spectrum diff = mat.D();
                float T12 = rtransEval(mat, Frame::cosTheta(wi), distr.getAlpha());
                float T21 = rtransEval(mat, Frame::cosTheta(wo), distr.getAlpha());
                float Fdr = 1 - rtransDiffuseInterior(mat, distr.getAlpha());
                if (m.variant) diff = diff / (Spectrum(1.0f) - diff * Fdr);
                else diff /= 1 - Fdr;
                result += diff * (INV_PI_F * Frame::cosTheta(wo) * T12 * T21 * mat.invEta2);
            }
            return result;
        }
        case ORC_BSDF_DIELECTRIC:   // dielectric.cpp:228-253: zero for the solid-angle measure
            return Spectrum(0.0f);
        case ORC_BSDF_WARD: {   // ward.cpp:178-226
            if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0) return Spectrum(0.0f);
            Spectrum result(0.0f);
            Vec3 H = wi + wo;
            float alphaU = m.alpha_u, alphaV = m.alpha_v;
            float factor1 = 0.0f;
            switch (m.variant) {
                case 0: factor1 = 1.0f / (4.0f * M_PI_F * alphaU * alphaV * std::sqrt(Frame::cosTheta(wi) * Frame::cosTheta(wo))); break;
                case 1: factor1 = 1.0f / (4.0f * M_PI_F * alphaU * alphaV * Frame::cosTheta(wi) * Frame::cosTheta(wo)); break;
                default:   // std::pow(Float, int) promotes to double in the reference expression
                    factor1 = (float) ((double) dot(H, H) / ((double) (M_PI_F * alphaU * alphaV) * std::pow((double) Frame::cosTheta(H), 4.0))); break;
            }
            float factor2 = H.x / alphaU, factor3 = H.y / alphaV;
            float exponent = -(factor2 * factor2 + factor3 * factor3) / (H.z * H.z);
            float specRef = factor1 * fastexp(exponent);
            if (specRef > 1e-10f) result += mat.S() * specRef;
            result += mat.D() * INV_PI_F;
            return result * Frame::cosTheta(wo);
        }
        case ORC_BSDF_ROUGHDIELECTRIC: {   // roughdielectric.cpp:276-346
            if (Frame::cosTheta(wi) == 0) return Spectrum(0.0f);
            bool reflect_ = Frame::cosTheta(wi) * Frame::cosTheta(wo) > 0;
            Vec3 H;
            if (reflect_) {
                H = normalize(wo + wi);
            } else {
                float eta_ = Frame::cosTheta(wi) > 0 ? mat.eta : mat.invEta;
                H = normalize(wi + wo * eta_);
            }
            H = H * signum(Frame::cosTheta(H));
            MicrofacetDistribution distr(m.distribution, mat.AU(), mat.AV(), m.sample_visible != 0);
            const float D = distr.eval(H);
            if (D == 0) return Spectrum(0.0f);
            const float F = fresnelDielectricExt(dot(wi, H), mat.eta);
            const float G = distr.G(wi, wo, H);
            if (reflect_) {
                float value = F * D * G / (4.0f * std::abs(Frame::cosTheta(wi)));
                return mat.S() * value;
            } else {
                float eta_ = Frame::cosTheta(wi) > 0.0f ? mat.eta : mat.invEta;
                float sqrtDenom = dot(wi, H) + eta_ * dot(wo, H);
                float value = ((1 - F) * D * G * eta_ * eta_ * dot(wi, H) * dot(wo, H)) / (Frame::cosTheta(wi) * sqrtDenom * sqrtDenom);
                float factor = Frame::cosTheta(wi) > 0 ? mat.invEta : mat.eta;     // ERadiance
                return mat.T() * std::abs(value * factor * factor);
            }
        }
        case ORC_BSDF_CONDUCTOR:   // conductor.cpp:225-239: a delta lobe only, zero for the solid-angle measure
            return Spectrum(0.0f);
        case ORC_BSDF_PLASTIC: {   // plastic.cpp:243-276, solid-angle measure: the diffuse lobe
            if (Frame::cosTheta(wo) <= 0 || Frame::cosTheta(wi) <= 0) return Spectrum(0.0f);
            float Fi = fresnelDielectricExt(Frame::cosTheta(wi), mat.eta);
            float Fo = fresnelDielectricExt(Frame::cosTheta(wo), mat.eta);
            Spectrum diff = mat.D();
            if (m.variant) diff = diff / (Spectrum(1.0f) - diff * mat.fdrInt);
            else diff /= 1 - mat.fdrInt;
            return diff * (squareToCosineHemispherePdf(wo) * mat.invEta2 * (1 - Fi) * (1 - Fo));
        }
    }
    return Spectrum(0.0f);
}

// ---- pdf ------------------------------------------------------------------------------------------
static float pdfOneSided(const Material &mat, const Vec3 &wi, const Vec3 &wo) {
    const orc_material &m = mat.m;
    switch (m.type) {
        case ORC_BSDF_DIFFUSE:   // diffuse.cpp:120-127
            if (!(mat.own_flags & EDiffuseReflection)) return 0.0f;
            if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0) return 0.0f;
            return squareToCosineHemispherePdf(wo);
        case ORC_BSDF_ROUGHCONDUCTOR: {   // roughconductor.cpp:296-321
            if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0) return 0.0f;
            Vec3 H = normalize(wo + wi);
            MicrofacetDistribution distr(m.distribution, mat.AU(), mat.AV(), m.sample_visible != 0);
            if (distr.sampleVisible)
                return distr.eval(H) * distr.smithG1(wi, H) / (4.0f * Frame::cosTheta(wi));
            else
                return distr.pdf(wi, H) / (4 * absDot(wo, H));
        }
        case ORC_BSDF_ROUGHPLASTIC: {   // roughplastic.cpp:384-437
            if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0) return 0.0f;
            MicrofacetDistribution distr(m.distribution, mat.AU(), mat.AU(), m.sample_visible != 0);
            const Vec3 H = normalize(wo + wi);
            float probSpecular = 1 - rtransEval(mat, Frame::cosTheta(wi), distr.getAlpha());
            probSpecular = (probSpecular * mat.specularSamplingWeight) /
                (probSpecular * mat.specularSamplingWeight + (1 - probSpecular) * (1 - mat.specularSamplingWeight));
            float probDiffuse = 1 - probSpecular;
            const float dwh_dwo = 1.0f / (4.0f * dot(wo, H));
            const float prob = distr.pdf(wi, H);
            float result = prob * dwh_dwo * probSpecular;
            result += probDiffuse * squareToCosineHemispherePdf(wo);
            return result;
        }
        case ORC_BSDF_DIELECTRIC:
            return 0.0f;
        case ORC_BSDF_WARD: {   // ward.cpp:228-264
            if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0) return 0.0f;
            float alphaU = m.alpha_u, alphaV = m.alpha_v;
            Vec3 H = normalize(wi + wo);
            float factor1 = (float) (1.0 / ((double) (4.0f * M_PI_F * alphaU * alphaV * dot(H, wi)) * std::pow((double) Frame::cosTheta(H), 3.0)));
            float factor2 = H.x / alphaU, factor3 = H.y / alphaV;
            float exponent = -(factor2 * factor2 + factor3 * factor3) / (H.z * H.z);
            float specProb = factor1 * fastexp(exponent);
            float diffuseProb = squareToCosineHemispherePdf(wo);
            return mat.specularSamplingWeight * specProb + (1 - mat.specularSamplingWeight) * diffuseProb;
        }
        case ORC_BSDF_ROUGHDIELECTRIC: {   // roughdielectric.cpp:348-414
            bool reflect_ = Frame::cosTheta(wi) * Frame::cosTheta(wo) > 0;
            Vec3 H;
            float dwh_dwo;
            if (reflect_) {
                H = normalize(wo + wi);
                dwh_dwo = 1.0f / (4.0f * dot(wo, H));
            } else {
                float eta_ = Frame::cosTheta(wi) > 0 ? mat.eta : mat.invEta;
                H = normalize(wi + wo * eta_);
                float sqrtDenom = dot(wi, H) + eta_ * dot(wo, H);
                dwh_dwo = (eta_ * eta_ * dot(wo, H)) / (sqrtDenom * sqrtDenom);
            }
            H = H * signum(Frame::cosTheta(H));
            MicrofacetDistribution sampleDistr(m.distribution, mat.AU(), mat.AV(), m.sample_visible != 0);
            if (!sampleDistr.sampleVisible) sampleDistr.scaleAlpha(1.2f - 0.2f * std::sqrt(std::abs(Frame::cosTheta(wi))));
            float prob = sampleDistr.pdf(wi * signum(Frame::cosTheta(wi)), H);
            float F = fresnelDielectricExt(dot(wi, H), mat.eta);
            prob *= reflect_ ? F : (1 - F);
            return std::abs(prob * dwh_dwo);
        }
        case ORC_BSDF_CONDUCTOR:   // conductor.cpp:241-255: zero for the solid-angle measure
            return 0.0f;
        case ORC_BSDF_PLASTIC: {   // plastic.cpp:278-306, solid-angle measure
            if (Frame::cosTheta(wo) <= 0 || Frame::cosTheta(wi) <= 0) return 0.0f;
            float Fi = fresnelDielectricExt(Frame::cosTheta(wi), mat.eta);
            float probSpecular = (Fi * mat.specularSamplingWeight) /
                (Fi * mat.specularSamplingWeight + (1 - Fi) * (1 - mat.specularSamplingWeight));
            return squareToCosineHemispherePdf(wo) * (1 - probSpecular);
        }
    }
    return 0.0f;
}

// ---- sample (the pdf-returning overload the integrator calls, path.cpp:209) --------------------------
static Spectrum sampleOneSided(const Material &mat, const Vec3 &wi, const Vec2 &_sample, BSDFSample &out, float &pdf, Sampler1D *sampler1D) {
    const orc_material &m = mat.m;
    pdf = 0;
    out.eta = 1.0f;
    out.sampledType = 0;
    switch (m.type) {
        case ORC_BSDF_DIFFUSE: {   // diffuse.cpp:140-150
            if (!(mat.own_flags & EDiffuseReflection) || Frame::cosTheta(wi) <= 0) return Spectrum(0.0f);
            out.wo = squareToCosineHemisphere(_sample);
            out.eta = 1.0f;
            out.sampledType = EDiffuseReflection;
            pdf = squareToCosineHemispherePdf(out.wo);
            return mat.D();
        }
        case ORC_BSDF_ROUGHCONDUCTOR: {   // roughconductor.cpp:369-415
            if (Frame::cosTheta(wi) < 0) return Spectrum(0.0f);
            MicrofacetDistribution distr(m.distribution, mat.AU(), mat.AV(), m.sample_visible != 0);
            Vec3 mm = distr.sample(wi, _sample, pdf);
            if (pdf == 0) return Spectrum(0.0f);
            out.wo = reflect(wi, mm);
            out.eta = 1.0f;
            out.sampledType = EGlossyReflection;
            if (Frame::cosTheta(out.wo) <= 0) return Spectrum(0.0f);
            Spectrum F = fresnelConductorExact(dot(wi, mm), mat.condEta, mat.condK) * mat.S();
            float weight;
            if (distr.sampleVisible) weight = distr.smithG1(out.wo, mm);
            else weight = distr.eval(mm) * distr.G(wi, out.wo, mm) * dot(wi, mm) / (pdf * Frame::cosTheta(wi));
            pdf /= 4.0f * dot(out.wo, mm);
            return F * weight;
        }
        case ORC_BSDF_ROUGHPLASTIC: {   // roughplastic.cpp:439-501
            if (Frame::cosTheta(wi) <= 0) return Spectrum(0.0f);
            bool choseSpecular = true;
            Vec2 sample = _sample;
            MicrofacetDistribution distr(m.distribution, mat.AU(), mat.AU(), m.sample_visible != 0);
            float probSpecular = 1 - rtransEval(mat, Frame::cosTheta(wi), distr.getAlpha());
            probSpecular = (probSpecular * mat.specularSamplingWeight) /
                (probSpecular * mat.specularSamplingWeight + (1 - probSpecular) * (1 - mat.specularSamplingWeight));
            if (sample.y < probSpecular) {
                sample.y /= probSpecular;
            } else {
                sample.y = (sample.y - probSpecular) / (1 - probSpecular);
                choseSpecular = false;
            }
            if (choseSpecular) {
                Vec3 mm = distr.sample(wi, sample);
                out.wo = reflect(wi, mm);
                out.sampledType = EGlossyReflection;
                if (Frame::cosTheta(out.wo) <= 0) return Spectrum(0.0f);
            } else {
                out.sampledType = EDiffuseReflection;
                out.wo = squareToCosineHemisphere(sample);
            }
            out.eta = 1.0f;
            pdf = pdfOneSided(mat, wi, out.wo);
            if (pdf == 0) return Spectrum(0.0f);
            return evalOneSided(mat, wi, out.wo) / pdf;
        }
        case ORC_BSDF_DIELECTRIC: {   // dielectric.cpp:277-333 (both lobes enabled, ERadiance)
            float cosThetaT;
            float F = fresnelDielectricExt(Frame::cosTheta(wi), cosThetaT, mat.eta);
            if (_sample.x <= F) {
                out.sampledType = EDeltaReflection;
                out.wo = Vec3(-wi.x, -wi.y, wi.z);
                out.eta = 1.0f;
                pdf = F;
                return mat.S();
            } else {
                out.sampledType = EDeltaTransmission;
                float scale = -(cosThetaT < 0 ? mat.invEta : mat.eta);
                out.wo = Vec3(scale * wi.x, scale * wi.y, cosThetaT);
                out.eta = cosThetaT < 0 ? mat.eta : mat.invEta;
                pdf = 1 - F;
                float factor = cosThetaT < 0 ? mat.invEta : mat.eta;
                return mat.T() * (factor * factor);
            }
        }
        case ORC_BSDF_WARD: {   // ward.cpp:266-326
            Vec2 sample = _sample;
            bool choseSpecular = true;
            if (sample.x <= mat.specularSamplingWeight) {
                sample.x /= mat.specularSamplingWeight;
            } else {
                sample.x = (sample.x - mat.specularSamplingWeight) / (1 - mat.specularSamplingWeight);
                choseSpecular = false;
            }
            if (choseSpecular) {
                float alphaU = m.alpha_u, alphaV = m.alpha_v;
                float phiH = std::atan(alphaV / alphaU * std::tan(2.0f * M_PI_F * sample.y));
                if (sample.y > 0.5f) phiH += M_PI_F;
                float cosPhiH = std::cos(phiH);
                float sinPhiH = safe_sqrt(1.0f - cosPhiH * cosPhiH);
                float thetaH = std::atan(safe_sqrt(-fastlog(sample.x) /
                    ((cosPhiH * cosPhiH) / (alphaU * alphaU) + (sinPhiH * sinPhiH) / (alphaV * alphaV))));
                Vec3 H = sphericalDirection(thetaH, phiH);
                out.wo = H * (2.0f * dot(wi, H)) - wi;
                out.sampledType = EGlossyReflection;
                if (Frame::cosTheta(out.wo) <= 0.0f) return Spectrum(0.0f);
            } else {
                out.wo = squareToCosineHemisphere(sample);
                out.sampledType = EDiffuseReflection;
            }
            out.eta = 1.0f;
            pdf = pdfOneSided(mat, wi, out.wo);
            if (pdf == 0) return Spectrum(0.0f);
            return evalOneSided(mat, wi, out.wo) / pdf;
        }
        case ORC_BSDF_ROUGHDIELECTRIC: {   // roughdielectric.cpp:511-613 (both lobes enabled, ERadiance)
            MicrofacetDistribution distr(m.distribution, mat.AU(), mat.AV(), m.sample_visible != 0);
            MicrofacetDistribution sampleDistr(distr);
            if (!distr.sampleVisible) sampleDistr.scaleAlpha(1.2f - 0.2f * std::sqrt(std::abs(Frame::cosTheta(wi))));
            float microfacetPDF;
            const Vec3 mm = sampleDistr.sample(wi * signum(Frame::cosTheta(wi)), _sample, microfacetPDF);
            if (microfacetPDF == 0) return Spectrum(0.0f);
            pdf = microfacetPDF;
            float cosThetaT;
            float F = fresnelDielectricExt(dot(wi, mm), cosThetaT, mat.eta);
            Spectrum weight(1.0f);
            bool sampleReflection = true;
            if ((sampler1D ? sampler1D->next1D() : 0.5f) > F) { sampleReflection = false; pdf *= 1 - F; }
            else pdf *= F;
            float dwh_dwo;
            if (sampleReflection) {
                out.wo = reflect(wi, mm);
                out.eta = 1.0f;
                out.sampledType = EGlossyReflection;
                if (Frame::cosTheta(wi) * Frame::cosTheta(out.wo) <= 0) return Spectrum(0.0f);
                weight *= mat.S();
                dwh_dwo = 1.0f / (4.0f * dot(out.wo, mm));
            } else {
                if (cosThetaT == 0) return Spectrum(0.0f);
                {   // refract(wi, m, eta, cosThetaT) (src/libcore/util.cpp:767-772)
                    float e = cosThetaT < 0 ? 1 / mat.eta : mat.eta;
                    out.wo = mm * (dot(wi, mm) * e + cosThetaT) - wi * e;
                }
                out.eta = cosThetaT < 0 ? mat.eta : mat.invEta;
                out.sampledType = EGlossyTransmission;
                if (Frame::cosTheta(wi) * Frame::cosTheta(out.wo) >= 0) return Spectrum(0.0f);
                float factor = cosThetaT < 0 ? mat.invEta : mat.eta;
                weight *= mat.T() * (factor * factor);
                float sqrtDenom = dot(wi, mm) + out.eta * dot(out.wo, mm);
                dwh_dwo = (out.eta * out.eta * dot(out.wo, mm)) / (sqrtDenom * sqrtDenom);
            }
            if (distr.sampleVisible) weight *= distr.smithG1(out.wo, mm);
            else weight *= std::abs(distr.eval(mm) * distr.G(wi, out.wo, mm) * dot(wi, mm) / (microfacetPDF * Frame::cosTheta(wi)));
            pdf *= std::abs(dwh_dwo);
            return weight;
        }
        case ORC_BSDF_CONDUCTOR: {   // conductor.cpp:273-288
            if (Frame::cosTheta(wi) <= 0) return Spectrum(0.0f);
            out.sampledType = EDeltaReflection;
            out.wo = Vec3(-wi.x, -wi.y, wi.z);
            out.eta = 1.0f;
            pdf = 1;
            return mat.S() * fresnelConductorExact(Frame::cosTheta(wi), mat.condEta, mat.condK);
        }
        case ORC_BSDF_PLASTIC: {   // plastic.cpp:373-440 (both lobes enabled)
            if (Frame::cosTheta(wi) <= 0) return Spectrum(0.0f);
            float Fi = fresnelDielectricExt(Frame::cosTheta(wi), mat.eta);
            out.eta = 1.0f;
            float probSpecular = (Fi * mat.specularSamplingWeight) /
                (Fi * mat.specularSamplingWeight + (1 - Fi) * (1 - mat.specularSamplingWeight));
            if (_sample.x < probSpecular) {
                out.sampledType = EDeltaReflection;
                out.wo = Vec3(-wi.x, -wi.y, wi.z);
                pdf = probSpecular;
                return mat.S() * Fi / probSpecular;
            } else {
                out.sampledType = EDiffuseReflection;
                out.wo = squareToCosineHemisphere(Vec2{ (_sample.x - probSpecular) / (1 - probSpecular), _sample.y });
                float Fo = fresnelDielectricExt(Frame::cosTheta(out.wo), mat.eta);
                Spectrum diff = mat.D();
                if (m.variant) diff = diff / (Spectrum(1.0f) - diff * mat.fdrInt);
                else diff /= 1 - mat.fdrInt;
                pdf = (1 - probSpecular) * squareToCosineHemispherePdf(out.wo);
                return diff * (mat.invEta2 * (1 - Fi) * (1 - Fo) / (1 - probSpecular));
            }
        }
    }
    return Spectrum(0.0f);
}

// ---- eval / pdf in the discrete measure: the Dirac lobes of dielectric (dielectric.cpp:228-275), conductor
// (conductor.cpp:221-250) and plastic (plastic.cpp:243-300); every other lobe is zero there. A mixture asks for them
// after one of its nested BSDFs has sampled such a lobe.

static inline Vec3 reflectZ(const Vec3 &wi) { return Vec3(-wi.x, -wi.y, wi.z); }

static Spectrum evalDiscrete(const Material &mat, const Vec3 &wi, const Vec3 &wo) {
    switch (mat.m.type) {
        case ORC_BSDF_DIELECTRIC: {
            float cosThetaT;
            const float F = fresnelDielectricExt(Frame::cosTheta(wi), cosThetaT, mat.eta);
            if (Frame::cosTheta(wi) * Frame::cosTheta(wo) >= 0) {
                if (std::abs(dot(reflectZ(wi), wo) - 1) > DeltaEpsilon) return Spectrum(0.0f);
                return mat.S() * F;
            }
            const float scale = -(cosThetaT < 0 ? mat.invEta : mat.eta);
            const Vec3 wt(scale * wi.x, scale * wi.y, cosThetaT);
            if (std::abs(dot(wt, wo) - 1) > DeltaEpsilon) return Spectrum(0.0f);
            const float factor = cosThetaT < 0 ? mat.invEta : mat.eta;      // ERadiance
            return mat.T() * (factor * factor * (1 - F));
        }
        case ORC_BSDF_CONDUCTOR:
            if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0 || std::abs(dot(reflectZ(wi), wo) - 1) > DeltaEpsilon) return Spectrum(0.0f);
            return mat.S() * fresnelConductorExact(Frame::cosTheta(wi), mat.condEta, mat.condK);
        case ORC_BSDF_PLASTIC: {
            if (Frame::cosTheta(wo) <= 0 || Frame::cosTheta(wi) <= 0) return Spectrum(0.0f);
            const float Fi = fresnelDielectricExt(Frame::cosTheta(wi), mat.eta);
            if (std::abs(dot(reflectZ(wi), wo) - 1) < DeltaEpsilon) return mat.S() * Fi;
            return Spectrum(0.0f);
        }
    }
    return Spectrum(0.0f);
}

static float pdfDiscrete(const Material &mat, const Vec3 &wi, const Vec3 &wo) {
    switch (mat.m.type) {
        case ORC_BSDF_DIELECTRIC: {
            float cosThetaT;
            const float F = fresnelDielectricExt(Frame::cosTheta(wi), cosThetaT, mat.eta);
            if (Frame::cosTheta(wi) * Frame::cosTheta(wo) >= 0)
                return std::abs(dot(reflectZ(wi), wo) - 1) > DeltaEpsilon ? 0.0f : F;
            const float scale = -(cosThetaT < 0 ? mat.invEta : mat.eta);
            const Vec3 wt(scale * wi.x, scale * wi.y, cosThetaT);
            return std::abs(dot(wt, wo) - 1) > DeltaEpsilon ? 0.0f : 1 - F;
        }
        case ORC_BSDF_CONDUCTOR:
            if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0 || std::abs(dot(reflectZ(wi), wo) - 1) > DeltaEpsilon) return 0.0f;
            return 1.0f;
        case ORC_BSDF_PLASTIC: {
            if (Frame::cosTheta(wo) <= 0 || Frame::cosTheta(wi) <= 0) return 0.0f;
            const float Fi = fresnelDielectricExt(Frame::cosTheta(wi), mat.eta);
            const float probSpecular = (Fi * mat.specularSamplingWeight) / (Fi * mat.specularSamplingWeight + (1 - Fi) * (1 - mat.specularSamplingWeight));
            return std::abs(dot(reflectZ(wi), wo) - 1) < DeltaEpsilon ? probSpecular : 0.0f;
        }
    }
    return 0.0f;
}

static Spectrum evalFramed(const Material &mat, const Vec3 &wi, const Vec3 &wo);
static float pdfFramed(const Material &mat, const Vec3 &wi, const Vec3 &wo);
static Spectrum sampleFramed(const Material &mat, const Vec3 &wi, const Vec2 &sample, BSDFSample &out, float &pdf, Sampler1D *sampler1D);

// ---- MixtureBSDF::eval / pdf / sample (src/bsdfs/mixturebsdf.cpp:174-277), all components requested -------------
static Spectrum evalPlain(const Material &mat, const Vec3 &wi, const Vec3 &wo) {
    if (mat.m.type != ORC_BSDF_MIXTURE) return evalFramed(mat, wi, wo);
    Spectrum result(0.0f);
    for (size_t i = 0; i < mat.mixChildren.size(); ++i)
        result += evalOneSided(*mat.mixChildren[i], wi, wo) * mat.mixWeights[i];
    return result;
}

static float pdfPlain(const Material &mat, const Vec3 &wi, const Vec3 &wo) {
    if (mat.m.type != ORC_BSDF_MIXTURE) return pdfFramed(mat, wi, wo);
    float result = 0.0f;
    for (size_t i = 0; i < mat.mixChildren.size(); ++i)
        result += pdfOneSided(*mat.mixChildren[i], wi, wo) * mat.mixPdf[i];
    return result;
}

static Spectrum samplePlain(const Material &mat, const Vec3 &wi, const Vec2 &sample_, BSDFSample &out, float &pdf, Sampler1D *sampler1D) {
    if (mat.m.type != ORC_BSDF_MIXTURE) return sampleFramed(mat, wi, sample_, out, pdf, sampler1D);
    Vec2 sample(sample_);
    const size_t entry = mat.mixPdf.sampleReuse(sample.x);
    Spectrum result = sampleOneSided(*mat.mixChildren[entry], wi, sample, out, pdf, sampler1D);
    if (result.isZero()) return result;     // sampling failed
    result *= mat.mixWeights[entry] * pdf;
    pdf *= mat.mixPdf[entry];
    const bool discrete = (out.sampledType & EDelta) != 0;      // BSDF::getMeasure(bRec.sampledType)
    for (size_t i = 0; i < mat.mixChildren.size(); ++i) {
        if (i == entry) continue;
        const Material &c = *mat.mixChildren[i];
        pdf += (discrete ? pdfDiscrete(c, wi, out.wo) : pdfOneSided(c, wi, out.wo)) * mat.mixPdf[i];
        result += (discrete ? evalDiscrete(c, wi, out.wo) : evalOneSided(c, wi, out.wo)) * mat.mixWeights[i];
    }
    return result / pdf;
}

// ---- BumpMap / NormalMap (src/bsdfs/bumpmap.cpp:135-237, src/bsdfs/normalmap.cpp:107-260): the nested BSDF is asked in
// a perturbed shading frame; directions that change sides between the two frames carry nothing ----------------
static const Intersection &queryIntersection() {
    if (g_bsdfIts) return *g_bsdfIts;
    static thread_local Intersection canonical;
    canonical.shFrame.s = Vec3(1, 0, 0); canonical.shFrame.t = Vec3(0, 1, 0); canonical.shFrame.n = Vec3(0, 0, 1);
    canonical.geoFrame = canonical.shFrame;
    canonical.dpdu = Vec3(1, 0, 0); canonical.dpdv = Vec3(0, 1, 0);
    canonical.uv = g_textureUV;
    return canonical;
}

static Frame perturbedFrame(const Material &mat, const Intersection &its) {
    Frame result;
    if (mat.frameKind == 1) {           // BumpMap::getFrame (bumpmap.cpp:135-159)
        Spectrum grad[2];
        mat.frameTex.evalGradient(its.uv, grad);
        const float dDispDu = luminance(grad[0]), dDispDv = luminance(grad[1]);
        const Vec3 dpdu = its.dpdu + its.shFrame.n * (dDispDu - dot(its.shFrame.n, its.dpdu));
        const Vec3 dpdv = its.dpdv + its.shFrame.n * (dDispDv - dot(its.shFrame.n, its.dpdv));
        result.n = normalize(cross(dpdu, dpdv));
        result.s = normalize(dpdu - result.n * dot(result.n, dpdu));
        result.t = cross(result.n, result.s);
        if (dot(result.n, its.geoFrame.n) < 0) result.n *= -1;
    } else {                            // NormalMap::getFrame (normalmap.cpp:107-124): an unfiltered lookup
        const UVPartials saved = g_texturePartials;
        g_texturePartials = UVPartials();
        const Spectrum c = mat.frameTex.eval(its.uv);
        g_texturePartials = saved;
        const Vec3 n(2 * c.x - 1, 2 * c.y - 1, 2 * c.z - 1);
        result.n = normalize(its.shFrame.toWorld(n));
        result.s = normalize(its.dpdu - result.n * dot(result.n, its.dpdu));
        result.t = cross(result.n, result.s);
    }
    return result;
}

static Spectrum evalFramed(const Material &mat, const Vec3 &wi, const Vec3 &wo) {
    if (!mat.frameKind) return evalOneSided(mat, wi, wo);
    const Intersection &its = queryIntersection();
    const Frame p = perturbedFrame(mat, its);
    const Vec3 pwi = p.toLocal(its.shFrame.toWorld(wi)), pwo = p.toLocal(its.shFrame.toWorld(wo));
    if (Frame::cosTheta(wo) * Frame::cosTheta(pwo) <= 0) return Spectrum(0.0f);
    return evalOneSided(mat, pwi, pwo);
}

static float pdfFramed(const Material &mat, const Vec3 &wi, const Vec3 &wo) {
    if (!mat.frameKind) return pdfOneSided(mat, wi, wo);
    const Intersection &its = queryIntersection();
    const Frame p = perturbedFrame(mat, its);
    const Vec3 pwi = p.toLocal(its.shFrame.toWorld(wi)), pwo = p.toLocal(its.shFrame.toWorld(wo));
    if (Frame::cosTheta(wo) * Frame::cosTheta(pwo) <= 0) return 0.0f;
    return pdfOneSided(mat, pwi, pwo);
}

static Spectrum sampleFramed(const Material &mat, const Vec3 &wi, const Vec2 &sample, BSDFSample &out, float &pdf, Sampler1D *sampler1D) {
    if (!mat.frameKind) return sampleOneSided(mat, wi, sample, out, pdf, sampler1D);
    const Intersection &its = queryIntersection();
    const Frame p = perturbedFrame(mat, its);
    Spectrum result = sampleOneSided(mat, p.toLocal(its.shFrame.toWorld(wi)), sample, out, pdf, sampler1D);
    if (!result.isZero()) {
        const Vec3 pwo = out.wo;
        out.wo = its.shFrame.toLocal(p.toWorld(pwo));
        if (Frame::cosTheta(out.wo) * Frame::cosTheta(pwo) <= 0) return Spectrum(0.0f);
    }
    return result;
}

// ---- the entry points the integrator calls: the BSDF itself, or the same through the twosided adapter
// (src/bsdfs/twosided.cpp:108-183: seen from behind, wi and wo are mirrored and the back BRDF answers) ----------
Spectrum bsdfEval(const Material &mat, const Vec3 &wi, const Vec3 &wo) {
    if (!mat.m.two_sided || Frame::cosTheta(wi) > 0) return evalPlain(mat, wi, wo);
    return evalPlain(*mat.back, Vec3(wi.x, wi.y, -wi.z), Vec3(wo.x, wo.y, -wo.z));
}

float bsdfPdf(const Material &mat, const Vec3 &wi, const Vec3 &wo) {
    if (!mat.m.two_sided || wi.z > 0) return pdfPlain(mat, wi, wo);
    return pdfPlain(*mat.back, Vec3(wi.x, wi.y, -wi.z), Vec3(wo.x, wo.y, -wo.z));
}

Spectrum bsdfSample(const Material &mat, const Vec3 &wi, const Vec2 &sample, BSDFSample &out, float &pdf, Sampler1D *sampler1D) {
    if (!mat.m.two_sided || !(Frame::cosTheta(wi) < 0)) return samplePlain(mat, wi, sample, out, pdf, sampler1D);
    Spectrum result = samplePlain(*mat.back, Vec3(wi.x, wi.y, -wi.z), sample, out, pdf, sampler1D);
    if (!result.isZero() && pdf != 0) out.wo.z *= -1;
    return result;
}

}  // namespace orc
