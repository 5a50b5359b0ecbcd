// Device-side shading: intersection records, the five BSDF models, microfacet machinery and next-event
// emitter sampling. Each function names the reference code it reproduces.
#pragma once
#include "device_math.h"
#include "uv_tangents.h"

namespace mtsamd {

struct Isect {
    V3 p, geoN, wi;
    V3 dpdu, dpdv;      // position partials (read only where uv partials are computed)
    Frame3 sh;
    float uvx, uvy;
    uint32_t shape, material, flags;
    int32_t emitter;
};

// ShapeKDTree::fillIntersectionRecord<true> (include/mitsuba/render/skdtree.h:343-428) + computeShadingFrame
// (src/libcore/util.cpp:603-608). tri is the global triangle index; everything comes from the triangle's packed
// shading record, so the six loads go out together.
MTS_DEV void fillIntersection(const DevScene &sc, uint32_t tri, float u, float v, V3 rayD, Isect &its) {
    const float4 *g = sc.shadeTris + kShadeTriStride * (size_t) tri;
    const float4 g0 = g[0], g1 = g[1], g2 = g[2], h0 = g[3], h1 = g[4], h2 = g[5];
    const float bx = 1 - u - v, by = u, bz = v;
    const V3 p0 = mk(g0.x, g0.y, g0.z), p1 = mk(g1.x, g1.y, g1.z), p2 = mk(g2.x, g2.y, g2.z);
    its.material = __float_as_uint(g0.w) & 0xffffffu;
    its.flags = __float_as_uint(g0.w) >> 24;
    its.emitter = (int32_t) __float_as_uint(g1.w);
    its.shape = __float_as_uint(g2.w);
    its.p = p0 * bx + p1 * by + p2 * bz;
    const V3 side1 = p1 - p0, side2 = p2 - p0;
    V3 faceNormal = cross(side1, side2);
    const float len = length(faceNormal);
    if (!isZero(faceNormal)) faceNormal = faceNormal / len;
    V3 dpdu = side1;
    its.dpdv = side2;
    its.uvx = by; its.uvy = bz;
    if (its.flags & 2u) {       // TriMesh::computeUVTangents (trimesh.cpp:683-735) and interpolated texture coordinates
        const float4 q0 = g[6], q1 = g[7];      // uv0, uv1 | uv2: the tail of the record's own line
        const float v0[3] = { p0.x, p0.y, p0.z }, v1[3] = { p1.x, p1.y, p1.z }, v2[3] = { p2.x, p2.y, p2.z };
        const float t0[2] = { q0.x, q0.y }, t1[2] = { q0.z, q0.w }, t2[2] = { q1.x, q1.y };
        float du[3], dv[3];
        uvTangents(v0, v1, v2, t0, t1, t2, du, dv);
        dpdu = mk(du[0], du[1], du[2]);
        its.dpdv = mk(dv[0], dv[1], dv[2]);
        its.uvx = q0.x * bx + q0.z * by + q1.x * bz;
        its.uvy = q0.y * bx + q0.w * by + q1.y * bz;
    }
    V3 shN;
    if (its.flags & 1u) {
        const V3 n0 = mk(h0.x, h0.y, h0.z), n1 = mk(h1.x, h1.y, h1.z), n2 = mk(h2.x, h2.y, h2.z);
        shN = normalize(n0 * bx + n1 * by + n2 * bz);
        if (dot(faceNormal, shN) < 0) faceNormal = -faceNormal;
    } else {
        shN = faceNormal;
    }
    its.geoN = faceNormal;
    its.dpdu = dpdu;
    its.sh.n = shN;
    its.sh.s = normalize(dpdu - shN * dot(shN, dpdu));
    its.sh.t = cross(shN, its.sh.s);
    its.wi = toLocal(its.sh, -rayD);
}

// ---- warps (src/libcore/warp.cpp:51-60,84-110) -----------------------------------------------------------
MTS_DEV void squareToUniformDiskConcentric(float sx, float sy, float &px, float &py) {
    const float r1 = 2.0f * sx - 1.0f, r2 = 2.0f * sy - 1.0f;
    float phi, r;
    if (r1 == 0 && r2 == 0) { r = phi = 0; }
    else if (r1 * r1 > r2 * r2) { r = r1; phi = (kPi / 4.0f) * (r2 / r1); }
    else { r = r2; phi = (kPi / 2.0f) - (r1 / r2) * (kPi / 4.0f); }
    float s, c;
    sincosM(phi, &s, &c);
    px = r * c; py = r * s;
}

MTS_DEV V3 squareToCosineHemisphere(float sx, float sy) {
    float px, py;
    squareToUniformDiskConcentric(sx, sy, px, py);
    float z = safeSqrt(1.0f - px * px - py * py);
    if (z == 0) z = 1e-10f;
    return mk(px, py, z);
}

// ---- math::erf / erfinv / hypot2 (src/libcore/math.cpp:25-86). Exponentials and logarithms go through expRef / logRef:
//      the reference's fastexp / fastlog are the double-precision functions rounded to float (math.h:185-209), and its
//      std::exp on a float is glibc's correctly rounded expf; the device's single-precision versions differ from both in
//      the last bit, which is enough to flip the exit test of the Newton iteration below ----
MTS_DEV float erfinvM(float x) {
    float w = -logRef((1.0f - x) * (1.0f + x));
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
        w = sqrtf(w) - 3.0f;
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

MTS_DEV float erfM(float x) {
    const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
    const float sign = copysignf(1.0f, x);
    x = fabsf(x);
    const float t = 1.0f / (1.0f + p * x);
    const float y = 1.0f - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * expRef(-x * x);
    return sign * y;
}

MTS_DEV float hypot2(float a, float b) {
    float r;
    if (fabsf(a) > fabsf(b)) { r = b / a; r = fabsf(a) * sqrtf(1.0f + r * r); }
    else if (b != 0.0f) { r = a / b; r = fabsf(b) * sqrtf(1.0f + r * r); }
    else r = 0.0f;
    return r;
}

// ---- Fresnel (src/libcore/util.cpp:651-681, 739-761) -----------------------------------------------------
MTS_DEV float fresnelDielectricExt(float cosThetaI_, float &cosThetaT_, float eta) {
    if (eta == 1) { cosThetaT_ = -cosThetaI_; return 0.0f; }
    const float scale = (cosThetaI_ > 0) ? 1 / eta : eta;
    const float cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) { cosThetaT_ = 0.0f; return 1.0f; }
    const float cosThetaI = fabsf(cosThetaI_);
    const float cosThetaT = sqrtf(cosThetaTSqr);
    const float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    const float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}

MTS_DEV float fresnelConductorExact1(float cosThetaI, float eta, float k) {
    const float cosThetaI2 = cosThetaI * cosThetaI, sinThetaI2 = 1 - cosThetaI2, sinThetaI4 = sinThetaI2 * sinThetaI2;
    const float temp1 = eta * eta - k * k - sinThetaI2;
    const float a2pb2 = safeSqrt(temp1 * temp1 + k * k * eta * eta * 4.0f);
    const float a = safeSqrt((a2pb2 + temp1) * 0.5f);
    const float term1 = a2pb2 + cosThetaI2, term2 = a * (2 * cosThetaI);
    const float Rs2 = (term1 - term2) / (term1 + term2);
    const float term3 = a2pb2 * cosThetaI2 + sinThetaI4, term4 = term2 * sinThetaI2;
    const float Rp2 = Rs2 * (term3 - term4) / (term3 + term4);
    return 0.5f * (Rp2 + Rs2);
}

MTS_DEV V3 fresnelConductorExact(float cosThetaI, V3 eta, V3 k) {
    return mk(fresnelConductorExact1(cosThetaI, eta.x, k.x), fresnelConductorExact1(cosThetaI, eta.y, k.y),
              fresnelConductorExact1(cosThetaI, eta.z, k.z));
}

// ---- evalCubicInterp1D on [0,1] (src/libcore/spline.cpp:23-60) ---------------------------------------------
MTS_DEV float evalCubicInterp1D01(float x, const float *values, uint32_t size) {
    if (!(x >= 0.0f && x <= 1.0f)) return 0.0f;
    float t = ((x - 0.0f) * (float) (size - 1)) / (1.0f - 0.0f);
    uint32_t k = (uint32_t) t;
    if (k > size - 2) k = size - 2;
    const float f0 = values[k], f1 = values[k + 1];
    float d0, d1;
    if (k > 0) d0 = 0.5f * (values[k + 1] - values[k - 1]); else d0 = values[k + 1] - values[k];
    if (k + 2 < size) d1 = 0.5f * (values[k + 2] - values[k]); else d1 = values[k + 1] - values[k];
    t = t - (float) k;
    const float t2 = t * t, t3 = t2 * t;
    return (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
}

// One dimension of evalCubicInterp2D on [0,1] (src/libcore/spline.cpp:242-287): the knot and the four node weights
// around it, derivative stencils folded in. False outside [0,1].
MTS_DEV bool cubicWeights01(float p, uint32_t size, float (&w)[4], uint32_t &knot) {
    if (!(p >= 0.0f && p <= 1.0f)) return false;
    float t = ((p - 0.0f) * (float) (size - 1)) / (1.0f - 0.0f);
    knot = (uint32_t) t;
    if (knot > size - 2) knot = size - 2;
    t = t - (float) knot;
    const float t2 = t * t, t3 = t2 * t;
    w[0] = 0.0f; w[1] = 2 * t3 - 3 * t2 + 1; w[2] = -2 * t3 + 3 * t2; w[3] = 0.0f;
    const float d0 = t3 - 2 * t2 + t, d1 = t3 - t2;
    if (knot > 0) { w[2] += 0.5f * d0; w[0] -= 0.5f * d0; } else { w[2] += d0; w[1] -= d0; }
    if (knot + 2 < size) { w[3] += 0.5f * d1; w[1] -= 0.5f * d1; } else { w[2] += d1; w[1] -= d1; }
    return true;
}
// evalCubicInterp2D on [0,1]^2 (spline.cpp:236-304); values[y * sx + x]
MTS_DEV float evalCubicInterp2D01(float px, float py, const float *values, uint32_t sx, uint32_t sy) {
    float wx[4], wy[4];
    uint32_t kx, ky;
    if (!cubicWeights01(px, sx, wx, kx)) return 0.0f;
    if (!cubicWeights01(py, sy, wy, ky)) return 0.0f;
    float result = 0.0f;
    for (int y = -1; y <= 2; ++y)
        for (int x = -1; x <= 2; ++x) {
            const float wxy = wx[x + 1] * wy[y + 1];
            if (wxy == 0) continue;     // (the nodes beyond the table's edge carry zero weight)
            result += values[(size_t) ((int) ky + y) * sx + (size_t) ((int) kx + x)] * wxy;
        }
    return result;
}

// A roughplastic whose roughness is a texture keeps the transmittance table two-dimensional (variant bit 1): the block
// thetaSamples, alphaSamples, alphaMin, alphaMax, T[alpha][theta], diffuse exterior [alpha], diffuse interior [alpha]
MTS_DEV bool rtransTextured(const DevMaterial &m) { return (m.variant & 2) != 0; }
// (the roughness as the microfacet distribution keeps it: distr.getAlpha(), clamped from below, microfacet.h:67-74)
MTS_DEV float rtransWarpedAlpha(const float *block, float alpha) { return powM((fmaxf(alpha, 1e-4f) - block[2]) / (block[3] - block[2]), 0.25f); }

// RoughTransmittance::eval with the index of refraction fixed: fully reduced (src/bsdfs/rtrans.h:183-194,233) or, for
// a textured roughness, still a function of alpha (rtrans.h:195-205); m.alphaU is the roughness at this intersection
MTS_DEV float rtransEval(const DevScene &sc, const DevMaterial &m, float cosTheta) {
    const float warped = powM(fabsf(cosTheta), 0.25f);
    if (!(cosTheta >= 0)) return 0.f;
    const float *table = sc.rtrans + m.rtransOffset;
    float result;
    if (rtransTextured(m)) result = evalCubicInterp2D01(warped, rtransWarpedAlpha(table, m.alphaU), table + 4, (uint32_t) table[0], (uint32_t) table[1]);
    else result = evalCubicInterp1D01(warped, table, m.rtransN);
    return fminf(1.0f, fmaxf(0.0f, result));
}
// m_internalRoughTransmittance->evalDiffuse(alpha) (rtrans.h:249-259)
MTS_DEV float rtransDiffuseInterior(const DevScene &sc, const DevMaterial &m) {
    if (!rtransTextured(m)) return m.fdrInt;
    const float *table = sc.rtrans + m.rtransOffset;
    const uint32_t nTheta = (uint32_t) table[0], nAlpha = (uint32_t) table[1];
    const float result = evalCubicInterp1D01(rtransWarpedAlpha(table, m.alphaU), table + 4 + nTheta * nAlpha + nAlpha, nAlpha);
    return fminf(1.0f, fmaxf(0.0f, result));
}

// ---- MicrofacetDistribution (src/bsdfs/microfacet.h) -------------------------------------------------------
struct Microfacet {
    int type;
    float alphaU, alphaV;
    bool sampleVisible;

    MTS_DEV void init(int type_, float au, float av, bool sv) {
        type = type_; alphaU = fmaxf(au, 1e-4f); alphaV = fmaxf(av, 1e-4f); sampleVisible = sv;    // :67-74
    }
    MTS_DEV bool isotropic() const { return alphaU == alphaV; }
    MTS_DEV void scaleAlpha(float v) { alphaU *= v; alphaV *= v; }     // :178-183

    MTS_DEV float eval(V3 m) const {   // :191-234
        if (m.z <= 0) return 0.0f;
        const float cosTheta2 = m.z * m.z;
        const float beckmannExponent = ((m.x * m.x) / (alphaU * alphaU) + (m.y * m.y) / (alphaV * alphaV)) / cosTheta2;
        float result;
        if (type == 0) {
            result = expRef(-beckmannExponent) / (kPi * alphaU * alphaV * cosTheta2 * cosTheta2);
        } else {
            const float root = (1.0f + beckmannExponent) * cosTheta2;
            result = 1.0f / (kPi * alphaU * alphaV * root * root);
        }
        if (result * m.z < 1e-20f) result = 0;
        return result;
    }

    MTS_DEV float projectRoughness(V3 v) const {   // :541-551
        const float invSinTheta2 = 1 / (1.0f - v.z * v.z);
        if (isotropic() || invSinTheta2 <= 0) return alphaU;
        const float cosPhi2 = v.x * v.x * invSinTheta2, sinPhi2 = v.y * v.y * invSinTheta2;
        return sqrtf(cosPhi2 * alphaU * alphaU + sinPhi2 * alphaV * alphaV);
    }

    MTS_DEV float smithG1(V3 v, V3 m) const {   // :477-514
        if (dot(v, m) * v.z <= 0) return 0.0f;
        const float temp = 1 - v.z * v.z;
        const float tanTheta = temp <= 0.0f ? 0.0f : fabsf(sqrtf(temp) / v.z);   // |Frame::tanTheta|
        if (tanTheta == 0.0f) return 1.0f;
        const float alpha = projectRoughness(v);
        if (type == 0) {
            const float a = 1.0f / (alpha * tanTheta);
            if (a >= 1.6f) return 1.0f;
            const float aSqr = a * a;
            return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
        } else {
            const float root = alpha * tanTheta;
            return 2.0f / (1.0f + hypot2(1.0f, root));
        }
    }
    MTS_DEV float G(V3 wi, V3 wo, V3 m) const { return smithG1(wi, m) * smithG1(wo, m); }

    MTS_DEV float pdfVisible(V3 wi, V3 m) const {   // :462-466
        if (wi.z == 0) return 0.0f;
        return smithG1(wi, m) * fabsf(dot(wi, m)) * eval(m) / fabsf(wi.z);
    }
    MTS_DEV float pdfAll(V3 m) const { return eval(m) * m.z; }
    MTS_DEV float pdf(V3 wi, V3 m) const { return sampleVisible ? pdfVisible(wi, m) : pdfAll(m); }

    MTS_DEV V3 sampleAll(float sx, float sy, float &pdf_) const {   // :287-395
        float cosThetaM, sinPhiM, cosPhiM, alphaSqr;
        if (isotropic()) {
            sincosM((2.0f * kPi) * sy, &sinPhiM, &cosPhiM);
            alphaSqr = alphaU * alphaU;
        } else {
            const float phiM = atanM(alphaV / alphaU * tanM(kPi + 2 * kPi * sy)) + kPi * floorf(2 * sy + 0.5f);
            sincosM(phiM, &sinPhiM, &cosPhiM);
            const float cosSc = cosPhiM / alphaU, sinSc = sinPhiM / alphaV;
            alphaSqr = 1.0f / (cosSc * cosSc + sinSc * sinSc);
        }
        if (type == 0) {
            const float tanThetaMSqr = alphaSqr * -logRef(1.0f - sx);
            cosThetaM = 1.0f / sqrtf(1.0f + tanThetaMSqr);
            pdf_ = (1.0f - sx) / (kPi * alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM);
        } else {
            const float tanThetaMSqr = alphaSqr * sx / (1.0f - sx);
            cosThetaM = 1.0f / sqrtf(1.0f + tanThetaMSqr);
            const float temp = 1 + tanThetaMSqr / alphaSqr;
            pdf_ = kInvPi / (alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM * temp * temp);
        }
        if (pdf_ < 1e-20f) pdf_ = 0;
        const float sinThetaM = sqrtf(fmaxf(0.0f, 1 - cosThetaM * cosThetaM));
        return mk(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
    }

    MTS_DEV void sampleVisible11(float thetaI, float sx, float sy, float &slopeX, float &slopeY) const {   // :573-697
        const float SQRT_PI_INV = 1 / sqrtf(kPi);
        if (type == 0) {
            if (thetaI < 1e-4f) {
                float sinPhi, cosPhi;
                const float r = sqrtf(-logRef(1.0f - sx));
                sincosM(2 * kPi * sy, &sinPhi, &cosPhi);
                slopeX = r * cosPhi; slopeY = r * sinPhi;
                return;
            }
            const float tanThetaI = tanM(thetaI);
            const float cotThetaI = 1 / tanThetaI;
            float a = -1, c = erfM(cotThetaI);
            const float sample_x = fmaxf(sx, 1e-6f);
            const float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
            float b = c - (1 + c) * powM(1 - sample_x, fit);
            const float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * expRef(-cotThetaI * cotThetaI));
            int it = 0;
            while (++it < 10) {
                if (!(b >= a && b <= c)) b = 0.5f * (a + c);
                const float invErf = erfinvM(b);
                const float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * expRef(-invErf * invErf)) - sample_x;
                const float derivative = normalization * (1 - invErf * tanThetaI);
                if (fabsf(value) < 1e-5f) break;
                if (value > 0) c = b; else a = b;
                b -= value / derivative;
            }
            slopeX = erfinvM(b);
            slopeY = erfinvM(2.0f * fmaxf(sy, 1e-6f) - 1.0f);
        } else {
            if (thetaI < 1e-4f) {
                float sinPhi, cosPhi;
                const float r = safeSqrt(sx / (1 - sx));
                sincosM(2 * kPi * sy, &sinPhi, &cosPhi);
                slopeX = r * cosPhi; slopeY = r * sinPhi;
                return;
            }
            const float tanThetaI = tanM(thetaI);
            const float a = 1 / tanThetaI;
            const float G1 = 2.0f / (1.0f + safeSqrt(1.0f + 1.0f / (a * a)));
            float A = 2.0f * sx / G1 - 1.0f;
            if (fabsf(A) == 1) A -= copysignf(1.0f, A) * kEpsilon;
            const float tmp = 1.0f / (A * A - 1.0f);
            const float B = tanThetaI;
            const float D = safeSqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
            const float slope_x_1 = B * tmp - D;
            const float slope_x_2 = B * tmp + D;
            slopeX = (A < 0.0f || slope_x_2 > 1.0f / tanThetaI) ? slope_x_1 : slope_x_2;
            float S;
            if (sy > 0.5f) { S = 1.0f; sy = 2.0f * (sy - 0.5f); }
            else { S = -1.0f; sy = 2.0f * (0.5f - sy); }
            const float z = (sy * (sy * (sy * (-0.365728915865723f) + 0.790235037209296f) - 0.424965825137544f) + 0.000152998850436920f) /
                            (sy * (sy * (sy * (sy * 0.169507819808272f - 0.397203533833404f) - 0.232500544458471f) + 1.0f) - 0.539825872510702f);
            slopeY = S * z * sqrtf(1.0f + slopeX * slopeX);
        }
    }

    MTS_DEV V3 sampleVisibleN(V3 _wi, float sx, float sy) const {   // :421-459
        const V3 wi = normalize(mk(alphaU * _wi.x, alphaV * _wi.y, _wi.z));
        float theta = 0, phi = 0;
        if (wi.z < 0.99999f) { theta = acosM(wi.z); phi = atan2M(wi.y, wi.x); }
        float sinPhi, cosPhi;
        sincosM(phi, &sinPhi, &cosPhi);
        float slx, sly;
        sampleVisible11(theta, sx, sy, slx, sly);
        float rx = cosPhi * slx - sinPhi * sly, ry = sinPhi * slx + cosPhi * sly;
        rx *= alphaU; ry *= alphaV;
        const float normalization = 1.0f / sqrtf(rx * rx + ry * ry + 1.0f);
        return mk(-rx * normalization, -ry * normalization, normalization);
    }

    MTS_DEV V3 sample(V3 wi, float sx, float sy, float &pdf_) const {   // :240-249
        if (sampleVisible) { const V3 m = sampleVisibleN(wi, sx, sy); pdf_ = pdfVisible(wi, m); return m; }
        return sampleAll(sx, sy, pdf_);
    }
    MTS_DEV V3 sample(V3 wi, float sx, float sy) const {   // :255-264
        if (sampleVisible) return sampleVisibleN(wi, sx, sy);
        float p; return sampleAll(sx, sy, p);
    }
};

MTS_DEV V3 reflectM(V3 wi, V3 m) { return 2 * dot(wi, m) * m - wi; }
MTS_DEV float cosHemiPdf(V3 d) { return kInvPi * d.z; }

// ---- BSDF::eval (value includes the cosine factor) -----------------------------------------------------------
// RD = false compiles the rough dielectric out (kernels of scenes that have none: fewer live registers).
template <bool RD = true>
MTS_DEV V3 bsdfEval(const DevScene &sc, const DevMaterial &m, V3 wi, V3 wo) {
    const V3 zero = mk(0.0f);
    if (m.type == 3 || m.type == 5) return zero;                // dielectric.cpp:228-253, conductor.cpp:225-239: no solid-angle lobe
    if (RD && m.type == 7) {                                    // roughdielectric.cpp:276-346
        if (wi.z == 0) return zero;
        const bool reflect = wi.z * wo.z > 0;
        const float etaSide = wi.z > 0 ? m.eta : m.invEta;
        V3 H = reflect ? normalize(wo + wi) : normalize(wi + wo * etaSide);
        H = H * copysignf(1.0f, H.z);
        Microfacet distr; distr.init(m.distribution, m.alphaU, m.alphaV, m.sampleVisible != 0);
        const float D = distr.eval(H);
        if (D == 0) return zero;
        float ct;
        const float F = fresnelDielectricExt(dot(wi, H), ct, m.eta);
        const float G = distr.G(wi, wo, H);
        if (reflect) return ld3(m.specular) * (F * D * G / (4.0f * fabsf(wi.z)));
        const float sqrtDenom = dot(wi, H) + etaSide * dot(wo, H);
        const float value = ((1 - F) * D * G * etaSide * etaSide * dot(wi, H) * dot(wo, H)) / (wi.z * sqrtDenom * sqrtDenom);
        const float factor = wi.z > 0 ? m.invEta : m.eta;      // ERadiance
        return ld3(m.transmittance) * fabsf(value * factor * factor);
    }
    if (wi.z <= 0 || wo.z <= 0) return zero;
    const V3 diff = ld3(m.diffuse), spec = ld3(m.specular);
    switch (m.type) {
        case 0:                                                 // diffuse.cpp:110-118
            if (!(m.ownFlags & EDiffuseReflection)) return zero;
            return diff * (kInvPi * wo.z);
        case 1: {                                               // roughconductor.cpp:257-294
            const V3 H = normalize(wo + wi);
            Microfacet distr; distr.init(m.distribution, m.alphaU, m.alphaV, m.sampleVisible != 0);
            const float D = distr.eval(H);
            if (D == 0) return zero;
            const V3 F = fresnelConductorExact(dot(wi, H), ld3(m.condEta), ld3(m.condK)) * spec;
            const float G = distr.G(wi, wo, H);
            const float model = D * G / (4.0f * wi.z);
            return F * model;
        }
        case 2: {                                               // roughplastic.cpp:326-382
            Microfacet distr; distr.init(m.distribution, m.alphaU, m.alphaU, m.sampleVisible != 0);
            V3 result = zero;
            {
                const V3 H = normalize(wo + wi);
                const float D = distr.eval(H);
                float ct;
                const float F = fresnelDielectricExt(dot(wi, H), ct, m.eta);
                const float G = distr.G(wi, wo, H);
                const float value = F * D * G / (4.0f * wi.z);
                result = result + spec * value;
            }
            {
                V3 d = diff;
                const float T12 = rtransEval(sc, m, wi.z);
                const float T21 = rtransEval(sc, m, wo.z);
                const float Fdr = 1 - rtransDiffuseInterior(sc, m);
                if (m.variant & 1) d = d / (mk(1.0f) - d * Fdr);
                else d = d / (1 - Fdr);
                result = result + d * (kInvPi * wo.z * T12 * T21 * m.invEta2);
            }
            return result;
        }
        case 6: {                                               // plastic.cpp:243-276, the diffuse lobe
            float ct;
            const float Fi = fresnelDielectricExt(wi.z, ct, m.eta), Fo = fresnelDielectricExt(wo.z, ct, m.eta);
            V3 d = diff;
            if (m.variant) d = d / (mk(1.0f) - d * m.fdrInt);
            else d = d / (1 - m.fdrInt);
            return d * (cosHemiPdf(wo) * m.invEta2 * (1 - Fi) * (1 - Fo));
        }
        default: {                                              // ward.cpp:178-226
            V3 result = zero;
            const V3 H = wi + wo;
            float factor1;
            if (m.variant == 0) factor1 = 1.0f / (4.0f * kPi * m.alphaU * m.alphaV * sqrtf(wi.z * wo.z));
            else if (m.variant == 1) factor1 = 1.0f / (4.0f * kPi * m.alphaU * m.alphaV * wi.z * wo.z);
            else {
                // std::pow(Float, int) runs in double in the reference; so does this quotient
                const double c = (double) H.z;
                factor1 = (float) ((double) dot(H, H) / ((double) (kPi * m.alphaU * m.alphaV) * (c * c * c * c)));
            }
            const float factor2 = H.x / m.alphaU, factor3 = H.y / m.alphaV;
            const float exponent = -(factor2 * factor2 + factor3 * factor3) / (H.z * H.z);
            const float specRef = factor1 * expRef(exponent);
            if (specRef > 1e-10f) result = result + spec * specRef;
            result = result + diff * kInvPi;
            return result * wo.z;
        }
    }
}

// ---- BSDF::pdf -----------------------------------------------------------------------------------------------
template <bool RD = true>
MTS_DEV float bsdfPdf(const DevScene &sc, const DevMaterial &m, V3 wi, V3 wo) {
    if (m.type == 3 || m.type == 5) return 0.0f;
    if (RD && m.type == 7) {                                    // roughdielectric.cpp:348-414
        const bool reflect = wi.z * wo.z > 0;
        const float etaSide = wi.z > 0 ? m.eta : m.invEta;
        V3 H;
        float dwh_dwo;
        if (reflect) {
            H = normalize(wo + wi);
            dwh_dwo = 1.0f / (4.0f * dot(wo, H));
        } else {
            H = normalize(wi + wo * etaSide);
            const float sqrtDenom = dot(wi, H) + etaSide * dot(wo, H);
            dwh_dwo = (etaSide * etaSide * dot(wo, H)) / (sqrtDenom * sqrtDenom);
        }
        H = H * copysignf(1.0f, H.z);
        Microfacet sampleDistr; sampleDistr.init(m.distribution, m.alphaU, m.alphaV, m.sampleVisible != 0);
        if (!sampleDistr.sampleVisible) sampleDistr.scaleAlpha(1.2f - 0.2f * sqrtf(fabsf(wi.z)));
        float prob = sampleDistr.pdf(wi * copysignf(1.0f, wi.z), H);
        float ct;
        const float F = fresnelDielectricExt(dot(wi, H), ct, m.eta);
        prob *= reflect ? F : (1 - F);
        return fabsf(prob * dwh_dwo);
    }
    if (wi.z <= 0 || wo.z <= 0) return 0.0f;
    switch (m.type) {
        case 0:                                                 // diffuse.cpp:120-127
            if (!(m.ownFlags & EDiffuseReflection)) return 0.0f;
            return cosHemiPdf(wo);
        case 1: {                                               // roughconductor.cpp:296-321
            const V3 H = normalize(wo + wi);
            Microfacet distr; distr.init(m.distribution, m.alphaU, m.alphaV, m.sampleVisible != 0);
            if (distr.sampleVisible) return distr.eval(H) * distr.smithG1(wi, H) / (4.0f * wi.z);
            return distr.pdf(wi, H) / (4 * fabsf(dot(wo, H)));
        }
        case 2: {                                               // roughplastic.cpp:384-437
            Microfacet distr; distr.init(m.distribution, m.alphaU, m.alphaU, m.sampleVisible != 0);
            const V3 H = normalize(wo + wi);
            float probSpecular = 1 - rtransEval(sc, m, wi.z);
            probSpecular = (probSpecular * m.specSamplingWeight) /
                (probSpecular * m.specSamplingWeight + (1 - probSpecular) * (1 - m.specSamplingWeight));
            const float probDiffuse = 1 - probSpecular;
            const float dwh_dwo = 1.0f / (4.0f * dot(wo, H));
            const float prob = distr.pdf(wi, H);
            float result = prob * dwh_dwo * probSpecular;
            result += probDiffuse * cosHemiPdf(wo);
            return result;
        }
        case 6: {                                               // plastic.cpp:278-306
            float ct;
            const float Fi = fresnelDielectricExt(wi.z, ct, m.eta);
            const float probSpecular = (Fi * m.specSamplingWeight) / (Fi * m.specSamplingWeight + (1 - Fi) * (1 - m.specSamplingWeight));
            return cosHemiPdf(wo) * (1 - probSpecular);
        }
        default: {                                              // ward.cpp:228-264
            const V3 H = normalize(wi + wo);
            const double c = (double) H.z;
            const float factor1 = (float) (1.0 / ((double) (4.0f * kPi * m.alphaU * m.alphaV * dot(H, wi)) * (c * c * c)));
            const float factor2 = H.x / m.alphaU, factor3 = H.y / m.alphaV;
            const float exponent = -(factor2 * factor2 + factor3 * factor3) / (H.z * H.z);
            const float specProb = factor1 * expRef(exponent);
            const float diffuseProb = cosHemiPdf(wo);
            return m.specSamplingWeight * specProb + (1 - m.specSamplingWeight) * diffuseProb;
        }
    }
}

// ---- BSDF::sample(bRec, pdf, sample) -----------------------------------------------------------------------------
// su: the number a BSDF flagged EUsesSampler (roughdielectric) takes from the sampler itself, bRec.sampler->next1D()
MTS_DEV bool bsdfUsesSampler(const DevMaterial &m) { return m.type == 7; }
template <bool RD = true>
MTS_DEV V3 bsdfSample(const DevScene &sc, const DevMaterial &m, V3 wi, float sx, float sy, float su, V3 &wo, float &pdf, float &eta,
                      uint32_t &sampledType) {
    const V3 zero = mk(0.0f);
    pdf = 0; eta = 1.0f; sampledType = 0; wo = zero;
    const V3 diff = ld3(m.diffuse), spec = ld3(m.specular);
    switch (m.type) {
        case 0: {                                               // diffuse.cpp:140-150
            if (!(m.ownFlags & EDiffuseReflection) || wi.z <= 0) return zero;
            wo = squareToCosineHemisphere(sx, sy);
            sampledType = EDiffuseReflection;
            pdf = cosHemiPdf(wo);
            return diff;
        }
        case 1: {                                               // roughconductor.cpp:369-415
            if (wi.z < 0) return zero;
            Microfacet distr; distr.init(m.distribution, m.alphaU, m.alphaV, m.sampleVisible != 0);
            const V3 mm = distr.sample(wi, sx, sy, pdf);
            if (pdf == 0) return zero;
            wo = reflectM(wi, mm);
            sampledType = EGlossyReflection;
            if (wo.z <= 0) return zero;
            const V3 F = fresnelConductorExact(dot(wi, mm), ld3(m.condEta), ld3(m.condK)) * spec;
            float weight;
            if (distr.sampleVisible) weight = distr.smithG1(wo, mm);
            else weight = distr.eval(mm) * distr.G(wi, wo, mm) * dot(wi, mm) / (pdf * wi.z);
            pdf /= 4.0f * dot(wo, mm);
            return F * weight;
        }
        case 2: {                                               // roughplastic.cpp:439-501
            if (wi.z <= 0) return zero;
            bool choseSpecular = true;
            Microfacet distr; distr.init(m.distribution, m.alphaU, m.alphaU, m.sampleVisible != 0);
            float probSpecular = 1 - rtransEval(sc, m, wi.z);
            probSpecular = (probSpecular * m.specSamplingWeight) /
                (probSpecular * m.specSamplingWeight + (1 - probSpecular) * (1 - m.specSamplingWeight));
            if (sy < probSpecular) sy /= probSpecular;
            else { sy = (sy - probSpecular) / (1 - probSpecular); choseSpecular = false; }
            if (choseSpecular) {
                const V3 mm = distr.sample(wi, sx, sy);
                wo = reflectM(wi, mm);
                sampledType = EGlossyReflection;
                if (wo.z <= 0) return zero;
            } else {
                sampledType = EDiffuseReflection;
                wo = squareToCosineHemisphere(sx, sy);
            }
            pdf = bsdfPdf<false>(sc, m, wi, wo);
            if (pdf == 0) return zero;
            return bsdfEval<false>(sc, m, wi, wo) / pdf;
        }
        case 3: {                                               // dielectric.cpp:277-308 (both lobes, ERadiance)
            float cosThetaT;
            const float F = fresnelDielectricExt(wi.z, cosThetaT, m.eta);
            if (sx <= F) {
                sampledType = EDeltaReflection;
                wo = mk(-wi.x, -wi.y, wi.z);
                pdf = F;
                return spec;
            } else {
                sampledType = EDeltaTransmission;
                const float scale = -(cosThetaT < 0 ? m.invEta : m.eta);
                wo = mk(scale * wi.x, scale * wi.y, cosThetaT);
                eta = cosThetaT < 0 ? m.eta : m.invEta;
                pdf = 1 - F;
                const float factor = cosThetaT < 0 ? m.invEta : m.eta;
                return ld3(m.transmittance) * (factor * factor);
            }
        }
        case 7: {                                               // roughdielectric.cpp:511-613 (both lobes, ERadiance)
            if (!RD) return zero;
            Microfacet distr; distr.init(m.distribution, m.alphaU, m.alphaV, m.sampleVisible != 0);
            Microfacet sampleDistr = distr;
            if (!distr.sampleVisible) sampleDistr.scaleAlpha(1.2f - 0.2f * sqrtf(fabsf(wi.z)));
            float microfacetPdf;
            const V3 mm = sampleDistr.sample(wi * copysignf(1.0f, wi.z), sx, sy, microfacetPdf);
            if (microfacetPdf == 0) return zero;
            pdf = microfacetPdf;
            float cosThetaT;
            const float F = fresnelDielectricExt(dot(wi, mm), cosThetaT, m.eta);
            V3 weight = mk(1.0f);
            const bool sampleReflection = !(su > F);
            pdf *= sampleReflection ? F : 1 - F;
            float dwh_dwo;
            if (sampleReflection) {
                wo = reflectM(wi, mm);
                sampledType = EGlossyReflection;
                if (wi.z * wo.z <= 0) return zero;
                weight = weight * spec;
                dwh_dwo = 1.0f / (4.0f * dot(wo, mm));
            } else {
                if (cosThetaT == 0) return zero;
                const float e = cosThetaT < 0 ? 1 / m.eta : m.eta;      // refract(wi, m, eta, cosThetaT) (util.cpp:767-772)
                wo = mm * (dot(wi, mm) * e + cosThetaT) - wi * e;
                eta = cosThetaT < 0 ? m.eta : m.invEta;
                sampledType = EGlossyTransmission;
                if (wi.z * wo.z >= 0) return zero;
                const float factor = cosThetaT < 0 ? m.invEta : m.eta;
                weight = weight * (ld3(m.transmittance) * (factor * factor));
                const float sqrtDenom = dot(wi, mm) + eta * dot(wo, mm);
                dwh_dwo = (eta * eta * dot(wo, mm)) / (sqrtDenom * sqrtDenom);
            }
            if (distr.sampleVisible) weight = weight * distr.smithG1(wo, mm);
            else weight = weight * fabsf(distr.eval(mm) * distr.G(wi, wo, mm) * dot(wi, mm) / (microfacetPdf * wi.z));
            pdf *= fabsf(dwh_dwo);
            return weight;
        }
        case 5: {                                               // conductor.cpp:273-288
            if (wi.z <= 0) return zero;
            sampledType = EDeltaReflection;
            wo = mk(-wi.x, -wi.y, wi.z);
            pdf = 1;
            return spec * fresnelConductorExact(wi.z, ld3(m.condEta), ld3(m.condK));
        }
        case 6: {                                               // plastic.cpp:373-440
            if (wi.z <= 0) return zero;
            float ct;
            const float Fi = fresnelDielectricExt(wi.z, ct, m.eta);
            const float probSpecular = (Fi * m.specSamplingWeight) / (Fi * m.specSamplingWeight + (1 - Fi) * (1 - m.specSamplingWeight));
            if (sx < probSpecular) {
                sampledType = EDeltaReflection;
                wo = mk(-wi.x, -wi.y, wi.z);
                pdf = probSpecular;
                return spec * Fi / probSpecular;
            }
            sampledType = EDiffuseReflection;
            wo = squareToCosineHemisphere((sx - probSpecular) / (1 - probSpecular), sy);
            const float Fo = fresnelDielectricExt(wo.z, ct, m.eta);
            V3 d = diff;
            if (m.variant) d = d / (mk(1.0f) - d * m.fdrInt);
            else d = d / (1 - m.fdrInt);
            pdf = (1 - probSpecular) * cosHemiPdf(wo);
            return d * (m.invEta2 * (1 - Fi) * (1 - Fo) / (1 - probSpecular));
        }
        default: {                                              // ward.cpp:266-326
            bool choseSpecular = true;
            if (sx <= m.specSamplingWeight) sx /= m.specSamplingWeight;
            else { sx = (sx - m.specSamplingWeight) / (1 - m.specSamplingWeight); choseSpecular = false; }
            if (choseSpecular) {
                float phiH = atanM(m.alphaV / m.alphaU * tanM(2.0f * kPi * sy));
                if (sy > 0.5f) phiH += kPi;
                const float cosPhiH = cosM(phiH);
                const float sinPhiH = safeSqrt(1.0f - cosPhiH * cosPhiH);
                const float thetaH = atanM(safeSqrt(-logRef(sx) /
                    ((cosPhiH * cosPhiH) / (m.alphaU * m.alphaU) + (sinPhiH * sinPhiH) / (m.alphaV * m.alphaV))));
                float sinTheta, cosTheta, sinPhi, cosPhi;
                sincosM(thetaH, &sinTheta, &cosTheta);
                sincosM(phiH, &sinPhi, &cosPhi);
                const V3 H = mk(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
                wo = H * (2.0f * dot(wi, H)) - wi;
                sampledType = EGlossyReflection;
                if (wo.z <= 0.0f) return zero;
            } else {
                wo = squareToCosineHemisphere(sx, sy);
                sampledType = EDiffuseReflection;
            }
            pdf = bsdfPdf<false>(sc, m, wi, wo);
            if (pdf == 0) return zero;
            return bsdfEval<false>(sc, m, wi, wo) / pdf;
        }
    }
}

// ---- TwoSidedBRDF (src/bsdfs/twosided.cpp:108-183) ------------------------------------------------------------
// m holds the record of the hit's material (index `self`). Seen from behind, a two-sided record hands over to its
// back record (loaded over m when it is a different one) and the caller mirrors wi and wo in z.
MTS_DEV void loadMaterial(const DevScene &sc, uint32_t index, DevMaterial &m) {
    const float4 *mp = reinterpret_cast<const float4 *>(sc.materials + index);
    float4 mv[sizeof(DevMaterial) / 16];
#pragma unroll
    for (int k = 0; k < (int) (sizeof(DevMaterial) / 16); ++k) mv[k] = mp[k];
    memcpy(&m, mv, sizeof(DevMaterial));
}

// `self` becomes the index of the record that answers
MTS_DEV bool resolveTwoSided(const DevScene &sc, uint32_t &self, float wiZ, DevMaterial &m) {
    if (!m.twoSided || !(wiZ < 0)) return false;
    if ((uint32_t) m.backMaterial != self) { self = (uint32_t) m.backMaterial; loadMaterial(sc, self, m); }
    return true;
}

// ---- bitmap textures: lookups into the MIP pyramid (include/mitsuba/render/mipmap.h:503-834) -------------------------
struct MipRef { const DevScene *sc; uint32_t first, levels, bcu, bcv; float maxAnisotropy; };

MTS_DEV int wrapTexel(int x, int n, uint32_t bc, int &constant) {
    if (x >= 0 && x < n) return x;
    switch (bc) {
        case 1: { const int r = x % n; return r < 0 ? r + n : r; }                          // repeat
        case 0: return min(n - 1, max(0, x));                                              // clamp
        case 2: { int r = x % (2 * n); if (r < 0) r += 2 * n; return r >= n ? 2 * n - r - 1 : r; }     // mirror
        case 3: constant = 0; return 0;
        default: constant = 1; return 0;
    }
}

MTS_DEV V3 mipTexel(const MipRef &m, uint32_t level, int x, int y) {        // evalTexel (mipmap.h:503-563)
    const DevMipLevel L = m.sc->mipLevels[m.first + level];
    int constant = -1;
    x = wrapTexel(x, L.w, m.bcu, constant);
    if (constant >= 0) return mk((float) constant);
    y = wrapTexel(y, L.h, m.bcv, constant);
    if (constant >= 0) return mk((float) constant);
    const uint2 t = m.sc->texels[(size_t) L.offset + (size_t) y * (size_t) L.w + (size_t) x];
    union { uint32_t u; _Float16 h[2]; } a, b;
    a.u = t.x; b.u = t.y;
    return mk((float) a.h[0], (float) a.h[1], (float) b.h[0]);
}

MTS_DEV V3 mipBox(const MipRef &m, uint32_t level, float u, float v) {       // evalBox (mipmap.h:566-569)
    const DevMipLevel L = m.sc->mipLevels[m.first + level];
    return mipTexel(m, level, (int) floorf(u * (float) L.w), (int) floorf(v * (float) L.h));
}

MTS_DEV V3 mipBilinear(const MipRef &m, uint32_t level, float uu, float vv) {    // evalBilinear (mipmap.h:575-596)
    if (!isfinite(uu) || !isfinite(vv)) return mk(0.0f);
    if (level >= m.levels) return mipBox(m, m.levels - 1, uu, vv);
    const DevMipLevel L = m.sc->mipLevels[m.first + level];
    const float u = uu * (float) L.w - 0.5f, v = vv * (float) L.h - 0.5f;
    const int xPos = (int) floorf(u), yPos = (int) floorf(v);
    const float dx1 = u - (float) xPos, dx2 = 1.0f - dx1, dy1 = v - (float) yPos, dy2 = 1.0f - dy1;
    return mipTexel(m, level, xPos, yPos) * dx2 * dy2 + mipTexel(m, level, xPos, yPos + 1) * dx2 * dy1
         + mipTexel(m, level, xPos + 1, yPos) * dx1 * dy2 + mipTexel(m, level, xPos + 1, yPos + 1) * dx1 * dy1;
}

MTS_DEV V3 mipEWA(const MipRef &m, uint32_t level, float uu, float vv, float A, float B, float C) {    // evalEWA (mipmap.h:764-834)
    if (!isfinite(A + B + C + uu + vv)) return mk(0.0f);
    if (level >= m.levels) return mipBox(m, m.levels - 1, uu, vv);
    const DevMipLevel L = m.sc->mipLevels[m.first + level];
    const float u = uu * (float) L.w - 0.5f, v = vv * (float) L.h - 0.5f;
    A /= L.ratioX * L.ratioX; B /= L.ratioX * L.ratioY; C /= L.ratioY * L.ratioY;
    const float invDet = 1.0f / (-B * B + 4.0f * A * C), deltaU = 2.0f * sqrtf(C * invDet), deltaV = 2.0f * sqrtf(A * invDet);
    const int u0 = (int) ceilf(u - deltaU), u1 = (int) floorf(u + deltaU), v0 = (int) ceilf(v - deltaV), v1 = (int) floorf(v + deltaV);
    const float As = A * 64, Bs = B * 64, Cs = C * 64;
    V3 result = mk(0.0f);
    float denominator = 0.0f;
    const float ddq = 2 * As, uu0 = (float) u0 - u;
    for (int vt = v0; vt <= v1; ++vt) {
        const float vd = (float) vt - v;
        float q = As * uu0 * uu0 + (Bs * uu0 + Cs * vd) * vd;
        float dq = As * (2 * uu0 + 1) + Bs * vd;
        for (int ut = u0; ut <= u1; ++ut) {
            // the reference's test is on (uint32_t) q, which lets values in (-1, 0) through as entry 0
            if (q < 64.0f && q > -1.0f) {
                const float weight = m.sc->ewaLut[(int) q];
                result = result + mipTexel(m, level, ut, vt) * weight;
                denominator += weight;
            }
            q += dq;
            dq += ddq;
        }
    }
    if (denominator == 0) return mipBilinear(m, level, uu, vv);
    return result / denominator;
}

MTS_DEV float log2Ref(float value) {        // math::log2 (src/libcore/math.cpp:103-106): fastlog times 1 / log(2.0f)
    return logRef(value) * (1.0f / 0.693147182464599609375f);
}

// TMIPMap::eval(uv, d0, d1) (mipmap.h:629-715)
MTS_DEV V3 mipFiltered(const MipRef &m, uint32_t filter, float u, float v, float d0x, float d0y, float d1x, float d1y) {
    if (filter == 0) return mipBox(m, 0, u, v);
    if (filter == 1) return mipBilinear(m, 0, u, v);
    const DevMipLevel L0 = m.sc->mipLevels[m.first];
    const float du0 = d0x * (float) L0.w, dv0 = d0y * (float) L0.h, du1 = d1x * (float) L0.w, dv1 = d1y * (float) L0.h;
    float A = dv0 * dv0 + dv1 * dv1, B = -2.0f * (du0 * dv0 + du1 * dv1), C = du0 * du0 + du1 * du1, F = A * C - B * B * 0.25f;
    const float root = hypot2(A - C, B), Aprime = 0.5f * (A + C - root), Cprime = 0.5f * (A + C + root);
    float majorRadius = Aprime != 0 ? sqrtf(F / Aprime) : 0, minorRadius = Cprime != 0 ? sqrtf(F / Cprime) : 0;
    if (filter == 2 || !(minorRadius > 0) || !(majorRadius > 0) || F < 0) {
        const float level = log2Ref(fmaxf(majorRadius, kEpsilon));
        const int ilevel = (int) floorf(level);
        if (ilevel < 0) return mipBilinear(m, 0, u, v);
        const float a = level - (float) ilevel;
        return mipBilinear(m, (uint32_t) ilevel, u, v) * (1.0f - a) + mipBilinear(m, (uint32_t) ilevel + 1, u, v) * a;
    }
    if (minorRadius * m.maxAnisotropy < majorRadius) {
        minorRadius = majorRadius / m.maxAnisotropy;
        const float theta = 0.5f * atanM(B / (A - C));
        float sinTheta, cosTheta;
        sincosM(theta, &sinTheta, &cosTheta);
        const float a2 = majorRadius * majorRadius, b2 = minorRadius * minorRadius, sinTheta2 = sinTheta * sinTheta,
                    cosTheta2 = cosTheta * cosTheta, sin2Theta = 2 * sinTheta * cosTheta;
        A = a2 * cosTheta2 + b2 * sinTheta2;
        B = (a2 - b2) * sin2Theta;
        C = a2 * sinTheta2 + b2 * cosTheta2;
        F = a2 * b2;
    }
    const float scale = 1.0f / F;
    A *= scale; B *= scale; C *= scale;
    const float level = fmaxf(0.0f, log2Ref(minorRadius));
    const int ilevel = (int) level;
    const float a = level - (float) ilevel;
    if (majorRadius < 1 || !(A > 0 && C > 0)) return mipBilinear(m, (uint32_t) ilevel, u, v);
    return mipEWA(m, (uint32_t) ilevel, u, v, A, B, C) * (1.0f - a) + mipEWA(m, (uint32_t) ilevel + 1, u, v, A, B, C) * a;
}

// BitmapTexture::eval (src/textures/bitmap.cpp:431-454 without partials, :486-499 with) at the transformed uv. A real
// call: the lookups are long, and the shading kernels evaluate textures in many places.
__device__ __attribute__((noinline)) inline float4 bitmapEval(const DevScene *sc, const DevTexture *tp, float x, float y, float4 duv, uint32_t filtered) {
    const DevTexture &t = *tp;
    const uint32_t mode = __float_as_uint(t.color1[2]), filter = mode & 0xffu;
    MipRef m;
    m.sc = sc; m.first = __float_as_uint(t.color1[0]); m.levels = __float_as_uint(t.color1[1]);
    m.bcu = (mode >> 8) & 0xffu; m.bcv = (mode >> 16) & 0xffu; m.maxAnisotropy = t.lineWidth;
    V3 value;
    if (filtered) value = mipFiltered(m, filter, x, y, duv.x, duv.y, duv.z, duv.w);
    else value = filter != 0 ? mipBilinear(m, 0, x, y) : mipBox(m, 0, x, y);
    value = value * ld3(t.color0);
    return make_float4(value.x, value.y, value.z, 0.0f);
}

// evalGradientBilinear (mipmap.h:601-626) of level 0, times the texture's factor
__device__ __attribute__((noinline)) inline void bitmapGradient(const DevScene *sc, const DevTexture *tp, float uu, float vv, float *out /* 6 */) {
    const DevTexture &t = *tp;
    const uint32_t mode = __float_as_uint(t.color1[2]), filter = mode & 0xffu;
    for (int i = 0; i < 6; ++i) out[i] = 0.0f;
    if (filter == 0 || !isfinite(uu) || !isfinite(vv)) return;      // bitmap.cpp:456-480: no gradient with the nearest filter
    MipRef m;
    m.sc = sc; m.first = __float_as_uint(t.color1[0]); m.levels = __float_as_uint(t.color1[1]);
    m.bcu = (mode >> 8) & 0xffu; m.bcv = (mode >> 16) & 0xffu; m.maxAnisotropy = t.lineWidth;
    const DevMipLevel L = sc->mipLevels[m.first];
    const float u = uu * (float) L.w - 0.5f, v = vv * (float) L.h - 0.5f;
    const int xPos = (int) floorf(u), yPos = (int) floorf(v);
    const float dx = u - (float) xPos, dy = v - (float) yPos;
    const V3 p00 = mipTexel(m, 0, xPos, yPos), p10 = mipTexel(m, 0, xPos + 1, yPos), p01 = mipTexel(m, 0, xPos, yPos + 1),
             p11 = mipTexel(m, 0, xPos + 1, yPos + 1);
    const V3 tmp = p01 + p10 - p11;
    const V3 gu = (p10 + p00 * (dy - 1) - tmp * dy) * (float) L.w, gv = (p01 + p00 * (dx - 1) - tmp * dx) * (float) L.h;
    out[0] = gu.x; out[1] = gu.y; out[2] = gu.z; out[3] = gv.x; out[4] = gv.y; out[5] = gv.z;
}

// ---- Texture2D::eval(its, filter = true) (src/librender/texture.cpp:112-121) ------------------------------------------
// checkerboard / gridtexture at transformed coordinates, before the texture's scale factor
MTS_DEV V3 proceduralEval(const DevTexture &t, float x, float y) {
    bool second;
    if (t.type == 1) {      // Checkerboard::eval (src/textures/checkerboard.cpp:66-75); math::modulo of a truncated int
        int xi = (int) (x * 2) % 2, yi = (int) (y * 2) % 2;
        if (xi < 0) xi += 2;
        if (yi < 0) yi += 2;
        second = (2 * xi - 1) * (2 * yi - 1) != 1;
    } else {                // GridTexture::eval (src/textures/gridtexture.cpp:75-88)
        float fx = x - (float) (int) floorf(x), fy = y - (float) (int) floorf(y);
        if (fx > 0.5f) fx -= 1;
        if (fy > 0.5f) fy -= 1;
        second = fabsf(fx) < t.lineWidth || fabsf(fy) < t.lineWidth;
    }
    return second ? ld3(t.color1) : ld3(t.color0);
}
MTS_DEV V3 textureEval(const DevScene &sc, const DevTexture &t, const SurfUV &q) {
    const float x = q.u * t.uScale + t.uOffset, y = q.v * t.vScale + t.vOffset;
    if (t.type == 3) {
        const float4 r = bitmapEval(sc.self, &t, x, y, make_float4(q.dudx * t.uScale, q.dvdx * t.vScale, q.dudy * t.uScale, q.dvdy * t.vScale), q.filtered);
        return mk(r.x, r.y, r.z) * t.scale;
    }
    return proceduralEval(t, x, y) * t.scale;
}

// Texture2D::evalGradient(its, gradient) (texture.cpp:123-141): the bitmap's bilinear gradient (bitmap.cpp:456-480) or
// forward differences of the unfiltered lookup, times the uv scale
MTS_DEV void textureGradient(const DevScene &sc, const DevTexture &t, float u, float v, V3 &gu, V3 &gv) {
    const float x = u * t.uScale + t.uOffset, y = v * t.vScale + t.vOffset;
    if (t.type == 3) {
        float g[6];
        bitmapGradient(sc.self, &t, x, y, g);
        const V3 f = ld3(t.color0);
        gu = mk(g[0], g[1], g[2]) * f; gv = mk(g[3], g[4], g[5]) * f;
    } else {
        const float eps = kEpsilon;
        const V3 value = proceduralEval(t, x, y), valueU = proceduralEval(t, x + eps, y), valueV = proceduralEval(t, x, y + eps);
        gu = (valueU - value) * (1 / eps); gv = (valueV - value) * (1 / eps);
    }
    gu = gu * t.uScale * t.scale; gv = gv * t.vScale * t.scale;    // Texture2D::evalGradient, then ScaleTexture::evalGradient
}

// BumpMap::getFrame (src/bsdfs/bumpmap.cpp:135-159) / NormalMap::getFrame (src/bsdfs/normalmap.cpp:107-124): the shading
// frame the nested BSDF is asked in. `frameWord` = kind << 1 | texture << 8 (see DevScene::materialTextures).
MTS_DEV Frame3 perturbedFrame(const DevScene &sc, uint32_t frameWord, const Frame3 &sh, V3 geoN, V3 dpdu0, V3 dpdv0, float u, float v) {
    const DevTexture &t = sc.plainTextures[(frameWord >> 8) - 1];
    Frame3 r;
    if (((frameWord >> 1) & 3u) == 1u) {
        V3 gu, gv;
        textureGradient(sc, t, u, v, gu, gv);
        const float dDispDu = luminance(gu), dDispDv = luminance(gv);
        const V3 dpdu = dpdu0 + sh.n * (dDispDu - dot(sh.n, dpdu0)), dpdv = dpdv0 + sh.n * (dDispDv - dot(sh.n, dpdv0));
        r.n = normalize(cross(dpdu, dpdv));
        r.s = normalize(dpdu - r.n * dot(r.n, dpdu));
        r.t = cross(r.n, r.s);
        if (dot(r.n, geoN) < 0) r.n = -r.n;
    } else {
        const V3 c = textureEval(sc, t, surfUV(u, v));
        const V3 n = mk(2 * c.x - 1, 2 * c.y - 1, 2 * c.z - 1);
        r.n = normalize(toWorld(sh, n));
        r.s = normalize(dpdu0 - r.n * dot(r.n, dpdu0));
        r.t = cross(r.n, r.s);
    }
    return r;
}

// m_alphaU->eval(bRec.its).average() (roughconductor.cpp:264-268; Spectrum::average sums the channels, then multiplies by 1 / 3)
MTS_DEV float texturedRoughness(const DevScene &sc, uint32_t texture /* 1-based in plainTextures */, const SurfUV &q) {
    const V3 c = textureEval(sc, sc.plainTextures[texture - 1], q);
    float r = 0.0f;
    r += c.x; r += c.y; r += c.z;
    return r * (1.0f / 3.0f);
}

// The colours of record `index` that are textured, evaluated at the intersection's uv and written over the constants
// in m: what m_reflectance->eval(bRec.its) etc. return inside the BSDF (e.g. src/bsdfs/diffuse.cpp:110-118)
MTS_DEV void applyTextures(const DevScene &sc, uint32_t index, const SurfUV &q, DevMaterial &m) {
    if (!sc.materialTextures) return;
    const uint4 b = sc.materialTextures[index];
    if (b.x) { const V3 c = textureEval(sc, sc.textures[b.x - 1], q); m.diffuse[0] = c.x; m.diffuse[1] = c.y; m.diffuse[2] = c.z; }
    if (b.y) { const V3 c = textureEval(sc, sc.textures[b.y - 1], q); m.specular[0] = c.x; m.specular[1] = c.y; m.specular[2] = c.z; }
    if (b.z) { const V3 c = textureEval(sc, sc.textures[b.z - 1], q); m.transmittance[0] = c.x; m.transmittance[1] = c.y; m.transmittance[2] = c.z; }
    if (b.w & 8u) {
        const uint2 a = sc.materialAlphaTextures[index];
        if (a.x) m.alphaU = texturedRoughness(sc, a.x, q);
        if (a.y) m.alphaV = texturedRoughness(sc, a.y, q);
    }
}
MTS_DEV void loadMaterialAt(const DevScene &sc, uint32_t index, const SurfUV &uv, DevMaterial &m) {
    loadMaterial(sc, index, m);
    applyTextures(sc, index, uv, m);
}

MTS_DEV V3 mirrorZ(V3 v) { return mk(v.x, v.y, -v.z); }

#ifndef MTS_KARY
#define MTS_KARY 1
#endif
// ---- DiscreteDistribution::sampleReuse on a device CDF (include/mitsuba/core/pmf.h:124-188) ---------------------
MTS_DEV uint32_t cdfSampleReuse(const float *cdf, uint32_t n /* entries */, float &sample, float *pdfOut) {
    // lower_bound over cdf[0..n] = the number of entries below the sample (the entries never decrease). Eight-way
    // steps: the seven pivots of a step are independent loads, so a search costs log8 instead of log2 memory round trips.
    uint32_t lo = 0, count = n + 1;
#if MTS_KARY
    while (count > 8) {
        uint32_t start = lo, end = lo + count;
        bool open = true;       // the pivots so far are all below the sample
#pragma unroll
        for (uint32_t j = 1; j < 8; ++j) {
            const uint32_t p = lo + (count * j) / 8;
            const bool below = cdf[p] < sample;
            if (open && below) start = p + 1;
            if (open && !below) { end = p; open = false; }
        }
        lo = start; count = end - start;
    }
    {
        uint32_t below = 0;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j)
            if (j < count && cdf[lo + j] < sample) ++below;
        lo += below;
    }
#else
    while (count > 0) {
        const uint32_t step = count / 2, mid = lo + step;
        if (cdf[mid] < sample) { lo = mid + 1; count -= step + 1; } else count = step;
    }
#endif
    int idx = (int) lo - 1;
    if (idx < 0) idx = 0;
    uint32_t index = (uint32_t) idx < n - 1 ? (uint32_t) idx : n - 1;
    while (cdf[index + 1] - cdf[index] == 0 && index < n) ++index;
    if (pdfOut) *pdfOut = cdf[index + 1] - cdf[index];
    sample = (sample - cdf[index]) / (cdf[index + 1] - cdf[index]);
    return index;
}

// ---- MixtureBSDF (src/bsdfs/mixturebsdf.cpp:174-277), all components requested ---------------------------------
// A mixture record keeps its count in `variant`, its weights in diffuse[0..2] + rtransOffset, the upper ends of the
// normalised choice intervals in specular[0..2] + rtransN and the nested records' indices in condEta[0..2] + condK[0].
// `distribution` holds the nested records' BSDF types, a byte each. Fields are picked with selects, not indexed, so that
// they stay in registers.
MTS_DEV float mixWeight(const DevMaterial &m, int i) {
    return i == 0 ? m.diffuse[0] : i == 1 ? m.diffuse[1] : i == 2 ? m.diffuse[2] : __uint_as_float(m.rtransOffset);
}
MTS_DEV float mixCdf(const DevMaterial &m, int i) {      // cdf[0] = 0, cdf[count] = 1
    return i <= 0 ? 0.0f : i == 1 ? m.specular[0] : i == 2 ? m.specular[1] : i == 3 ? m.specular[2] : __uint_as_float(m.rtransN);
}
MTS_DEV uint32_t mixChild(const DevMaterial &m, int i) {
    return __float_as_uint(i == 0 ? m.condEta[0] : i == 1 ? m.condEta[1] : i == 2 ? m.condEta[2] : m.condK[0]);
}

// eval / pdf in the discrete measure: the Dirac lobes of dielectric (dielectric.cpp:228-275), conductor
// (conductor.cpp:221-250) and plastic (plastic.cpp:243-300); every other lobe is zero there. A mixture asks for them
// after one of its nested BSDFs has sampled such a lobe.
static constexpr float kDeltaEpsilon = 1e-3f;       // include/mitsuba/core/constants.h:31
MTS_DEV V3 bsdfEvalDiscrete(const DevMaterial &m, V3 wi, V3 wo) {
    const V3 wr = mk(-wi.x, -wi.y, wi.z);
    switch (m.type) {
        case 3: {
            float cosThetaT;
            const float F = fresnelDielectricExt(wi.z, cosThetaT, m.eta);
            if (wi.z * wo.z >= 0) {
                if (fabsf(dot(wr, wo) - 1) > kDeltaEpsilon) return mk(0.0f);
                return ld3(m.specular) * F;
            }
            const float scale = -(cosThetaT < 0 ? m.invEta : m.eta);
            const V3 wt = mk(scale * wi.x, scale * wi.y, cosThetaT);
            if (fabsf(dot(wt, wo) - 1) > kDeltaEpsilon) return mk(0.0f);
            const float factor = cosThetaT < 0 ? m.invEta : m.eta;
            return ld3(m.transmittance) * (factor * factor * (1 - F));
        }
        case 5:
            if (wi.z <= 0 || wo.z <= 0 || fabsf(dot(wr, wo) - 1) > kDeltaEpsilon) return mk(0.0f);
            return ld3(m.specular) * fresnelConductorExact(wi.z, ld3(m.condEta), ld3(m.condK));
        case 6: {
            if (wo.z <= 0 || wi.z <= 0) return mk(0.0f);
            float cosThetaT;
            const float Fi = fresnelDielectricExt(wi.z, cosThetaT, m.eta);
            if (fabsf(dot(wr, wo) - 1) < kDeltaEpsilon) return ld3(m.specular) * Fi;
            return mk(0.0f);
        }
    }
    return mk(0.0f);
}

MTS_DEV float bsdfPdfDiscrete(const DevMaterial &m, V3 wi, V3 wo) {
    const V3 wr = mk(-wi.x, -wi.y, wi.z);
    switch (m.type) {
        case 3: {
            float cosThetaT;
            const float F = fresnelDielectricExt(wi.z, cosThetaT, m.eta);
            if (wi.z * wo.z >= 0) return fabsf(dot(wr, wo) - 1) > kDeltaEpsilon ? 0.0f : F;
            const float scale = -(cosThetaT < 0 ? m.invEta : m.eta);
            const V3 wt = mk(scale * wi.x, scale * wi.y, cosThetaT);
            return fabsf(dot(wt, wo) - 1) > kDeltaEpsilon ? 0.0f : 1 - F;
        }
        case 5:
            return (wi.z <= 0 || wo.z <= 0 || fabsf(dot(wr, wo) - 1) > kDeltaEpsilon) ? 0.0f : 1.0f;
        case 6: {
            if (wo.z <= 0 || wi.z <= 0) return 0.0f;
            float cosThetaT;
            const float Fi = fresnelDielectricExt(wi.z, cosThetaT, m.eta);
            const float probSpecular = (Fi * m.specSamplingWeight) / (Fi * m.specSamplingWeight + (1 - Fi) * (1 - m.specSamplingWeight));
            return fabsf(dot(wr, wo) - 1) < kDeltaEpsilon ? probSpecular : 0.0f;
        }
    }
    return 0.0f;
}

MTS_DEV V3 mixEval(const DevScene &sc, const DevMaterial &m, V3 wi, V3 wo, const SurfUV &uv) {
    V3 result = mk(0.0f);
#pragma unroll 1
    for (int i = 0; i < m.variant; ++i) {
        DevMaterial c;
        loadMaterialAt(sc, mixChild(m, i), uv, c);
        result = result + bsdfEval<true>(sc, c, wi, wo) * mixWeight(m, i);
    }
    return result;
}

MTS_DEV float mixPdf(const DevScene &sc, const DevMaterial &m, V3 wi, V3 wo, const SurfUV &uv) {
    float result = 0.0f;
#pragma unroll 1
    for (int i = 0; i < m.variant; ++i) {
        DevMaterial c;
        loadMaterialAt(sc, mixChild(m, i), uv, c);
        result += bsdfPdf<true>(sc, c, wi, wo) * (mixCdf(m, i + 1) - mixCdf(m, i));
    }
    return result;
}

// DiscreteDistribution::sampleReuse (include/mitsuba/core/pmf.h:124-188) over the five-entry table: the nested BSDF
// that the first sample coordinate picks
MTS_DEV int mixEntry(const DevMaterial &m, float sx) {
    const int n = m.variant;
    int lo = 0;
#pragma unroll
    for (int k = 0; k <= 4; ++k) lo += (k <= n && mixCdf(m, k) < sx) ? 1 : 0;
    int entry = lo - 1 < 0 ? 0 : lo - 1;
    entry = entry < n - 1 ? entry : n - 1;
    while (mixCdf(m, entry + 1) - mixCdf(m, entry) == 0 && entry < n) ++entry;
    return entry;
}

// Whether BSDF::sample with first sample coordinate sx takes a number from the sampler itself (roughdielectric does,
// bRec.sampler->next1D(); a mixture does when the nested BSDF it picks does)
template <bool MIX = true>
MTS_DEV bool bsdfDrawsFromSampler(const DevMaterial &m, float sx) {
    if (MIX && m.type == 8) return (((uint32_t) m.distribution >> (8 * mixEntry(m, sx))) & 0xffu) == 7u;
    return m.type == 7;
}

MTS_DEV V3 mixSample(const DevScene &sc, const DevMaterial &m, V3 wi, float sx, float sy, float su, V3 &wo, float &pdf, float &eta, uint32_t &sampledType, const SurfUV &uv) {
    const int n = m.variant;
    const int entry = mixEntry(m, sx);
    const float lower = mixCdf(m, entry), width = mixCdf(m, entry + 1) - lower;
    sx = (sx - lower) / width;
    DevMaterial c;
    loadMaterialAt(sc, mixChild(m, entry), uv, c);
    V3 result = bsdfSample<true>(sc, c, wi, sx, sy, su, wo, pdf, eta, sampledType);
    if (isZero(result)) return result;      // sampling failed
    result = result * (mixWeight(m, entry) * pdf);
    pdf *= width;
    const bool discrete = (sampledType & EDelta) != 0;     // BSDF::getMeasure(bRec.sampledType)
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        if (i == entry) continue;
        loadMaterialAt(sc, mixChild(m, i), uv, c);
        pdf += (discrete ? bsdfPdfDiscrete(c, wi, wo) : bsdfPdf<true>(sc, c, wi, wo)) * (mixCdf(m, i + 1) - mixCdf(m, i));
        result = result + (discrete ? bsdfEvalDiscrete(c, wi, wo) : bsdfEval<true>(sc, c, wi, wo)) * mixWeight(m, i);
    }
    return result / pdf;
}

// The three mixture functions are real calls: each holds a copy of every BSDF model, and inlined into the shading
// kernels they would triple the code there and spill hundreds of registers. What they need of the mixture record travels by
// value (13 words), the rest of the arguments likewise, so nothing of the caller has to live in memory.
struct MixRec { int n; float w0, w1, w2, w3, c1, c2, c3, c4; uint32_t k0, k1, k2, k3, types; };
MTS_DEV MixRec mixRec(const DevMaterial &m) {
    MixRec r;
    r.n = m.variant; r.w0 = m.diffuse[0]; r.w1 = m.diffuse[1]; r.w2 = m.diffuse[2]; r.w3 = __uint_as_float(m.rtransOffset);
    r.c1 = m.specular[0]; r.c2 = m.specular[1]; r.c3 = m.specular[2]; r.c4 = __uint_as_float(m.rtransN);
    r.k0 = __float_as_uint(m.condEta[0]); r.k1 = __float_as_uint(m.condEta[1]); r.k2 = __float_as_uint(m.condEta[2]); r.k3 = __float_as_uint(m.condK[0]);
    r.types = (uint32_t) m.distribution;
    return r;
}
MTS_DEV void mixMaterial(const MixRec &r, DevMaterial &m) {
    m.type = 8; m.variant = r.n; m.diffuse[0] = r.w0; m.diffuse[1] = r.w1; m.diffuse[2] = r.w2; m.rtransOffset = __float_as_uint(r.w3);
    m.specular[0] = r.c1; m.specular[1] = r.c2; m.specular[2] = r.c3; m.rtransN = __float_as_uint(r.c4);
    m.condEta[0] = __uint_as_float(r.k0); m.condEta[1] = __uint_as_float(r.k1); m.condEta[2] = __uint_as_float(r.k2); m.condK[0] = __uint_as_float(r.k3);
    m.distribution = (decltype(m.distribution)) r.types;
}
template <int TAG>      // see resolveSurfaceCall
__device__ __attribute__((noinline)) inline float4 mixEvalCall(const DevScene *scp, MixRec r, V3 wi, V3 wo, SurfUV uv) {
    DevMaterial m;
    mixMaterial(r, m);
    const V3 v = mixEval(*scp, m, wi, wo, uv);
    return make_float4(v.x, v.y, v.z, 0.0f);
}
template <int TAG>
__device__ __attribute__((noinline)) inline float mixPdfCall(const DevScene *scp, MixRec r, V3 wi, V3 wo, SurfUV uv) {
    DevMaterial m;
    mixMaterial(r, m);
    return mixPdf(*scp, m, wi, wo, uv);
}
struct MixSampled { float4 weightPdf, woEta; uint32_t sampledType; };
template <int TAG>
__device__ __attribute__((noinline)) inline void mixSampleCall(const DevScene *scp, MixRec r, V3 wi, float sx, float sy, float su, SurfUV uv, MixSampled *out) {
    DevMaterial m;
    mixMaterial(r, m);
    V3 wo; float pdf, eta; uint32_t sampledType;
    const V3 w = mixSample(*scp, m, wi, sx, sy, su, wo, pdf, eta, sampledType, uv);
    out->weightPdf = make_float4(w.x, w.y, w.z, pdf);
    out->woEta = make_float4(wo.x, wo.y, wo.z, eta);
    out->sampledType = sampledType;
}

// What the integrator calls on the record that answers for this side: a mixture (only compiled in with MIX) or a plain model
// (the overloads that take the uv record by address read it only for a mixture: callers keep it in memory, not in registers)
template <bool RD = true, bool MIX = true, int TAG = 0>
MTS_DEV V3 bsdfEvalAny(const DevScene &sc, const DevMaterial &m, V3 wi, V3 wo, const SurfUV *uv) {
    if (MIX && m.type == 8) { const float4 r = mixEvalCall<TAG>(sc.self, mixRec(m), wi, wo, *uv); return mk(r.x, r.y, r.z); }
    return bsdfEval<RD>(sc, m, wi, wo);
}
template <bool RD = true, bool MIX = true, int TAG = 0>
MTS_DEV float bsdfPdfAny(const DevScene &sc, const DevMaterial &m, V3 wi, V3 wo, const SurfUV *uv) {
    if (MIX && m.type == 8) return mixPdfCall<TAG>(sc.self, mixRec(m), wi, wo, *uv);
    return bsdfPdf<RD>(sc, m, wi, wo);
}
template <bool RD = true, bool MIX = true, int TAG = 0>
MTS_DEV V3 bsdfEvalAny(const DevScene &sc, const DevMaterial &m, V3 wi, V3 wo, const SurfUV &uv = surfUV(0, 0)) {
    if (MIX && m.type == 8) { const float4 r = mixEvalCall<TAG>(sc.self, mixRec(m), wi, wo, uv); return mk(r.x, r.y, r.z); }
    return bsdfEval<RD>(sc, m, wi, wo);
}
template <bool RD = true, bool MIX = true, int TAG = 0>
MTS_DEV float bsdfPdfAny(const DevScene &sc, const DevMaterial &m, V3 wi, V3 wo, const SurfUV &uv = surfUV(0, 0)) {
    if (MIX && m.type == 8) return mixPdfCall<TAG>(sc.self, mixRec(m), wi, wo, uv);
    return bsdfPdf<RD>(sc, m, wi, wo);
}
template <bool RD = true, bool MIX = true, int TAG = 0>
MTS_DEV V3 bsdfSampleAny(const DevScene &sc, const DevMaterial &m, V3 wi, float sx, float sy, float su, V3 &wo, float &pdf, float &eta, uint32_t &sampledType,
                         const SurfUV *uv) {
    if (MIX && m.type == 8) {
        MixSampled r;
        mixSampleCall<TAG>(sc.self, mixRec(m), wi, sx, sy, su, *uv, &r);
        wo = mk(r.woEta.x, r.woEta.y, r.woEta.z); pdf = r.weightPdf.w; eta = r.woEta.w; sampledType = r.sampledType;
        return mk(r.weightPdf.x, r.weightPdf.y, r.weightPdf.z);
    }
    return bsdfSample<RD>(sc, m, wi, sx, sy, su, wo, pdf, eta, sampledType);
}
template <bool RD = true, bool MIX = true, int TAG = 0>
MTS_DEV V3 bsdfSampleAny(const DevScene &sc, const DevMaterial &m, V3 wi, float sx, float sy, float su, V3 &wo, float &pdf, float &eta, uint32_t &sampledType,
                         const SurfUV &uv = surfUV(0, 0)) {
    if (MIX && m.type == 8) {
        MixSampled r;
        mixSampleCall<TAG>(sc.self, mixRec(m), wi, sx, sy, su, uv, &r);
        wo = mk(r.woEta.x, r.woEta.y, r.woEta.z); pdf = r.weightPdf.w; eta = r.woEta.w; sampledType = r.sampledType;
        return mk(r.weightPdf.x, r.weightPdf.y, r.weightPdf.z);
    }
    return bsdfSample<RD>(sc, m, wi, sx, sy, su, wo, pdf, eta, sampledType);
}

#ifndef MTS_ENV_SCENE
#define MTS_ENV_SCENE(sc) (sc).self
#endif
struct DirectRec { V3 d; float dist, pdf; bool delta; };

// warp::squareToUniformSphere (src/libcore/warp.cpp:25-31)
MTS_DEV V3 squareToUniformSphere(float sx, float sy) {
    const float z = 1.0f - 2.0f * sy;
    const float r = safeSqrt(1.0f - z * z);
    float sinPhi, cosPhi;
    sincosM(2.0f * kPi * sx, &sinPhi, &cosPhi);
    return mk(r * cosPhi, r * sinPhi, z);
}

// Far intersection of a ray with the environment sphere, or false when the origin is not inside it: solveQuadratic
// (src/libcore/util.cpp:447-485) through BSphere::rayIntersect (include/mitsuba/core/bsphere.h:88-95) with the
// "nearT < 0 && farT > 0" test of constant.cpp:201-206
MTS_DEV bool environmentExit(const DevScene &sc, V3 ro, V3 rd, float &farT) {
    const V3 o = ro - ld3(sc.envCenter);
    const float A = dot(rd, rd), B = 2 * dot(o, rd), C = dot(o, o) - sc.envRadius * sc.envRadius;
    float x0, x1;
    if (A == 0) {
        if (B == 0) return false;
        x0 = x1 = -C / B;
    } else {
        const float discrim = B * B - 4.0f * A * C;
        if (discrim < 0) return false;
        const float sqrtDiscrim = sqrtf(discrim);
        const float temp = B < 0 ? -0.5f * (B - sqrtDiscrim) : -0.5f * (B + sqrtDiscrim);
        x0 = temp / A;
        x1 = C / temp;
        if (x0 > x1) { const float t = x0; x0 = x1; x1 = t; }
    }
    farT = x1;
    return x0 < 0 && x1 > 0;
}

__device__ __attribute__((noinline)) inline float4 envSampleDirect(const DevScene *scp, V3 ref, float sx, float sy, float emPdf, float4 *dOut);

// ConstantBackgroundEmitter::sampleDirect (src/emitters/constant.cpp:178-216) inside Scene::sampleEmitterDirect
// (src/librender/scene.cpp:828-852): a sample on the back side of the reference normal keeps its pdf (so its
// shadow ray is still traced and counted) but carries no radiance.
MTS_DEV V3 sampleEnvironmentDirect(const DevScene &sc, V3 ref, V3 refN, float sx, float sy, float emPdf, DirectRec &dRec) {
    if (sc.envIsMap) {
        float4 dd;
        const float4 r = envSampleDirect(MTS_ENV_SCENE(sc), ref, sx, sy, emPdf, &dd);
        dRec.d = mk(dd.x, dd.y, dd.z); dRec.dist = dd.w; dRec.pdf = r.w;
        return mk(r.x, r.y, r.z);
    }
    V3 d;
    float pdf;
    const bool haveN = !isZero(refN);
    if (haveN) {
        d = squareToCosineHemisphere(sx, sy);
        pdf = cosHemiPdf(d);
        Frame3 f;
        f.n = refN;
        coordinateSystem(refN, f.s, f.t);
        d = toWorld(f, d);
    } else {
        d = squareToUniformSphere(sx, sy);
        pdf = 0.07957747154594766788f;
    }
    float farT;
    if (!environmentExit(sc, ref, d, farT)) return mk(0.0f);
    dRec.d = d; dRec.dist = farT;
    dRec.pdf = pdf * emPdf;
    if (haveN && dot(d, refN) <= 0) return mk(0.0f);
    return (ld3(sc.envRadiance) / pdf) / emPdf;
}

// ConstantBackgroundEmitter::pdfDirect, solid-angle measure (constant.cpp:218-233), times pdfEmitterDiscrete;
// refCos is the cosine to the reference normal or -2 when there is none
MTS_DEV float environmentPdfDirect(const DevScene &sc, float refCos) {
    const float pdfSA = refCos > -1.5f ? kInvPi * fmaxf(0.0f, refCos) : 0.07957747154594766788f;
    return pdfSA * sc.envPdfDiscrete;
}

// ---- EnvironmentMap (src/emitters/envmap.cpp) ------------------------------------------------------------------------
MTS_DEV V3 envToWorld(const DevScene &sc, V3 v) {
    const float *T = sc.envToWorld;
    return mk(T[0] * v.x + T[1] * v.y + T[2] * v.z, T[3] * v.x + T[4] * v.y + T[5] * v.z, T[6] * v.x + T[7] * v.y + T[8] * v.z);
}
MTS_DEV V3 envToLocal(const DevScene &sc, V3 v) {
    const float *T = sc.envToLocal;
    return mk(T[0] * v.x + T[1] * v.y + T[2] * v.z, T[3] * v.x + T[4] * v.y + T[5] * v.z, T[6] * v.x + T[7] * v.y + T[8] * v.z);
}
MTS_DEV MipRef envMipRef(const DevScene &sc) {
    MipRef m;
    m.sc = MTS_ENV_SCENE(sc); m.first = sc.envFirstLevel; m.levels = sc.envLevels; m.bcu = 1u; m.bcv = 0u; m.maxAnisotropy = 10.0f;    // repeat / clamp (envmap.cpp:141-146)
    return m;
}
MTS_DEV float intervalToTent(float sample) {        // src/libcore/warp.cpp:151-163
    float sign;
    if (sample < 0.5f) { sign = 1; sample *= 2; }
    else { sign = -1; sample = 2 * (sample - 0.5f); }
    return sign * (1 - sqrtf(sample));
}
// EnvironmentMap::sampleReuse (envmap.cpp:657-662): lower_bound over cdf[0 .. size]
MTS_DEV uint32_t envSampleReuse(const float *cdf, uint32_t size, float &sample) {
    uint32_t lo = 0, count = size + 1;
    while (count > 0) {
        const uint32_t step = count / 2;
        if (cdf[lo + step] < sample) { lo += step + 1; count -= step + 1; }
        else count = step;
    }
    const uint32_t index = min((uint32_t) max((int) lo - 1, 0), size - 1);
    sample = (sample - cdf[index]) / (cdf[index + 1] - cdf[index]);
    return index;
}
// the two rows of the bilinear interpolation around a fractional pixel position (envmap.cpp:577-591, :618-632)
MTS_DEV void envRows(const DevScene &sc, int xPos, int yPos, float dx1, float dy1, V3 &value1, V3 &value2) {
    const MipRef m = envMipRef(sc);
    const float dx2 = 1.0f - dx1, dy2 = 1.0f - dy1;
    value1 = mipTexel(m, 0, xPos, yPos) * dx2 * dy2 + mipTexel(m, 0, xPos + 1, yPos) * dx1 * dy2;
    value2 = mipTexel(m, 0, xPos, yPos + 1) * dx2 * dy1 + mipTexel(m, 0, xPos + 1, yPos + 1) * dx1 * dy1;
}
MTS_DEV float envRowPdf(const DevScene &sc, V3 value1, V3 value2, int yPos) {
    const int h = (int) sc.envH;
    return (luminance(value1) * sc.envRowWeights[min(h - 1, max(0, yPos))] + luminance(value2) * sc.envRowWeights[min(h - 1, max(0, yPos + 1))]) * sc.envNormalization;
}
// EnvironmentMap::internalPdfDirection (envmap.cpp:603-633), d in the emitter's frame
__device__ __attribute__((noinline)) inline float envPdfDirection(const DevScene *scp, float dx, float dy, float dz) {
    const DevScene &sc = *scp;
    const float uu = atan2M(dx, -dz) * 0.15915494309189533577f, vv = acosM(fminf(1.0f, fmaxf(-1.0f, dy))) * kInvPi;
    if (!isfinite(uu) || !isfinite(vv)) return 0.0f;
    const float u = uu * (float) sc.envW - 0.5f, v = vv * (float) sc.envH - 0.5f;
    const int xPos = (int) floorf(u), yPos = (int) floorf(v);
    V3 value1, value2;
    envRows(sc, xPos, yPos, u - (float) xPos, v - (float) yPos, value1, value2);
    const float sinTheta = safeSqrt(1 - dy * dy);
    return envRowPdf(sc, value1, value2, yPos) / fmaxf(fabsf(sinTheta), kEpsilon);
}
// EnvironmentMap::evalEnvironment (envmap.cpp:380-410) for a world-space direction; with offset rays (a camera ray) the
// lookup is filtered
__device__ __attribute__((noinline)) inline float4 envEvalMap(const DevScene *scp, V3 rayD, bool filtered, V3 rxD, V3 ryD) {
    const DevScene &sc = *scp;
    const V3 v = envToLocal(sc, rayD);
    const float uu = atan2M(v.x, -v.z) * 0.15915494309189533577f, vv = acosM(fminf(1.0f, fmaxf(-1.0f, v.y))) * kInvPi;
    const MipRef m = envMipRef(sc);
    V3 value;
    if (!filtered) value = mipBilinear(m, 0, uu, vv);
    else {
        const V3 dvdx = envToLocal(sc, rxD) - v, dvdy = envToLocal(sc, ryD) - v;
        const float t1 = 0.15915494309189533577f / (v.x * v.x + v.z * v.z), t2 = -kInvPi / fmaxf(safeSqrt(1.0f - v.y * v.y), kEpsilon);
        value = mipFiltered(m, 3u, uu, vv, t1 * (dvdx.z * v.x - dvdx.x * v.z), t2 * dvdx.y, t1 * (dvdy.z * v.x - dvdy.x * v.z), t2 * dvdy.y);
    }
    value = value * sc.envScale;
    return make_float4(value.x, value.y, value.z, 0.0f);
}
// EnvironmentMap::sampleDirect (envmap.cpp:516-543, internalSampleDirection :567-600) inside Scene::sampleEmitterDirect
__device__ __attribute__((noinline)) inline float4 envSampleDirect(const DevScene *scp, V3 ref, float sx, float sy, float emPdf, float4 *dOut /* d.xyz, dist */) {
    const DevScene &sc = *scp;
    const uint32_t row = envSampleReuse(sc.envCdfRows, sc.envH, sy), col = envSampleReuse(sc.envCdfCols + (size_t) row * (sc.envW + 1), sc.envW, sx);
    const float px = (float) col + intervalToTent(sx), py = (float) row + intervalToTent(sy);
    const int xPos = (int) floorf(px), yPos = (int) floorf(py);
    V3 value1, value2;
    envRows(sc, xPos, yPos, px - (float) xPos, py - (float) yPos, value1, value2);
    const V3 value = (value1 + value2) * sc.envScale;
    float pdf = envRowPdf(sc, value1, value2, yPos);
    float sinPhi, cosPhi, sinTheta, cosTheta;
    sincosM(sc.envPixelX * (px + 0.5f), &sinPhi, &cosPhi);
    sincosM(sc.envPixelY * (py + 0.5f), &sinTheta, &cosTheta);
    pdf /= fmaxf(fabsf(sinTheta), kEpsilon);
    const V3 d = envToWorld(sc, mk(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta));
    *dOut = make_float4(d.x, d.y, d.z, 0.0f);
    // the sample counts only if the reference point lies inside the environment sphere (nearT < 0 < farT, envmap.cpp:527-533)
    float farT;
    if (isZero(value) || pdf == 0 || !environmentExit(sc, ref, d, farT)) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    dOut->w = farT;
    const V3 r = (value / pdf) / emPdf;
    return make_float4(r.x, r.y, r.z, pdf * emPdf);
}

// The radiance a ray that leaves the scene collects (Scene::evalEnvironment), and the density with which emitter
// sampling would have produced its direction (pdfEmitterDirect), for the constant emitter or the map
MTS_DEV V3 environmentRadiance(const DevScene &sc, V3 rayD) {
    if (!sc.envIsMap) return ld3(sc.envRadiance);
    const float4 r = envEvalMap(MTS_ENV_SCENE(sc), rayD, false, rayD, rayD);
    return mk(r.x, r.y, r.z);
}
MTS_DEV float environmentPdf(const DevScene &sc, V3 rayD, float refCos) {
    if (!sc.envIsMap) return environmentPdfDirect(sc, refCos);
    const V3 v = envToLocal(sc, rayD);
    return envPdfDirection(MTS_ENV_SCENE(sc), v.x, v.y, v.z) * sc.envPdfDiscrete;     // EnvironmentMap::pdfDirect (envmap.cpp:545-551)
}

// PointEmitter::sampleDirect (src/emitters/point.cpp:131-147), SpotEmitter::sampleDirect with falloffCurve
// (src/emitters/spot.cpp:105-125,184-200) and DirectionalEmitter::sampleDirect (src/emitters/directional.cpp:159-180)
// inside Scene::sampleEmitterDirect: one position or direction, discrete measure
MTS_DEV V3 sampleDeltaDirect(const DevScene &sc, const DevEmitter &em, V3 ref, float emPdf, DirectRec &dRec) {
    const DevDeltaEmitter de = sc.deltaEmitters[em.delta];
    V3 value = ld3(em.radiance);
    if (em.kind == 4) {
        const V3 d = ld3(de.direction);
        const float distance = dot(ref - ld3(de.position), d);
        if (distance < 0) return mk(0.0f);
        dRec.d = -d;
        dRec.dist = distance;
    } else {
        V3 d = ld3(de.position) - ref;
        const float dist = length(d);
        const float invDist = 1.0f / dist;
        d = d * invDist;
        dRec.d = d; dRec.dist = dist;
        if (em.kind == 3) {
            const float cosTheta = dot(ld3(de.toLocalZ), -d);
            float falloff = 1.0f;
            if (cosTheta <= de.cosCutoff) falloff = 0.0f;
            else if (cosTheta < de.cosBeam) falloff = (de.cutoffAngle - acosM(cosTheta)) * de.invTransitionWidth;
            value = value * falloff;
        }
        value = value * (invDist * invDist);
    }
    dRec.pdf = emPdf;
    dRec.delta = true;
    return value / emPdf;
}

// Scene::sampleEmitterDirect without the shadow ray (src/librender/scene.cpp:828-852) -> AreaLight::sampleDirect
// (src/emitters/area.cpp:158-173) -> Shape::sampleDirect (src/librender/shape.cpp:102-115) -> TriMesh::samplePosition
// (src/librender/trimesh.cpp:412-423) -> Triangle::sample (src/libcore/triangle.cpp:24-59)
// EXT: the scene has emitters that are not area lights.
// The emitters that are not area lights (the environment, point / spot / directional lights) behind one real call:
// inlined, their sampling code costs every vertex of a scene that has such an emitter a hundred spilled registers.
// Returns value.rgb and the pdf; dOut receives the direction and the distance. TAG: see resolveSurfaceCall.
template <int TAG>
__device__ __attribute__((noinline)) inline float4 sampleOtherEmitterCall(const DevScene *scp, uint32_t index, V3 ref, V3 refN, float sx, float sy, float emPdf,
                                                                         float4 *dOut) {
    const DevScene &sc = *scp;
    DirectRec dRec;
    dRec.pdf = 0; dRec.dist = 0; dRec.d = mk(0.0f); dRec.delta = false;
    const V3 value = (int32_t) index == sc.envEmitter ? sampleEnvironmentDirect(sc, ref, refN, sx, sy, emPdf, dRec)
                                                      : sampleDeltaDirect(sc, sc.emitters[index], ref, emPdf, dRec);
    *dOut = make_float4(dRec.d.x, dRec.d.y, dRec.d.z, dRec.dist);
    return make_float4(value.x, value.y, value.z, dRec.pdf);
}

// The emitter tables of a scene with few area lights, as the shading kernel holds them in workgroup memory (DevScene::emitterTris)
struct EmitterTables {
    float cdf[kLdsEmitters + 1];
    float areaCdf[kLdsEmitterTris + kLdsEmitters];
    uint32_t firstTri[kLdsEmitters];        // an emitter's first triangle in `tris`
    DevEmitter em[kLdsEmitters];
    float4 tris[kLdsEmitterTris * 6];       // shading records: three positions, three normals
    // every thread of the workgroup calls this; a barrier must follow before the tables are read
    MTS_DEV void fill(const DevScene &sc, uint32_t tid, uint32_t nThreads) {
        for (uint32_t i = tid; i <= sc.nEmitters; i += nThreads) cdf[i] = sc.emitterCdf[i];
        for (uint32_t i = tid; i < sc.areaCdfLen; i += nThreads) areaCdf[i] = sc.areaCdf[i];
        for (uint32_t i = tid; i < sc.nEmitters; i += nThreads) {
            em[i] = sc.emitters[i];
            uint32_t first = 0;
            for (uint32_t j = 0; j < i; ++j) first += sc.emitters[j].nTris;
            firstTri[i] = first;
        }
        for (uint32_t k = tid; k < sc.emitterTris * 6u; k += nThreads) {
            uint32_t t = k / 6u, e = 0;
            while (t >= sc.emitters[e].nTris) { t -= sc.emitters[e].nTris; ++e; }
            tris[k] = sc.shadeTris[kShadeTriStride * (size_t) (sc.emitters[e].firstTri + t) + k % 6u];
        }
    }
};

// The area-light branch of sampleEmitterDirect, given the emitter's record, its triangle (six float4) and the choice
// probability: AreaLight::sampleDirect (src/emitters/area.cpp:158-173)
MTS_DEV V3 sampleAreaLight(const DevEmitter &em, const float4 g0, const float4 g1, const float4 g2, const float4 h0, const float4 h1, const float4 h2,
                           V3 ref, V3 refN, float sx, float sy, float emPdf, DirectRec &dRec) {
    const V3 p0 = mk(g0.x, g0.y, g0.z), p1 = mk(g1.x, g1.y, g1.z), p2 = mk(g2.x, g2.y, g2.z);
    const float a = safeSqrt(1.0f - sx);                  // squareToUniformTriangle (warp.cpp:84-87)
    const float baryX = 1 - a, baryY = a * sy;
    const V3 sideA = p1 - p0, sideB = p2 - p0;
    const V3 p = p0 + (sideA * baryX) + (sideB * baryY);
    V3 n;
    if ((__float_as_uint(g0.w) >> 24) & 1u) {
        const V3 n0 = mk(h0.x, h0.y, h0.z), n1 = mk(h1.x, h1.y, h1.z), n2 = mk(h2.x, h2.y, h2.z);
        n = normalize(n0 * (1.0f - baryX - baryY) + n1 * baryX + n2 * baryY);
    } else {
        n = normalize(cross(sideA, sideB));
    }
    float pdf = em.invArea;
    V3 d = p - ref;
    const float distSquared = dot(d, d);
    const float dist = sqrtf(distSquared);
    d = d / dist;
    const float dp = fabsf(dot(d, n));
    pdf *= dp != 0 ? (distSquared / dp) : 0.0f;
    dRec.d = d; dRec.dist = dist;
    if (dot(d, refN) >= 0 && dot(d, n) < 0 && pdf != 0) {
        V3 value = ld3(em.radiance) / pdf;
        dRec.pdf = pdf * emPdf;
        return value / emPdf;
    }
    dRec.pdf = 0.0f;
    return mk(0.0f);
}

template <bool EXT = true, int TAG = 0>
MTS_DEV V3 sampleEmitterDirect(const DevScene &sc, V3 ref, V3 refN, float sx, float sy, DirectRec &dRec, const EmitterTables *tables = nullptr, bool useTables = false) {
    dRec.pdf = 0; dRec.dist = 0; dRec.d = mk(0.0f); dRec.delta = false;
    if (sc.nEmitters == 0) return mk(0.0f);
    float emPdf;
    if (!EXT && useTables) {
        // the same steps on the copies in workgroup memory
        const uint32_t index = cdfSampleReuse(tables->cdf, sc.nEmitters, sx, &emPdf);
        const DevEmitter em = tables->em[index];
        const uint32_t triIdx = cdfSampleReuse(tables->areaCdf + em.cdfOffset, em.nTris, sy, nullptr);
        const float4 *g = tables->tris + 6u * (tables->firstTri[index] + triIdx);
        return sampleAreaLight(em, g[0], g[1], g[2], g[3], g[4], g[5], ref, refN, sx, sy, emPdf, dRec);
    }
    const uint32_t index = cdfSampleReuse(sc.emitterCdf, sc.nEmitters, sx, &emPdf);
    if (EXT) {
        const bool environment = (int32_t) index == sc.envEmitter;
        if (environment || sc.emitters[index].kind >= 2) {
            float4 dd;
            const float4 r = sampleOtherEmitterCall<TAG>(sc.self, index, ref, refN, sx, sy, emPdf, &dd);
            dRec.d = mk(dd.x, dd.y, dd.z); dRec.dist = dd.w; dRec.pdf = r.w; dRec.delta = !environment;
            return mk(r.x, r.y, r.z);
        }
    }
    const DevEmitter em = sc.emitters[index];
    const uint32_t triIdx = cdfSampleReuse(sc.areaCdf + em.cdfOffset, em.nTris, sy, nullptr);
    const uint32_t t = em.firstTri + triIdx;
    const float4 *g = sc.shadeTris + kShadeTriStride * (size_t) t;
    return sampleAreaLight(em, g[0], g[1], g[2], g[3], g[4], g[5], ref, refN, sx, sy, emPdf, dRec);
}

MTS_DEV float miWeight(float pdfA, float pdfB) {   // path.cpp:296-300
    pdfA *= pdfA; pdfB *= pdfB;
    return pdfA / (pdfA + pdfB);
}

// PerspectiveCameraImpl::sampleRayDifferential, ray part (src/sensors/perspective.cpp:271-298)
MTS_DEV void cameraRay(const DevScene &sc, float px, float py, V3 &o, V3 &d, float &mint, float &maxt) {
    const float *M = sc.sampleToCamera;
    const float sx = px * sc.invResX, sy = py * sc.invResY;
    const float x = M[0] * sx + M[1] * sy + M[3];
    const float y = M[4] * sx + M[5] * sy + M[7];
    const float z = M[8] * sx + M[9] * sy + M[11];
    const float w = M[12] * sx + M[13] * sy + M[15];
    V3 nearP = mk(x, y, z);
    if (w != 1.0f) nearP = nearP / w;
    const V3 dl = normalize(nearP);
    const float invZ = 1.0f / dl.z;
    const float *W = sc.camToWorld;
    o = mk(W[3], W[7], W[11]);
    d = mk(W[0] * dl.x + W[1] * dl.y + W[2] * dl.z, W[4] * dl.x + W[5] * dl.y + W[6] * dl.z, W[8] * dl.x + W[9] * dl.y + W[10] * dl.z);
    mint = sc.nearClip * invZ;
    maxt = sc.farClip * invZ;
}

// Intersection::computePartials (src/librender/intersection.cpp:5-76) for the vertex a camera ray reaches: the offset
// rays of PerspectiveCameraImpl::sampleRayDifferential (perspective.cpp:289-293; position differentials on the near
// plane :158-162) scaled by 1 / sqrt(sampleCount) (integrator.cpp:150,177; ray.h:163-168), intersected with the
// tangent plane; two of the three equations p' - p = dpdu * du + dpdv * dv are solved for (du, dv).
MTS_DEV V3 sampleToCameraPoint(const DevScene &sc, float sx, float sy) {
    const float *M = sc.sampleToCamera;
    V3 p = mk(M[0] * sx + M[1] * sy + M[3], M[4] * sx + M[5] * sy + M[7], M[8] * sx + M[9] * sy + M[11]);
    const float w = M[12] * sx + M[13] * sy + M[15];
    if (w != 1.0f) p = p / w;
    return p;
}

// The directions of a camera ray's offset rays after RayDifferential::scaleDifferential (their origins are the ray's)
MTS_DEV void cameraOffsetRays(const DevScene &sc, const PassParams &pp, float2 pos, V3 d, V3 &rxD, V3 &ryD) {
    const V3 nearP = sampleToCameraPoint(sc, pos.x * sc.invResX, pos.y * sc.invResY);
    const V3 origin = sampleToCameraPoint(sc, 0.0f, 0.0f);
    const V3 dx = sampleToCameraPoint(sc, sc.invResX, 0.0f) - origin, dy = sampleToCameraPoint(sc, 0.0f, sc.invResY) - origin;
    const float *W = sc.camToWorld;
    const V3 lx = normalize(nearP + dx), ly = normalize(nearP + dy);
    const V3 rx = mk(W[0] * lx.x + W[1] * lx.y + W[2] * lx.z, W[4] * lx.x + W[5] * lx.y + W[6] * lx.z, W[8] * lx.x + W[9] * lx.y + W[10] * lx.z);
    const V3 ry = mk(W[0] * ly.x + W[1] * ly.y + W[2] * ly.z, W[4] * ly.x + W[5] * ly.y + W[6] * ly.z, W[8] * ly.x + W[9] * ly.y + W[10] * ly.z);
    const float amount = 1.0f / sqrtf((float) pp.sppTotal);
    rxD = d + (rx - d) * amount; ryD = d + (ry - d) * amount;
}

// Scene::evalEnvironment for a camera ray that leaves the scene: the map through its filtered lookup (envmap.cpp:391-407)
MTS_DEV V3 cameraEnvironmentRadiance(const DevScene &sc, const PassParams &pp, float2 pos, V3 d) {
    V3 rxD, ryD;
    cameraOffsetRays(sc, pp, pos, d, rxD, ryD);
    const float4 r = envEvalMap(MTS_ENV_SCENE(sc), d, true, rxD, ryD);
    return mk(r.x, r.y, r.z);
}

MTS_DEV void cameraUVPartials(const DevScene &sc, const PassParams &pp, float2 pos, const Isect &its, float4 rayO, V3 d, SurfUV &uv) {
    uv.filtered = 1u;
    uv.dudx = uv.dvdx = uv.dudy = uv.dvdy = 0.0f;
    if (isZero(its.dpdu) && isZero(its.dpdv)) return;
    const V3 o = mk(rayO.x, rayO.y, rayO.z);
    V3 rxD, ryD;
    cameraOffsetRays(sc, pp, pos, d, rxD, ryD);
    const V3 n = its.geoN;
    const float ppn = dot(n, its.p), po = dot(n, o), prx = dot(n, rxD), pry = dot(n, ryD);
    if (prx == 0 || pry == 0) return;
    const float tx = (ppn - po) / prx, ty = (ppn - po) / pry;
    const float absX = fabsf(n.x), absY = fabsf(n.y), absZ = fabsf(n.z);
    float a00, a01, a10, a11, bx0, bx1, by0, by1;
    const V3 px = o + rxD * tx, py = o + ryD * ty;
    const V3 ex = px - its.p, ey = py - its.p;
    if (absX > absY && absX > absZ) { a00 = its.dpdu.y; a01 = its.dpdv.y; a10 = its.dpdu.z; a11 = its.dpdv.z; bx0 = ex.y; bx1 = ex.z; by0 = ey.y; by1 = ey.z; }
    else if (absY > absZ) { a00 = its.dpdu.x; a01 = its.dpdv.x; a10 = its.dpdu.z; a11 = its.dpdv.z; bx0 = ex.x; bx1 = ex.z; by0 = ey.x; by1 = ey.z; }
    else { a00 = its.dpdu.x; a01 = its.dpdv.x; a10 = its.dpdu.y; a11 = its.dpdv.y; bx0 = ex.x; bx1 = ex.y; by0 = ey.x; by1 = ey.y; }
    // solveLinearSystem2x2 (src/libcore/util.cpp:527-539)
    const float det = a00 * a11 - a01 * a10;
    if (fabsf(det) <= 0x1p-128f) { uv.dudx = 1; uv.dudy = 1; return; }     // intersection.cpp:68-75 (dvdy is left unset there)
    const float inverse = 1.0f / det;
    uv.dudx = (a11 * bx0 - a01 * bx1) * inverse; uv.dvdx = (a00 * bx1 - a10 * bx0) * inverse;
    uv.dudy = (a11 * by0 - a01 * by1) * inverse; uv.dvdy = (a00 * by1 - a10 * by0) * inverse;
}

// Everything a hit needs from the texture machinery, behind ONE real call: the uv partials of a camera vertex, the
// textured colours of the answering record and the frame of a bumpmap / normalmap around it. Inlined into the shading
// kernel these pieces cost every path a few hundred register spills whether or not its material has a texture; as a
// call they cost the paths that need them a trip through memory (the query record) and the others one test.
struct SurfaceQuery {
    // in
    V3 p, geoN, dpdu, dpdv, rayO, rayD;
    Frame3 sh;
    float u, v;
    float2 pos;
    int32_t sppTotal;
    uint32_t material, first;
    // out
    SurfUV uv;
    float colour[9];        // diffuse, specular, transmittance
    float alpha[2];         // roughness along u, v
    uint32_t textured;      // bit i: colour[3 i ..] is set; bits 3, 4: alpha[0], alpha[1]
    uint32_t framed;
    Frame3 pf;
};
// the textured colours and roughness of record `material` at uv, into the query record
MTS_DEV void resolveColours(const DevScene &sc, uint32_t material, const uint4 b, const SurfUV &uv, SurfaceQuery *q) {
    uint32_t textured = 0;
    if (b.x) { const V3 c = textureEval(sc, sc.textures[b.x - 1], uv); q->colour[0] = c.x; q->colour[1] = c.y; q->colour[2] = c.z; textured |= 1u; }
    if (b.y) { const V3 c = textureEval(sc, sc.textures[b.y - 1], uv); q->colour[3] = c.x; q->colour[4] = c.y; q->colour[5] = c.z; textured |= 2u; }
    if (b.z) { const V3 c = textureEval(sc, sc.textures[b.z - 1], uv); q->colour[6] = c.x; q->colour[7] = c.y; q->colour[8] = c.z; textured |= 4u; }
    if (b.w & 8u) {
        const uint2 a = sc.materialAlphaTextures[material];
        if (a.x) { q->alpha[0] = texturedRoughness(sc, a.x, uv); textured |= 8u; }
        if (a.y) { q->alpha[1] = texturedRoughness(sc, a.y, uv); textured |= 16u; }
    }
    q->textured = textured;
}
// TAG: one instance per register budget of the calling kernels (a function shared by kernels with different launch
// bounds is compiled for the largest budget, and its register count then becomes that of every kernel that calls it)
template <int TAG>
__device__ __attribute__((noinline)) inline void resolveSurfaceCall(const DevScene *scp, SurfaceQuery *q) {
    const DevScene &sc = *scp;
    SurfUV uv = surfUV(q->u, q->v);
    if (q->first) {
        Isect its;
        its.p = q->p; its.geoN = q->geoN; its.dpdu = q->dpdu; its.dpdv = q->dpdv;
        PassParams pp;
        pp.sppTotal = q->sppTotal;
        cameraUVPartials(sc, pp, q->pos, its, make_float4(q->rayO.x, q->rayO.y, q->rayO.z, 0.0f), q->rayD, uv);
    }
    q->uv = uv;
    const uint4 b = sc.materialTextures[q->material];
    resolveColours(sc, q->material, b, uv, q);
    q->framed = 0;
    if (b.w & 6u) { q->framed = 1; q->pf = perturbedFrame(sc, b.w, q->sh, q->geoN, q->dpdu, q->dpdv, q->u, q->v); }
}
// The common case of the above, a vertex after the first of a record without a frame adapter: nothing goes in through
// memory, only the colours and the roughness come back through it.
template <int TAG>
__device__ __attribute__((noinline)) inline void resolveColoursCall(const DevScene *scp, uint32_t material, float u, float v, SurfaceQuery *q) {
    const DevScene &sc = *scp;
    resolveColours(sc, material, sc.materialTextures[material], surfUV(u, v), q);
}

}  // namespace mtsamd
