// TEST INFRASTRUCTURE ONLY -- CPU oracle: RoughTransmittance table handling
// (src/bsdfs/rtrans.h:81-149 file layout, :183-284 lookups, :292-388 setEta/setAlpha).
#include "bsdf.h"
#include <cstring>
#include <string>

namespace orc {

struct RoughTransmittance {
    size_t etaSamples = 0, alphaSamples = 0, thetaSamples = 0;
    float etaMin = 0, etaMax = 0, alphaMin = 0, alphaMax = 0;
    bool etaFixed = false, alphaFixed = false;
    std::vector<float> trans, diffTrans;

    bool load(const void *dat, uint64_t size) {
        const char header[] = "MTS_TRANSMITTANCE";
        const size_t hlen = strlen(header);
        const uint8_t *p = (const uint8_t *) dat;
        if (size < hlen + 24 + 16 || memcmp(p, header, hlen) != 0) return false;
        p += hlen;
        uint64_t s[3];
        memcpy(s, p, 24); p += 24;
        etaSamples = s[0]; alphaSamples = s[1]; thetaSamples = s[2];
        float r[4];
        memcpy(r, p, 16); p += 16;
        etaMin = r[0]; etaMax = r[1]; alphaMin = r[2]; alphaMax = r[3];
        size_t transSize = 2 * etaSamples * alphaSamples * thetaSamples;
        size_t diffSize = 2 * etaSamples * alphaSamples;
        if (size != hlen + 24 + 16 + (transSize + diffSize) * 4) return false;
        trans.resize(transSize); diffTrans.resize(diffSize);
        const float *ptr = (const float *) p;   // little-endian host assumed
        std::vector<float> temp(transSize + diffSize);
        memcpy(temp.data(), ptr, temp.size() * 4);
        size_t q = 0, fdr = 0, de = 0;
        for (size_t i = 0; i < 2 * etaSamples; ++i)
            for (size_t j = 0; j < alphaSamples; ++j) {
                for (size_t k = 0; k < thetaSamples; ++k) trans[de++] = temp[q++];
                diffTrans[fdr++] = temp[q++];
            }
        return true;
    }

    float eval(float cosTheta, float alpha = 0, float eta = 0) const {
        float warpedCosTheta = std::pow(std::abs(cosTheta), 0.25f), result;
        if (alphaFixed && etaFixed) {
            if (!(cosTheta >= 0)) return 0.f;
            result = evalCubicInterp1D(warpedCosTheta, trans.data(), thetaSamples, 0.0f, 1.0f);
        } else if (etaFixed) {
            if (!(cosTheta >= 0)) return 0.f;
            float warpedAlpha = std::pow((alpha - alphaMin) / (alphaMax - alphaMin), 0.25f);
            result = evalCubicInterp2D(warpedCosTheta, warpedAlpha, trans.data(), thetaSamples, alphaSamples);
        } else {
            if (cosTheta < 0) { cosTheta = -cosTheta; eta = 1.0f / eta; }
            const float *data = trans.data();
            if (eta < 1) { data += etaSamples * alphaSamples * thetaSamples; eta = 1.0f / eta; }
            if (eta < etaMin) eta = etaMin;
            float warpedAlpha = std::pow((alpha - alphaMin) / (alphaMax - alphaMin), 0.25f);
            float warpedEta = std::pow((eta - etaMin) / (etaMax - etaMin), 0.25f);
            result = evalCubicInterp3D(warpedCosTheta, warpedAlpha, warpedEta, data, thetaSamples, alphaSamples, etaSamples);
        }
        return std::min(1.0f, std::max(0.0f, result));
    }

    float evalDiffuse(float alpha = 0, float eta = 0) const {
        float result;
        if (alphaFixed && etaFixed) {
            result = diffTrans[0];
        } else if (etaFixed) {
            float warpedAlpha = std::pow((alpha - alphaMin) / (alphaMax - alphaMin), 0.25f);
            result = evalCubicInterp1D(warpedAlpha, diffTrans.data(), alphaSamples, 0.0f, 1.0f);
        } else {
            const float *data = diffTrans.data();
            if (eta < 1) { data += etaSamples * alphaSamples; eta = 1.0f / eta; }
            if (eta < etaMin) eta = etaMin;
            float warpedAlpha = std::pow((alpha - alphaMin) / (alphaMax - alphaMin), 0.25f);
            float warpedEta = std::pow((eta - etaMin) / (etaMax - etaMin), 0.25f);
            result = evalCubicInterp2D(warpedAlpha, warpedEta, data, alphaSamples, etaSamples);
        }
        return std::min(1.0f, std::max(0.0f, result));
    }

    void setEta(float eta) {
        if (etaFixed) return;
        const float *tr = trans.data(), *dt = diffTrans.data();
        if (eta < 1) {
            tr += etaSamples * alphaSamples * thetaSamples;
            dt += etaSamples * alphaSamples;
            eta = 1.0f / eta;
        }
        if (eta < etaMin) eta = etaMin;
        float warpedEta = std::pow((eta - etaMin) / (etaMax - etaMin), 0.25f);
        std::vector<float> newTrans(alphaSamples * thetaSamples), newDiff(alphaSamples);
        float dAlpha = 1.0f / (alphaSamples - 1), dTheta = 1.0f / (thetaSamples - 1);
        for (size_t i = 0; i < alphaSamples; ++i) {
            for (size_t j = 0; j < thetaSamples; ++j)
                newTrans[i * thetaSamples + j] = evalCubicInterp3D(j * dTheta, i * dAlpha, warpedEta, tr,
                                                                   thetaSamples, alphaSamples, etaSamples);
            newDiff[i] = evalCubicInterp2D(i * dAlpha, warpedEta, dt, alphaSamples, etaSamples);
        }
        trans.swap(newTrans);
        diffTrans.swap(newDiff);
        etaFixed = true;
    }

    void setAlpha(float alpha) {
        if (!etaFixed || alphaFixed) return;
        float warpedAlpha = std::pow((alpha - alphaMin) / (alphaMax - alphaMin), 0.25f);
        std::vector<float> newTrans(thetaSamples), newDiff(1);
        float dTheta = 1.0f / (thetaSamples - 1);
        for (size_t i = 0; i < thetaSamples; ++i)
            newTrans[i] = evalCubicInterp2D(i * dTheta, warpedAlpha, trans.data(), thetaSamples, alphaSamples);
        newDiff[0] = evalCubicInterp1D(warpedAlpha, diffTrans.data(), alphaSamples, 0.0f, 1.0f);
        trans.swap(newTrans);
        diffTrans.swap(newDiff);
        alphaFixed = true;
    }
};

}  // namespace orc

using namespace orc;

extern "C" int orc_rtrans_reduce(const void *dat, uint64_t size, float eta, float alpha, float *out_T,
                                 uint32_t out_cap, uint32_t *theta_samples, float *diffuse_ext, float *diffuse_int) {
    // RoughPlastic::configure (src/bsdfs/roughplastic.cpp:281-299): external copy gets setEta(eta)
    // then setAlpha(alpha); the internal copy only setEta(1/eta) and is queried with evalDiffuse(alpha).
    RoughTransmittance ext;
    if (!ext.load(dat, size)) return -1;
    RoughTransmittance in = ext;
    ext.setEta(eta);
    in.setEta(1 / eta);
    ext.setAlpha(alpha);
    if (out_cap < ext.thetaSamples) return -2;
    for (size_t i = 0; i < ext.thetaSamples; ++i) out_T[i] = ext.trans[i];
    *theta_samples = (uint32_t) ext.thetaSamples;
    *diffuse_ext = ext.evalDiffuse(alpha);
    *diffuse_int = in.evalDiffuse(alpha);
    return 0;
}

extern "C" int orc_rtrans_reduce_eta(const void *dat, uint64_t size, float eta, float *out, uint32_t out_cap, uint32_t *n_out) {
    // RoughPlastic::configure with a textured roughness (roughplastic.cpp:281-299): both copies only get setEta. The block:
    // thetaSamples, alphaSamples, alphaMin, alphaMax, T[alpha][theta] exterior, diffuse exterior [alpha], diffuse interior [alpha]
    RoughTransmittance ext;
    if (!ext.load(dat, size)) return -1;
    RoughTransmittance in = ext;
    ext.setEta(eta);
    in.setEta(1 / eta);
    const size_t n = 4 + ext.trans.size() + 2 * ext.alphaSamples;
    *n_out = (uint32_t) n;
    if (out_cap < n) return -2;
    out[0] = (float) ext.thetaSamples; out[1] = (float) ext.alphaSamples; out[2] = ext.alphaMin; out[3] = ext.alphaMax;
    size_t k = 4;
    for (float v : ext.trans) out[k++] = v;
    for (float v : ext.diffTrans) out[k++] = v;
    for (float v : in.diffTrans) out[k++] = v;
    return 0;
}

extern "C" int orc_rtrans_eval3d(const void *dat, uint64_t size, uint64_t n, const float *cos_theta,
                                 const float *alpha, const float *eta, float *out) {
    RoughTransmittance rt;
    if (!rt.load(dat, size)) return -1;
    for (uint64_t i = 0; i < n; ++i) out[i] = rt.eval(cos_theta[i], alpha[i], eta[i]);
    return 0;
}
