// Host-side helpers of the C ABI that need no device: rough-transmittance table reduction and the discretised
// reconstruction filters.
#include "../../include/mtsamd.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

// Catmull-Rom style cubic weights for one dimension of evalCubicInterp2D/3D on [0,1]
// (src/libcore/spline.cpp:242-287): four node weights around the knot, derivative stencils folded in.
struct Cubic1 {
    float w[4];
    size_t knot;
    bool ok;
    Cubic1(float p, size_t size) {
        ok = p >= 0.0f && p <= 1.0f;
        knot = 0;
        w[0] = w[1] = w[2] = w[3] = 0;
        if (!ok) return;
        float t = ((p - 0.0f) * (float) (size - 1)) / (1.0f - 0.0f);
        knot = std::min((size_t) t, size - 2);
        t = t - (float) knot;
        const float t2 = t * t, t3 = t2 * t;
        w[1] = 2 * t3 - 3 * t2 + 1;
        w[2] = -2 * t3 + 3 * t2;
        const float d0 = t3 - 2 * t2 + t, d1 = t3 - t2;
        if (knot > 0) { w[2] += 0.5f * d0; w[0] -= 0.5f * d0; } else { w[2] += d0; w[1] -= d0; }
        if (knot + 2 < size) { w[3] += 0.5f * d1; w[1] -= 0.5f * d1; } else { w[2] += d1; w[1] -= d1; }
    }
};

float cubic1D(float x, const float *v, size_t size) {   // spline.cpp:23-60 with min = 0, max = 1
    if (!(x >= 0.0f && x <= 1.0f)) return 0.0f;
    float t = ((x - 0.0f) * (float) (size - 1)) / (1.0f - 0.0f);
    size_t k = std::min((size_t) t, size - 2);
    const float f0 = v[k], f1 = v[k + 1];
    const float d0 = k > 0 ? 0.5f * (v[k + 1] - v[k - 1]) : v[k + 1] - v[k];
    const float d1 = k + 2 < size ? 0.5f * (v[k + 2] - v[k]) : v[k + 1] - v[k];
    t = t - (float) k;
    const float t2 = t * t, t3 = t2 * t;
    return (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
}

float cubic2D(float px, float py, const float *v, size_t sx, size_t sy) {     // spline.cpp:236-304
    Cubic1 cx(px, sx), cy(py, sy);
    if (!cx.ok || !cy.ok) return 0.0f;
    float result = 0.0f;
    for (int y = -1; y <= 2; ++y)
        for (int x = -1; x <= 2; ++x) {
            const float wxy = cx.w[x + 1] * cy.w[y + 1];
            if (wxy == 0) continue;
            result += v[(cy.knot + y) * sx + cx.knot + x] * wxy;
        }
    return result;
}

float cubic3D(float px, float py, float pz, const float *v, size_t sx, size_t sy, size_t sz) {    // spline.cpp:379-451
    Cubic1 cx(px, sx), cy(py, sy), cz(pz, sz);
    if (!cx.ok || !cy.ok || !cz.ok) return 0.0f;
    float result = 0.0f;
    for (int z = -1; z <= 2; ++z)
        for (int y = -1; y <= 2; ++y) {
            const float wyz = cy.w[y + 1] * cz.w[z + 1];
            for (int x = -1; x <= 2; ++x) {
                const float wxyz = cx.w[x + 1] * wyz;
                if (wxyz == 0) continue;
                result += v[((cz.knot + z) * sy + (cy.knot + y)) * sx + cx.knot + x] * wxyz;
            }
        }
    return result;
}

inline float clamp01(float v) { return std::min(1.0f, std::max(0.0f, v)); }

// The image of a data/microfacet/<distribution>.dat file (src/bsdfs/rtrans.h:100-148): magic, 3 x uint64 sample counts,
// 4 x float ranges, then for each of the 2*etaSamples blocks and each alpha: thetaSamples transmittances followed by
// one diffuse transmittance
struct TransmittanceFile {
    size_t nEta = 0, nAlpha = 0, nTheta = 0;
    float etaMin = 0, etaMax = 0, alphaMin = 0, alphaMax = 0;
    std::vector<float> trans, diff;

    int load(const void *dat, uint64_t size) {
        static const char magic[] = "MTS_TRANSMITTANCE";
        const size_t ml = sizeof(magic) - 1;
        const uint8_t *p = (const uint8_t *) dat;
        if (size < ml + 40 || memcmp(p, magic, ml) != 0) return MTSAMD_EINVAL;
        uint64_t cnt[3];
        float range[4];
        memcpy(cnt, p + ml, 24);
        memcpy(range, p + ml + 24, 16);
        nEta = cnt[0]; nAlpha = cnt[1]; nTheta = cnt[2];
        etaMin = range[0]; etaMax = range[1]; alphaMin = range[2]; alphaMax = range[3];
        const size_t nT = 2 * nEta * nAlpha * nTheta, nD = 2 * nEta * nAlpha;
        if (nEta < 2 || nAlpha < 2 || nTheta < 2 || size != ml + 40 + (nT + nD) * 4) return MTSAMD_EINVAL;
        trans.resize(nT); diff.resize(nD);
        const float *src = (const float *) (p + ml + 40);
        size_t q = 0, a = 0, b = 0;
        std::vector<float> tmp(nT + nD);
        memcpy(tmp.data(), src, tmp.size() * 4);
        for (size_t i = 0; i < 2 * nEta; ++i)
            for (size_t j = 0; j < nAlpha; ++j) {
                for (size_t k = 0; k < nTheta; ++k) trans[a++] = tmp[q++];
                diff[b++] = tmp[q++];
            }
        return MTSAMD_OK;
    }
    // RoughTransmittance::checkEta (rtrans.h:390-398)
    bool etaInRange(float eta) const { const float e = eta < 1 ? 1 / eta : eta; return e >= etaMin && e <= etaMax; }
    // setEta (rtrans.h:292-345): the slice at one relative index of refraction, [alpha][theta] and [alpha]
    void sliceEta(float e, std::vector<float> &t2, std::vector<float> &d2) const {
        const float *tr = trans.data(), *dt = diff.data();
        if (e < 1) { tr += nEta * nAlpha * nTheta; dt += nEta * nAlpha; e = 1.0f / e; }
        if (e < etaMin) e = etaMin;
        const float warpedEta = std::pow((e - etaMin) / (etaMax - etaMin), 0.25f);
        t2.resize(nAlpha * nTheta); d2.resize(nAlpha);
        const float dAlpha = 1.0f / (float) (nAlpha - 1), dTheta = 1.0f / (float) (nTheta - 1);
        for (size_t i = 0; i < nAlpha; ++i) {
            for (size_t j = 0; j < nTheta; ++j)
                t2[i * nTheta + j] = cubic3D((float) j * dTheta, (float) i * dAlpha, warpedEta, tr, nTheta, nAlpha, nEta);
            d2[i] = cubic2D((float) i * dAlpha, warpedEta, dt, nAlpha, nEta);
        }
    }
};

}  // namespace

extern "C" int mtsamd_rtrans_reduce(const void *dat, uint64_t size, float eta, float alpha, float *out_T, uint32_t out_cap,
                                    uint32_t *theta_samples, float *diffuse_ext, float *diffuse_int) {
    if (!dat || !out_T || !theta_samples || !diffuse_ext || !diffuse_int) return MTSAMD_EINVAL;
    TransmittanceFile f;
    const int rc = f.load(dat, size);
    if (rc != MTSAMD_OK) return rc;
    if (out_cap < f.nTheta) return MTSAMD_EINVAL;
    // RoughTransmittance::checkEta / checkAlpha (rtrans.h:390-407)
    if (!f.etaInRange(eta) || alpha < f.alphaMin || alpha > f.alphaMax) return MTSAMD_EUNSUPPORTED;
    // setEta for the exterior (eta) and interior (1/eta) copies
    std::vector<float> extT, extD, intT, intD;
    f.sliceEta(eta, extT, extD);
    f.sliceEta(1 / eta, intT, intD);
    // setAlpha on the exterior copy (rtrans.h:353-388); evalDiffuse(alpha) on both (rtrans.h:249-259)
    const float warpedAlpha = std::pow((alpha - f.alphaMin) / (f.alphaMax - f.alphaMin), 0.25f);
    const float dTheta = 1.0f / (float) (f.nTheta - 1);
    for (size_t i = 0; i < f.nTheta; ++i) out_T[i] = cubic2D((float) i * dTheta, warpedAlpha, extT.data(), f.nTheta, f.nAlpha);
    *theta_samples = (uint32_t) f.nTheta;
    *diffuse_ext = clamp01(cubic1D(warpedAlpha, extD.data(), f.nAlpha));
    *diffuse_int = clamp01(cubic1D(warpedAlpha, intD.data(), f.nAlpha));
    return MTSAMD_OK;
}

extern "C" int mtsamd_rtrans_reduce_eta(const void *dat, uint64_t size, float eta, float *out, uint32_t out_cap, uint32_t *n_out) {
    if (!dat || !out || !n_out) return MTSAMD_EINVAL;
    TransmittanceFile f;
    const int rc = f.load(dat, size);
    if (rc != MTSAMD_OK) return rc;
    if (!f.etaInRange(eta)) return MTSAMD_EUNSUPPORTED;
    const size_t n = 4 + f.nAlpha * f.nTheta + 2 * f.nAlpha;
    *n_out = (uint32_t) n;
    if (out_cap < n) return MTSAMD_EINVAL;
    std::vector<float> extT, extD, intT, intD;
    f.sliceEta(eta, extT, extD);
    f.sliceEta(1 / eta, intT, intD);
    out[0] = (float) f.nTheta; out[1] = (float) f.nAlpha; out[2] = f.alphaMin; out[3] = f.alphaMax;
    memcpy(out + 4, extT.data(), extT.size() * 4);
    memcpy(out + 4 + extT.size(), extD.data(), f.nAlpha * 4);
    memcpy(out + 4 + extT.size() + f.nAlpha, intD.data(), f.nAlpha * 4);
    return MTSAMD_OK;
}

static void normalizeFilter(float radius, float values[32]) {
    // ReconstructionFilter::configure (src/libcore/rfilter.cpp:37-55)
    float sum = 0.0f;
    for (int i = 0; i < 31; ++i) sum += values[i];
    values[31] = 0.0f;
    sum *= 2 * radius / 31;
    const float normalization = 1.0f / sum;
    for (int i = 0; i < 31; ++i) values[i] *= normalization;
}

extern "C" int mtsamd_filter_gaussian(float stddev, float *radius, float values[32]) {
    if (!(stddev > 0) || !radius || !values) return MTSAMD_EINVAL;
    const float r = 4 * stddev;                                     // gaussian.cpp:38
    const float alpha = -1.0f / (2.0f * stddev * stddev);
    for (int i = 0; i < 31; ++i) {
        const float x = (r * (float) i) / 31;
        values[i] = std::max(0.0f, (float) std::exp((double) (alpha * x * x)) - (float) std::exp((double) (alpha * r * r)));   // gaussian.cpp:52-57
    }
    normalizeFilter(r, values);
    *radius = r;
    return MTSAMD_OK;
}

// The other reconstruction filters of the reference, tabulated the same way (ReconstructionFilter::configure):
// tent (src/rfilters/tent.cpp:34-44), Mitchell-Netravali (mitchell.cpp:35-68), Catmull-Rom (catmullrom.cpp:32-55),
// Lanczos sinc (lanczos.cpp:35-55)
extern "C" int mtsamd_filter_table(int kind, float p0, float p1, float *radius, float values[32]) {
    if (!radius || !values) return MTSAMD_EINVAL;
    if (kind == MTSAMD_FILTER_GAUSSIAN) return mtsamd_filter_gaussian(p0, radius, values);
    if (kind == MTSAMD_FILTER_BOX) return mtsamd_filter_box(p0, radius, values);
    float r;
    float B = 0.0f, C = 0.5f;
    switch (kind) {
        case MTSAMD_FILTER_TENT: r = 1.0f; break;
        case MTSAMD_FILTER_MITCHELL: r = 2.0f; B = p0; C = p1; break;
        case MTSAMD_FILTER_CATMULLROM: r = 2.0f; break;
        case MTSAMD_FILTER_LANCZOS: r = (float) (int) p0; if (!(r >= 1)) return MTSAMD_EINVAL; break;
        default: return MTSAMD_EINVAL;
    }
    for (int i = 0; i < 31; ++i) {
        float x = std::fabs((r * (float) i) / 31);
        float v;
        if (kind == MTSAMD_FILTER_TENT) {
            v = std::max(0.0f, 1.0f - std::fabs(x / r));
        } else if (kind == MTSAMD_FILTER_LANCZOS) {
            if (x < 1e-4f) v = 1.0f;
            else if (x > r) v = 0.0f;
            else {
                const float x1 = 3.14159265358979323846f * x, x2 = x1 / r;
                v = (std::sin(x1) * std::sin(x2)) / (x1 * x2);
            }
        } else {
            const float x2 = x * x, x3 = x2 * x;
            if (x < 1) v = 1.0f / 6.0f * ((12 - 9 * B - 6 * C) * x3 + (-18 + 12 * B + 6 * C) * x2 + (6 - 2 * B));
            else if (x < 2) v = 1.0f / 6.0f * ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * x + (8 * B + 24 * C));
            else v = 0.0f;
        }
        values[i] = v;
    }
    normalizeFilter(r, values);
    *radius = r;
    return MTSAMD_OK;
}

extern "C" int mtsamd_filter_box(float radius_in, float *radius, float values[32]) {
    if (!(radius_in > 0) || !radius || !values) return MTSAMD_EINVAL;
    const float r = radius_in + 1e-5f;      // box.cpp:38
    for (int i = 0; i < 31; ++i) values[i] = 1.0f;
    normalizeFilter(r, values);
    *radius = r;
    return MTSAMD_OK;
}
