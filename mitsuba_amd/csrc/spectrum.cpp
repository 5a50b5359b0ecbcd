// Measured spectra (wavelength / value pairs) to the renderer's linear Rec. 709 RGB, as the reference converts the
// data/ior/*.spd material presets when it is built with three spectral samples. Single precision throughout, with
// the operations in the reference's order, so that the adaptive quadrature takes the same decisions.
//   src/libcore/spectrum.cpp:172-184   Spectrum::fromContinuousSpectrum (product with the matching functions)
//   src/libcore/spectrum.cpp:546-568   ContinuousSpectrum::average (50 nm pieces, Gauss-Lobatto)
//   src/libcore/spectrum.cpp:575-602   InterpolatedSpectrum file format
//   src/libcore/spectrum.cpp:650-714   InterpolatedSpectrum::average / eval
//   src/libcore/quad.cpp:305-409       GaussLobattoIntegrator
// and the <spectrum> / <blackbody> values of a scene file (src/librender/scenehandler.cpp:534-620), which add
// InterpolatedSpectrum::zeroExtend (spectrum.cpp:630-648), BlackBodySpectrum::eval (:483-495) and a clamp at zero.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <fstream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/mtsamd.h"

namespace {

struct Spd {
    std::vector<float> lambda, value;

    // Between two entries the reference weights the RIGHT neighbour with (1 - t) and the left one with t
    // (spectrum.cpp:706 passes the values to lerp in that order); entries themselves are returned as they are.
    float eval(float l) const {
        if (lambda.size() < 2 || l < lambda.front() || l > lambda.back()) return 0.0f;
        auto range = std::equal_range(lambda.begin(), lambda.end(), l);
        const size_t i1 = range.first - lambda.begin(), i2 = range.second - lambda.begin();
        if (i1 != i2) return value[i1];
        const float a = lambda[i1 - 1], b = lambda[i1], fa = value[i1 - 1], fb = value[i1];
        const float t = (l - a) / (b - a);
        return (1.0f - t) * fb + t * fa;
    }

    // trapezoids over the entries, clipped to [lo, hi], divided by the full width (spectrum.cpp:650-686)
    float average(float lo, float hi) const {
        if (lambda.size() < 2) return 0.0f;
        const float start = std::max(lo, lambda.front()), end = std::min(hi, lambda.back());
        if (end <= start) return 0.0f;
        size_t e = std::max((size_t) (std::lower_bound(lambda.begin(), lambda.end(), start) - lambda.begin()), (size_t) 1) - 1;
        float result = 0.0f;
        for (; e + 1 < lambda.size() && end >= lambda[e]; ++e) {
            const float a = lambda[e], b = lambda[e + 1], ca = std::max(a, start), cb = std::min(b, end);
            const float fa = value[e], fb = value[e + 1], invAB = 1.0f / (b - a);
            if (cb <= ca) continue;
            const float ta = (ca - a) * invAB, tb = (cb - a) * invAB;
            const float va = (1.0f - ta) * fa + ta * fb, vb = (1.0f - tb) * fa + tb * fb;
            result += 0.5f * (va + vb) * (cb - ca);
        }
        return result / (hi - lo);
    }
};

const float kCie[][4] = {
#include "cie1931.inc"
};
constexpr int kCieSamples = sizeof(kCie) / sizeof(kCie[0]);

Spd matchingFunction(int column) {
    Spd s;
    for (int i = 0; i < kCieSamples; ++i) { s.lambda.push_back(kCie[i][0]); s.value.push_back(kCie[i][column]); }
    return s;
}

// Planck's law in the reference's units and operation order (BlackBodySpectrum::eval, spectrum.cpp:483-495)
struct BlackBody {
    float temperature;
    float eval(float l) const {
        const double c = 299792458, k = 1.3806488e-23, h = 6.62606957e-34;
        const double lambda = l * 1e-9;
        const double I = (2 * h * c * c) * std::pow(lambda, -5.0) / ((std::exp((h / k) * c / (lambda * temperature)) - 1.0) * 1e9);
        return (float) I;
    }
};

// Adaptive Gauss-Lobatto quadrature of a product of two spectra (quad.cpp:305-409, without the convergence estimate).
template <typename First> struct Lobatto {
    const First &s1;
    const Spd &s2;
    size_t maxEvals = 10000, evals = 0;
    float absError = 1e-4f, relError = 1e-4f;   // the reference's Epsilon

    float f(float l) const { return s1.eval(l) * s2.eval(l); }

    float tolerance(float a, float b) {
        const float alpha = (float) std::sqrt(2.0 / 3.0), beta = (float) (1.0 / std::sqrt(5.0));
        const float x1 = (float) 0.94288241569547971906, x2 = (float) 0.64185334234578130578, x3 = (float) 0.23638319966214988028;
        const float m = (a + b) / 2, h = (b - a) / 2;
        const float y1 = f(a), y3 = f(m - alpha * h), y5 = f(m - beta * h), y7 = f(m), y9 = f(m + beta * h), y11 = f(m + alpha * h),
                    y13 = f(b);
        const float acc = h * ((float) 0.0158271919734801831 * (y1 + y13)
                               + (float) 0.0942738402188500455 * (f(m - x1 * h) + f(m + x1 * h))
                               + (float) 0.1550719873365853963 * (y3 + y11)
                               + (float) 0.1888215739601824544 * (f(m - x2 * h) + f(m + x2 * h))
                               + (float) 0.1997734052268585268 * (y5 + y9)
                               + (float) 0.2249264653333395270 * (f(m - x3 * h) + f(m + x3 * h))
                               + (float) 0.2426110719014077338 * y7);
        evals += 13;
        const float eps = std::numeric_limits<float>::epsilon();
        float result = std::numeric_limits<float>::infinity();
        if (relError != 0 && acc != 0) result = acc * std::max(relError, eps) / eps;
        if (absError != 0) result = std::min(result, absError / eps);
        return result;
    }

    float step(float a, float b, float fa, float fb, float acc) {
        const float alpha = (float) std::sqrt(2.0 / 3.0), beta = (float) (1.0 / std::sqrt(5.0));
        const float h = (b - a) / 2, m = (a + b) / 2;
        const float mll = m - alpha * h, ml = m - beta * h, mr = m + beta * h, mrr = m + alpha * h;
        const float fmll = f(mll), fml = f(ml), fm = f(m), fmr = f(mr), fmrr = f(mrr);
        const float integral2 = (h / 6) * (fa + fb + 5 * (fml + fmr));
        const float integral1 = (h / 1470) * (77 * (fa + fb) + 432 * (fmll + fmrr) + 625 * (fml + fmr) + 672 * fm);
        evals += 5;
        if (evals >= maxEvals) return integral1;
        // volatile: the comparison below asks whether the difference vanishes IN SINGLE PRECISION next to acc
        volatile float dist = acc + (integral1 - integral2);
        if (dist == acc || mll <= a || b <= mrr) return integral1;
        return step(a, mll, fa, fmll, acc) + step(mll, ml, fmll, fml, acc) + step(ml, m, fml, fm, acc)
             + step(m, mr, fm, fmr, acc) + step(mr, mrr, fmr, fmrr, acc) + step(mrr, b, fmrr, fb, acc);
    }

    float integrate(float a, float b) {
        if (a == b) return 0;
        evals = 0;
        const float acc = tolerance(a, b);
        evals += 2;
        return step(a, b, f(a), f(b), acc);
    }
};

template <typename First> float productAverage(const First &s1, const Spd &s2, float lo, float hi) {
    if (hi <= lo) return 0.0f;
    Lobatto<First> q{s1, s2};
    float integral = 0;
    const size_t nSteps = std::max((size_t) 1, (size_t) std::ceil((hi - lo) / 50));
    const float stepSize = (hi - lo) / nSteps;
    float pos = lo;
    for (size_t i = 0; i < nSteps; ++i) {
        integral += q.integrate(pos, pos + stepSize);
        pos += stepSize;
    }
    return integral / (hi - lo);
}

template <typename Spectrum> void toRgb(const Spectrum &s, float *rgb) {
    static const Spd cieX = matchingFunction(1), cieY = matchingFunction(2), cieZ = matchingFunction(3);
    const float start = kCie[0][0], end = kCie[kCieSamples - 1][0];
    float X = productAverage(s, cieX, start, end), Y = productAverage(s, cieY, start, end), Z = productAverage(s, cieZ, start, end);
    const float normalization = 1.0f / cieY.average(start, end);
    X *= normalization; Y *= normalization; Z *= normalization;
    rgb[0] =  3.240479f * X + -1.537150f * Y + -0.498535f * Z;
    rgb[1] = -0.969256f * X +  1.875991f * Y +  0.041556f * Z;
    rgb[2] =  0.055648f * X + -0.204043f * Y +  1.057311f * Z;
}

// InterpolatedSpectrum::zeroExtend (spectrum.cpp:630-648): one entry of value zero at the average spacing before a
// non-zero first entry and after a non-zero last one
bool zeroExtend(Spd &s) {
    if (s.lambda.size() < 2) return false;
    float avgSpacing = 0;
    for (size_t i = 0; i + 1 < s.lambda.size(); ++i) avgSpacing += s.lambda[i + 1] - s.lambda[i];
    avgSpacing /= s.lambda.size() - 1;
    if (s.value.front() != 0) {
        s.lambda.insert(s.lambda.begin(), s.lambda.front() - avgSpacing);
        s.value.insert(s.value.begin(), 0.0f);
    }
    if (s.value.back() != 0) {
        s.lambda.push_back(s.lambda.back() + avgSpacing);
        s.value.push_back(0.0f);
    }
    return true;
}

int convert(const float *wavelengths, const float *values, uint32_t n, bool sceneValue, float *rgb) {
    if (!wavelengths || !values || !rgb) return MTSAMD_EINVAL;
    Spd s;
    for (uint32_t i = 0; i < n; ++i) {
        if (i > 0 && wavelengths[i - 1] >= wavelengths[i]) return MTSAMD_EINVAL;   // must increase (spectrum.cpp:610-612)
        s.lambda.push_back(wavelengths[i]);
        s.value.push_back(values[i]);
    }
    if (sceneValue && !zeroExtend(s)) return MTSAMD_EINVAL;
    toRgb(s, rgb);
    if (sceneValue) for (int i = 0; i < 3; ++i) rgb[i] = std::max(0.0f, rgb[i]);       // Spectrum::clampNegative
    return MTSAMD_OK;
}

int convertFile(const char *path, bool sceneValue, float *rgb);

}  // namespace

extern "C" int mtsamd_spd_to_rgb(const float *wavelengths, const float *values, uint32_t n, float *rgb) {
    return convert(wavelengths, values, n, false, rgb);
}

extern "C" int mtsamd_spectrum_value_to_rgb(const float *wavelengths, const float *values, uint32_t n, float *rgb) {
    return convert(wavelengths, values, n, true, rgb);
}

extern "C" int mtsamd_blackbody_to_rgb(float temperature, float scale, float *rgb) {
    if (!rgb || !(temperature > 0)) return MTSAMD_EINVAL;
    toRgb(BlackBody{temperature}, rgb);
    for (int i = 0; i < 3; ++i) rgb[i] = std::max(0.0f, rgb[i]) * scale;
    return MTSAMD_OK;
}

extern "C" int mtsamd_spd_file_to_rgb(const char *path, float *rgb) { return convertFile(path, false, rgb); }
extern "C" int mtsamd_spectrum_file_to_rgb(const char *path, float *rgb) { return convertFile(path, true, rgb); }

namespace {
int convertFile(const char *path, bool sceneValue, float *rgb) {
    if (!path || !rgb) return MTSAMD_EINVAL;
    std::ifstream is(path);
    if (!is) return MTSAMD_EINVAL;
    std::vector<float> lambda, value;
    std::string line;
    while (std::getline(is, line)) {
        const size_t b = line.find_first_not_of(" \t\r\n");
        if (b == std::string::npos || line[b] == '#') continue;
        std::istringstream iss(line.substr(b));
        float l, v;
        if (!(iss >> l >> v)) break;
        lambda.push_back(l);
        value.push_back(v);
    }
    if (lambda.empty()) return MTSAMD_EINVAL;
    return convert(lambda.data(), value.data(), (uint32_t) lambda.size(), sceneValue, rgb);
}
}  // namespace
