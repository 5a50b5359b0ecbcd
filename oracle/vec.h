// TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the Mitsuba 0.5 `path` hot path.
// Nothing under oracle/ is part of the shipped GPU product; it is the checker the HIP
// kernels are compared against and the host-CPU baseline that bench.py times.
//
// Small vector / frame / ray types. Single precision throughout, mirroring the reference
// build flavour (-DSINGLE_PRECISION -DSPECTRUM_SAMPLES=3, build/config-linux-gcc.py:7):
// Float = float, Spectrum = 3 linear-RGB floats (include/mitsuba/core/spectrum.h:816-826).
#pragma once
#include <cmath>
#include <cstdint>
#include <algorithm>
#include <limits>

namespace orc {

// include/mitsuba/core/constants.h:28-31,62-68
static const float Epsilon = 1e-4f;
static const float ShadowEpsilon = 1e-3f;
static const float DeltaEpsilon = 1e-3f;
static const float M_PI_F = 3.14159265358979323846f;
static const float INV_PI_F = 0.31830988618379067154f;
static const float INV_TWOPI_F = 0.15915494309189533577f;
static const float INV_FOURPI_F = 0.07957747154594766788f;
static const float RCPOVERFLOW_F = 0x1p-128f;

struct Vec2 { float x, y; };

struct Vec3 {
    float x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    explicit Vec3(float v) : x(v), y(v), z(v) {}
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float operator[](int i) const { return (&x)[i]; }
    float &operator[](int i) { return (&x)[i]; }
    Vec3 operator+(const Vec3 &b) const { return Vec3(x + b.x, y + b.y, z + b.z); }
    Vec3 operator-(const Vec3 &b) const { return Vec3(x - b.x, y - b.y, z - b.z); }
    Vec3 operator*(const Vec3 &b) const { return Vec3(x * b.x, y * b.y, z * b.z); }
    Vec3 operator/(const Vec3 &b) const { return Vec3(x / b.x, y / b.y, z / b.z); }
    Vec3 operator*(float s) const { return Vec3(x * s, y * s, z * s); }
    // TVector3::operator/ multiplies by the reciprocal (include/mitsuba/core/vector.h)
    Vec3 operator/(float s) const { float r = 1.0f / s; return Vec3(x * r, y * r, z * r); }
    Vec3 operator-() const { return Vec3(-x, -y, -z); }
    Vec3 &operator+=(const Vec3 &b) { x += b.x; y += b.y; z += b.z; return *this; }
    Vec3 &operator*=(const Vec3 &b) { x *= b.x; y *= b.y; z *= b.z; return *this; }
    Vec3 &operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
    Vec3 &operator/=(float s) { float r = 1.0f / s; x *= r; y *= r; z *= r; return *this; }
    float lengthSquared() const { return x * x + y * y + z * z; }
    float length() const { return std::sqrt(lengthSquared()); }
    bool isZero() const { return x == 0 && y == 0 && z == 0; }
    float max() const { return std::max(std::max(x, y), z); }
};

inline Vec3 operator*(float s, const Vec3 &v) { return v * s; }
inline float dot(const Vec3 &a, const Vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float absDot(const Vec3 &a, const Vec3 &b) { return std::abs(dot(a, b)); }
inline Vec3 cross(const Vec3 &a, const Vec3 &b) {
    return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline Vec3 normalize(const Vec3 &v) { return v / v.length(); }

typedef Vec3 Spectrum;  // linear RGB

// include/mitsuba/core/spectrum.h:754-756
inline float luminance(const Spectrum &s) {
    return s.x * 0.212671f + s.y * 0.715160f + s.z * 0.072169f;
}

inline float safe_sqrt(float v) { return std::sqrt(std::max(0.0f, v)); }
inline float signum(float v) { return copysignf(1.0f, v); }
// include/mitsuba/core/math.h:195-205: on Linux/x86-64 exp/log run in double and round to float
inline float fastexp(float v) { return (float) ::exp((double) v); }
inline float fastlog(float v) { return (float) ::log((double) v); }

// src/libcore/util.cpp:592-601
inline void coordinateSystem(const Vec3 &a, Vec3 &b, Vec3 &c) {
    if (std::abs(a.x) > std::abs(a.y)) {
        float invLen = 1.0f / std::sqrt(a.x * a.x + a.z * a.z);
        c = Vec3(a.z * invLen, 0.0f, -a.x * invLen);
    } else {
        float invLen = 1.0f / std::sqrt(a.y * a.y + a.z * a.z);
        c = Vec3(0.0f, a.z * invLen, -a.y * invLen);
    }
    b = cross(c, a);
}

// include/mitsuba/core/frame.h:37-85
struct Frame {
    Vec3 s, t, n;
    Frame() {}
    explicit Frame(const Vec3 &n_) : n(n_) { coordinateSystem(n, s, t); }
    Vec3 toLocal(const Vec3 &v) const { return Vec3(dot(v, s), dot(v, t), dot(v, n)); }
    Vec3 toWorld(const Vec3 &v) const { return s * v.x + t * v.y + n * v.z; }
    static float cosTheta(const Vec3 &v) { return v.z; }
    static float cosTheta2(const Vec3 &v) { return v.z * v.z; }
    static float sinTheta2(const Vec3 &v) { return 1.0f - v.z * v.z; }
    static float tanTheta(const Vec3 &v) {
        float temp = 1 - v.z * v.z;
        if (temp <= 0.0f) return 0.0f;
        return std::sqrt(temp) / v.z;
    }
};

// src/libcore/util.cpp:603-608
inline void computeShadingFrame(const Vec3 &n, const Vec3 &dpdu, Frame &frame) {
    frame.n = n;
    frame.s = normalize(dpdu - frame.n * dot(frame.n, dpdu));
    frame.t = cross(frame.n, frame.s);
}

// include/mitsuba/core/ray.h:44-96
struct Ray {
    Vec3 o; float mint;
    Vec3 d; float maxt;
    Vec3 dRcp;
    Ray() : mint(Epsilon), maxt(std::numeric_limits<float>::infinity()) {}
    Ray(const Vec3 &o_, const Vec3 &d_)
        : o(o_), mint(Epsilon), d(d_), maxt(std::numeric_limits<float>::infinity()) { setRcp(); }
    Ray(const Vec3 &o_, const Vec3 &d_, float mint_, float maxt_)
        : o(o_), mint(mint_), d(d_), maxt(maxt_) { setRcp(); }
    void setRcp() { dRcp = Vec3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }
    Vec3 operator()(float t) const { return o + d * t; }
};

struct AABB {
    Vec3 min, max;
    AABB() { reset(); }
    void reset() {
        min = Vec3(std::numeric_limits<float>::infinity());
        max = Vec3(-std::numeric_limits<float>::infinity());
    }
    void expandBy(const Vec3 &p) {
        for (int i = 0; i < 3; ++i) { min[i] = std::min(min[i], p[i]); max[i] = std::max(max[i], p[i]); }
    }
    void expandBy(const AABB &b) { expandBy(b.min); expandBy(b.max); }
    void clip(const AABB &b) {
        for (int i = 0; i < 3; ++i) { min[i] = std::max(min[i], b.min[i]); max[i] = std::min(max[i], b.max[i]); }
    }
    bool isValid() const { return max.x >= min.x && max.y >= min.y && max.z >= min.z; }
    float surfaceArea() const {
        Vec3 d = max - min;
        return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
    }
    // include/mitsuba/core/aabb.h:308-338
    bool rayIntersect(const Ray &ray, float &nearT, float &farT) const {
        nearT = -std::numeric_limits<float>::infinity();
        farT = std::numeric_limits<float>::infinity();
        for (int i = 0; i < 3; i++) {
            const float origin = ray.o[i];
            const float minVal = min[i], maxVal = max[i];
            if (ray.d[i] == 0) {
                if (origin < minVal || origin > maxVal) return false;
            } else {
                float t1 = (minVal - origin) * ray.dRcp[i];
                float t2 = (maxVal - origin) * ray.dRcp[i];
                if (t1 > t2) std::swap(t1, t2);
                nearT = std::max(t1, nearT);
                farT = std::min(t2, farT);
                if (!(nearT <= farT)) return false;
            }
        }
        return true;
    }
};

}  // namespace orc
