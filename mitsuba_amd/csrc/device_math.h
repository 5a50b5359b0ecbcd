// Device-side vector helpers, the shared sample stream and small numeric routines.
// All arithmetic is single precision; the translation unit is compiled with -ffp-contract=off so that
// products and sums round like the scalar SSE code of the reference build (build/config-linux-gcc.py:7).
#pragma once
#include "device_types.h"

namespace mtsamd {

#define MTS_DEV __device__ __forceinline__

static constexpr float kEpsilon = 1e-4f;          // include/mitsuba/core/constants.h:28
static constexpr float kShadowEpsilon = 1e-3f;    // constants.h:29
static constexpr float kPi = 3.14159265358979323846f;
static constexpr float kInvPi = 0.31830988618379067154f;
static constexpr float kInf = __builtin_huge_valf();

// exp / log as the reference computes them: in double precision, rounded to float once. Real calls, not inlined:
// each expansion is about a hundred double-precision instructions, and a dozen of them inside the shading kernel
// cost it registers (spills) and instruction-cache room on paths that never run them.
#ifndef MTS_REF_NOINLINE
#define MTS_REF_NOINLINE 1
#endif
#if MTS_REF_NOINLINE
#define MTS_REF_FN __device__ __attribute__((noinline)) inline
#else
#define MTS_REF_FN MTS_DEV
#endif
MTS_REF_FN float expRef(float x) { return (float) exp((double) x); }
MTS_REF_FN float logRef(float x) { return (float) log((double) x); }
// The library's trigonometric and power functions behind real calls for the same reason (sincosf alone brings a
// large-argument reduction of several hundred instructions to every place it is inlined into).
#ifndef MTS_MATH_NOINLINE
#define MTS_MATH_NOINLINE 1
#endif
#if MTS_MATH_NOINLINE
#define MTS_MATH_FN __device__ __attribute__((noinline)) inline
#else
#define MTS_MATH_FN MTS_DEV
#endif
MTS_MATH_FN float2 sinCosM(float x) { float s, c; sincosf(x, &s, &c); return make_float2(s, c); }
MTS_DEV void sincosM(float x, float *s, float *c) { const float2 r = sinCosM(x); *s = r.x; *c = r.y; }
MTS_MATH_FN float cosM(float x) { return cosf(x); }
MTS_MATH_FN float tanM(float x) { return tanf(x); }
MTS_MATH_FN float atanM(float x) { return atanf(x); }
MTS_MATH_FN float atan2M(float y, float x) { return atan2f(y, x); }
MTS_MATH_FN float acosM(float x) { return acosf(x); }
MTS_MATH_FN float powM(float x, float y) { return powf(x, y); }

struct V3 { float x, y, z; };

MTS_DEV V3 mk(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
MTS_DEV V3 mk(float s) { return mk(s, s, s); }
MTS_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
MTS_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
MTS_DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
MTS_DEV V3 operator/(V3 a, V3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
MTS_DEV V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
MTS_DEV V3 operator*(float s, V3 a) { return mk(a.x * s, a.y * s, a.z * s); }
// division by a scalar multiplies by the reciprocal, as TVector3::operator/ does (include/mitsuba/core/vector.h)
MTS_DEV V3 operator/(V3 a, float s) { float r = 1.0f / s; return mk(a.x * r, a.y * r, a.z * r); }
MTS_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
MTS_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MTS_DEV V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MTS_DEV float length(V3 a) { return sqrtf(dot(a, a)); }
MTS_DEV V3 normalize(V3 a) { return a / length(a); }
MTS_DEV bool isZero(V3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
MTS_DEV float maxComp(V3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
MTS_DEV V3 ld3(const float *p, uint32_t i) { return mk(p[3 * (size_t) i], p[3 * (size_t) i + 1], p[3 * (size_t) i + 2]); }
MTS_DEV V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
MTS_DEV float safeSqrt(float v) { return sqrtf(fmaxf(0.0f, v)); }
// Spectrum::getLuminance (include/mitsuba/core/spectrum.h:754-756)
MTS_DEV float luminance(V3 s) { return s.x * 0.212671f + s.y * 0.715160f + s.z * 0.072169f; }

struct Frame3 { V3 s, t, n; };

// coordinateSystem (src/libcore/util.cpp:592-601)
MTS_DEV void coordinateSystem(V3 a, V3 &b, V3 &c) {
    if (fabsf(a.x) > fabsf(a.y)) {
        float invLen = 1.0f / sqrtf(a.x * a.x + a.z * a.z);
        c = mk(a.z * invLen, 0.0f, -a.x * invLen);
    } else {
        float invLen = 1.0f / sqrtf(a.y * a.y + a.z * a.z);
        c = mk(0.0f, a.z * invLen, -a.y * invLen);
    }
    b = cross(c, a);
}

MTS_DEV V3 toLocal(const Frame3 &f, V3 v) { return mk(dot(v, f.s), dot(v, f.t), dot(v, f.n)); }
MTS_DEV V3 toWorld(const Frame3 &f, V3 v) { return f.s * v.x + f.t * v.y + f.n * v.z; }

// ---- sample stream: PCG32 keyed by (seed, pixel_index * spp_total + sample_index) --------------------
MTS_DEV uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct Rng {
    uint64_t state, inc;
    MTS_DEV uint32_t nextU32() {
        uint64_t old = state;
        state = old * 6364136223846793005ull + inc;
        uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t) (old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
    }
    MTS_DEV float next1D() { return (float) (nextU32() >> 8) * 0x1p-24f; }
    MTS_DEV void init(uint32_t seed, uint64_t key) {
        state = 0;
        inc = (key << 1) | 1u;
        nextU32();
        state += splitmix64(seed);
        nextU32();
    }
};

}  // namespace mtsamd
