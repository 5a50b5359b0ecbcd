// TEST INFRASTRUCTURE ONLY -- the counter-based sample stream shared by the CPU oracle and the
// GPU kernels. It stands in for the reference's Sampler::next1D/next2D contract
// (src/samplers/independent.cpp:95-103): one independent uniform [0,1) stream per camera sample.
// The reference's own streams (SFMT state cloned per worker thread) depend on thread scheduling,
// so image parity with the reference is statistical; using one keyed stream on both sides here makes
// oracle-vs-GPU comparisons per path instead.
//
// Stream definition: PCG32 (XSH-RR 64/32) seeded with initstate = splitmix64(seed),
// initseq = key, where key = pixel_index * spp_total + sample_index.
// next1D = (u32 >> 8) * 2^-24; next2D = (next1D, next1D) in that order.
#pragma once
#include "vec.h"

namespace orc {

inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct Pcg32 {
    uint64_t state, inc;
    Pcg32(uint32_t seed, uint64_t key) {
        state = 0;
        inc = (key << 1) | 1u;
        nextU32();
        state += splitmix64(seed);
        nextU32();
    }
    uint32_t nextU32() {
        uint64_t old = state;
        state = old * 6364136223846793005ull + inc;
        uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t) (old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
    }
    float next1D() { return (float) (nextU32() >> 8) * 0x1p-24f; }
    Vec2 next2D() { float a = next1D(); float b = next1D(); return Vec2{ a, b }; }
};

}  // namespace orc
