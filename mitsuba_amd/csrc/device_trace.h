// Ray traversal of the 8-wide quantised BVH on wave64: persistent wavefronts that pull rays from a global counter
// (idle lanes are refilled as soon as enough of them have finished), one lane per ray, node and triangle work done
// in separate wave-uniform phases so that lanes with triangles to test wait for each other instead of stalling the
// box tests, and a short per-lane stack in LDS that spills to scratch.
// Replaces ShapeKDTree::rayIntersect + rayIntersectHavran + TriAccel::rayIntersect
// (src/librender/skdtree.cpp:112-142,207-226; include/mitsuba/render/sahkdtree3.h:178-308;
//  include/mitsuba/render/triaccel.h:96-158).
#pragma once
#include "device_math.h"
#include "bvh.h"

namespace mtsamd {

struct HitRec { float t, u, v; uint32_t tri; };     // tri = global triangle index | BSDF type << 28, 0xFFFFFFFF = miss
static constexpr uint32_t kTriIndexMask = 0x0fffffffu;

// kLdsStack entries per lane of the traversal stack live in LDS, kSpillStack more in scratch (bvh.h, next to the depth
// limit of the builders that they imply)
static constexpr int kNodeFloat4 = 5;       // a wide node is 5 x float4
// LDS of a traversal workgroup: kLdsStack stack entries per lane and a buffer of one prepared ray (two float4 and its
// index) per lane from which idle lanes are refilled.
static constexpr int kRayBufBytes = 36;     // per lane
static constexpr int kFetchHeads = 8, kFetchHeadStride = 16;    // one hand-out counter per XCD, 64 bytes apart
inline size_t traceLdsBytes(uint32_t blockThreads, int ldsEntries) {
    return (size_t) ldsEntries * blockThreads * 8 + (size_t) kRayBufBytes * blockThreads;
}

// Ray interval exactly as the reference prepares it before walking the tree: clip against the enlarged
// scene bounds (AABB::rayIntersect, include/mitsuba/core/aabb.h:308-338), apply the adaptive epsilon when
// mint == Epsilon (skdtree.cpp:126-129; the shadow query omits the max(..., Epsilon) clamp, :214-216).
MTS_DEV bool rayInterval(const DevScene &sc, V3 o, V3 d, float rayMint, float rayMaxt, bool shadow, float &mint, float &maxt) {
    float nearT = -kInf, farT = kInf;
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float minVal = sc.aabbMin[i], maxVal = sc.aabbMax[i];
        if (dd[i] == 0) {
            if (oo[i] < minVal || oo[i] > maxVal) return false;
        } else {
            const float rcp = 1.0f / dd[i];
            float t1 = (minVal - oo[i]) * rcp;
            float t2 = (maxVal - oo[i]) * rcp;
            if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; }
            nearT = fmaxf(t1, nearT);
            farT = fminf(t2, farT);
            if (!(nearT <= farT)) return false;
        }
    }
    float rayMinT = rayMint;
    if (rayMinT == kEpsilon) {
        float m = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
        if (!shadow) m = fmaxf(m, kEpsilon);
        rayMinT *= m;
    }
    mint = nearT; maxt = farT;
    if (rayMinT > mint) mint = rayMinT;
    if (rayMaxt < maxt) maxt = rayMaxt;
    return maxt > mint;
}

// Three-input bit operations (full rate, where compares and conditional moves run at half rate): (a & m) | (b & ~m),
// and a | b | c. Written out because the compiler turns the first back into a conditional move and the second into v_or3_b32.
MTS_DEV uint32_t bitSelect(uint32_t m, uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(d) : "v"(m), "v"(a), "v"(b));
    return d;
}
MTS_DEV uint32_t bitOr3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xfe" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// The reference's projection test on a precomputed triangle record, same operation order, no contraction. Written
// without early exits so that the three loads of the record are issued together and the wave takes no branches;
// a degenerate record (k == 3) computes garbage that the final predicate rejects.
MTS_DEV bool triTest(const float4 r0, const float4 r1, const float4 r2, V3 o, V3 d, float mint, float maxt,
                     float &t, float &u, float &v) {
#pragma clang fp contract(off)
    const uint32_t k = __float_as_uint(r0.x);
    const float n_u = r0.y, n_v = r0.z, n_d = r0.w;
    const float a_u = r1.x, a_v = r1.y, b_nu = r1.z, b_nv = r1.w;
    const float c_nu = r2.x, c_nv = r2.y;
    // the ray's components in the record's (u, v, k) order, picked with bit masks (all ones for k == 0, for k == 1)
    const uint32_t m0 = (uint32_t) ((int) (k - 1u) >> 31), m1 = (uint32_t) ((int) ((k ^ 1u) - 1u) >> 31);
#define MTS_PICK3(a0, a1, a2) __uint_as_float(bitSelect(m0, __float_as_uint(a0), bitSelect(m1, __float_as_uint(a1), __float_as_uint(a2))))
    const float o_u = MTS_PICK3(o.y, o.z, o.x), o_v = MTS_PICK3(o.z, o.x, o.y), o_k = MTS_PICK3(o.x, o.y, o.z);
    const float d_u = MTS_PICK3(d.y, d.z, d.x), d_v = MTS_PICK3(d.z, d.x, d.y), d_k = MTS_PICK3(d.x, d.y, d.z);
#undef MTS_PICK3
    t = (n_d - o_u * n_u - o_v * n_v - o_k) / (d_u * n_u + d_v * n_v + d_k);
    const float hu = o_u + t * d_u - a_u;
    const float hv = o_v + t * d_v - a_v;
    u = hv * b_nu + hu * b_nv;
    v = hu * c_nu + hv * c_nv;
    return (k <= 2u) & !(t < mint) & !(t > maxt) & (u >= 0) & (v >= 0) & (u + v <= 1.0f);
}

typedef __attribute__((address_space(3))) uint2 LdsUint2;
typedef float __attribute__((ext_vector_type(4))) RawFloat4;
typedef __attribute__((address_space(3))) RawFloat4 LdsFloat4;
MTS_DEV float4 toFloat4(RawFloat4 v) { return make_float4(v.x, v.y, v.z, v.w); }

// Per-lane traversal stack: first kLdsStack entries in LDS (lane-strided, conflict free; `stackStride` = threads of the
// workgroup, handed in by the caller -- read through blockDim.x it is fetched from the dispatch packet on every use), the rest in scratch. The
// pieces are separate variables on purpose: kept in one struct, the dynamically indexed spill array drags the stack
// pointer and the LDS base into scratch memory with it.
// Index of this lane's column in the LDS stack, recomputed where it is used (two VALU instructions) instead of being
// carried in a register across the traversal loop. `waveFirst` (wave-uniform, a scalar register) is the workgroup
// thread index of the wave's lane 0.
MTS_DEV uint32_t stackLane(uint32_t waveFirst) {
    uint32_t lane;      // volatile: computed where it is used, never hoisted out of the loop into a register of its own
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return waveFirst + lane;
}
MTS_DEV uint32_t waveFirstThread() { return (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x & ~63u)); }

// STACK selects the layout: STACK == 0 keeps kLdsStack entries in LDS and kSpillStack more in scratch (any tree the
// builders return); STACK = N > 0 keeps N entries in LDS and none in scratch, for trees of depth <= N + 2 -- without the
// spill path the loop has fewer branches and the kernel no scratch memory (5 % faster on a 1M-triangle scene).
template <int STACK> struct StackLayout {
    static constexpr int kLds = STACK > 0 ? STACK : kLdsStack;
    static constexpr int kSpill = STACK > 0 ? 1 : kSpillStack;
    static constexpr bool kSpills = STACK == 0;
};

template <int STACK>
MTS_DEV void stackPush(LdsUint2 *ldsStack, uint32_t stackStride, uint32_t waveFirst, uint2 (&spill)[StackLayout<STACK>::kSpill], int &sp, uint2 e) {
    typedef StackLayout<STACK> L;
    if (!L::kSpills || sp < L::kLds) { LdsUint2 *slot = ldsStack + (__umul24((uint32_t) sp, stackStride) + stackLane(waveFirst)); slot->x = e.x; slot->y = e.y; }
    else if (sp - L::kLds < L::kSpill) spill[sp - L::kLds] = e;
    ++sp;
}

template <int STACK>
MTS_DEV uint2 stackPop(LdsUint2 *ldsStack, uint32_t stackStride, uint32_t waveFirst, uint2 (&spill)[StackLayout<STACK>::kSpill], int &sp) {
    typedef StackLayout<STACK> L;
    --sp;
    if (!L::kSpills || sp < L::kLds) { const LdsUint2 *slot = ldsStack + (__umul24((uint32_t) sp, stackStride) + stackLane(waveFirst)); return make_uint2(slot->x, slot->y); }
    return spill[sp - L::kLds < L::kSpill ? sp - L::kLds : L::kSpill - 1];
}

// State of the ray a lane is working on.
struct LaneRay {
    V3 o, d;
    float idx, idy, idz;        // clamped reciprocal direction (slab tests only)
    float mint, maxt;           // maxt shrinks to the closest hit found so far
    uint32_t oct;               // sign octant of the direction
    HitRec hit;
    uint2 ng;                   // node group: x = index of the first internal child, y = imask | pending (priority order) << 8
    uint2 tg;                   // triangle group: x = first triangle record of the node, y = mask of triangles still to test (3 bits per slot)
    uint32_t tv;                // triangle group: which of those 3 x 8 bits stand for a record (the node's `leafTris`)
};

MTS_DEV void laneStart(LaneRay &r, V3 o, V3 d, float mint, float maxt) {
    r.o = o; r.d = d; r.mint = mint; r.maxt = maxt;
    // zero components stay finite; signs are taken from the reciprocals so that -0 picks consistent near/far planes
    const float big = 1e30f;
    // v_rcp_f32 (1 ulp) is enough here: the reciprocals only feed the padded slab tests
    r.idx = fabsf(d.x) > 1e-30f ? __builtin_amdgcn_rcpf(d.x) : copysignf(big, d.x);
    r.idy = fabsf(d.y) > 1e-30f ? __builtin_amdgcn_rcpf(d.y) : copysignf(big, d.y);
    r.idz = fabsf(d.z) > 1e-30f ? __builtin_amdgcn_rcpf(d.z) : copysignf(big, d.z);
    r.oct = (r.idx < 0 ? 1u : 0u) | (r.idy < 0 ? 2u : 0u) | (r.idz < 0 ? 4u : 0u);
    r.hit.t = kInf; r.hit.u = 0; r.hit.v = 0; r.hit.tri = 0xFFFFFFFFu;
    // Priority bit k of a node group stands for slot k ^ oct; the root is a pseudo group whose only member sits in slot 0.
    r.ng = make_uint2(0u, 1u | ((1u << r.oct) << 8));
    r.tg = make_uint2(0u, 0u);
    r.tv = 0u;
}

#define MTS_UBYTE(w, i) ((float) (((w) >> (8 * (i))) & 0xffu))

typedef float __attribute__((ext_vector_type(2))) Float2;

// Bit s of an 8-bit mask moves to bit s ^ oct: three conditional exchanges whose shift distances are 0 or 4, 2, 1.
MTS_DEV uint32_t xorPermute8(uint32_t x, uint32_t oct) {
    const uint32_t s4 = oct & 4u, s2 = oct & 2u, s1 = oct & 1u;
    x = ((x << s4) & 0xf0u) | ((x >> s4) & 0x0fu);
    x = ((x << s2) & 0xccu) | ((x >> s2) & 0x33u);
    x = ((x << s1) & 0xaau) | ((x >> s1) & 0x55u);
    return x;
}

// Bit s of an 8-bit mask to the kLeafTris bits from bit kLeafTris * s on
MTS_DEV uint32_t tripleBits8(uint32_t x) {
    if (kLeafTris == 3) {
        x = (x | (x << 8)) & 0x00f00fu;
        x = (x | (x << 4)) & 0x0c30c3u;
        x = (x | (x << 2)) & 0x249249u;
        return x * 7u;
    }
    x = (x | (x << 12)) & 0x000f000fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x * 15u;
}

typedef __attribute__((address_space(3))) uint32_t LdsUint;

// The record of bit b of the lane's triangle group: the node's records lie one after the other in the order of their bits
MTS_DEV const float4 *triRecord(const DevScene &sc, const LaneRay &r, uint32_t b) {
    return sc.tris + sc.triStride * (size_t) (r.tg.x + __popc(r.tv & ((1u << b) - 1u)));
}

// How a node visit turns its eight slab tests into a bit mask: MTS_VISIT_SIGNS (the shipped form) ORs the sign bits of
// three differences per slot and shifts them in with v_alignbit_b32; without it, max / min / compare, and
// MTS_VISIT_ADDC feeds the comparison's lane mask straight into the carry input of an add.
#ifndef MTS_VISIT_SIGNS
#define MTS_VISIT_SIGNS 1
#endif
#ifndef MTS_VISIT_ADDC
#define MTS_VISIT_ADDC 1
#endif
// mask = 2 * mask + (a <= b)
MTS_DEV void shiftInLessEqual(uint32_t &mask, float a, float b) {
    if (!MTS_VISIT_ADDC) { mask = (mask << 1) | (a <= b ? 1u : 0u); return; }
    asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(a), "v"(b) : "vcc");
}

// Visits the next pending internal child of the lane's node group: box tests of its 8 children, branch free, two
// children per packed multiply-add. The tests leave one bit per slot; the slots that hold internal children go into the
// node group in this ray's visiting order (slot ^ oct), the others select their (up to three) bits of the node's
// triangle mask. Empty slots have degenerate planes that no ray passes and no triangle bits.
template <int STACK>
MTS_DEV uint32_t visitNode(const DevScene &sc, LaneRay &r, LdsUint2 *ldsStack, uint32_t stackStride, uint32_t waveFirst, uint2 (&spill)[StackLayout<STACK>::kSpill], int &sp) {
    const uint32_t pending = (r.ng.y >> 8) & 0xffu;
    const uint32_t kbit = __ffs((int) pending) - 1;
    r.ng.y &= ~(0x100u << kbit);
    const uint32_t slot = kbit ^ r.oct;
    const uint32_t nodeIdx = r.ng.x + __popc(r.ng.y & 0xffu & ((1u << slot) - 1u));

    // (a copy of the top levels of the tree in LDS was measured and bought nothing: those nodes are L1 hits anyway)
    const float4 *np = sc.nodes + sc.nodeStride * (size_t) nodeIdx;
    const float4 n0 = np[0], n2 = np[2], n3 = np[3], n4 = np[4];
    const float3 n1 = *reinterpret_cast<const float3 *>(np + 1);      // child base, triangle base, triangle bits
    const uint32_t ebits = __float_as_uint(n0.w);
    const float sx = __uint_as_float((ebits & 0xffu) << 23), sy = __uint_as_float(((ebits >> 8) & 0xffu) << 23),
                sz = __uint_as_float(((ebits >> 16) & 0xffu) << 23);
    const uint32_t imask = ebits >> 24;
    // t = q * as + ap per plane. The far side is scaled up by a few ulps so that rounding in the slab arithmetic can
    // never cull a box the exact arithmetic would enter (the boxes themselves are padded at build time).
    const float kFar = 1.0f + 12.0f * 0x1p-23f;
    const float apx = (n0.x - r.o.x) * r.idx, apy = (n0.y - r.o.y) * r.idy, apz = (n0.z - r.o.z) * r.idz;
    const float asx = sx * r.idx, asy = sy * r.idy, asz = sz * r.idz;
    const float fpx = apx * kFar, fpy = apy * kFar, fpz = apz * kFar;
    const float fsx = asx * kFar, fsy = asy * kFar, fsz = asz * kFar;
    const float maxtPad = r.maxt * kFar;
    // all ones where the direction component is negative: there the high plane is the near one
    const uint32_t nx = (uint32_t) ((int) __float_as_uint(r.idx) >> 31), ny = (uint32_t) ((int) __float_as_uint(r.idy) >> 31),
                   nz = (uint32_t) ((int) __float_as_uint(r.idz) >> 31);
#define MTS_PICK(m, a, b) bitSelect(m, a, b)
    // near/far plane bytes per axis, two dwords (children 0-3, 4-7) each
    const uint32_t lox0 = __float_as_uint(n2.x), lox1 = __float_as_uint(n2.y), loy0 = __float_as_uint(n2.z), loy1 = __float_as_uint(n2.w);
    const uint32_t loz0 = __float_as_uint(n3.x), loz1 = __float_as_uint(n3.y), hix0 = __float_as_uint(n3.z), hix1 = __float_as_uint(n3.w);
    const uint32_t hiy0 = __float_as_uint(n4.x), hiy1 = __float_as_uint(n4.y), hiz0 = __float_as_uint(n4.z), hiz1 = __float_as_uint(n4.w);
    const uint32_t nearx[2] = { MTS_PICK(nx, hix0, lox0), MTS_PICK(nx, hix1, lox1) }, farx[2] = { MTS_PICK(nx, lox0, hix0), MTS_PICK(nx, lox1, hix1) };
    const uint32_t neary[2] = { MTS_PICK(ny, hiy0, loy0), MTS_PICK(ny, hiy1, loy1) }, fary[2] = { MTS_PICK(ny, loy0, hiy0), MTS_PICK(ny, loy1, hiy1) };
    const uint32_t nearz[2] = { MTS_PICK(nz, hiz0, loz0), MTS_PICK(nz, hiz1, loz1) }, farz[2] = { MTS_PICK(nz, loz0, hiz0), MTS_PICK(nz, loz1, hiz1) };
#undef MTS_PICK

    const Float2 asx2 = { asx, asx }, apx2 = { apx, apx }, fsx2 = { fsx, fsx }, fpx2 = { fpx, fpx };
    const Float2 asy2 = { asy, asy }, apy2 = { apy, apy }, fsy2 = { fsy, fsy }, fpy2 = { fpy, fpy };
    const Float2 asz2 = { asz, asz }, apz2 = { apz, apz }, fsz2 = { fsz, fsz }, fpz2 = { fpz, fpz };
    // (gathering the bits in two or four independent compare chains instead of one was measured: no faster)
    uint32_t hitSlots = 0;      // bit s: the box in slot s is entered within [mint, maxt]
    uint32_t missSlots = 0;     // (MTS_VISIT_SIGNS: the complement is gathered)
#pragma unroll
    for (int h = 1; h >= 0; --h) {
#pragma unroll
        for (int p = 1; p >= 0; --p) {
            const Float2 qx0 = { MTS_UBYTE(nearx[h], 2 * p), MTS_UBYTE(nearx[h], 2 * p + 1) }, qx1 = { MTS_UBYTE(farx[h], 2 * p), MTS_UBYTE(farx[h], 2 * p + 1) };
            const Float2 qy0 = { MTS_UBYTE(neary[h], 2 * p), MTS_UBYTE(neary[h], 2 * p + 1) }, qy1 = { MTS_UBYTE(fary[h], 2 * p), MTS_UBYTE(fary[h], 2 * p + 1) };
            const Float2 qz0 = { MTS_UBYTE(nearz[h], 2 * p), MTS_UBYTE(nearz[h], 2 * p + 1) }, qz1 = { MTS_UBYTE(farz[h], 2 * p), MTS_UBYTE(farz[h], 2 * p + 1) };
            const Float2 tx0 = __builtin_elementwise_fma(qx0, asx2, apx2), tx1 = __builtin_elementwise_fma(qx1, fsx2, fpx2);
            const Float2 ty0 = __builtin_elementwise_fma(qy0, asy2, apy2), ty1 = __builtin_elementwise_fma(qy1, fsy2, fpy2);
            const Float2 tz0 = __builtin_elementwise_fma(qz0, asz2, apz2), tz1 = __builtin_elementwise_fma(qz1, fsz2, fpz2);
#if MTS_VISIT_SIGNS
            // entered within [mint, maxt]  <=>  none of (far - near), (maxt - near), (far - mint) is negative: the sign
            // bits of the three differences (a difference of two floats is negative exactly when the second is larger),
            // ORed, go into the mask as a "missed" bit -- subtractions and bit operations run at twice the rate of compares
            const float tnB = fmaxf(fmaxf(tx0.y, ty0.y), tz0.y), tfB = fminf(fminf(tx1.y, ty1.y), tz1.y);
            const float tnA = fmaxf(fmaxf(tx0.x, ty0.x), tz0.x), tfA = fminf(fminf(tx1.x, ty1.x), tz1.x);
            const uint32_t sB = bitOr3(__float_as_uint(tfB - tnB), __float_as_uint(maxtPad - tnB), __float_as_uint(tfB - r.mint));
            const uint32_t sA = bitOr3(__float_as_uint(tfA - tnA), __float_as_uint(maxtPad - tnA), __float_as_uint(tfA - r.mint));
            missSlots = __builtin_amdgcn_alignbit(missSlots, sB, 31);
            missSlots = __builtin_amdgcn_alignbit(missSlots, sA, 31);
#else
            const float tnB = fmaxf(fmaxf(tx0.y, ty0.y), fmaxf(tz0.y, r.mint)), tfB = fminf(fminf(tx1.y, ty1.y), fminf(tz1.y, maxtPad));
            const float tnA = fmaxf(fmaxf(tx0.x, ty0.x), fmaxf(tz0.x, r.mint)), tfA = fminf(fminf(tx1.x, ty1.x), fminf(tz1.x, maxtPad));
            shiftInLessEqual(hitSlots, tnB, tfB);
            shiftInLessEqual(hitSlots, tnA, tfA);
#endif
        }
    }

    if (MTS_VISIT_SIGNS) hitSlots = ~missSlots & 0xffu;
    const uint32_t leafTris = __float_as_uint(n1.z);        // bit 3s + j: slot s is a leaf with more than j triangles
    const uint32_t triMask = tripleBits8(hitSlots) & leafTris;
    const uint32_t childPending = xorPermute8(hitSlots & imask, r.oct);
    r.tg = make_uint2(__float_as_uint(n1.y), triMask);     // the lane's triangle group is empty whenever it visits a node
    r.tv = leafTris;
    if (childPending) {
        // descend; siblings still pending in the current group wait on the stack
        if ((r.ng.y >> 8) & 0xffu) stackPush<STACK>(ldsStack, stackStride, waveFirst, spill, sp, r.ng);
        r.ng = make_uint2(__float_as_uint(n1.x), imask | (childPending << 8));
    }
    return triMask | (childPending << 24);
}

// One triangle record against the lane's ray. Equal-t ties are resolved towards the lowest global triangle index, so
// the result does not depend on traversal order (the reference keeps the last equal-t triangle in kd-leaf order,
// sahkdtree3.h:286-291, which depends on its tree). Returns true when the hit was accepted.
MTS_DEV bool testRecord(LaneRay &r, const float4 r0, const float4 r1, const float4 r2, bool enabled) {
    float t, u, v;
    // global triangle index in bits 0..27, BSDF type of its material in bits 28..30 (carried along for the shading kernel)
    const uint32_t tri = __float_as_uint(r2.z);
    const bool inside = triTest(r0, r1, r2, r.o, r.d, r.mint, r.maxt, t, u, v);
    // (without a hit so far hit.t is infinite and hit.tri all ones: the tie rule cannot fire)
    const bool accept = enabled & inside & !((t == r.hit.t) & ((tri & kTriIndexMask) >= (r.hit.tri & kTriIndexMask)));
    r.hit.t = accept ? t : r.hit.t; r.hit.u = accept ? u : r.hit.u; r.hit.v = accept ? v : r.hit.v;
    r.hit.tri = accept ? tri : r.hit.tri;
    r.maxt = accept ? t : r.maxt;
    return accept;
}

// (non-temporal loads of the records were measured: the closest-hit kernel half as slow again -- it lives on the caches)
MTS_DEV void loadTriangle(const float4 *tp, float4 &r0, float4 &r1, float4 &r2) {
    // 44 of the record's 48 bytes: the last word is not used
    r0 = tp[0]; r1 = tp[1];
    const float3 last = *reinterpret_cast<const float3 *>(tp + 2);
    r2 = make_float4(last.x, last.y, last.z, 0.0f);
}

// Tests the next triangle record of the lane's triangle group. Returns true when a hit was accepted.
MTS_DEV bool testNextTriangle(const DevScene &sc, LaneRay &r) {
    const uint32_t b = __ffs((int) r.tg.y) - 1;
    r.tg.y &= r.tg.y - 1;
    float4 r0, r1, r2;
    loadTriangle(triRecord(sc, r, b), r0, r1, r2);
    return testRecord(r, r0, r1, r2, true);
}

// Tests the next TWO records of the lane's triangle group (the second only if there is one) behind one wait for memory:
// both records are requested before the first is tested. An any-hit query stops after the first accepted hit.
#ifndef MTS_TRI_PAIRS
#define MTS_TRI_PAIRS 1
#endif
template <bool ANY_HIT>
MTS_DEV bool testNextTriangles(const DevScene &sc, LaneRay &r) {
    if (!MTS_TRI_PAIRS) return testNextTriangle(sc, r);
    const uint32_t b0 = __ffs((int) r.tg.y) - 1;
    r.tg.y &= r.tg.y - 1;
    const bool two = r.tg.y != 0;
    const uint32_t b1 = two ? __ffs((int) r.tg.y) - 1 : b0;
    r.tg.y &= r.tg.y - 1;       // (zero stays zero)
    float4 p0, p1, p2, q0, q1, q2;
    loadTriangle(triRecord(sc, r, b0), p0, p1, p2);
    loadTriangle(triRecord(sc, r, b1), q0, q1, q2);
    const bool first = testRecord(r, p0, p1, p2, true);
    const bool second = testRecord(r, q0, q1, q2, two & !(ANY_HIT & first));
    return first | second;
}

// What a traversal kernel does with a finished ray
enum TraceMode { TRACE_CLOSEST = 0, TRACE_OCCLUDED = 1 };

struct TraceArgs {
    const float4 *rayO, *rayD;      // origin.xyz + mint, direction.xyz + maxt of ray i at [i * stride]
    uint32_t stride, n;
    uint32_t blockThreads;          // threads per workgroup of the launch (the LDS stack's stride)
    // Ray hand-out: the rays are cut into `segments` contiguous ranges of `segmentLen` rays, each with a head counter
    // of its own (zero at launch, kFetchHeadStride words apart). A wave starts on the range of its own XCD, so that
    // the rays one XCD's L2 sees stay neighbours in the queue, takes chunks of `chunk` rays from it and moves on to the
    // next range when that one is empty.
    uint32_t *fetchHeads;
    uint32_t segments, segmentLen;
    uint32_t chunk;
    float4 *hits;                   // TRACE_CLOSEST: (t, u, v, triangle) per ray
    uint8_t *occluded;              // TRACE_OCCLUDED: one byte per ray
    unsigned long long *stats;      // COUNT: [0] node visits, [1] triangle tests
    uint32_t refillBelow;           // refill idle lanes when fewer than this many lanes hold a ray
    uint32_t triShare;              // run the triangle phase once this many 64ths of the busy lanes wait for it
};

template <int MODE>
MTS_DEV void laneFinish(const TraceArgs &a, uint32_t ray, const LaneRay &r, bool occluded) {
    if (MODE == TRACE_CLOSEST) {
        a.hits[ray] = make_float4(r.hit.t, r.hit.u, r.hit.v, __uint_as_float(r.hit.tri));
    } else {
        a.occluded[ray] = occluded ? 1 : 0;
    }
}

// The traversal loop shared by the closest-hit and the any-hit kernels. Every wavefront runs until the ray counter
// is exhausted and its own lanes have drained, so the grid may be any size >= 1.
// Rays reach the lanes in two steps. A producer step, run by all 64 lanes together, takes the next 64 rays of the
// wave's chunk, clips them against the scene bounds (rays that miss them are finished on the spot) and parks the
// survivors in the wave's LDS buffer. Lanes that run out of work pick their next ray from that buffer, which costs a
// couple of LDS reads, so they can be refilled while most of the wave is still busy.
template <int MODE, bool COUNT, int STACK>
MTS_DEV void traceLoop(const DevScene &sc, const TraceArgs &a, LdsUint2 *ldsStack, LdsFloat4 *rayBuf, LdsUint *rayBufIdx) {
    constexpr bool ANY_HIT = MODE != TRACE_CLOSEST;
    const uint32_t stackStride = a.blockThreads;
    const uint32_t lane = __lane_id();
    const unsigned long long laneBelow = (1ull << lane) - 1ull;
    // this wave's part of the ray buffer: 64 x (origin+mint, direction+maxt) and 64 ray indices
    const uint32_t waveFirst = waveFirstThread();
    LdsFloat4 *bufRay = rayBuf + 2 * waveFirst;
    LdsUint *bufIdx = rayBufIdx + waveFirst;
    LaneRay r;
    r.tg = make_uint2(0u, 0u); r.ng = make_uint2(0u, 0u); r.tv = 0u;
    uint2 spill[StackLayout<STACK>::kSpill];
    int sp = 0;
    int ray = -1;               // index of the ray this lane holds, -1 = idle
    bool exhausted = false;     // wave-uniform: the counter has run past the last ray
    uint32_t chunkNext = 0, chunkEnd = 0;   // wave-uniform: the part of the wave's current chunk not prepared yet
    // wave-uniform: the range this wave draws from (first its XCD's: HW_REG_XCC_ID, bits 3:0) and how many it found empty
    uint32_t seg = ((uint32_t) __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) % a.segments, segTried = 0;
    uint32_t bufHead = 0, bufCount = 0;     // wave-uniform: prepared rays waiting in LDS
    uint32_t nv = 0, nt = 0, ne = 0, ns = 0;
    // Diagnostic build (-DMTS_COUNT_ITERATIONS): the two counters that otherwise hold the visits without a child hit
    // count the wave iterations of the node phase and of the triangle phase instead (lane utilisation = visits or
    // records per iteration / 64; measured on the 1M-triangle scene: 78 % and about 30 %).
#ifdef MTS_COUNT_ITERATIONS
    uint32_t wn = 0, wt = 0;    // node phase: counted by its first active lane; triangle phase: wave-uniform
#endif

    while (true) {
        const unsigned long long idleMask = __ballot(ray < 0);
        // ---- producer: prepare the next 64 rays of the chunk when the buffer has run dry
        if (idleMask && bufCount == 0 && !exhausted) {
            while (chunkNext >= chunkEnd && segTried < a.segments) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(a.fetchHeads + seg * kFetchHeadStride, a.chunk);
                base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
                const uint32_t segFirst = seg * a.segmentLen;
                const uint32_t segEnd = segFirst + a.segmentLen < a.n ? segFirst + a.segmentLen : a.n;
                if (segFirst < a.n && base < segEnd - segFirst) {
                    chunkNext = segFirst + base;
                    chunkEnd = chunkNext + a.chunk < segEnd ? chunkNext + a.chunk : segEnd;
                } else {        // this range is handed out: on to the next one
                    seg = seg + 1 < a.segments ? seg + 1 : 0;
                    ++segTried;
                }
            }
            if (chunkNext >= chunkEnd) {
                exhausted = true;
            } else {
                const uint32_t i = chunkNext + lane;
                chunkNext = chunkNext + 64u < chunkEnd ? chunkNext + 64u : chunkEnd;
                bool ok = false;
                float4 ro = make_float4(0, 0, 0, 0), rd = ro;
                if (i < chunkEnd) {
                    ro = a.rayO[(size_t) i * a.stride]; rd = a.rayD[(size_t) i * a.stride];
                    float mint, maxt;
                    ok = rayInterval(sc, mk(ro.x, ro.y, ro.z), mk(rd.x, rd.y, rd.z), ro.w, rd.w, ANY_HIT, mint, maxt);
                    ro.w = mint; rd.w = maxt;
                    if (!ok) {      // misses the scene bounds: finished without a traversal
                        LaneRay miss;
                        miss.hit.t = kInf; miss.hit.u = 0; miss.hit.v = 0; miss.hit.tri = 0xFFFFFFFFu;
                        laneFinish<MODE>(a, i, miss, false);
                    }
                }
                const unsigned long long okMask = __ballot(ok);
                if (ok) {
                    const uint32_t pos = (uint32_t) __popcll(okMask & laneBelow);
                    const RawFloat4 wo = { ro.x, ro.y, ro.z, ro.w }, wd = { rd.x, rd.y, rd.z, rd.w };
                    bufRay[2 * pos] = wo; bufRay[2 * pos + 1] = wd; bufIdx[pos] = i;
                }
                bufHead = 0; bufCount = (uint32_t) __popcll(okMask);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- consumer: idle lanes take prepared rays
        if (idleMask && bufCount) {
            const uint32_t k = (uint32_t) __popcll(idleMask & laneBelow);
            if (ray < 0 && k < bufCount) {
                const RawFloat4 wo = bufRay[2 * (bufHead + k)], wd = bufRay[2 * (bufHead + k) + 1];
                laneStart(r, mk(wo.x, wo.y, wo.z), mk(wd.x, wd.y, wd.z), wo.w, wd.w);
                sp = 0;
                ray = (int) bufIdx[bufHead + k];
            }
            const uint32_t taken = min((uint32_t) __popcll(idleMask), bufCount);
            bufHead += taken; bufCount -= taken;
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t nHeld = (uint32_t) __popcll(__ballot(ray >= 0));
        const bool moreRays = bufCount != 0 || !exhausted;
        if (nHeld == 0) {
            if (!moreRays) break;
            continue;
        }
        // leave the traversal loop to refill as soon as this many lanes are idle, as long as there is something to refill with
        const uint32_t keepAbove = moreRays ? a.refillBelow : 1u;
        if (moreRays && nHeld < keepAbove && nHeld < 64u && bufCount == 0) continue;    // buffer ran dry mid-refill: produce first

        // ---- traverse until too few lanes hold a ray
        while (true) {
            const bool busy = ray >= 0;
            const bool hasTri = busy && r.tg.y != 0;
            const uint32_t nTri = (uint32_t) __popcll(__ballot(hasTri));
            const uint32_t nBusy = (uint32_t) __popcll(__ballot(busy));
            if (nBusy < keepAbove) break;
            if (nTri == nBusy || nTri * 64u >= a.triShare * nBusy) {
#ifdef MTS_COUNT_ITERATIONS
                if (COUNT) wt++;
#endif
                // triangle phase: one record for every lane that has some
                if (hasTri) {
                    if (COUNT) nt += (MTS_TRI_PAIRS && (r.tg.y & (r.tg.y - 1u))) ? 2u : 1u;        // records fetched
                    const bool accepted = testNextTriangles<ANY_HIT>(sc, r);
                    if (ANY_HIT && accepted) { laneFinish<MODE>(a, (uint32_t) ray, r, true); ray = -1; r.tg.y = 0; }
                }
            } else if (busy && !hasTri) {
#ifdef MTS_COUNT_ITERATIONS
                if (COUNT && __builtin_amdgcn_readfirstlane((int) __lane_id()) == (int) __lane_id()) wn++;
#endif
                // node phase: lanes without triangles waiting pop if needed and visit one node
                bool go = true;
                if (!((r.ng.y >> 8) & 0xffu)) {
                    if (sp == 0) { laneFinish<MODE>(a, (uint32_t) ray, r, false); ray = -1; go = false; }
                    else r.ng = stackPop<STACK>(ldsStack, stackStride, waveFirst, spill, sp);
                }
                if (go) {
                    if (COUNT) nv++;
                    const uint32_t hm = visitNode<STACK>(sc, r, ldsStack, stackStride, waveFirst, spill, sp);
                    if (COUNT && hm == 0) { ne++; if (r.hit.tri != 0xFFFFFFFFu) ns++; }
                }
            }
        }
    }
    if (COUNT) {
        atomicAdd(&a.stats[0], (unsigned long long) nv);
        atomicAdd(&a.stats[1], (unsigned long long) nt);
#ifdef MTS_COUNT_ITERATIONS
        if (MODE == TRACE_CLOSEST) { atomicAdd(&a.stats[4], (unsigned long long) wn); if (lane == 0) atomicAdd(&a.stats[5], (unsigned long long) wt); }
#else
        if (MODE == TRACE_CLOSEST) { atomicAdd(&a.stats[4], (unsigned long long) ne); atomicAdd(&a.stats[5], (unsigned long long) ns); }
#endif
    }
}

// Traversal of the rays the lanes of one wavefront hold right now, nothing fetched and nothing refilled: the same
// node / triangle phases as traceLoop, run until every lane has finished. `busy` says which lanes hold a ray (prepared
// with laneStart); for ANY_HIT it comes back true on the lanes whose ray was occluded. BVH nodes are read from global
// memory.
template <bool ANY_HIT>
MTS_DEV void traceHeldRays(const DevScene &sc, LaneRay &r, bool &busy, LdsUint2 *ldsStack, uint32_t stackStride, uint32_t triShare) {
    uint2 spill[StackLayout<0>::kSpill];
    int sp = 0;
    bool occluded = false;
    const uint32_t waveFirst = waveFirstThread();
    while (true) {
        const bool hasTri = busy && r.tg.y != 0;
        const uint32_t nTri = (uint32_t) __popcll(__ballot(hasTri));
        const uint32_t nBusy = (uint32_t) __popcll(__ballot(busy));
        if (nBusy == 0) break;
        if (nTri == nBusy || nTri * 64u >= triShare * nBusy) {
            if (hasTri) {
                const bool accepted = testNextTriangles<ANY_HIT>(sc, r);
                if (ANY_HIT && accepted) { occluded = true; busy = false; r.tg.y = 0; }
            }
        } else if (busy && !hasTri) {
            bool go = true;
            if (!((r.ng.y >> 8) & 0xffu)) {
                if (sp == 0) { busy = false; go = false; }
                else r.ng = stackPop<0>(ldsStack, stackStride, waveFirst, spill, sp);
            }
            if (go) visitNode<0>(sc, r, ldsStack, stackStride, waveFirst, spill, sp);
        }
    }
    if (ANY_HIT) busy = occluded;
}

}  // namespace mtsamd
