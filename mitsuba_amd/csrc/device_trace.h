// Ray traversal of the 8-wide quantised BVH: one lane per ray, wave64, a short per-lane stack in LDS that
// spills to scratch. Replaces ShapeKDTree::rayIntersect + rayIntersectHavran + TriAccel::rayIntersect
// (src/librender/skdtree.cpp:112-142,207-226; include/mitsuba/render/sahkdtree3.h:178-308;
//  include/mitsuba/render/triaccel.h:96-158).
#pragma once
#include "device_math.h"

namespace mtsamd {

struct HitRec { float t, u, v; uint32_t tri; };     // tri = global triangle index, 0xFFFFFFFF = miss

static constexpr int kLdsStack = 12;        // entries per lane kept in LDS
static constexpr int kSpillStack = 16;      // further entries in scratch
static constexpr int kTraceBlock = 256;     // threads per workgroup of the traversal kernels

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

// The reference's projection test on a precomputed triangle record, same operation order, no contraction.
MTS_DEV bool triTest(const float4 r0, const float4 r1, const float4 r2, V3 o, V3 d, float mint, float maxt,
                     float &t, float &u, float &v) {
#pragma clang fp contract(off)
    const uint32_t k = __float_as_uint(r0.x);
    if (k > 2u) return false;
    const float n_u = r0.y, n_v = r0.z, n_d = r0.w;
    const float a_u = r1.x, a_v = r1.y, b_nu = r1.z, b_nv = r1.w;
    const float c_nu = r2.x, c_nv = r2.y;
    const float o_u = k == 0 ? o.y : (k == 1 ? o.z : o.x);
    const float o_v = k == 0 ? o.z : (k == 1 ? o.x : o.y);
    const float o_k = k == 0 ? o.x : (k == 1 ? o.y : o.z);
    const float d_u = k == 0 ? d.y : (k == 1 ? d.z : d.x);
    const float d_v = k == 0 ? d.z : (k == 1 ? d.x : d.y);
    const float d_k = k == 0 ? d.x : (k == 1 ? d.y : d.z);
    t = (n_d - o_u * n_u - o_v * n_v - o_k) / (d_u * n_u + d_v * n_v + d_k);
    if (t < mint || t > maxt) return false;
    const float hu = o_u + t * d_u - a_u;
    const float hv = o_v + t * d_v - a_v;
    u = hv * b_nu + hu * b_nv;
    v = hu * c_nu + hv * c_nv;
    return u >= 0 && v >= 0 && u + v <= 1.0f;
}

// Per-lane traversal stack: first kLdsStack entries in LDS (lane-strided, conflict free), the rest in scratch.
struct TravStack {
    uint2 *lds;                 // &ldsStack[0][threadIdx.x]
    uint2 spill[kSpillStack];
    int sp;
    MTS_DEV void push(uint2 e) {
        if (sp < kLdsStack) lds[sp * kTraceBlock] = e;
        else if (sp - kLdsStack < kSpillStack) spill[sp - kLdsStack] = e;
        ++sp;
    }
    MTS_DEV uint2 pop() {
        --sp;
        if (sp < kLdsStack) return lds[sp * kTraceBlock];
        return spill[sp - kLdsStack < kSpillStack ? sp - kLdsStack : kSpillStack - 1];
    }
};

#define MTS_UBYTE(w, i) ((float) (((w) >> (8 * (i))) & 0xffu))

// Walks the BVH for one ray. ANY_HIT: stop at the first accepted triangle. COUNT: accumulate node visits and
// triangle tests (profiling builds). Equal-t ties are resolved towards the lowest global triangle index, so the
// result does not depend on traversal order (the reference keeps the last equal-t triangle in kd-leaf order,
// sahkdtree3.h:286-291, which depends on its tree).
template <bool ANY_HIT, bool COUNT>
MTS_DEV bool traverse(const DevScene &sc, V3 o, V3 d, float mint, float maxt, HitRec &hit, uint2 *ldsStack,
                      uint32_t &nodeVisits, uint32_t &triTests) {
    hit.t = kInf; hit.u = 0; hit.v = 0; hit.tri = 0xFFFFFFFFu;
    bool found = false;

    // reciprocal direction for the slab tests only (clamped so that zero components stay finite)
    const float big = 1e30f;
    const float idx = fabsf(d.x) > 1e-30f ? 1.0f / d.x : copysignf(big, d.x);
    const float idy = fabsf(d.y) > 1e-30f ? 1.0f / d.y : copysignf(big, d.y);
    const float idz = fabsf(d.z) > 1e-30f ? 1.0f / d.z : copysignf(big, d.z);
    // signs are taken from the reciprocals so that -0 components pick consistent near/far planes
    const bool nx = idx < 0, ny = idy < 0, nz = idz < 0;
    const uint32_t oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);

    TravStack st;
    st.lds = ldsStack;
    st.sp = 0;

    // current node group: x = index of the first internal child, y = imask | pending (priority space) << 8.
    // Priority bit k stands for slot k ^ oct; the root is a pseudo group whose only member sits in slot 0.
    uint2 group = make_uint2(0u, 1u | ((1u << oct) << 8));
    const float kSlabPad = 4.0f * 0x1p-23f;

    while (true) {
        uint32_t pending = group.y >> 8;
        if (pending == 0) {
            if (st.sp == 0) break;
            group = st.pop();
            continue;
        }
        // next internal child in priority order
        const uint32_t kbit = __ffs((int) pending) - 1;
        pending &= pending - 1;
        group.y = (group.y & 0xffu) | (pending << 8);
        const uint32_t slot = kbit ^ oct;
        const uint32_t nodeIdx = group.x + __popc(group.y & 0xffu & ((1u << slot) - 1u));

        const float4 *np = sc.nodes + 5 * (size_t) nodeIdx;
        const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
        if (COUNT) nodeVisits++;
        const uint32_t ebits = __float_as_uint(n0.w);
        const float sx = __uint_as_float((ebits & 0xffu) << 23), sy = __uint_as_float(((ebits >> 8) & 0xffu) << 23),
                    sz = __uint_as_float(((ebits >> 16) & 0xffu) << 23);
        const uint32_t imask = ebits >> 24;
        const float apx = (n0.x - o.x) * idx, apy = (n0.y - o.y) * idy, apz = (n0.z - o.z) * idz;
        const float asx = sx * idx, asy = sy * idy, asz = sz * idz;
        const uint32_t meta0 = __float_as_uint(n1.z), meta1 = __float_as_uint(n1.w);
        // near/far plane bytes per axis, two dwords (children 0-3, 4-7) each
        const uint32_t lox0 = __float_as_uint(n2.x), lox1 = __float_as_uint(n2.y), loy0 = __float_as_uint(n2.z), loy1 = __float_as_uint(n2.w);
        const uint32_t loz0 = __float_as_uint(n3.x), loz1 = __float_as_uint(n3.y), hix0 = __float_as_uint(n3.z), hix1 = __float_as_uint(n3.w);
        const uint32_t hiy0 = __float_as_uint(n4.x), hiy1 = __float_as_uint(n4.y), hiz0 = __float_as_uint(n4.z), hiz1 = __float_as_uint(n4.w);
        const uint32_t nearx[2] = { nx ? hix0 : lox0, nx ? hix1 : lox1 }, farx[2] = { nx ? lox0 : hix0, nx ? lox1 : hix1 };
        const uint32_t neary[2] = { ny ? hiy0 : loy0, ny ? hiy1 : loy1 }, fary[2] = { ny ? loy0 : hiy0, ny ? loy1 : hiy1 };
        const uint32_t nearz[2] = { nz ? hiz0 : loz0, nz ? hiz1 : loz1 }, farz[2] = { nz ? loz0 : hiz0, nz ? loz1 : hiz1 };
        const uint32_t metaw[2] = { meta0, meta1 };

        uint32_t childPending = 0, triMask = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t meta = (metaw[h] >> (8 * i)) & 0xffu;
                const float tx0 = __builtin_fmaf(MTS_UBYTE(nearx[h], i), asx, apx), tx1 = __builtin_fmaf(MTS_UBYTE(farx[h], i), asx, apx);
                const float ty0 = __builtin_fmaf(MTS_UBYTE(neary[h], i), asy, apy), ty1 = __builtin_fmaf(MTS_UBYTE(fary[h], i), asy, apy);
                const float tz0 = __builtin_fmaf(MTS_UBYTE(nearz[h], i), asz, apz), tz1 = __builtin_fmaf(MTS_UBYTE(farz[h], i), asz, apz);
                float tn = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, mint));
                float tf = fminf(fminf(tx1, ty1), fminf(tz1, maxt));
                // widen by a few ulps so that rounding in the slab arithmetic can never cull a true hit
                tn = tn - fabsf(tn) * kSlabPad;
                tf = tf + fabsf(tf) * kSlabPad;
                if (meta != 0 && tn <= tf) {
                    const int s = 4 * h + i;
                    if ((imask >> s) & 1u) childPending |= 1u << ((uint32_t) s ^ oct);
                    else triMask |= (meta >> 5) << (meta & 31u);
                }
            }
        }

        // triangles of the leaf children that were hit
        const uint32_t triBase = __float_as_uint(n1.y);
        while (triMask) {
            const uint32_t b = __ffs((int) triMask) - 1;
            triMask &= triMask - 1;
            const float4 *tp = sc.tris + 3 * (size_t) (triBase + b);
            const float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
            if (COUNT) triTests++;
            float t, u, v;
            if (triTest(r0, r1, r2, o, d, mint, maxt, t, u, v)) {
                if (ANY_HIT) { hit.t = t; return true; }
                const uint32_t tri = __float_as_uint(r2.z);
                if (!(found && t == hit.t && tri >= hit.tri)) {
                    hit.t = t; hit.u = u; hit.v = v; hit.tri = tri;
                    maxt = t;
                    found = true;
                }
            }
        }

        if (childPending) {
            // descend; siblings still pending in the current group wait on the stack
            if (group.y >> 8) st.push(group);
            group = make_uint2(__float_as_uint(n1.x), imask | (childPending << 8));
        }
    }
    return found;
}

// Closest-hit query with the reference's interval preparation.
template <bool COUNT>
MTS_DEV bool sceneIntersect(const DevScene &sc, V3 o, V3 d, float rayMint, float rayMaxt, HitRec &hit, uint2 *ldsStack,
                            uint32_t &nv, uint32_t &nt) {
    float mint, maxt;
    hit.t = kInf; hit.u = 0; hit.v = 0; hit.tri = 0xFFFFFFFFu;
    if (!rayInterval(sc, o, d, rayMint, rayMaxt, false, mint, maxt)) return false;
    return traverse<false, COUNT>(sc, o, d, mint, maxt, hit, ldsStack, nv, nt);
}

template <bool COUNT>
MTS_DEV bool sceneOccluded(const DevScene &sc, V3 o, V3 d, float rayMint, float rayMaxt, uint2 *ldsStack, uint32_t &nv, uint32_t &nt) {
    float mint, maxt;
    if (!rayInterval(sc, o, d, rayMint, rayMaxt, true, mint, maxt)) return false;
    HitRec hit;
    return traverse<true, COUNT>(sc, o, d, mint, maxt, hit, ldsStack, nv, nt);
}

}  // namespace mtsamd
