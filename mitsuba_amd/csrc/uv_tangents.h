// TriMesh::computeUVTangents for one triangle (src/librender/trimesh.cpp:683-735): the position partials dpdu, dpdv
// from the vertex positions and texture coordinates. One source for the host (scene preparation, batch queries) and
// the device (every hit on a mesh with texture coordinates): both are compiled without contraction, with IEEE
// division and square root, so they give the same floats.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define MTS_UVT_HD __host__ __device__
#else
#define MTS_UVT_HD
#endif

namespace mtsamd {

MTS_UVT_HD inline void uvTangents(const float v0[3], const float v1[3], const float v2[3], const float uv0[2], const float uv1[2], const float uv2[2],
                                  float dpdu[3], float dpdv[3]) {
    for (int k = 0; k < 3; ++k) dpdu[k] = dpdv[k] = 0.0f;
    const float dP1[3] = { v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2] }, dP2[3] = { v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2] };
    const float dUV1[2] = { uv1[0] - uv0[0], uv1[1] - uv0[1] }, dUV2[2] = { uv2[0] - uv0[0], uv2[1] - uv0[1] };
    const float n[3] = { dP1[1] * dP2[2] - dP1[2] * dP2[1], dP1[2] * dP2[0] - dP1[0] * dP2[2], dP1[0] * dP2[1] - dP1[1] * dP2[0] };
    const float length = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (length == 0) return;
    const float determinant = dUV1[0] * dUV2[1] - dUV1[1] * dUV2[0];
    if (determinant == 0) {
        // coordinateSystem(n / length, dpdu, dpdv) (src/libcore/util.cpp:592-601): dpdu = cross(c, a)
        const float r = 1.0f / length;
        const float a[3] = { n[0] * r, n[1] * r, n[2] * r };
        float c[3];
        if (fabsf(a[0]) > fabsf(a[1])) {
            const float invLen = 1.0f / sqrtf(a[0] * a[0] + a[2] * a[2]);
            c[0] = a[2] * invLen; c[1] = 0.0f; c[2] = -a[0] * invLen;
        } else {
            const float invLen = 1.0f / sqrtf(a[1] * a[1] + a[2] * a[2]);
            c[0] = 0.0f; c[1] = a[2] * invLen; c[2] = -a[1] * invLen;
        }
        dpdu[0] = c[1] * a[2] - c[2] * a[1]; dpdu[1] = c[2] * a[0] - c[0] * a[2]; dpdu[2] = c[0] * a[1] - c[1] * a[0];
        dpdv[0] = c[0]; dpdv[1] = c[1]; dpdv[2] = c[2];
    } else {
        const float invDet = 1.0f / determinant;
        for (int k = 0; k < 3; ++k) dpdu[k] = (dUV2[1] * dP1[k] - dUV1[1] * dP2[k]) * invDet;
        for (int k = 0; k < 3; ++k) dpdv[k] = (-dUV2[0] * dP1[k] + dUV1[0] * dP2[k]) * invDet;
    }
}

}  // namespace mtsamd
