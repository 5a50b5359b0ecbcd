// Wide (8-ary) bounding volume hierarchy with quantised child boxes -- the acceleration structure the
// GPU traversal kernels walk. It replaces the SAH kd-tree of the reference
// (include/mitsuba/render/gkdtree.h:958-1263, built by ShapeKDTree::build, src/librender/skdtree.cpp:68-110);
// only query results are contractual, not the topology.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace mtsamd {

// float4s from one triangle's shading record to the next: eight, i.e. every record in a 128-byte line of its own
// (positions + words, normals, texture coordinates), so that a gathered record costs one line from the fabric.
constexpr int kShadeTriStride = 8;


// 80-byte node. Child boxes are stored on an 8-bit grid local to the node:
//   lo = p + qlo * 2^e,  hi = p + qhi * 2^e   (rounded outwards, so decoded boxes contain the true ones)
// Children occupy 8 slots; slot s is visited first by rays whose direction-sign octant equals s, the other
// slots follow in the order of (slot XOR octant) -- an approximate front-to-back order that needs no sort.
//   imask bit s             internal child; node index = childBase + popcount(imask & ((1<<s)-1))
//   leafTris bits Ls..Ls+L-1  (L = kLeafTris) leaf child with 1 .. L triangles (001, 011, 111 ...). The records of a node's leaves
//                           lie one after the other in slot order: the record of bit b is triBase + popcount(leafTris & ((1<<b)-1))
//   neither                 empty slot: whatever its box test says selects nothing
// Most triangles of a leaf child: 3 (24 bits of leafTris) or 4 (all 32)
#ifndef MTS_LEAF_TRIS
#define MTS_LEAF_TRIS 4
#endif
constexpr int kLeafTris = MTS_LEAF_TRIS;
static_assert(kLeafTris == 3 || kLeafTris == 4, "a node's triangle mask has 8 fields of 3 or 4 bits");

struct alignas(16) WideNode {
    float p[3];
    uint8_t e[3];
    uint8_t imask;
    uint32_t childBase;
    uint32_t triBase;
    uint32_t leafTris;
    uint32_t reserved;
    uint8_t qlo[3][8];
    uint8_t qhi[3][8];
};
static_assert(sizeof(WideNode) == 80, "WideNode must be 80 bytes");

// 48-byte triangle record in BVH leaf order: the precomputed projection test of the reference
// (TriAccel, include/mitsuba/render/triaccel.h:37-94) plus the global triangle index.
struct alignas(16) TriRecord {
    uint32_t k;
    float n_u, n_v, n_d;
    float a_u, a_v, b_nu, b_nv;
    float c_nu, c_nv;
    uint32_t tri;       // index into the scene's global triangle array
    uint32_t pad;
};
static_assert(sizeof(TriRecord) == 48, "TriRecord must be 48 bytes");

struct BVH {
    std::vector<WideNode> nodes;        // empty when the tree was emitted straight into device memory
    std::vector<TriRecord> tris;
    uint32_t nodeCount = 0, triCount = 0;
    float aabbMin[3], aabbMax[3];       // tight bounds
    uint32_t depth = 0;
    double sahCost = 0;
    double buildSeconds = 0;        // whole build
    int builder = 0;
    double binarySeconds = 0;       // of which: the binary tree (host SAH sweep or device clustering)
};

// Which builder makes the binary tree that is then collapsed to 8-wide nodes: a binned-SAH sweep on the host cores, or
// Morton sort + parallel locally-ordered clustering on the current HIP device.
// BVH_BUILDER_DEVICE_BINARY builds only the binary tree on the device and collapses it on the host (for comparisons).
enum { BVH_BUILDER_HOST = 0, BVH_BUILDER_DEVICE = 1, BVH_BUILDER_DEVICE_BINARY = 2 };

// positions: nVerts*3 floats, indices: nTris*3. Returns an empty string or an error message.
const char *buildBVH(const float *positions, uint32_t nVerts, const uint32_t *indices, uint32_t nTris, BVH &out, int builder = BVH_BUILDER_HOST);

// TriAccel::load restated (triaccel.h:61-94); k = 3 marks degenerate triangles. One source for the host and the device
// builder: both are compiled without contraction and with IEEE division, so the records are bit-identical.
#if defined(__HIPCC__)
#define MTS_HD __host__ __device__
#else
#define MTS_HD
#endif
MTS_HD inline void loadTriRecordHD(const float *A, const float *B, const float *C, TriRecord &t) {
    const int waldModulo[4] = { 1, 2, 0, 1 };
    const float b[3] = { C[0] - A[0], C[1] - A[1], C[2] - A[2] };
    const float c[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] };
    const float N[3] = { c[1] * b[2] - c[2] * b[1], c[2] * b[0] - c[0] * b[2], c[0] * b[1] - c[1] * b[0] };
    uint32_t k = 0;
    for (int j = 0; j < 3; j++)
        if (fabsf(N[j]) > fabsf(N[k])) k = j;
    const uint32_t u = waldModulo[k], v = waldModulo[k + 1];
    const float n_k = N[k], denom = b[u] * c[v] - b[v] * c[u];
    t.k = 3; t.n_u = t.n_v = t.n_d = t.a_u = t.a_v = t.b_nu = t.b_nv = t.c_nu = t.c_nv = 0.0f; t.tri = 0; t.pad = 0;
    if (denom == 0) return;
    t.k = k;
    t.n_u = N[u] / n_k;
    t.n_v = N[v] / n_k;
    t.n_d = (A[0] * N[0] + A[1] * N[1] + A[2] * N[2]) / n_k;
    t.b_nu = b[u] / denom;
    t.b_nv = -b[v] / denom;
    t.a_u = A[u];
    t.a_v = A[v];
    t.c_nu = c[v] / denom;
    t.c_nv = -c[u] / denom;
}
void loadTriRecord(const float *A, const float *B, const float *C, TriRecord &t);

// Per-lane traversal stack of the GPU kernels: kLdsStack entries in LDS, kSpillStack more in scratch memory. A ray
// pushes at most one entry per level of wide nodes it descends through with siblings still pending, and neither the
// root (a group of one) nor the deepest level (no internal children) pushes: a tree of depth D (root = 1) needs D - 2
// entries. The builders refuse deeper trees, so the stack cannot overflow.
#ifndef MTS_LDS_STACK
#define MTS_LDS_STACK 4
#endif
static constexpr int kLdsStack = MTS_LDS_STACK;
static constexpr int kSpillStack = 26;
static constexpr int kMaxWideDepth = kLdsStack + kSpillStack + 2;
// A binary tree split at the median all the way down is at most 27 levels deep below its root (fewer than 2^27
// triangles), and collapsing can only remove levels: the builders' last resort always fits.
static_assert(kMaxWideDepth >= 29, "the traversal stack must hold the tree of a balanced binary build");

}  // namespace mtsamd
