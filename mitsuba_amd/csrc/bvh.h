// Wide (8-ary) bounding volume hierarchy with quantised child boxes -- the acceleration structure the
// GPU traversal kernels walk. It replaces the SAH kd-tree of the reference
// (include/mitsuba/render/gkdtree.h:958-1263, built by ShapeKDTree::build, src/librender/skdtree.cpp:68-110);
// only query results are contractual, not the topology.
#pragma once
#include <cstdint>
#include <vector>

namespace mtsamd {

// 80-byte node. Child boxes are stored on an 8-bit grid local to the node:
//   lo = p + qlo * 2^e,  hi = p + qhi * 2^e   (rounded outwards, so decoded boxes contain the true ones)
// Children occupy 8 slots; slot s is visited first by rays whose direction-sign octant equals s, the other
// slots follow in the order of (slot XOR octant) -- an approximate front-to-back order that needs no sort.
//   meta[s] == 0            empty slot
//   meta[s] & 0x20, low 5 bits >= 24   internal child; node index = childBase + popcount(imask & ((1<<s)-1))
//   otherwise               leaf child: triangles [triBase + (meta & 31), +popcount(meta >> 5)) (1..3 triangles)
struct alignas(16) WideNode {
    float p[3];
    uint8_t e[3];
    uint8_t imask;
    uint32_t childBase;
    uint32_t triBase;
    uint8_t meta[8];
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
    std::vector<WideNode> nodes;
    std::vector<TriRecord> tris;
    float aabbMin[3], aabbMax[3];       // tight bounds
    uint32_t depth = 0;
    double sahCost = 0;
    double buildSeconds = 0;        // whole build
    int builder = 0;
    double binarySeconds = 0;       // of which: the binary tree (host SAH sweep or device clustering)
};

// Which builder makes the binary tree that is then collapsed to 8-wide nodes: a binned-SAH sweep on the host cores, or
// Morton sort + parallel locally-ordered clustering on the current HIP device.
enum { BVH_BUILDER_HOST = 0, BVH_BUILDER_DEVICE = 1 };

// positions: nVerts*3 floats, indices: nTris*3. Returns an empty string or an error message.
const char *buildBVH(const float *positions, uint32_t nVerts, const uint32_t *indices, uint32_t nTris, BVH &out, int builder = BVH_BUILDER_HOST);

// TriAccel::load restated (triaccel.h:61-94); returns k = 3 for degenerate triangles.
void loadTriRecord(const float *A, const float *B, const float *C, TriRecord &t);

static const int kMaxWideDepth = 23;    // the traversal stack holds one entry per level

}  // namespace mtsamd
