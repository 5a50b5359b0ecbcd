// Shared between the binary-tree builders (host SAH, device PLOC) and the collapse to the wide BVH.
#pragma once
#include "bvh.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace mtsamd {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int i = 0; i < 3; ++i) { lo[i] = std::numeric_limits<float>::infinity(); hi[i] = -lo[i]; }
    }
    void grow(const float *p) {
        for (int i = 0; i < 3; ++i) { lo[i] = std::min(lo[i], p[i]); hi[i] = std::max(hi[i], p[i]); }
    }
    void grow(const Box &b) {
        for (int i = 0; i < 3; ++i) { lo[i] = std::min(lo[i], b.lo[i]); hi[i] = std::max(hi[i], b.hi[i]); }
    }
    float area() const {
        float d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
        if (!(d0 >= 0 && d1 >= 0 && d2 >= 0)) return 0.0f;
        return 2.0f * (d0 * d1 + d0 * d2 + d1 * d2);
    }
};

struct Node2 {
    Box box;
    uint32_t left = 0, right = 0;   // children; left == 0 marks a single-triangle leaf of the binary tree
    uint32_t first = 0, count = 0;  // primitive range of the whole subtree
    bool isLeaf() const { return left == 0; }
};

// A binary BVH in the form the collapse expects: node 0 is the root, children carry larger indices than their
// parents, and the triangles of every subtree are contiguous in primTri (first .. first + count).
struct BinaryTree {
    std::vector<Node2> nodes;
    uint32_t nNodes = 0;
    std::vector<uint32_t> primTri;      // leaf order -> global triangle index
    Box sceneBox;
};

// What a builder returns for a tree deeper than kMaxWideDepth (compared by content)
static const char *const kTooDeep = "BVH too deep for the traversal stack";
// Host builder: multithreaded binned SAH down to single triangles. Nodes at depth >= sahDepthLimit are split at the
// median instead (0: a balanced tree throughout).
const char *buildBinaryTreeHost(const float *positions, uint32_t nVerts, const uint32_t *indices, uint32_t nTris, BinaryTree &out,
                                int sahDepthLimit = std::numeric_limits<int>::max());
// The host sweep over arbitrary boxes; primTri holds the box order.
const char *buildBinaryOverBoxes(const std::vector<Box> &boxes, BinaryTree &out);
// Device builder: Morton sort + parallel locally-ordered clustering on the current HIP device.
const char *buildBinaryTreeDevice(const float *positions, uint32_t nVerts, const uint32_t *indices, uint32_t nTris, BinaryTree &out);
// The whole build on the current HIP device: the wide nodes and triangle records are written into two device buffers
// (hipMalloc'ed, handed to the caller); `info` receives counts, depth, bounds and timings, no host copies.
const char *buildWideDevice(const float *positions, uint32_t nVerts, const uint32_t *indices, uint32_t nTris, uint32_t nodeStride, uint32_t triStride,
                            void **outNodes, void **outTris, BVH &info);
void setDeviceBuilderCosts(float costNode, float costTri);
// Optimal collapse to 8-wide nodes, slot assignment, quantisation, triangle records.
const char *collapseToWide(const BinaryTree &tree, const float *positions, const uint32_t *indices, BVH &out);

}  // namespace mtsamd
