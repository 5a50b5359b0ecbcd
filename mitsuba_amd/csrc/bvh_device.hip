// Binary BVH built on the GPU: triangle boxes and 63-bit Morton codes of their centroids, a radix sort, then parallel
// locally-ordered clustering (each cluster looks for the neighbour within a window of the sorted order whose union
// with it has the least surface area; mutual nearest neighbours merge; survivors are compacted; repeat until one
// cluster is left). The tree comes back to the host, is renumbered depth first (so that children follow their parents
// and the triangles of a subtree are contiguous) and goes through the same optimal collapse as the host-built tree.
// It replaces the SAH kd-tree construction of the reference (include/mitsuba/render/gkdtree.h:958-1263); only the
// query results of the final structure are contractual.
#include "bvh_internal.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace mtsamd {

namespace {

#define DEV_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { lastError = std::string(#expr " failed: ") + hipGetErrorString(e_); return lastError.c_str(); } } while (0)

thread_local std::string lastError;

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void) hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
    template <typename T> T *as() const { return (T *) p; }
};

int gRadius = 16;                // half width of the neighbour search window (MTSAMD_PLOC_RADIUS)
constexpr uint32_t kNone = 0xFFFFFFFFu;

__device__ inline unsigned long long spread21(unsigned long long x) {     // 21 bits -> every third bit
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void k_prim_boxes(const float *positions, const uint32_t *indices, uint32_t nVerts, uint32_t nTris, float4 *lo, float4 *hi, int *bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nTris) return;
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int j = 0; j < 3; ++j) {
        const uint32_t vi = indices[3 * (size_t) t + j];
        if (vi >= nVerts) { atomicOr(bad, 1); return; }
        for (int k = 0; k < 3; ++k) {
            const float v = positions[3 * (size_t) vi + k];
            if (!isfinite(v)) atomicOr(bad, 2);
            mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v);
        }
    }
    lo[t] = make_float4(mn[0], mn[1], mn[2], 0); hi[t] = make_float4(mx[0], mx[1], mx[2], 0);
}

// scene bounds: block reduction, then float atomics on the sign-corrected integer image of the floats
__device__ inline void atomicMinF(float *addr, float v) { if (v >= 0) atomicMin((int *) addr, __float_as_int(v)); else atomicMax((unsigned *) addr, __float_as_uint(v)); }
__device__ inline void atomicMaxF(float *addr, float v) { if (v >= 0) atomicMax((int *) addr, __float_as_int(v)); else atomicMin((unsigned *) addr, __float_as_uint(v)); }

__global__ void k_scene_box(const float4 *lo, const float4 *hi, uint32_t n, float *box /* 6 */) {
    __shared__ float s[6][256];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float v[6] = { INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY };
    if (t < n) { const float4 a = lo[t], b = hi[t]; v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = b.x; v[4] = b.y; v[5] = b.z; }
    for (int k = 0; k < 6; ++k) s[k][threadIdx.x] = v[k];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int) threadIdx.x < w)
            for (int k = 0; k < 6; ++k)
                s[k][threadIdx.x] = k < 3 ? fminf(s[k][threadIdx.x], s[k][threadIdx.x + w]) : fmaxf(s[k][threadIdx.x], s[k][threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; ++k) atomicMinF(&box[k], s[k][0]);
        for (int k = 3; k < 6; ++k) atomicMaxF(&box[k], s[k][0]);
    }
}

__global__ void k_morton(const float4 *lo, const float4 *hi, uint32_t n, const float *box, unsigned long long *keys, uint32_t *vals) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float4 a = lo[t], b = hi[t];
    const float c[3] = { 0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z) };
    unsigned long long q[3];
    for (int k = 0; k < 3; ++k) {
        const float ext = box[3 + k] - box[k];
        float u = ext > 0 ? (c[k] - box[k]) / ext : 0.0f;
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        q[k] = (unsigned long long) fminf(u * 2097152.0f, 2097151.0f);
    }
    keys[t] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
    vals[t] = t;
}

// device node: children (leaf: left = kNone, right = triangle), box and number of triangles below
struct DNode { uint32_t left, right; float lo[3], hi[3]; uint32_t count, pad; };

// Collapse table of a binary node, filled when the node is created (its children are final by then):
// c[i] = least SAH cost of representing the subtree by at most i roots of the wide tree, k[j] = share of the left child
// in the best split into j roots, take bit j = splitting into j beats j-1, leaf = the node becomes a leaf of the wide tree.
struct DDp { float c[8]; uint8_t k[9]; uint8_t take; uint8_t leaf; uint8_t pad; };

constexpr uint32_t kMaxLeafTris = kLeafTris;

MTS_HD inline float boxArea(const float *lo, const float *hi) {
    const float d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
    if (!(d0 >= 0 && d1 >= 0 && d2 >= 0)) return 0.0f;
    return 2.0f * (d0 * d1 + d0 * d2 + d1 * d2);
}

MTS_HD inline void dpLeaf(const DNode &nd, float costTri, DDp &d) {
    const float area = boxArea(nd.lo, nd.hi);
    for (int i = 1; i < 8; ++i) d.c[i] = area * costTri;
    d.c[0] = 0; d.take = 0; d.leaf = 1; d.pad = 0;
    for (int j = 0; j < 9; ++j) d.k[j] = 0;
}

MTS_HD inline void dpInner(const DNode &nd, const DDp &L, const DDp &R, float costNode, float costTri, DDp &d) {
    const float area = boxArea(nd.lo, nd.hi);
    float dist[9];
    d.k[0] = d.k[1] = 0; d.c[0] = 0; d.pad = 0;
    for (int j = 2; j <= 8; ++j) {
        float best = INFINITY; int bk = 1;
        for (int k = 1; k < j; ++k) {
            if (k > 7 || j - k > 7) continue;
            const float c = L.c[k] + R.c[j - k];
            if (c < best) { best = c; bk = k; }
        }
        dist[j] = best; d.k[j] = (uint8_t) bk;
    }
    const float leafCost = nd.count <= kMaxLeafTris ? area * costTri * (float) nd.count : INFINITY;
    const float nodeCost = area * costNode + dist[8];
    d.leaf = nd.count <= kMaxLeafTris && leafCost <= nodeCost;    // (the count guard matters when the areas overflow)
    d.c[1] = leafCost < nodeCost ? leafCost : nodeCost;
    d.take = 0;
    for (int j = 2; j <= 7; ++j) {
        if (dist[j] < d.c[j - 1]) { d.c[j] = dist[j]; d.take |= (uint8_t) (1u << j); }
        else d.c[j] = d.c[j - 1];
    }
}

__global__ void k_init_clusters(const uint32_t *sortedTri, const float4 *lo, const float4 *hi, uint32_t n, DNode *nodes, DDp *dp, float costTri, uint32_t *cnode, float4 *clo, float4 *chi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = sortedTri[i];
    const float4 a = lo[t], b = hi[t];
    DNode nd;
    nd.left = kNone; nd.right = t; nd.count = 1; nd.pad = 0;
    nd.lo[0] = a.x; nd.lo[1] = a.y; nd.lo[2] = a.z; nd.hi[0] = b.x; nd.hi[1] = b.y; nd.hi[2] = b.z;
    nodes[i] = nd;
    DDp d;
    dpLeaf(nd, costTri, d);
    dp[i] = d;
    cnode[i] = i; clo[i] = a; chi[i] = b;
}

__device__ inline float unionArea(float4 a0, float4 a1, float4 b0, float4 b1) {
    const float dx = fmaxf(a1.x, b1.x) - fminf(a0.x, b0.x), dy = fmaxf(a1.y, b1.y) - fminf(a0.y, b0.y), dz = fmaxf(a1.z, b1.z) - fminf(a0.z, b0.z);
    return dx * dy + dx * dz + dy * dz;
}

__global__ void k_nearest(const float4 *clo, const float4 *chi, uint32_t n, uint32_t *nn, uint32_t kRadius) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a0 = clo[i], a1 = chi[i];
    const uint32_t b = i > kRadius ? i - kRadius : 0u, e = min(n - 1, i + kRadius);
    float best = INFINITY; uint32_t bj = kNone;
    for (uint32_t j = b; j <= e; ++j) {
        if (j == i) continue;
        const float a = unionArea(a0, a1, clo[j], chi[j]);
        if (a < best) { best = a; bj = j; }     // ties go to the lower index, the same way from both sides of a pair
    }
    nn[i] = bj;
}

__global__ void k_merge(const uint32_t *nn, uint32_t n, uint32_t nLeaves, uint32_t *nodeCounter, DNode *nodes, DDp *dp, float costNode, float costTri,
                        uint32_t *cnode, float4 *clo, float4 *chi, uint32_t *keep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = nn[i];
    uint32_t k = 1;
    if (j != kNone && nn[j] == i) {
        if (i < j) {
            const uint32_t id = nLeaves + atomicAdd(nodeCounter, 1u);
            const float4 a0 = clo[i], a1 = chi[i], b0 = clo[j], b1 = chi[j];
            DNode nd;
            nd.left = cnode[i]; nd.right = cnode[j];
            nd.lo[0] = fminf(a0.x, b0.x); nd.lo[1] = fminf(a0.y, b0.y); nd.lo[2] = fminf(a0.z, b0.z);
            nd.hi[0] = fmaxf(a1.x, b1.x); nd.hi[1] = fmaxf(a1.y, b1.y); nd.hi[2] = fmaxf(a1.z, b1.z);
            nd.count = nodes[nd.left].count + nodes[nd.right].count; nd.pad = 0;
            nodes[id] = nd;
            DDp d;
            dpInner(nd, dp[nd.left], dp[nd.right], costNode, costTri, d);   // the children were finished by earlier launches
            dp[id] = d;
            // the merged cluster takes the place of its first member; written after every lane has read (separate arrays below)
            cnode[i] = id;
            clo[i] = make_float4(nd.lo[0], nd.lo[1], nd.lo[2], 0); chi[i] = make_float4(nd.hi[0], nd.hi[1], nd.hi[2], 0);
        } else {
            k = 0;
        }
    }
    keep[i] = k;
}

__global__ void k_compact(const uint32_t *keep, const uint32_t *pos, uint32_t n, const uint32_t *cnode, const float4 *clo, const float4 *chi,
                          uint32_t *cnode2, float4 *clo2, float4 *chi2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const uint32_t p = pos[i];
    cnode2[p] = cnode[i]; clo2[p] = clo[i]; chi2[p] = chi[i];
}

__global__ void k_gather_clusters(const uint32_t *cnode, uint32_t n, const DNode *nodes, const DDp *dp, DNode *outNodes, DDp *outDp) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    outNodes[i] = nodes[cnode[i]]; outDp[i] = dp[cnode[i]];
}

inline unsigned blocks(size_t n) { return (unsigned) ((n + 255) / 256); }


// Everything the clustering stage leaves on the device
struct ClusterTree {
    DevBuf pos, idx, nodes, dp;
    uint32_t root = 0, nTris = 0, nVerts = 0;
    float box[6] = { 0, 0, 0, 0, 0, 0 };
    int rounds = 0;
};

static bool gVerbose = false;
static std::chrono::steady_clock::time_point gLap;
static void lap(const char *what) {
    if (!gVerbose) return;
    (void) hipDeviceSynchronize();
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "mtsamd: device BVH %-28s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - gLap).count());
    gLap = now;
}

static const char *clusterOnDevice(const float *positions, uint32_t nVerts, const uint32_t *indices, uint32_t nTris, float costNode, float costTri, ClusterTree &ct) {
    const size_t N = nTris;
    gVerbose = getenv("MTSAMD_VERBOSE") != nullptr;
    gLap = std::chrono::steady_clock::now();
    if (const char *e = getenv("MTSAMD_PLOC_RADIUS")) gRadius = std::min(256, std::max(1, atoi(e)));
    ct.nTris = nTris; ct.nVerts = nVerts;
    DevBuf dLo, dHi, dBox, dBad, dKeys, dKeys2, dVals, dVals2, dTemp, dCnode[2], dClo[2], dChi[2], dNn, dKeep, dScan, dCounter;
    DEV_TRY(ct.pos.alloc((size_t) nVerts * 12)); DEV_TRY(ct.idx.alloc(N * 12));
    DEV_TRY(hipMemcpy(ct.pos.p, positions, (size_t) nVerts * 12, hipMemcpyHostToDevice));
    DEV_TRY(hipMemcpy(ct.idx.p, indices, N * 12, hipMemcpyHostToDevice));
    DEV_TRY(dLo.alloc(N * 16)); DEV_TRY(dHi.alloc(N * 16)); DEV_TRY(dBox.alloc(24)); DEV_TRY(dBad.alloc(4));
    const float inf = std::numeric_limits<float>::infinity();
    const float boxInit[6] = { inf, inf, inf, -inf, -inf, -inf };
    DEV_TRY(hipMemcpy(dBox.p, boxInit, 24, hipMemcpyHostToDevice));
    DEV_TRY(hipMemset(dBad.p, 0, 4));
    hipLaunchKernelGGL(k_prim_boxes, dim3(blocks(N)), dim3(256), 0, 0, ct.pos.as<float>(), ct.idx.as<uint32_t>(), nVerts, nTris, dLo.as<float4>(), dHi.as<float4>(), dBad.as<int>());
    int bad = 0;
    DEV_TRY(hipMemcpy(&bad, dBad.p, 4, hipMemcpyDeviceToHost));
    if (bad & 1) return "vertex index out of range";
    if (bad & 2) return "non-finite vertex position";
    hipLaunchKernelGGL(k_scene_box, dim3(blocks(N)), dim3(256), 0, 0, dLo.as<float4>(), dHi.as<float4>(), nTris, dBox.as<float>());
    DEV_TRY(hipMemcpy(ct.box, dBox.p, 24, hipMemcpyDeviceToHost));
    lap("upload + boxes");

    // Morton codes and sort
    DEV_TRY(dKeys.alloc(N * 8)); DEV_TRY(dKeys2.alloc(N * 8)); DEV_TRY(dVals.alloc(N * 4)); DEV_TRY(dVals2.alloc(N * 4));
    hipLaunchKernelGGL(k_morton, dim3(blocks(N)), dim3(256), 0, 0, dLo.as<float4>(), dHi.as<float4>(), nTris, dBox.as<float>(), dKeys.as<unsigned long long>(), dVals.as<uint32_t>());
    size_t tempBytes = 0;
    DEV_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, dKeys.as<unsigned long long>(), dKeys2.as<unsigned long long>(), dVals.as<uint32_t>(), dVals2.as<uint32_t>(), (int) N, 0, 63));
    size_t scanBytes = 0;
    DEV_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, (uint32_t *) nullptr, (uint32_t *) nullptr, (int) N + 1));
    DEV_TRY(dTemp.alloc(std::max(tempBytes, scanBytes)));
    DEV_TRY(hipcub::DeviceRadixSort::SortPairs(dTemp.p, tempBytes, dKeys.as<unsigned long long>(), dKeys2.as<unsigned long long>(), dVals.as<uint32_t>(), dVals2.as<uint32_t>(), (int) N, 0, 63));
    lap("morton + sort");

    // clustering
    DEV_TRY(ct.nodes.alloc((2 * N) * sizeof(DNode))); DEV_TRY(ct.dp.alloc((2 * N) * sizeof(DDp)));
    for (int k = 0; k < 2; ++k) { DEV_TRY(dCnode[k].alloc(N * 4)); DEV_TRY(dClo[k].alloc(N * 16)); DEV_TRY(dChi[k].alloc(N * 16)); }
    DEV_TRY(dNn.alloc(N * 4)); DEV_TRY(dKeep.alloc((N + 1) * 4)); DEV_TRY(dScan.alloc((N + 1) * 4)); DEV_TRY(dCounter.alloc(4));
    DEV_TRY(hipMemset(dCounter.p, 0, 4));
    hipLaunchKernelGGL(k_init_clusters, dim3(blocks(N)), dim3(256), 0, 0, dVals2.as<uint32_t>(), dLo.as<float4>(), dHi.as<float4>(), nTris, ct.nodes.as<DNode>(),
                       ct.dp.as<DDp>(), costTri, dCnode[0].as<uint32_t>(), dClo[0].as<float4>(), dChi[0].as<float4>());
    uint32_t n = nTris;
    int cur = 0;
    ct.rounds = 0;
    // Clustering stops when about N/32 clusters are left (never for small inputs): the top of the tree, where most
    // node visits happen, is then built by a SAH sweep over the clusters on the host cores (a few milliseconds).
    uint32_t stopAt = 1;
    if (const char *e = getenv("MTSAMD_TOP_CLUSTERS")) stopAt = (uint32_t) std::max(1, atoi(e));
    else if (nTris >= 65536) stopAt = nTris / 32;
    uint32_t merged = 0;
    while (n > stopAt) {
        if (++ct.rounds > 4096) return "device BVH build did not converge";
        hipLaunchKernelGGL(k_nearest, dim3(blocks(n)), dim3(256), 0, 0, dClo[cur].as<float4>(), dChi[cur].as<float4>(), n, dNn.as<uint32_t>(), (uint32_t) gRadius);
        hipLaunchKernelGGL(k_merge, dim3(blocks(n)), dim3(256), 0, 0, dNn.as<uint32_t>(), n, nTris, dCounter.as<uint32_t>(), ct.nodes.as<DNode>(), ct.dp.as<DDp>(),
                           costNode, costTri, dCnode[cur].as<uint32_t>(), dClo[cur].as<float4>(), dChi[cur].as<float4>(), dKeep.as<uint32_t>());
        DEV_TRY(hipMemsetAsync(dKeep.as<uint32_t>() + n, 0, 4, 0));
        size_t sb = scanBytes;
        DEV_TRY(hipcub::DeviceScan::ExclusiveSum(dTemp.p, sb, dKeep.as<uint32_t>(), dScan.as<uint32_t>(), (int) n + 1));
        hipLaunchKernelGGL(k_compact, dim3(blocks(n)), dim3(256), 0, 0, dKeep.as<uint32_t>(), dScan.as<uint32_t>(), n, dCnode[cur].as<uint32_t>(), dClo[cur].as<float4>(),
                           dChi[cur].as<float4>(), dCnode[cur ^ 1].as<uint32_t>(), dClo[cur ^ 1].as<float4>(), dChi[cur ^ 1].as<float4>());
        uint32_t n2 = 0;
        DEV_TRY(hipMemcpy(&n2, dScan.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost));
        if (n2 >= n) return "device BVH build made no progress";
        merged += n - n2;
        n = n2; cur ^= 1;
    }
    DEV_TRY(hipGetLastError());
    lap("clustering");
    if (n == 1) {
        DEV_TRY(hipMemcpy(&ct.root, dCnode[cur].p, 4, hipMemcpyDeviceToHost));
    } else {
        // top of the tree: SAH sweep over the remaining clusters on the host, collapse tables included
        DevBuf dGn, dGd;
        DEV_TRY(dGn.alloc((size_t) n * sizeof(DNode))); DEV_TRY(dGd.alloc((size_t) n * sizeof(DDp)));
        hipLaunchKernelGGL(k_gather_clusters, dim3(blocks(n)), dim3(256), 0, 0, dCnode[cur].as<uint32_t>(), n, ct.nodes.as<DNode>(), ct.dp.as<DDp>(), dGn.as<DNode>(), dGd.as<DDp>());
        std::vector<DNode> cn(n); std::vector<DDp> cd(n); std::vector<uint32_t> cid(n);
        DEV_TRY(hipMemcpy(cn.data(), dGn.p, (size_t) n * sizeof(DNode), hipMemcpyDeviceToHost));
        DEV_TRY(hipMemcpy(cd.data(), dGd.p, (size_t) n * sizeof(DDp), hipMemcpyDeviceToHost));
        DEV_TRY(hipMemcpy(cid.data(), dCnode[cur].p, (size_t) n * 4, hipMemcpyDeviceToHost));
        std::vector<Box> boxes(n);
        for (uint32_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { boxes[i].lo[k] = cn[i].lo[k]; boxes[i].hi[k] = cn[i].hi[k]; }
        BinaryTree top;
        const char *terr = buildBinaryOverBoxes(boxes, top);
        if (terr && *terr) return terr;
        // device ids: top leaves are the clusters, top internal nodes continue after the nodes made by clustering
        const uint32_t base = nTris + merged;
        std::vector<uint32_t> devId(top.nNodes);
        uint32_t nextInternal = 0;
        for (uint32_t t = 0; t < top.nNodes; ++t)
            devId[t] = top.nodes[t].isLeaf() ? cid[top.primTri[top.nodes[t].first]] : base + nextInternal++;
        if (base + nextInternal != 2 * nTris - 1) return "device BVH build: node count mismatch at the top of the tree";
        std::vector<DNode> tn(nextInternal); std::vector<DDp> td(nextInternal);
        std::vector<DDp> dpOf(top.nNodes); std::vector<uint32_t> cntOf(top.nNodes);
        for (uint32_t t = top.nNodes; t-- > 0;) {       // children follow their parents: descending order is bottom-up
            const Node2 &h = top.nodes[t];
            if (h.isLeaf()) { const uint32_t c = top.primTri[h.first]; dpOf[t] = cd[c]; cntOf[t] = cn[c].count; continue; }
            DNode d;
            d.left = devId[h.left]; d.right = devId[h.right]; d.pad = 0;
            for (int k = 0; k < 3; ++k) { d.lo[k] = h.box.lo[k]; d.hi[k] = h.box.hi[k]; }
            d.count = cntOf[t] = cntOf[h.left] + cntOf[h.right];
            dpInner(d, dpOf[h.left], dpOf[h.right], costNode, costTri, dpOf[t]);
            tn[devId[t] - base] = d; td[devId[t] - base] = dpOf[t];
        }
        DEV_TRY(hipMemcpy(ct.nodes.as<DNode>() + base, tn.data(), tn.size() * sizeof(DNode), hipMemcpyHostToDevice));
        DEV_TRY(hipMemcpy(ct.dp.as<DDp>() + base, td.data(), td.size() * sizeof(DDp), hipMemcpyHostToDevice));
        ct.root = devId[0];
        lap("top of the tree (host sweep)");
    }
    if (gVerbose) fprintf(stderr, "mtsamd: device BVH %d clustering rounds\n", ct.rounds);
    return "";
}

// ---- emission of the wide BVH on the device, one level per launch ---------------------------------------------------------
struct EmitArgs {
    const DNode *nodes; const DDp *dp;
    const float *positions; const uint32_t *indices;
    uint32_t *wideBin;          // wide node index -> binary node it represents
    float4 *outNodes; float4 *outTris;
    uint32_t nodeStride, triStride, nodeCapacity;
    uint32_t *counters;         // [0] wide nodes allocated, [1] triangle records allocated, [2] overflow flag
    double *sah; float rootArea, costNode, costTri;
    uint32_t root, levelStart, levelEnd;
};

// roots of the cheapest representation of subtree n by at most j roots (iterative form of the host's Collector)
__device__ inline void collectRoots(const DNode *nodes, const DDp *dp, uint32_t n, int j, uint32_t *kids, int &nk) {
    uint32_t stackN[16]; int stackJ[16]; int sp = 0;
    stackN[sp] = n; stackJ[sp++] = j;
    while (sp > 0) {
        n = stackN[--sp]; j = stackJ[sp];
        while (true) {
            const DNode &nd = nodes[n];
            if (nd.left == kNone || j == 1) { kids[nk++] = n; break; }
            if (j <= 7 && !((dp[n].take >> j) & 1)) { --j; continue; }
            const int k = dp[n].k[j];
            // the left part is expanded first so that the order of the roots matches the host builder's
            stackN[sp] = nd.right; stackJ[sp++] = j - k;
            n = nd.left; j = k;
        }
    }
}

__global__ void __launch_bounds__(64) k_emit_level(EmitArgs a) {
    const uint32_t w = a.levelStart + blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.levelEnd) return;
    const uint32_t bn = a.wideBin[w];
    const DNode &n2 = a.nodes[bn];
    uint32_t kids[8];
    int nk = 0;
    if (n2.left == kNone || (bn == a.root && a.dp[bn].leaf)) {
        kids[nk++] = bn;        // the whole scene fits one leaf
    } else {
        const int k = a.dp[bn].k[8];
        collectRoots(a.nodes, a.dp, n2.left, k, kids, nk);
        collectRoots(a.nodes, a.dp, n2.right, 8 - k, kids, nk);
    }
    // padded child boxes and node bounds
    float clo[8][3], chi[8][3], nlo[3] = { INFINITY, INFINITY, INFINITY }, nhi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int i = 0; i < nk; ++i) {
        const DNode &c = a.nodes[kids[i]];
        for (int ax = 0; ax < 3; ++ax) {
            const float m = fmaxf(fabsf(c.lo[ax]), fabsf(c.hi[ax]));
            const float pad = m * 0x1p-20f + 1e-30f;
            clo[i][ax] = c.lo[ax] - pad; chi[i][ax] = c.hi[ax] + pad;
            nlo[ax] = fminf(nlo[ax], clo[i][ax]); nhi[ax] = fmaxf(nhi[ax], chi[i][ax]);
        }
    }
    double sah = (double) (boxArea(nlo, nhi) / a.rootArea) * a.costNode;

    // slot s is the first one visited by rays of sign octant s: it should hold the child lying furthest "behind" along
    // the representative direction of that octant (greedy assignment, cheapest pair first)
    int slotOf[8], childAt[8];
    for (int i = 0; i < 8; ++i) { slotOf[i] = -1; childAt[i] = -1; }
    const float nc[3] = { 0.5f * (nlo[0] + nhi[0]), 0.5f * (nlo[1] + nhi[1]), 0.5f * (nlo[2] + nhi[2]) };
    for (int it = 0; it < nk; ++it) {
        int bi = -1, bs = -1; float bc = INFINITY;
        for (int i = 0; i < nk; ++i) {
            if (slotOf[i] >= 0) continue;
            for (int s = 0; s < 8; ++s) {
                if (childAt[s] >= 0) continue;
                float c = 0;
                for (int ax = 0; ax < 3; ++ax) {
                    const float d = 0.5f * (clo[i][ax] + chi[i][ax]) - nc[ax];
                    c += ((s >> ax) & 1) ? -d : d;
                }
                if (c < bc) { bc = c; bi = i; bs = s; }
            }
        }
        slotOf[bi] = bs; childAt[bs] = bi;
    }

    WideNode node;
    memset(&node, 0, sizeof(node));
    for (int ax = 0; ax < 3; ++ax) {
        node.p[ax] = nlo[ax];
        const float ext = nhi[ax] - nlo[ax];
        int e = 1;
        if (ext > 0) {
            int ex;
            frexpf(ext / 255.0f, &ex);      // ext/255 = m * 2^ex with m in [0.5,1)  => 2^ex >= ext/255
            e = min(254, max(1, ex + 127));
        }
        node.e[ax] = (uint8_t) e;
    }
    uint32_t nInternal = 0, nLeafTris = 0;
    for (int i = 0; i < nk; ++i) { if (a.dp[kids[i]].leaf) nLeafTris += a.nodes[kids[i]].count; else nInternal++; }
    const uint32_t childBase = nInternal ? atomicAdd(&a.counters[0], nInternal) : 0u;
    const uint32_t triBase = nLeafTris ? atomicAdd(&a.counters[1], nLeafTris) : 0u;
    if (childBase + nInternal > a.nodeCapacity) { atomicOr(&a.counters[2], 1u); return; }
    node.childBase = childBase; node.triBase = triBase;
    uint32_t rank = 0, triOfs = 0, leafTris = 0;
    for (int s = 0; s < 8; ++s) {
        const int i = childAt[s];
        if (i < 0) continue;
        const DNode &c = a.nodes[kids[i]];
        for (int ax = 0; ax < 3; ++ax) {
            const double scale = (double) __uint_as_float((uint32_t) node.e[ax] << 23);
            const double p = node.p[ax];
            int lo = (int) floor(((double) clo[i][ax] - p) / scale);
            int hi = (int) ceil(((double) chi[i][ax] - p) / scale);
            lo = min(255, max(0, lo));
            hi = min(255, max(0, hi));
            // the decoded planes must enclose the padded child box
            while (lo > 0 && p + lo * scale > (double) clo[i][ax]) --lo;
            while (hi < 255 && p + hi * scale < (double) chi[i][ax]) ++hi;
            node.qlo[ax][s] = (uint8_t) lo;
            node.qhi[ax][s] = (uint8_t) hi;
        }
        if (a.dp[kids[i]].leaf) {
            const uint32_t unary = (1u << c.count) - 1u;     // 1 -> 001, 2 -> 011, 3 -> 111
            leafTris |= unary << (kLeafTris * s);
            // the (at most kLeafTris) triangles of the leaf: walk its little subtree, left to right
            uint32_t st[8]; int sp = 0; uint32_t j = 0;
            st[sp++] = kids[i];
            while (sp > 0) {
                const DNode &ln = a.nodes[st[--sp]];
                if (ln.left != kNone) { st[sp++] = ln.right; st[sp++] = ln.left; continue; }
                const uint32_t tri = ln.right;
                TriRecord tr;
                loadTriRecordHD(a.positions + 3 * (size_t) a.indices[3 * (size_t) tri], a.positions + 3 * (size_t) a.indices[3 * (size_t) tri + 1],
                                a.positions + 3 * (size_t) a.indices[3 * (size_t) tri + 2], tr);
                tr.tri = tri;
                float4 *dst = a.outTris + (size_t) a.triStride * (triBase + triOfs + j);
                const float4 *src = reinterpret_cast<const float4 *>(&tr);
                dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
                ++j;
            }
            float plo[3] = { clo[i][0], clo[i][1], clo[i][2] }, phi[3] = { chi[i][0], chi[i][1], chi[i][2] };
            sah += (double) (boxArea(plo, phi) / a.rootArea) * a.costTri * c.count;
            triOfs += c.count;
        } else {
            node.imask |= (uint8_t) (1u << s);
            a.wideBin[childBase + rank] = kids[i];      // internal children are laid out contiguously in slot order
            ++rank;
        }
    }
    node.leafTris = leafTris;
    float4 *dst = a.outNodes + (size_t) a.nodeStride * w;
    const float4 *src = reinterpret_cast<const float4 *>(&node);
    for (int k = 0; k < 5; ++k) dst[k] = src[k];
    atomicAdd(a.sah, sah);
}

}  // namespace

static float gCostNode = 1.0f, gCostTri = 0.5f;
void setDeviceBuilderCosts(float costNode, float costTri) { gCostNode = costNode; gCostTri = costTri; }

const char *buildBinaryTreeDevice(const float *positions, uint32_t nVerts, const uint32_t *indices, uint32_t nTris, BinaryTree &out) {
    const size_t N = nTris;
    ClusterTree ct;
    const char *err = clusterOnDevice(positions, nVerts, indices, nTris, gCostNode, gCostTri, ct);
    if (err && *err) return err;
    for (int k = 0; k < 3; ++k) { out.sceneBox.lo[k] = ct.box[k]; out.sceneBox.hi[k] = ct.box[3 + k]; }
    const uint32_t root = ct.root;
    std::vector<DNode> dn(2 * N - 1);
    DEV_TRY(hipMemcpy(dn.data(), ct.nodes.p, dn.size() * sizeof(DNode), hipMemcpyDeviceToHost));
    lap("download");
    // depth-first renumbering: parents before children, subtree triangles contiguous
    out.nodes.assign(2 * N, Node2());
    out.primTri.resize(N);
    out.nNodes = (uint32_t) (2 * N - 1);
    struct Item { uint32_t dev, host; };
    std::vector<Item> stack;
    stack.push_back({ root, 0 });
    uint32_t nextNode = 1, nextPrim = 0;
    // first pass assigns indices in preorder and leaf slots in visiting order; counts are filled on the way back
    std::vector<uint32_t> order;     // host indices in preorder
    order.reserve(2 * N);
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        const DNode &d = dn[it.dev];
        Node2 &h = out.nodes[it.host];
        for (int k = 0; k < 3; ++k) { h.box.lo[k] = d.lo[k]; h.box.hi[k] = d.hi[k]; }
        order.push_back(it.host);
        if (d.left == kNone) {
            h.left = h.right = 0; h.first = nextPrim; h.count = 1;
            out.primTri[nextPrim++] = d.right;
        } else {
            h.left = nextNode; h.right = nextNode + 1; nextNode += 2;
            stack.push_back({ d.right, h.right });      // left is popped first: its leaves come first
            stack.push_back({ d.left, h.left });
        }
    }
    if (nextPrim != N) return "device BVH build lost triangles";
    for (size_t k = order.size(); k-- > 0;) {
        Node2 &h = out.nodes[order[k]];
        if (!h.isLeaf()) { h.first = out.nodes[h.left].first; h.count = out.nodes[h.left].count + out.nodes[h.right].count; }
    }
    lap("host renumbering");
    return "";
}

// The whole build on the device: clustering, collapse tables, then the wide nodes and triangle records are written
// level by level straight into device memory (the caller owns the two buffers afterwards).
const char *buildWideDevice(const float *positions, uint32_t nVerts, const uint32_t *indices, uint32_t nTris, uint32_t nodeStride, uint32_t triStride,
                            void **outNodes, void **outTris, BVH &info) {
    auto t0 = std::chrono::steady_clock::now();
    const size_t N = nTris;
    if (const char *e = getenv("MTSAMD_COST_TRI")) gCostTri = (float) atof(e);
    ClusterTree ct;
    const char *err = clusterOnDevice(positions, nVerts, indices, nTris, gCostNode, gCostTri, ct);
    if (err && *err) return err;
    info.binarySeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int k = 0; k < 3; ++k) { info.aabbMin[k] = ct.box[k]; info.aabbMax[k] = ct.box[3 + k]; }
    const uint32_t capacity = (uint32_t) (N / 2 + 64);      // a wide node has at least two children; typical use is N/8
    DevBuf dWideBin, dNodes, dTris, dCounters, dSah;
    DEV_TRY(dWideBin.alloc((size_t) capacity * 4));
    DEV_TRY(dNodes.alloc((size_t) capacity * nodeStride * 16)); DEV_TRY(dTris.alloc(N * triStride * 16));
    DEV_TRY(hipMemset(dNodes.p, 0, (size_t) capacity * nodeStride * 16)); DEV_TRY(hipMemset(dTris.p, 0, N * triStride * 16));
    DEV_TRY(dCounters.alloc(16)); DEV_TRY(dSah.alloc(8));
    const uint32_t init[4] = { 1, 0, 0, 0 };
    DEV_TRY(hipMemcpy(dCounters.p, init, 16, hipMemcpyHostToDevice));
    DEV_TRY(hipMemset(dSah.p, 0, 8));
    DEV_TRY(hipMemcpy(dWideBin.p, &ct.root, 4, hipMemcpyHostToDevice));
    EmitArgs a;
    a.nodes = ct.nodes.as<DNode>(); a.dp = ct.dp.as<DDp>(); a.positions = ct.pos.as<float>(); a.indices = ct.idx.as<uint32_t>();
    a.wideBin = dWideBin.as<uint32_t>(); a.outNodes = dNodes.as<float4>(); a.outTris = dTris.as<float4>();
    a.nodeStride = nodeStride; a.triStride = triStride; a.nodeCapacity = capacity;
    a.counters = dCounters.as<uint32_t>(); a.sah = dSah.as<double>();
    float lo[3] = { ct.box[0], ct.box[1], ct.box[2] }, hi[3] = { ct.box[3], ct.box[4], ct.box[5] };
    Box sb; for (int k = 0; k < 3; ++k) { sb.lo[k] = lo[k]; sb.hi[k] = hi[k]; }
    a.rootArea = std::max(sb.area(), 1e-30f); a.costNode = gCostNode; a.costTri = gCostTri;
    a.root = ct.root;
    uint32_t start = 0, end = 1, depth = 0;
    while (start < end) {
        if (++depth > (uint32_t) kMaxWideDepth) return kTooDeep;
        a.levelStart = start; a.levelEnd = end;
        hipLaunchKernelGGL(k_emit_level, dim3((end - start + 63) / 64), dim3(64), 0, 0, a);
        uint32_t c[4];
        DEV_TRY(hipMemcpy(c, dCounters.p, 16, hipMemcpyDeviceToHost));
        if (c[2]) return "device BVH build ran out of node space";
        start = end; end = c[0];
    }
    DEV_TRY(hipGetLastError());
    uint32_t c[4];
    DEV_TRY(hipMemcpy(c, dCounters.p, 16, hipMemcpyDeviceToHost));
    if (c[1] != nTris) return "device BVH build lost triangles";
    DEV_TRY(hipMemcpy(&info.sahCost, dSah.p, 8, hipMemcpyDeviceToHost));
    // hand over right-sized buffers
    void *nodesOut = nullptr;
    DEV_TRY(hipMalloc(&nodesOut, (size_t) c[0] * nodeStride * 16));
    DEV_TRY(hipMemcpy(nodesOut, dNodes.p, (size_t) c[0] * nodeStride * 16, hipMemcpyDeviceToDevice));
    *outNodes = nodesOut;
    *outTris = dTris.p; dTris.p = nullptr;
    info.nodes.clear(); info.tris.clear();
    info.nodeCount = c[0]; info.triCount = c[1]; info.depth = depth; info.builder = BVH_BUILDER_DEVICE;
    lap("collapse + emission");
    info.buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return "";
}

}  // namespace mtsamd
