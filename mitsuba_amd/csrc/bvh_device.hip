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

// device node: children (leaf: left = kNone, right = triangle) and box
struct DNode { uint32_t left, right; float lo[3], hi[3]; };

__global__ void k_init_clusters(const uint32_t *sortedTri, const float4 *lo, const float4 *hi, uint32_t n, DNode *nodes, uint32_t *cnode, float4 *clo, float4 *chi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = sortedTri[i];
    const float4 a = lo[t], b = hi[t];
    DNode nd;
    nd.left = kNone; nd.right = t;
    nd.lo[0] = a.x; nd.lo[1] = a.y; nd.lo[2] = a.z; nd.hi[0] = b.x; nd.hi[1] = b.y; nd.hi[2] = b.z;
    nodes[i] = nd;
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

__global__ void k_merge(const uint32_t *nn, uint32_t n, uint32_t nLeaves, uint32_t *nodeCounter, DNode *nodes, uint32_t *cnode, float4 *clo, float4 *chi, uint32_t *keep) {
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
            nodes[id] = nd;
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

inline unsigned blocks(size_t n) { return (unsigned) ((n + 255) / 256); }

}  // namespace

const char *buildBinaryTreeDevice(const float *positions, uint32_t nVerts, const uint32_t *indices, uint32_t nTris, BinaryTree &out) {
    const size_t N = nTris;
    const bool verbose = getenv("MTSAMD_VERBOSE") != nullptr;
    auto tStart = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        (void) hipDeviceSynchronize();
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "mtsamd: device BVH %-28s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tStart).count());
        tStart = now;
    };
    if (const char *e = getenv("MTSAMD_PLOC_RADIUS")) gRadius = std::min(256, std::max(1, atoi(e)));
    DevBuf dPos, dIdx, dLo, dHi, dBox, dBad, dKeys, dKeys2, dVals, dVals2, dTemp, dNodes, dCnode[2], dClo[2], dChi[2], dNn, dKeep, dScan, dCounter;
    DEV_TRY(dPos.alloc((size_t) nVerts * 12)); DEV_TRY(dIdx.alloc(N * 12));
    DEV_TRY(hipMemcpy(dPos.p, positions, (size_t) nVerts * 12, hipMemcpyHostToDevice));
    DEV_TRY(hipMemcpy(dIdx.p, indices, N * 12, hipMemcpyHostToDevice));
    DEV_TRY(dLo.alloc(N * 16)); DEV_TRY(dHi.alloc(N * 16)); DEV_TRY(dBox.alloc(24)); DEV_TRY(dBad.alloc(4));
    const float inf = std::numeric_limits<float>::infinity();
    const float boxInit[6] = { inf, inf, inf, -inf, -inf, -inf };
    DEV_TRY(hipMemcpy(dBox.p, boxInit, 24, hipMemcpyHostToDevice));
    DEV_TRY(hipMemset(dBad.p, 0, 4));
    hipLaunchKernelGGL(k_prim_boxes, dim3(blocks(N)), dim3(256), 0, 0, dPos.as<float>(), dIdx.as<uint32_t>(), nVerts, nTris, dLo.as<float4>(), dHi.as<float4>(), dBad.as<int>());
    int bad = 0;
    DEV_TRY(hipMemcpy(&bad, dBad.p, 4, hipMemcpyDeviceToHost));
    if (bad & 1) return "vertex index out of range";
    if (bad & 2) return "non-finite vertex position";
    hipLaunchKernelGGL(k_scene_box, dim3(blocks(N)), dim3(256), 0, 0, dLo.as<float4>(), dHi.as<float4>(), nTris, dBox.as<float>());
    float box[6];
    DEV_TRY(hipMemcpy(box, dBox.p, 24, hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) { out.sceneBox.lo[k] = box[k]; out.sceneBox.hi[k] = box[3 + k]; }

    lap("upload + boxes");
    // Morton codes and sort
    DEV_TRY(dKeys.alloc(N * 8)); DEV_TRY(dKeys2.alloc(N * 8)); DEV_TRY(dVals.alloc(N * 4)); DEV_TRY(dVals2.alloc(N * 4));
    hipLaunchKernelGGL(k_morton, dim3(blocks(N)), dim3(256), 0, 0, dLo.as<float4>(), dHi.as<float4>(), nTris, dBox.as<float>(), dKeys.as<unsigned long long>(), dVals.as<uint32_t>());
    size_t tempBytes = 0;
    DEV_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, dKeys.as<unsigned long long>(), dKeys2.as<unsigned long long>(), dVals.as<uint32_t>(), dVals2.as<uint32_t>(), (int) N, 0, 63));
    size_t scanBytes = 0;
    DEV_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, (uint32_t *) nullptr, (uint32_t *) nullptr, (int) N));
    DEV_TRY(dTemp.alloc(std::max(tempBytes, scanBytes)));
    DEV_TRY(hipcub::DeviceRadixSort::SortPairs(dTemp.p, tempBytes, dKeys.as<unsigned long long>(), dKeys2.as<unsigned long long>(), dVals.as<uint32_t>(), dVals2.as<uint32_t>(), (int) N, 0, 63));

    lap("morton + sort");
    // clustering
    DEV_TRY(dNodes.alloc((2 * N) * sizeof(DNode)));
    for (int k = 0; k < 2; ++k) { DEV_TRY(dCnode[k].alloc(N * 4)); DEV_TRY(dClo[k].alloc(N * 16)); DEV_TRY(dChi[k].alloc(N * 16)); }
    DEV_TRY(dNn.alloc(N * 4)); DEV_TRY(dKeep.alloc((N + 1) * 4)); DEV_TRY(dScan.alloc((N + 1) * 4)); DEV_TRY(dCounter.alloc(4));
    DEV_TRY(hipMemset(dCounter.p, 0, 4));
    hipLaunchKernelGGL(k_init_clusters, dim3(blocks(N)), dim3(256), 0, 0, dVals2.as<uint32_t>(), dLo.as<float4>(), dHi.as<float4>(), nTris, dNodes.as<DNode>(),
                       dCnode[0].as<uint32_t>(), dClo[0].as<float4>(), dChi[0].as<float4>());
    uint32_t n = nTris;
    int cur = 0, iterations = 0;
    while (n > 1) {
        if (++iterations > 4096) return "device BVH build did not converge";
        hipLaunchKernelGGL(k_nearest, dim3(blocks(n)), dim3(256), 0, 0, dClo[cur].as<float4>(), dChi[cur].as<float4>(), n, dNn.as<uint32_t>(), (uint32_t) gRadius);
        hipLaunchKernelGGL(k_merge, dim3(blocks(n)), dim3(256), 0, 0, dNn.as<uint32_t>(), n, nTris, dCounter.as<uint32_t>(), dNodes.as<DNode>(),
                           dCnode[cur].as<uint32_t>(), dClo[cur].as<float4>(), dChi[cur].as<float4>(), dKeep.as<uint32_t>());
        DEV_TRY(hipMemsetAsync(dKeep.as<uint32_t>() + n, 0, 4, 0));
        size_t sb = scanBytes;
        DEV_TRY(hipcub::DeviceScan::ExclusiveSum(dTemp.p, sb, dKeep.as<uint32_t>(), dScan.as<uint32_t>(), (int) n + 1));
        hipLaunchKernelGGL(k_compact, dim3(blocks(n)), dim3(256), 0, 0, dKeep.as<uint32_t>(), dScan.as<uint32_t>(), n, dCnode[cur].as<uint32_t>(), dClo[cur].as<float4>(),
                           dChi[cur].as<float4>(), dCnode[cur ^ 1].as<uint32_t>(), dClo[cur ^ 1].as<float4>(), dChi[cur ^ 1].as<float4>());
        uint32_t n2 = 0;
        DEV_TRY(hipMemcpy(&n2, dScan.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost));
        if (n2 >= n) return "device BVH build made no progress";
        n = n2; cur ^= 1;
    }
    DEV_TRY(hipGetLastError());
    lap("clustering");
    if (verbose) fprintf(stderr, "mtsamd: device BVH %d clustering rounds\n", iterations);
    uint32_t root = 0;
    DEV_TRY(hipMemcpy(&root, dCnode[cur].p, 4, hipMemcpyDeviceToHost));
    std::vector<DNode> dn(2 * N - 1);
    DEV_TRY(hipMemcpy(dn.data(), dNodes.p, dn.size() * sizeof(DNode), hipMemcpyDeviceToHost));

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

}  // namespace mtsamd
