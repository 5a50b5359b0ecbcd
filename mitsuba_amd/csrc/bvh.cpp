// Host-side construction of the 8-wide quantised BVH: binned-SAH binary build (multi-threaded), greedy
// surface-area collapse to 8-ary nodes, octant-ordered slot assignment and outward-rounded quantisation.
#include "bvh.h"
#include "bvh_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>

namespace mtsamd {

namespace {

struct Prim { Box box; float c[3]; uint32_t tri; };

float kCostNode = 1.0f, kCostTri = 0.5f;   // relative cost of visiting a wide node / testing a triangle
const int kBins = 16;
const uint32_t kMaxLeaf = kLeafTris;

struct Builder2 {
    std::vector<Prim> prims;
    std::vector<Node2> nodes;
    std::atomic<uint32_t> nextNode{ 0 };
    std::atomic<int> liveThreads{ 0 };
    int maxThreads = 1;
    int sahDepthLimit = std::numeric_limits<int>::max();   // below this depth: median splits (a balanced subtree)

    uint32_t alloc2() { return nextNode.fetch_add(2); }

    void build(uint32_t nodeIdx, uint32_t first, uint32_t count, int depth) {
        Node2 &node = nodes[nodeIdx];
        Box box, cbox;
        box.reset(); cbox.reset();
        for (uint32_t i = first; i < first + count; ++i) { box.grow(prims[i].box); cbox.grow(prims[i].c); }
        node.box = box;
        node.first = first; node.count = count;     // primitive range of the subtree (kept for internal nodes too)
        if (count == 1) return;

        // binned SAH over the centroid bounds
        int bestAxis = -1, bestBin = -1;
        float bestCost = std::numeric_limits<float>::infinity();
        for (int axis = 0; axis < 3 && depth < sahDepthLimit; ++axis) {
            float lo = cbox.lo[axis], hi = cbox.hi[axis];
            if (!(hi > lo)) continue;
            Box bb[kBins];
            uint32_t bc[kBins];
            for (int b = 0; b < kBins; ++b) { bb[b].reset(); bc[b] = 0; }
            float scale = (float) kBins / (hi - lo);
            for (uint32_t i = first; i < first + count; ++i) {
                int b = std::min(kBins - 1, std::max(0, (int) ((prims[i].c[axis] - lo) * scale)));
                bb[b].grow(prims[i].box); bc[b]++;
            }
            float rightArea[kBins];
            uint32_t rightCount[kBins];
            Box acc; acc.reset();
            uint32_t n = 0;
            for (int b = kBins - 1; b > 0; --b) {
                acc.grow(bb[b]); n += bc[b];
                rightArea[b] = acc.area(); rightCount[b] = n;
            }
            acc.reset(); n = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(bb[b]); n += bc[b];
                if (n == 0 || rightCount[b + 1] == 0) continue;
                float cost = acc.area() * (float) n + rightArea[b + 1] * (float) rightCount[b + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = b; }
            }
        }
        uint32_t mid;
        if (bestAxis >= 0) {
            float lo = cbox.lo[bestAxis], hi = cbox.hi[bestAxis];
            float scale = (float) kBins / (hi - lo);
            Prim *b = prims.data() + first, *e = b + count;
            Prim *m = std::partition(b, e, [&](const Prim &p) {
                int bin = std::min(kBins - 1, std::max(0, (int) ((p.c[bestAxis] - lo) * scale)));
                return bin <= bestBin;
            });
            mid = (uint32_t) (m - prims.data());
        } else {
            // all centroids coincide, or the SAH splits have used up their depth allowance: median split along the
            // widest axis of the centroid bounds, which halves the count whatever the geometry looks like
            mid = first + count / 2;
            int axis = 0;
            for (int a = 1; a < 3; ++a) if (cbox.hi[a] - cbox.lo[a] > cbox.hi[axis] - cbox.lo[axis]) axis = a;
            if (cbox.hi[axis] > cbox.lo[axis])
                std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                                 [axis](const Prim &a, const Prim &b) { return a.c[axis] < b.c[axis]; });
        }
        if (mid == first || mid == first + count) mid = first + count / 2;
        uint32_t c = alloc2();
        node.left = c; node.right = c + 1;
        uint32_t nl = mid - first, nr = count - nl;
        bool spawn = nl > 32768 && nr > 32768 && liveThreads.load() < maxThreads;
        if (spawn) {
            liveThreads.fetch_add(1);
            std::thread t([=]() { build(c, first, nl, depth + 1); liveThreads.fetch_sub(1); });
            build(c + 1, mid, nr, depth + 1);
            t.join();
        } else {
            build(c, first, nl, depth + 1);
            build(c + 1, mid, nr, depth + 1);
        }
    }
};

inline uint32_t floatBits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float bitsFloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

}  // namespace

void loadTriRecord(const float *A, const float *B, const float *C, TriRecord &t) { loadTriRecordHD(A, B, C, t); }

const char *buildBinaryTreeHost(const float *positions, uint32_t nVerts, const uint32_t *indices, uint32_t nTris, BinaryTree &out, int sahDepthLimit) {
    Builder2 b2;
    b2.sahDepthLimit = sahDepthLimit;
    b2.prims.resize(nTris);
    Box sceneBox; sceneBox.reset();
    for (uint32_t t = 0; t < nTris; ++t) {
        Prim &p = b2.prims[t];
        p.tri = t;
        p.box.reset();
        for (int j = 0; j < 3; ++j) {
            uint32_t vi = indices[3 * t + j];
            if (vi >= nVerts) return "vertex index out of range";
            const float *v = positions + 3 * (size_t) vi;
            if (!(std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]))) return "non-finite vertex position";
            p.box.grow(v);
        }
        for (int i = 0; i < 3; ++i) p.c[i] = 0.5f * (p.box.lo[i] + p.box.hi[i]);
        sceneBox.grow(p.box);
    }
    b2.nodes.resize(2 * (size_t) nTris + 2);
    b2.nextNode = 1;
    b2.maxThreads = (int) std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    b2.build(0, 0, nTris, 0);
    out.nNodes = b2.nextNode.load();
    out.nodes.swap(b2.nodes);
    out.primTri.resize(nTris);
    for (uint32_t i = 0; i < nTris; ++i) out.primTri[i] = b2.prims[i].tri;
    out.sceneBox = sceneBox;
    return "";
}

// The same SAH sweep over arbitrary boxes (the device builder uses it for the top of its tree, over clusters)
const char *buildBinaryOverBoxes(const std::vector<Box> &boxes, BinaryTree &out) {
    const uint32_t n = (uint32_t) boxes.size();
    Builder2 b2;
    b2.prims.resize(n);
    Box sceneBox; sceneBox.reset();
    for (uint32_t i = 0; i < n; ++i) {
        Prim &p = b2.prims[i];
        p.tri = i; p.box = boxes[i];
        for (int k = 0; k < 3; ++k) p.c[k] = 0.5f * (p.box.lo[k] + p.box.hi[k]);
        sceneBox.grow(p.box);
    }
    b2.nodes.resize(2 * (size_t) n + 2);
    b2.nextNode = 1;
    b2.maxThreads = (int) std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    b2.build(0, 0, n, 0);
    out.nNodes = b2.nextNode.load();
    out.nodes.swap(b2.nodes);
    out.primTri.resize(n);
    for (uint32_t i = 0; i < n; ++i) out.primTri[i] = b2.prims[i].tri;
    out.sceneBox = sceneBox;
    return "";
}

const char *buildBVH(const float *positions, uint32_t nVerts, const uint32_t *indices, uint32_t nTris, BVH &out, int builder) {
    auto t0 = std::chrono::steady_clock::now();
    out.nodes.clear(); out.tris.clear(); out.depth = 0;
    for (int i = 0; i < 3; ++i) { out.aabbMin[i] = 0; out.aabbMax[i] = 0; }
    if (nTris == 0) {
        WideNode n;
        memset(&n, 0, sizeof(n));
        n.e[0] = n.e[1] = n.e[2] = 1;
        out.nodes.push_back(n);
        out.nodeCount = 1; out.triCount = 0;
        return "";
    }
    if (nTris >= (1u << 27)) return "too many triangles";
    if (const char *e = getenv("MTSAMD_COST_TRI")) kCostTri = (float) atof(e);
    setDeviceBuilderCosts(kCostNode, kCostTri);
    out.builder = builder;
    // Surface-area splits follow the geometry and can produce a tree deeper than the traversal stack (long chains of
    // nested or geometrically growing triangles). Such a tree is rebuilt on the host with the surface-area splits
    // confined to the top levels and median splits below; with no surface-area levels at all the tree is balanced
    // and fits by construction (kMaxWideDepth).
    const int kNoLimit = std::numeric_limits<int>::max();
    const int limits[] = { kNoLimit, 16, 0 };
    const char *err = "";
    for (int limit : limits) {
        BinaryTree tree;
        const bool onDevice = builder == BVH_BUILDER_DEVICE_BINARY && limit == kNoLimit;
        err = onDevice ? buildBinaryTreeDevice(positions, nVerts, indices, nTris, tree)
                       : buildBinaryTreeHost(positions, nVerts, indices, nTris, tree, limit);
        if (err && *err) return err;
        if (!onDevice) out.builder = BVH_BUILDER_HOST;
        out.binarySeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        out.nodes.clear(); out.tris.clear(); out.depth = 0;
        err = collapseToWide(tree, positions, indices, out);
        if (!(err && strcmp(err, kTooDeep) == 0)) break;
    }
    out.buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return err;
}

const char *collapseToWide(const BinaryTree &tree, const float *positions, const uint32_t *indices, BVH &out) {
    const std::vector<Node2> &nodes2 = tree.nodes;
    const Box sceneBox = tree.sceneBox;
    const uint32_t nTris = (uint32_t) tree.primTri.size();
    for (int i = 0; i < 3; ++i) { out.aabbMin[i] = sceneBox.lo[i]; out.aabbMax[i] = sceneBox.hi[i]; }

    // ---- collapse to 8-wide: dynamic programme over the binary tree -----------------------------------
    // C(n,i) = least SAH cost of representing the subtree of n by at most i roots of the wide tree
    // (a root is a leaf of <= kMaxLeaf triangles or a wide node with up to 8 roots below it):
    //   C(n,1) = min(leaf(n), A(n) * kCostNode + D(n,8)),  D(n,j) = min_k C(left,k) + C(right,j-k),
    //   C(n,i) = min(D(n,i), C(n,i-1)).
    // Children carry larger indices than their parents, so a descending sweep is bottom-up.
    const uint32_t nNodes2 = tree.nNodes;
    struct Dp { float c[8]; uint8_t k[9]; uint8_t take; uint8_t leaf; };    // c[1..7], k[2..8] = left share, take bit j: D(n,j) < C(n,j-1)
    std::vector<Dp> dp(nNodes2);
    const float kInfCost = std::numeric_limits<float>::infinity();
    for (uint32_t n = nNodes2; n-- > 0;) {
        const Node2 &nd = nodes2[n];
        Dp &d = dp[n];
        const float area = nd.box.area();
        if (nd.isLeaf()) {
            for (int i = 1; i < 8; ++i) d.c[i] = area * kCostTri;
            d.leaf = 1; d.take = 0;
            continue;
        }
        const Dp &L = dp[nd.left], &R = dp[nd.right];
        float dist[9];
        for (int j = 2; j <= 8; ++j) {
            float best = kInfCost; int bk = 1;
            for (int k = 1; k < j; ++k) {
                if (k > 7 || j - k > 7) continue;
                const float c = L.c[k] + R.c[j - k];
                if (c < best) { best = c; bk = k; }
            }
            dist[j] = best; d.k[j] = (uint8_t) bk;
        }
        const float leafCost = nd.count <= kMaxLeaf ? area * kCostTri * (float) nd.count : kInfCost;
        const float nodeCost = area * kCostNode + dist[8];
        d.leaf = nd.count <= kMaxLeaf && leafCost <= nodeCost;    // (the count guard matters when the areas overflow)
        d.c[1] = std::min(leafCost, nodeCost);
        d.take = 0;
        for (int j = 2; j <= 7; ++j) {
            if (dist[j] < d.c[j - 1]) { d.c[j] = dist[j]; d.take |= (uint8_t) (1u << j); }
            else d.c[j] = d.c[j - 1];
        }
    }
    // roots of the cheapest representation of subtree n by at most j roots
    struct Collector {
        const std::vector<Node2> &nodes; const std::vector<Dp> &dp;
        void roots(uint32_t n, int j, uint32_t *kids, int &nk) const {
            while (true) {
                const Node2 &nd = nodes[n];
                if (nd.isLeaf() || j == 1) { kids[nk++] = n; return; }
                if (j <= 7 && !((dp[n].take >> j) & 1)) { --j; continue; }
                const int k = dp[n].k[j];
                roots(nd.left, k, kids, nk);
                n = nd.right; j -= k;
            }
        }
    } collector{ nodes2, dp };

    struct Work { uint32_t n2; uint32_t wide; uint32_t depth; };
    std::vector<Work> queue;
    out.nodes.reserve(nTris / 4 + 16);
    out.tris.reserve(nTris);
    out.nodes.push_back(WideNode());
    queue.push_back({ 0, 0, 1 });
    double sah = 0;
    const float rootArea = std::max(sceneBox.area(), 1e-30f);

    for (size_t qi = 0; qi < queue.size(); ++qi) {
        Work w = queue[qi];
        out.depth = std::max(out.depth, w.depth);
        const Node2 &n2 = nodes2[w.n2];
        uint32_t kids[8];
        int nk = 0;
        if (n2.isLeaf() || (w.n2 == 0 && dp[0].leaf)) {
            kids[nk++] = w.n2;      // the whole scene fits one leaf
        } else {
            const int k = dp[w.n2].k[8];
            collector.roots(n2.left, k, kids, nk);
            collector.roots(n2.right, 8 - k, kids, nk);
        }
        // padded child boxes and node bounds
        Box cb[8], nb;
        nb.reset();
        for (int i = 0; i < nk; ++i) {
            cb[i] = nodes2[kids[i]].box;
            for (int a = 0; a < 3; ++a) {
                float m = std::max(std::abs(cb[i].lo[a]), std::abs(cb[i].hi[a]));
                float pad = m * 0x1p-20f + 1e-30f;
                cb[i].lo[a] -= pad; cb[i].hi[a] += pad;
            }
            nb.grow(cb[i]);
        }
        sah += (double) (nb.area() / rootArea) * kCostNode;

        // slot assignment: slot s is the first one visited by rays whose sign octant is s, so it should
        // hold the child lying furthest "behind" along the representative direction of that octant
        int slotOf[8], childAt[8];
        for (int i = 0; i < 8; ++i) { slotOf[i] = -1; childAt[i] = -1; }
        float nc[3] = { 0.5f * (nb.lo[0] + nb.hi[0]), 0.5f * (nb.lo[1] + nb.hi[1]), 0.5f * (nb.lo[2] + nb.hi[2]) };
        float cost[8][8];
        for (int i = 0; i < nk; ++i)
            for (int s = 0; s < 8; ++s) {
                float c = 0;
                for (int a = 0; a < 3; ++a) {
                    float d = 0.5f * (cb[i].lo[a] + cb[i].hi[a]) - nc[a];
                    c += ((s >> a) & 1) ? -d : d;
                }
                cost[i][s] = c;
            }
        for (int it = 0; it < nk; ++it) {
            int bi = -1, bs = -1; float bc = std::numeric_limits<float>::infinity();
            for (int i = 0; i < nk; ++i) {
                if (slotOf[i] >= 0) continue;
                for (int s = 0; s < 8; ++s) {
                    if (childAt[s] >= 0) continue;
                    if (cost[i][s] < bc) { bc = cost[i][s]; bi = i; bs = s; }
                }
            }
            slotOf[bi] = bs; childAt[bs] = bi;
        }

        WideNode node;
        memset(&node, 0, sizeof(node));
        for (int a = 0; a < 3; ++a) {
            node.p[a] = nb.lo[a];
            float ext = nb.hi[a] - nb.lo[a];
            int e = 1;
            if (ext > 0) {
                int ex;
                std::frexp(ext / 255.0f, &ex);       // ext/255 = m * 2^ex with m in [0.5,1)  => 2^ex >= ext/255
                e = std::min(254, std::max(1, ex + 127));
            }
            node.e[a] = (uint8_t) e;
        }
        node.childBase = (uint32_t) out.nodes.size();
        node.triBase = (uint32_t) out.tris.size();
        uint32_t nInternal = 0, leafTris = 0;
        for (int s = 0; s < 8; ++s) {
            int i = childAt[s];
            if (i < 0) continue;
            const Node2 &c = nodes2[kids[i]];
            for (int a = 0; a < 3; ++a) {
                double scale = (double) bitsFloat((uint32_t) node.e[a] << 23);
                double p = node.p[a];
                int lo = (int) std::floor(((double) cb[i].lo[a] - p) / scale);
                int hi = (int) std::ceil(((double) cb[i].hi[a] - p) / scale);
                lo = std::min(255, std::max(0, lo));
                hi = std::min(255, std::max(0, hi));
                // the decoded planes must enclose the padded child box
                while (lo > 0 && p + lo * scale > (double) cb[i].lo[a]) --lo;
                while (hi < 255 && p + hi * scale < (double) cb[i].hi[a]) ++hi;
                node.qlo[a][s] = (uint8_t) lo;
                node.qhi[a][s] = (uint8_t) hi;
            }
            if (dp[kids[i]].leaf) {
                uint32_t unary = (1u << c.count) - 1u;     // 1 -> 001, 2 -> 011, 3 -> 111
                leafTris |= unary << (kLeafTris * s);
                for (uint32_t j = 0; j < c.count; ++j) {
                    uint32_t tri = tree.primTri[c.first + j];
                    TriRecord tr;
                    loadTriRecord(positions + 3 * (size_t) indices[3 * tri], positions + 3 * (size_t) indices[3 * tri + 1],
                                  positions + 3 * (size_t) indices[3 * tri + 2], tr);
                    tr.tri = tri;
                    out.tris.push_back(tr);
                }
                sah += (double) (cb[i].area() / rootArea) * kCostTri * c.count;
            } else {
                node.imask |= (uint8_t) (1u << s);
                nInternal++;
            }
        }
        node.leafTris = leafTris;
        out.nodes[w.wide] = node;
        // internal children are laid out contiguously in slot order
        for (int s = 0; s < 8; ++s) {
            int i = childAt[s];
            if (i < 0 || dp[kids[i]].leaf) continue;
            uint32_t idx = (uint32_t) out.nodes.size();
            out.nodes.push_back(WideNode());
            queue.push_back({ kids[i], idx, w.depth + 1 });
        }
        (void) nInternal;
    }
    if ((int) out.depth > kMaxWideDepth) return kTooDeep;
    out.sahCost = sah;
    out.nodeCount = (uint32_t) out.nodes.size(); out.triCount = (uint32_t) out.tris.size();
    return "";
}

}  // namespace mtsamd
