// TEST INFRASTRUCTURE ONLY -- CPU oracle: TriAccel, SAH kd-tree, Havran traversal and
// intersection-record construction, restated from the reference files cited per function.
#include "scene.h"
#include <cstring>
#include <thread>
#include <mutex>
#include <functional>

namespace orc {

// ---------------------------------------------------------------------------------------------
// TriAccel (include/mitsuba/render/triaccel.h:61-158)
// ---------------------------------------------------------------------------------------------
int TriAccel::load(const Vec3 &A, const Vec3 &B, const Vec3 &C) {
    static const int waldModulo[4] = { 1, 2, 0, 1 };
    Vec3 b = C - A, c = B - A, N = cross(c, b);
    k = 0;
    for (int j = 0; j < 3; j++)
        if (std::abs(N[j]) > std::abs(N[k])) k = j;
    uint32_t u = waldModulo[k], v = waldModulo[k + 1];
    const float n_k = N[k], denom = b[u] * c[v] - b[v] * c[u];
    if (denom == 0) { k = 3; return 1; }
    n_u = N[u] / n_k;
    n_v = N[v] / n_k;
    n_d = dot(A, N) / n_k;
    b_nu = b[u] / denom;
    b_nv = -b[v] / denom;
    a_u = A[u];
    a_v = A[v];
    c_nu = c[v] / denom;
    c_nv = -c[u] / denom;
    return 0;
}

inline bool TriAccel::rayIntersect(const Ray &ray, float mint, float maxt, float &u, float &v, float &t) const {
    float o_u, o_v, o_k, d_u, d_v, d_k;
    switch (k) {
        case 0: o_u = ray.o.y; o_v = ray.o.z; o_k = ray.o.x; d_u = ray.d.y; d_v = ray.d.z; d_k = ray.d.x; break;
        case 1: o_u = ray.o.z; o_v = ray.o.x; o_k = ray.o.y; d_u = ray.d.z; d_v = ray.d.x; d_k = ray.d.y; break;
        case 2: o_u = ray.o.x; o_v = ray.o.y; o_k = ray.o.z; d_u = ray.d.x; d_v = ray.d.y; d_k = ray.d.z; break;
        default: return false;
    }
    t = (n_d - o_u * n_u - o_v * n_v - o_k) / (d_u * n_u + d_v * n_v + d_k);
    if (t < mint || t > maxt) return false;
    const float hu = o_u + t * d_u - a_u;
    const float hv = o_v + t * d_v - a_v;
    u = hv * b_nu + hu * b_nv;
    v = hu * c_nu + hv * c_nv;
    return u >= 0 && v >= 0 && u + v <= 1.0f;
}

// ---------------------------------------------------------------------------------------------
// Triangle::getClippedAABB (src/libcore/triangle.cpp:71-147): double-precision
// Sutherland-Hodgman against the six box planes, conservative float rounding.
// ---------------------------------------------------------------------------------------------
struct P3d { double v[3]; };
static const int MAX_VERTS = 10;

static int sutherlandHodgman(const P3d *input, int inCount, P3d *output, int axis, double splitPos, bool isMinimum) {
    if (inCount < 3) return 0;
    P3d cur = input[0];
    double sign = isMinimum ? 1.0 : -1.0;
    double distance = sign * (cur.v[axis] - splitPos);
    bool curIsInside = (distance >= 0);
    int outCount = 0;
    for (int i = 0; i < inCount; ++i) {
        int nextIdx = i + 1;
        if (nextIdx == inCount) nextIdx = 0;
        P3d next = input[nextIdx];
        distance = sign * (next.v[axis] - splitPos);
        bool nextIsInside = (distance >= 0);
        if (curIsInside && nextIsInside) {
            output[outCount++] = next;
        } else if (curIsInside != nextIsInside) {
            double t = (splitPos - cur.v[axis]) / (next.v[axis] - cur.v[axis]);
            P3d p;
            for (int j = 0; j < 3; ++j) p.v[j] = cur.v[j] + (next.v[j] - cur.v[j]) * t;
            p.v[axis] = splitPos;
            output[outCount++] = p;
            if (nextIsInside) output[outCount++] = next;
        }
        cur = next;
        curIsInside = nextIsInside;
    }
    return outCount;
}

// include/mitsuba/core/math.h castflt_down / castflt_up
static float castflt_down(double val) {
    float a = (float) val;
    if ((double) a > val) a = std::nextafterf(a, -std::numeric_limits<float>::infinity());
    return a;
}
static float castflt_up(double val) {
    float a = (float) val;
    if ((double) a < val) a = std::nextafterf(a, std::numeric_limits<float>::infinity());
    return a;
}

AABB triangleClippedAABB(const Vec3 &p0, const Vec3 &p1, const Vec3 &p2, const AABB &box) {
    P3d v1[MAX_VERTS], v2[MAX_VERTS];
    int n = 3;
    const Vec3 *src[3] = { &p0, &p1, &p2 };
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v1[i].v[j] = (double) (*src[i])[j];
    for (int axis = 0; axis < 3; ++axis) {
        n = sutherlandHodgman(v1, n, v2, axis, box.min[axis], true);
        n = sutherlandHodgman(v2, n, v1, axis, box.max[axis], false);
    }
    AABB result;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j) {
            double pos = v1[i].v[j];
            result.min[j] = std::min(result.min[j], castflt_down(pos));
            result.max[j] = std::max(result.max[j], castflt_up(pos));
        }
    result.clip(box);
    return result;
}

// ---------------------------------------------------------------------------------------------
// SAH kd-tree construction. Same algorithm class and parameters as GenericKDTree
// (include/mitsuba/render/gkdtree.h:732-746: traversal 15, query 20, empty bonus 0.9,
// stop at 6 prims, depth cap 8+1.3*log2i(N) <= 48, 3 bad refines, exact event sweep with
// perfect-split clipping below 65536 prims, 128 min-max bins above; :1956-2114 sweep scoring).
// The tree topology is NOT contractual (the GPU side uses a BVH); only query results are.
// ---------------------------------------------------------------------------------------------
namespace {

const float kTraversalCost = 15, kQueryCost = 20, kEmptyBonus = 0.9f;
const uint32_t kStopPrims = 6, kMaxBadRefines = 3, kExactThreshold = 65536, kMinMaxBins = 128;

struct PrimRef { uint32_t idx; AABB box; };

struct SAH {
    Vec3 t0, t1;
    explicit SAH(const AABB &b) {
        Vec3 e = b.max - b.min;
        float temp = 1.0f / (e.x * e.y + e.y * e.z + e.x * e.z);
        t0 = Vec3(e.y * e.z, e.x * e.z, e.x * e.y) * temp;
        t1 = Vec3(e.y + e.z, e.x + e.z, e.x + e.y) * temp;
    }
    void prob(int axis, float lw, float rw, float &pl, float &pr) const {
        pl = t0[axis] + t1[axis] * lw;
        pr = t0[axis] + t1[axis] * rw;
    }
};

struct Split {
    float cost = std::numeric_limits<float>::infinity();
    float pos = 0;
    int axis = -1;
    bool planarLeft = false;
};

struct Event { float pos; uint32_t type; };  // 0 end, 1 planar, 2 start (sort order of the sweep)

struct SubTree {
    std::vector<KDNode> nodes;      // local indices; node 0 = subtree root
    std::vector<uint32_t> indices;
};

struct Builder {
    const Scene &scene;
    int maxDepth;
    explicit Builder(const Scene &s, int md) : scene(s), maxDepth(md) {}

    AABB primBox(uint32_t idx) const {
        AABB b;
        b.expandBy(scene.positions[scene.indices[3 * idx]]);
        b.expandBy(scene.positions[scene.indices[3 * idx + 1]]);
        b.expandBy(scene.positions[scene.indices[3 * idx + 2]]);
        return b;
    }
    AABB clipped(uint32_t idx, const AABB &box) const {
        return triangleClippedAABB(scene.positions[scene.indices[3 * idx]],
                                   scene.positions[scene.indices[3 * idx + 1]],
                                   scene.positions[scene.indices[3 * idx + 2]], box);
    }

    static void makeLeaf(SubTree &out, uint32_t node, const std::vector<PrimRef> &prims) {
        uint32_t start = (uint32_t) out.indices.size();
        for (const PrimRef &p : prims) out.indices.push_back(p.idx);
        out.nodes[node].combined = (start << 2) | 3u;
        out.nodes[node].primEnd = (uint32_t) out.indices.size();
    }

    Split findSplitExact(const AABB &box, const std::vector<PrimRef> &prims) const {
        Split best;
        SAH sah(box);
        std::vector<Event> ev;
        ev.reserve(prims.size() * 2);
        for (int axis = 0; axis < 3; ++axis) {
            ev.clear();
            for (const PrimRef &p : prims) {
                float lo = p.box.min[axis], hi = p.box.max[axis];
                if (lo == hi) ev.push_back({ lo, 1u });
                else { ev.push_back({ lo, 2u }); ev.push_back({ hi, 0u }); }
            }
            std::sort(ev.begin(), ev.end(), [](const Event &a, const Event &b) {
                return a.pos < b.pos || (a.pos == b.pos && a.type < b.type);
            });
            uint32_t nL = 0, nR = (uint32_t) prims.size();
            size_t i = 0;
            while (i < ev.size()) {
                float pos = ev[i].pos;
                uint32_t nEnd = 0, nPlanar = 0, nStart = 0;
                while (i < ev.size() && ev[i].pos == pos && ev[i].type == 0) { ++nEnd; ++i; }
                while (i < ev.size() && ev[i].pos == pos && ev[i].type == 1) { ++nPlanar; ++i; }
                while (i < ev.size() && ev[i].pos == pos && ev[i].type == 2) { ++nStart; ++i; }
                nR -= nPlanar + nEnd;
                if (pos > box.min[axis] && pos < box.max[axis]) {
                    float pl, pr;
                    sah.prob(axis, pos - box.min[axis], box.max[axis] - pos, pl, pr);
                    if (nPlanar == 0) {
                        float cost = kTraversalCost + kQueryCost * (pl * (float) nL + pr * (float) nR);
                        if (nL == 0 || nR == 0) cost *= kEmptyBonus;
                        if (cost < best.cost) { best.cost = cost; best.pos = pos; best.axis = axis; }
                    } else {
                        float cl = kTraversalCost + kQueryCost * (pl * (float) (nL + nPlanar) + pr * (float) nR);
                        float cr = kTraversalCost + kQueryCost * (pl * (float) nL + pr * (float) (nR + nPlanar));
                        if (nL + nPlanar == 0 || nR == 0) cl *= kEmptyBonus;
                        if (nL == 0 || nR + nPlanar == 0) cr *= kEmptyBonus;
                        if (cl < best.cost || cr < best.cost) {
                            best.pos = pos; best.axis = axis;
                            if (cl < cr) { best.cost = cl; best.planarLeft = true; }
                            else { best.cost = cr; best.planarLeft = false; }
                        }
                    }
                }
                nL += nStart + nPlanar;
            }
        }
        return best;
    }

    // Min-max binning (gkdtree.h:2400-2560): per axis, histogram of box minima and maxima.
    Split findSplitBinned(const AABB &box, const std::vector<PrimRef> &prims) const {
        Split best;
        SAH sah(box);
        std::vector<uint32_t> minBins(kMinMaxBins), maxBins(kMinMaxBins);
        for (int axis = 0; axis < 3; ++axis) {
            float lo = box.min[axis], hi = box.max[axis];
            if (!(hi > lo)) continue;
            float binSize = (hi - lo) / kMinMaxBins, invBin = 1.0f / binSize;
            std::fill(minBins.begin(), minBins.end(), 0u);
            std::fill(maxBins.begin(), maxBins.end(), 0u);
            for (const PrimRef &p : prims) {
                int a = std::min((int) kMinMaxBins - 1, std::max(0, (int) ((p.box.min[axis] - lo) * invBin)));
                int b = std::min((int) kMinMaxBins - 1, std::max(0, (int) ((p.box.max[axis] - lo) * invBin)));
                minBins[a]++; maxBins[b]++;
            }
            uint32_t nL = 0, nR = (uint32_t) prims.size();
            for (uint32_t i = 0; i + 1 < kMinMaxBins; ++i) {
                nL += minBins[i];
                nR -= maxBins[i];
                float pos = lo + binSize * (float) (i + 1);
                if (!(pos > lo && pos < hi)) continue;
                float pl, pr;
                sah.prob(axis, pos - lo, hi - pos, pl, pr);
                float cost = kTraversalCost + kQueryCost * (pl * (float) nL + pr * (float) nR);
                if (nL == 0 || nR == 0) cost *= kEmptyBonus;
                if (cost < best.cost) { best.cost = cost; best.pos = pos; best.axis = axis; }
            }
        }
        return best;
    }

    void build(SubTree &out, uint32_t node, const AABB &box, std::vector<PrimRef> &prims, int depth,
               uint32_t badRefines, std::vector<std::function<void()>> *deferred, std::mutex *mtx) const {
        uint32_t n = (uint32_t) prims.size();
        float leafCost = (float) n * kQueryCost;
        if (n <= kStopPrims || depth >= maxDepth) { makeLeaf(out, node, prims); return; }
        bool exact = n <= kExactThreshold;
        Split s = exact ? findSplitExact(box, prims) : findSplitBinned(box, prims);
        if (s.cost >= leafCost) {
            if ((s.cost > 4 * leafCost && n < 16) || badRefines >= kMaxBadRefines ||
                s.cost == std::numeric_limits<float>::infinity()) {
                makeLeaf(out, node, prims);
                return;
            }
            ++badRefines;
        }
        AABB lbox = box, rbox = box;
        lbox.max[s.axis] = s.pos;
        rbox.min[s.axis] = s.pos;
        std::vector<PrimRef> left, right;
        left.reserve(n / 2 + 8); right.reserve(n / 2 + 8);
        for (const PrimRef &p : prims) {
            float lo = p.box.min[s.axis], hi = p.box.max[s.axis];
            if (lo == hi && lo == s.pos) {
                (s.planarLeft || !exact ? left : right).push_back(p);
            } else if (hi <= s.pos) {
                left.push_back(p);
            } else if (lo >= s.pos) {
                right.push_back(p);
            } else if (exact) {
                // straddling: perfect split -- re-clip the triangle to both child boxes
                AABB lb = clipped(p.idx, lbox), rb = clipped(p.idx, rbox);
                if (lb.isValid() && lb.surfaceArea() >= 0) left.push_back({ p.idx, lb });
                if (rb.isValid() && rb.surfaceArea() >= 0) right.push_back({ p.idx, rb });
            } else {
                PrimRef l = p, r = p;
                l.box.max[s.axis] = s.pos;
                r.box.min[s.axis] = s.pos;
                left.push_back(l); right.push_back(r);
            }
        }
        std::vector<PrimRef>().swap(prims);
        uint32_t li = (uint32_t) out.nodes.size();
        out.nodes.push_back(KDNode()); out.nodes.push_back(KDNode());
        out.nodes[node].combined = (li << 2) | (uint32_t) s.axis;
        out.nodes[node].split = s.pos;
        build(out, li, lbox, left, depth + 1, badRefines, deferred, mtx);
        build(out, li + 1, rbox, right, depth + 1, badRefines, deferred, mtx);
    }
};

}  // namespace

void Scene::buildKDTree() {
    uint32_t primCount = (uint32_t) (indices.size() / 3);
    // TriAccel precompute (src/librender/skdtree.cpp:79-105)
    triAccel.resize(primCount);
    for (uint32_t s = 0; s < shapes.size(); ++s)
        for (uint32_t j = 0; j < shapes[s].n_triangles; ++j) {
            uint32_t idx = shapes[s].first_triangle + j;
            triAccel[idx].load(positions[indices[3 * idx]], positions[indices[3 * idx + 1]], positions[indices[3 * idx + 2]]);
            triAccel[idx].shapeIndex = s;
            triAccel[idx].primIndex = j;
        }

    int log2i = 0;
    while ((primCount >> log2i) != 0) log2i++;
    log2i -= 1;
    kdMaxDepth = std::min(48, (int) (8 + 1.3f * (float) std::max(log2i, 0)));   // gkdtree.h:986-988

    Builder builder(*this, kdMaxDepth);
    std::vector<PrimRef> prims(primCount);
    AABB box;
    for (uint32_t i = 0; i < primCount; ++i) {
        prims[i].idx = i;
        prims[i].box = builder.primBox(i);
        box.expandBy(prims[i].box);
    }
    // Slightly enlarged scene bounds (gkdtree.h:1213-1220, MTS_KD_AABB_EPSILON = 1e-3f)
    tightAABB = box;
    const float eps = 1e-3f;
    if (primCount > 0) {
        box.min = box.min - ((box.max - box.min) * eps + Vec3(eps));
        box.max = box.max + ((box.max - box.min) * eps + Vec3(eps));
    } else {
        box.min = Vec3(0.0f); box.max = Vec3(0.0f);
    }
    aabb = box;

    // Top of the tree is built serially down to a fixed number of independent subtrees, which
    // are then built in parallel and spliced in (the reference also builds subtrees in parallel).
    struct Job { uint32_t node; AABB box; std::vector<PrimRef> prims; int depth; uint32_t bad; SubTree sub; };
    std::vector<Job> jobs;
    SubTree top;
    top.nodes.push_back(KDNode());
    {
        struct Item { uint32_t node; AABB box; std::vector<PrimRef> prims; int depth; uint32_t bad; };
        std::vector<Item> stack;
        stack.push_back({ 0, box, std::move(prims), 0, 0 });
        const uint32_t parallelThreshold = std::max<uint32_t>(4096, primCount / 64);
        while (!stack.empty()) {
            Item it = std::move(stack.back());
            stack.pop_back();
            uint32_t n = (uint32_t) it.prims.size();
            if (n <= parallelThreshold || it.depth >= kdMaxDepth) {
                jobs.push_back({ it.node, it.box, std::move(it.prims), it.depth, it.bad, SubTree() });
                continue;
            }
            Split s = n <= kExactThreshold ? builder.findSplitExact(it.box, it.prims) : builder.findSplitBinned(it.box, it.prims);
            float leafCost = (float) n * kQueryCost;
            if (s.cost >= leafCost) {
                if (it.bad >= kMaxBadRefines || s.cost == std::numeric_limits<float>::infinity()) {
                    jobs.push_back({ it.node, it.box, std::move(it.prims), kdMaxDepth, it.bad, SubTree() });
                    continue;
                }
                ++it.bad;
            }
            bool exact = n <= kExactThreshold;
            AABB lbox = it.box, rbox = it.box;
            lbox.max[s.axis] = s.pos; rbox.min[s.axis] = s.pos;
            std::vector<PrimRef> left, right;
            for (const PrimRef &p : it.prims) {
                float lo = p.box.min[s.axis], hi = p.box.max[s.axis];
                if (lo == hi && lo == s.pos) (s.planarLeft || !exact ? left : right).push_back(p);
                else if (hi <= s.pos) left.push_back(p);
                else if (lo >= s.pos) right.push_back(p);
                else if (exact) {
                    AABB lb = builder.clipped(p.idx, lbox), rb = builder.clipped(p.idx, rbox);
                    if (lb.isValid()) left.push_back({ p.idx, lb });
                    if (rb.isValid()) right.push_back({ p.idx, rb });
                } else {
                    PrimRef l = p, r = p;
                    l.box.max[s.axis] = s.pos; r.box.min[s.axis] = s.pos;
                    left.push_back(l); right.push_back(r);
                }
            }
            uint32_t li = (uint32_t) top.nodes.size();
            top.nodes.push_back(KDNode()); top.nodes.push_back(KDNode());
            top.nodes[it.node].combined = (li << 2) | (uint32_t) s.axis;
            top.nodes[it.node].split = s.pos;
            stack.push_back({ li + 1, rbox, std::move(right), it.depth + 1, it.bad });
            stack.push_back({ li, lbox, std::move(left), it.depth + 1, it.bad });
        }
    }

    {
        std::atomic<size_t> next(0);
        unsigned nt = std::max(1u, std::thread::hardware_concurrency());
        nt = std::min<unsigned>(nt, 64);
        std::vector<std::thread> threads;
        for (unsigned t = 0; t < nt; ++t)
            threads.emplace_back([&]() {
                for (;;) {
                    size_t j = next.fetch_add(1);
                    if (j >= jobs.size()) break;
                    Job &job = jobs[j];
                    job.sub.nodes.push_back(KDNode());
                    builder.build(job.sub, 0, job.box, job.prims, job.depth, job.bad, nullptr, nullptr);
                }
            });
        for (auto &t : threads) t.join();
    }

    // Splice: subtree root replaces the placeholder node; its other nodes are appended.
    nodes = std::move(top.nodes);
    kdIndices.clear();
    for (Job &job : jobs) {
        uint32_t nodeBase = (uint32_t) nodes.size();      // local node i (i>=1) -> nodeBase + i - 1
        uint32_t idxBase = (uint32_t) kdIndices.size();
        auto remap = [&](const KDNode &n) {
            KDNode r = n;
            if (n.isLeaf()) {
                r.combined = ((n.primStart() + idxBase) << 2) | 3u;
                r.primEnd = n.primEnd + idxBase;
            } else {
                r.combined = ((n.left() - 1 + nodeBase) << 2) | (uint32_t) n.axis();
            }
            return r;
        };
        nodes[job.node] = remap(job.sub.nodes[0]);
        for (size_t i = 1; i < job.sub.nodes.size(); ++i) nodes.push_back(remap(job.sub.nodes[i]));
        kdIndices.insert(kdIndices.end(), job.sub.indices.begin(), job.sub.indices.end());
    }
}

// ---------------------------------------------------------------------------------------------
// Havran traversal (include/mitsuba/render/sahkdtree3.h:178-308) with the 8-entry hashed
// mailbox (sahkdtree3.h:30-32,138-152).
// ---------------------------------------------------------------------------------------------
template <bool shadowRay>
bool Scene::rayIntersectHavran(const Ray &ray, float mint, float maxt, float &t,
                               Intersection *its, TraversalCounters *c) const {
    struct StackEntry { uint32_t node; float t; uint32_t prev; Vec3 p; };
    const uint32_t NONE = 0xFFFFFFFFu;
    StackEntry stack[48 + 2];
    uint32_t mailbox[8];
    memset(mailbox, 0xFF, sizeof(mailbox));

    uint32_t enPt = 0;
    stack[enPt].t = mint;
    stack[enPt].p = ray(mint);
    uint32_t exPt = 1;
    stack[exPt].t = maxt;
    stack[exPt].p = ray(maxt);
    stack[exPt].node = NONE;

    bool found = false;
    uint32_t curr = 0;
    if (nodes.empty()) return false;
    while (curr != NONE) {
        while (!nodes[curr].isLeaf()) {
            const KDNode &nd = nodes[curr];
            const float splitVal = nd.split;
            const int axis = nd.axis();
            uint32_t farChild;
            if (c) c->steps++;
            if (stack[enPt].p[axis] <= splitVal) {
                if (stack[exPt].p[axis] <= splitVal) { curr = nd.left(); continue; }
                if (stack[enPt].p[axis] == splitVal) { curr = nd.left() + 1; continue; }
                curr = nd.left();
                farChild = curr + 1;
            } else {
                if (splitVal < stack[exPt].p[axis]) { curr = nd.left() + 1; continue; }
                farChild = nd.left();
                curr = farChild + 1;
            }
            float distToSplit = (splitVal - ray.o[axis]) * ray.dRcp[axis];
            const uint32_t tmp = exPt++;
            if (exPt == enPt) ++exPt;
            stack[exPt].prev = tmp;
            stack[exPt].t = distToSplit;
            stack[exPt].node = farChild;
            stack[exPt].p = ray(distToSplit);
            stack[exPt].p[axis] = splitVal;
        }
        const KDNode &leaf = nodes[curr];
        for (uint32_t entry = leaf.primStart(), last = leaf.primEnd; entry != last; entry++) {
            const uint32_t primIdx = kdIndices[entry];
            if (mailbox[primIdx & 7] == primIdx) continue;
            float tu, tv, tt;
            if (c) c->tests++;
            bool result = triAccel[primIdx].rayIntersect(ray, mint, maxt, tu, tv, tt);
            if (result) {
                if (shadowRay) return true;
                t = tt;
                its->u = tu; its->v = tv;
                its->shape = triAccel[primIdx].shapeIndex;
                its->prim = triAccel[primIdx].primIndex;
                maxt = t;
                found = true;
            }
            mailbox[primIdx & 7] = primIdx;
        }
        if (stack[exPt].t > maxt) break;
        enPt = exPt;
        curr = stack[exPt].node;
        exPt = stack[enPt].prev;
    }
    return found;
}

// src/librender/skdtree.cpp:112-142
bool Scene::rayIntersect(const Ray &ray, Intersection &its, TraversalCounters *c) const {
    its.t = std::numeric_limits<float>::infinity();
    its.shape = its.prim = 0xFFFFFFFFu;
    float mint, maxt;
    if (aabb.rayIntersect(ray, mint, maxt)) {
        float rayMinT = ray.mint;
        if (rayMinT == Epsilon)
            rayMinT *= std::max(std::max(std::max(std::abs(ray.o.x), std::abs(ray.o.y)), std::abs(ray.o.z)), Epsilon);
        if (rayMinT > mint) mint = rayMinT;
        if (ray.maxt < maxt) maxt = ray.maxt;
        if (maxt > mint) {
            if (rayIntersectHavran<false>(ray, mint, maxt, its.t, &its, c)) {
                fillIntersectionRecord(ray, its);
                return true;
            }
        }
    }
    return false;
}

// src/librender/skdtree.cpp:207-226 (note: no max(..., Epsilon) clamp here)
bool Scene::rayIntersectShadow(const Ray &ray, TraversalCounters *c) const {
    float mint, maxt, t = std::numeric_limits<float>::infinity();
    if (aabb.rayIntersect(ray, mint, maxt)) {
        float rayMinT = ray.mint;
        if (rayMinT == Epsilon)
            rayMinT *= std::max(std::max(std::abs(ray.o.x), std::abs(ray.o.y)), std::abs(ray.o.z));
        if (rayMinT > mint) mint = rayMinT;
        if (ray.maxt < maxt) maxt = ray.maxt;
        if (maxt > mint)
            if (rayIntersectHavran<true>(ray, mint, maxt, t, nullptr, c)) return true;
    }
    return false;
}

// Exhaustive variant of the two queries above (same interval logic, every TriAccel tested in
// index order, last accepted equal-t primitive wins like the leaf loop). Test aid only.
bool Scene::rayIntersectBrute(const Ray &ray, Intersection &its, bool shadow) const {
    its.t = std::numeric_limits<float>::infinity();
    its.shape = its.prim = 0xFFFFFFFFu;
    float mint, maxt;
    if (!aabb.rayIntersect(ray, mint, maxt)) return false;
    float rayMinT = ray.mint;
    if (rayMinT == Epsilon) {
        float m = std::max(std::max(std::abs(ray.o.x), std::abs(ray.o.y)), std::abs(ray.o.z));
        rayMinT *= shadow ? m : std::max(m, Epsilon);
    }
    if (rayMinT > mint) mint = rayMinT;
    if (ray.maxt < maxt) maxt = ray.maxt;
    if (!(maxt > mint)) return false;
    bool found = false;
    for (size_t i = 0; i < triAccel.size(); ++i) {
        float tu, tv, tt;
        if (triAccel[i].rayIntersect(ray, mint, maxt, tu, tv, tt)) {
            if (shadow) return true;
            its.t = tt; its.u = tu; its.v = tv;
            its.shape = triAccel[i].shapeIndex; its.prim = triAccel[i].primIndex;
            maxt = tt;
            found = true;
        }
    }
    if (found) fillIntersectionRecord(ray, its);
    return found;
}

// include/mitsuba/render/skdtree.h:343-428 (BarycentricPos = true)
void Scene::fillIntersectionRecord(const Ray &ray, Intersection &its) const {
    const orc_shape &sh = shapes[its.shape];
    uint32_t tri = sh.first_triangle + its.prim;
    const Vec3 b(1 - its.u - its.v, its.u, its.v);
    const uint32_t idx0 = indices[3 * tri], idx1 = indices[3 * tri + 1], idx2 = indices[3 * tri + 2];
    const Vec3 &p0 = positions[idx0], &p1 = positions[idx1], &p2 = positions[idx2];
    its.p = p0 * b.x + p1 * b.y + p2 * b.z;
    Vec3 side1(p1 - p0), side2(p2 - p0);
    Vec3 faceNormal(cross(side1, side2));
    float length = faceNormal.length();
    if (!faceNormal.isZero()) faceNormal /= length;
    if ((sh.flags & ORC_SHAPE_HAS_TEXCOORDS) && uvDpduValid[tri]) {
        // per-triangle UV tangents (src/librender/trimesh.cpp:683-735), computed in Scene::build
        its.dpdu = uvDpdu[tri];
        its.dpdv = uvDpdv[tri];
    } else if (sh.flags & ORC_SHAPE_HAS_TEXCOORDS) {
        its.dpdu = its.dpdv = Vec3(0.0f);   // zero-initialised TangentSpace of a degenerate triangle
    } else {
        its.dpdu = side1;
        its.dpdv = side2;
    }
    Vec3 shN;
    if (sh.flags & ORC_SHAPE_HAS_NORMALS) {
        const Vec3 &n0 = normals[idx0], &n1 = normals[idx1], &n2 = normals[idx2];
        shN = normalize(n0 * b.x + n1 * b.y + n2 * b.z);
        if (dot(faceNormal, shN) < 0) faceNormal = -faceNormal;
    } else {
        shN = faceNormal;
    }
    its.geoFrame = Frame(faceNormal);
    if (sh.flags & ORC_SHAPE_HAS_TEXCOORDS) {
        const Vec2 &t0 = texcoords[idx0], &t1 = texcoords[idx1], &t2 = texcoords[idx2];
        its.uv.x = t0.x * b.x + t1.x * b.y + t2.x * b.z;
        its.uv.y = t0.y * b.x + t1.y * b.y + t2.y * b.z;
    } else {
        its.uv.x = b.y; its.uv.y = b.z;
    }
    computeShadingFrame(shN, its.dpdu, its.shFrame);
    its.wi = its.toLocal(-ray.d);
}

}  // namespace orc
