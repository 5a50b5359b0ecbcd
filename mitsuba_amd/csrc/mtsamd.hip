// Host driver and C ABI of the MI355X path-tracing core: scene upload, the wavefront render loop
// (raygen -> [extend -> shade -> shadow-connect]* -> film) and the batch entry points of the parity harness.
#include "../../include/mtsamd.h"
#include "host_scene.h"
#include "mipmap.h"
#include "kernels.h"
#include "bvh_internal.h"

#include <atomic>
#include <memory>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace mtsamd;

namespace {

thread_local std::string g_error;

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

}  // namespace

// for the host-side helpers in the other translation units of the library
namespace mtsamd { void setLastError(const std::string &msg) { g_error = msg; } }

namespace {

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return fail(MTSAMD_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    ~DeviceBuffer() { if (ptr) (void) hipFree(ptr); }
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (ptr) { (void) hipFree(ptr); ptr = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&ptr, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <typename T> T *as() const { return (T *) ptr; }
};

template <typename T> hipError_t upload(DeviceBuffer &buf, const std::vector<T> &v) {
    size_t n = std::max<size_t>(v.size() * sizeof(T), 16);
    hipError_t e = buf.ensure(n);
    if (e != hipSuccess) return e;
    if (!v.empty()) e = hipMemcpy(buf.ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

inline unsigned blocksFor(size_t n, unsigned bs) { return (unsigned) ((n + bs - 1) / bs); }

}  // namespace

// ---- static device scratch of mtsamd_trace_device_static -----------------------------------------------------------------
// Callers that may not allocate device memory (tight serving loops, harnesses that forbid hipMalloc) get a closest-hit query whose work
// memory -- the ray hand-out counters and the raw (t, u, v, triangle) hits that k_hits_to_records turns into hit
// records -- lives in this array instead of buffers from hipMalloc. Every GPU holds its own copy at its own address;
// host code never uses the array's name as an address, only the pointer hipGetSymbolAddress returns for the device
// that is current (g_scratchPtr, filled once per device).
static constexpr uint64_t kStaticMaxRays = 1ull << 20;
static constexpr size_t kStaticHeadsBytes = 768;         // 8 counters 64 bytes apart, rounded up to the array's alignment
static_assert(kFetchHeads * kFetchHeadStride * sizeof(uint32_t) <= kStaticHeadsBytes && kStaticHeadsBytes % 256 == 0,
              "the hand-out counters that launchTrace clears must end before the raw hits of the static array begin");
static constexpr size_t kStaticScratchBytes = kStaticHeadsBytes + kStaticMaxRays * 16;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
unsigned char *g_scratchPtr[kMaxDevices] = {};
std::mutex g_scratchMutex;

// The address of `scratch` on the current device (looked up once per device, with that device current).
int staticScratch(unsigned char **out) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(MTSAMD_EDEVICE, "device index %d beyond the %d this library keeps scratch pointers for", dev, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if (!g_scratchPtr[dev]) {
        void *p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(MTSAMD_EDEVICE, "hipGetSymbolAddress returned a null address for the static scratch array");
        g_scratchPtr[dev] = (unsigned char *) p;
    }
    *out = g_scratchPtr[dev];
    return MTSAMD_OK;
}
}  // namespace

struct mtsamd_scene {
    HostScene host;
    DevScene dev;
    int device = 0;
    DeviceBuffer nodes, tris, shadeTris, shapes, materials, emitters, deltaEmitters,
        emitterCdf, areaCdf, rtrans, textures, materialTextures, materialAlphaTextures, plainTextures, mipLevels, texels, ewaLut, envCdfRows, envCdfCols, envRowWeights, devSelf;
    // work buffers of the wavefront loop
    DeviceBuffer fetchHeads, qbuf[2][6], hits, sbuf[3], sampleL, samplePos, shadowFlags, counters, scratch[8];
    size_t capacity = 0;
    hipStream_t stream = nullptr;
    hipStream_t shadowStream = nullptr;   // shadow rays of bounce k are traced next to the extension rays of bounce k+1
    hipEvent_t evShaded = nullptr, evShadowed = nullptr;
    uint32_t *hostCounters = nullptr;     // pinned
    unsigned long long *hostTail = nullptr;   // pinned: ray counts of the kernel that finishes a pass
    std::atomic<int> cancel{ 0 };
    bool countTraversal = false;
    // traversal launch shape: persistent workgroups (as many as fit on the device) and the lane-refill / triangle-phase
    // thresholds of the traversal loop
    unsigned traceGrid = 0, traceBlock = 256;
    int traceStack = 0;                   // StackLayout of the traversal kernels: 0, 8 or 10 (by the depth of the tree)
    uint32_t nodeStride = 5, triStride = 3;     // record strides in float4 units (padding buys line alignment)
    uint64_t overlapBelow = 8u << 20;     // rays (shadow + next extend) below which the two traversals run side by side
    float2 queryUv = make_float2(0.0f, 0.0f);     // where mtsamd_bsdf_eval / pdf / sample evaluate textures
    bool roughDielectric = false;         // some material is a rough dielectric (selects the kernels that know it)
    uint32_t xcdRanges = 8;               // ranges of rays with a hand-out counter each (8 = one per XCD, 1 = one for all)
    uint32_t tailBelow = 1u << 18;        // paths below which one launch finishes the pass (0 = never)
    int shadeRounds = 16;                 // sweeps per shade workgroup: 16, or 1 when set below 2 ...
    uint32_t shadeRoundsAbove = 4u << 20; // ... for queues of at least this many paths
    uint32_t refillBelow = 56, triShare = 16, chunkMax = 512;

    ~mtsamd_scene() {
        if (stream) (void) hipStreamDestroy(stream);
        if (shadowStream) (void) hipStreamDestroy(shadowStream);
        if (evShaded) (void) hipEventDestroy(evShaded);
        if (evShadowed) (void) hipEventDestroy(evShadowed);
        if (hostCounters) (void) hipHostFree(hostCounters);
        if (hostTail) (void) hipHostFree(hostTail);
    }

    PathQueue queue(int i) const {
        PathQueue q;
        q.rayO = qbuf[i][0].as<float4>(); q.rayD = qbuf[i][1].as<float4>(); q.thr = qbuf[i][2].as<float4>();
        q.misc = qbuf[i][3].as<uint4>(); q.bsdfPdf = qbuf[i][4].as<float>();
        q.refCos = (host.extEmitters || host.mixture || host.desc.integrator.type != MTSAMD_INTEGRATOR_PATH) ? qbuf[i][5].as<float>() : nullptr;
        return q;
    }
    ShadowQueue shadowQueue() const {
        ShadowQueue s;
        s.rayO = sbuf[0].as<float4>(); s.rayD = sbuf[1].as<float4>(); s.contrib = sbuf[2].as<float4>();
        return s;
    }
    // entries of the ray queues per camera sample: 1 for `path`, the larger sample count for `direct`
    size_t raysPerSample() const {
        const mtsamd_integrator &ig = host.desc.integrator;
        return ig.type == MTSAMD_INTEGRATOR_DIRECT ? (size_t) std::max(1, std::max(ig.emitter_samples, ig.bsdf_samples)) : 1;
    }
    int ensureWork(size_t nSamples) {
        const size_t n = nSamples * raysPerSample();
        if (n <= capacity) return MTSAMD_OK;
        // a failed growth leaves some buffers released: forget the old capacity so that the next call rebuilds all of them
        capacity = 0;
        for (int i = 0; i < 2; ++i) {
            for (int j = 0; j < 4; ++j) HIP_TRY(qbuf[i][j].ensure(n * 16));
            HIP_TRY(qbuf[i][4].ensure(n * 4));
            if (host.extEmitters || host.mixture || host.desc.integrator.type != MTSAMD_INTEGRATOR_PATH) HIP_TRY(qbuf[i][5].ensure(n * 4));
        }
        HIP_TRY(hits.ensure(n * 16));
        for (int j = 0; j < 3; ++j) HIP_TRY(sbuf[j].ensure(n * 16));
        HIP_TRY(sampleL.ensure(n * 16));
        HIP_TRY(samplePos.ensure(n * 8));
        HIP_TRY(shadowFlags.ensure(n));
        capacity = n;
        return MTSAMD_OK;
    }
};

extern "C" const char *mtsamd_last_error(void) { return g_error.c_str(); }

extern "C" int mtsamd_init(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) return fail(MTSAMD_EDEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(MTSAMD_EINVAL, "device %d out of range (%d devices)", device, count);
    HIP_TRY(hipSetDevice(device));
    unsigned char *p = nullptr;
    return staticScratch(&p);       // the static scratch array's address on this device, looked up once
}

extern "C" int mtsamd_scene_create(const mtsamd_scene_desc *desc, mtsamd_scene **out) {
    if (!out) return fail(MTSAMD_EINVAL, "null output handle");
    *out = nullptr;
    mtsamd_scene *s = new (std::nothrow) mtsamd_scene();
    if (!s) return fail(MTSAMD_EDEVICE, "out of host memory");
    std::unique_ptr<mtsamd_scene> owner(s);     // every early return below releases the scene and what it holds
    std::string err;
    try {
        err = prepareScene(desc, s->host);
    } catch (const std::exception &ex) {
        err = std::string("scene preparation failed: ") + ex.what();
    }
    if (!err.empty()) { return fail(MTSAMD_EINVAL, "%s", err.c_str()); }

    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { return fail(MTSAMD_EDEVICE, "no HIP device available"); }
    (void) hipGetDevice(&s->device);

    HostScene &h = s->host;
    for (const DevMaterial &m : h.materials) s->roughDielectric |= m.type == MTSAMD_BSDF_ROUGHDIELECTRIC;
    // kernel A/B runs: make a scene run a more general variant of the shading kernels than it needs
    if (const char *e = getenv("MTSAMD_SHADE_VARIANT")) {
        if (strstr(e, "env")) s->host.extEmitters = true;
        if (strstr(e, "rd")) s->roughDielectric = true;
        if (strstr(e, "mix")) s->host.mixture = s->host.blend = true;
        if (strstr(e, "tex")) s->host.mixture = true;       // the kernels with textures and frame adapters but no mixtures
    }
    if (const char *e = getenv("MTSAMD_NODE_STRIDE")) s->nodeStride = (uint32_t) std::min(16, std::max(5, atoi(e)));
    if (const char *e = getenv("MTSAMD_TRI_STRIDE")) s->triStride = (uint32_t) std::min(16, std::max(3, atoi(e)));
    // nodes and triangle records as raw float4 streams, each record padded to its stride
    auto uploadHostBvh = [&]() -> int {
        auto padded = [&](DeviceBuffer &buf, const void *src, size_t count, size_t recBytes, size_t strideBytes) -> int {
            std::vector<char> tmp(std::max<size_t>(count, 1) * strideBytes, 0);
            for (size_t i = 0; i < count; ++i) memcpy(&tmp[i * strideBytes], (const char *) src + i * recBytes, recBytes);
            HIP_TRY(buf.ensure(tmp.size()));
            HIP_TRY(hipMemcpy(buf.ptr, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
            return MTSAMD_OK;
        };
        int rc2 = padded(s->nodes, h.bvh.nodes.data(), h.bvh.nodes.size(), sizeof(WideNode), (size_t) s->nodeStride * 16);
        if (rc2 != MTSAMD_OK) return rc2;
        return padded(s->tris, h.bvh.tris.data(), h.bvh.tris.size(), sizeof(TriRecord), (size_t) s->triStride * 16);
    };
    auto up = [&]() -> int {
        if (!(h.bvh.builder == BVH_BUILDER_DEVICE && h.desc.n_triangles > 0)) {
            int rc2 = uploadHostBvh();
            if (rc2 != MTSAMD_OK) return rc2;
        }
        HIP_TRY(upload(s->shadeTris, h.shadeTris));
        HIP_TRY(upload(s->shapes, h.shapes));
        HIP_TRY(upload(s->materials, h.materials));
        HIP_TRY(upload(s->emitters, h.emitters));
        HIP_TRY(upload(s->deltaEmitters, h.deltaEmitters));
        HIP_TRY(upload(s->emitterCdf, h.emitterCdf));
        HIP_TRY(upload(s->areaCdf, h.areaCdf));
        if (!h.textures.empty()) HIP_TRY(upload(s->textures, h.textures));
        if (!h.materialTextures.empty()) HIP_TRY(upload(s->materialTextures, h.materialTextures));
        if (!h.materialAlphaTextures.empty()) HIP_TRY(upload(s->materialAlphaTextures, h.materialAlphaTextures));
        if (!h.plainTextures.empty()) HIP_TRY(upload(s->plainTextures, h.plainTextures));
        if (!h.mipLevels.empty()) {
            HIP_TRY(upload(s->mipLevels, h.mipLevels));
            HIP_TRY(upload(s->texels, h.texels));
            std::vector<float> lut(64);
            ewaWeightTable(lut.data());
            HIP_TRY(upload(s->ewaLut, lut));
        }
        if (!h.envCdfRows.empty()) {
            HIP_TRY(upload(s->envCdfRows, h.envCdfRows)); HIP_TRY(upload(s->envCdfCols, h.envCdfCols)); HIP_TRY(upload(s->envRowWeights, h.envRowWeights));
        }
        HIP_TRY(upload(s->rtrans, h.rtrans));
        if (h.bvh.builder == BVH_BUILDER_DEVICE && h.desc.n_triangles > 0) {
            // the acceleration structure is built where it will be used; the records then learn their BSDF types
            void *dn = nullptr, *dt = nullptr;
            const char *err = buildWideDevice(h.positions.data(), h.desc.n_vertices, h.indices.data(), h.desc.n_triangles, s->nodeStride, s->triStride, &dn, &dt, h.bvh);
            if (err && strcmp(err, kTooDeep) == 0) {
                // geometry whose clustering gives a tree deeper than the traversal stack: the host builder bounds the depth
                err = buildBVH(h.positions.data(), h.desc.n_vertices, h.indices.data(), h.desc.n_triangles, h.bvh, BVH_BUILDER_HOST);
                if (err && *err) return fail(MTSAMD_EINVAL, "%s", err);
                int rc2 = uploadHostBvh();
                if (rc2 != MTSAMD_OK) return rc2;
            } else {
                if (err && *err) return fail(MTSAMD_EDEVICE, "%s", err);
                s->nodes.ptr = dn; s->nodes.bytes = (size_t) h.bvh.nodeCount * s->nodeStride * 16;
                s->tris.ptr = dt; s->tris.bytes = (size_t) h.bvh.triCount * s->triStride * 16;
            }
            hipLaunchKernelGGL(k_tag_triangle_records, dim3(blocksFor(h.bvh.triCount, 256)), dim3(256), 0, 0, s->tris.as<float4>(), s->triStride, h.bvh.triCount,
                               s->shadeTris.as<float4>(), s->materials.as<DevMaterial>());
            HIP_TRY(hipDeviceSynchronize());
        }
        HIP_TRY(s->counters.ensure(CNT_WORDS * sizeof(uint32_t)));
        HIP_TRY(s->fetchHeads.ensure(2 * kFetchHeads * kFetchHeadStride * sizeof(uint32_t)));
        HIP_TRY(hipMemset(s->counters.ptr, 0, CNT_WORDS * sizeof(uint32_t)));
        HIP_TRY(hipStreamCreate(&s->stream));
        HIP_TRY(hipStreamCreate(&s->shadowStream));
        HIP_TRY(hipEventCreateWithFlags(&s->evShaded, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s->evShadowed, hipEventDisableTiming));
        HIP_TRY(hipHostMalloc((void **) &s->hostCounters, CNT_WORDS * sizeof(uint32_t)));
        HIP_TRY(hipHostMalloc((void **) &s->hostTail, 2 * sizeof(unsigned long long)));
        return MTSAMD_OK;
    };
    int rc = up();
    if (rc != MTSAMD_OK) { return rc; }

    DevScene d = h.params;
    d.nodes = s->nodes.as<float4>(); d.tris = s->tris.as<float4>();
    d.nodeStride = s->nodeStride; d.triStride = s->triStride;
    d.shadeTris = s->shadeTris.as<float4>();
    d.shapes = s->shapes.as<DevShape>(); d.materials = s->materials.as<DevMaterial>(); d.emitters = s->emitters.as<DevEmitter>();
    d.deltaEmitters = s->deltaEmitters.as<DevDeltaEmitter>();
    d.emitterCdf = s->emitterCdf.as<float>(); d.areaCdf = s->areaCdf.as<float>(); d.rtrans = s->rtrans.as<float>();
    d.textures = h.textures.empty() ? nullptr : s->textures.as<DevTexture>();
    d.materialTextures = h.materialTextures.empty() ? nullptr : s->materialTextures.as<uint4>();
    d.materialAlphaTextures = h.materialAlphaTextures.empty() ? nullptr : s->materialAlphaTextures.as<uint2>();
    d.plainTextures = h.plainTextures.empty() ? nullptr : s->plainTextures.as<DevTexture>();
    d.mipLevels = h.mipLevels.empty() ? nullptr : s->mipLevels.as<DevMipLevel>();
    d.texels = h.mipLevels.empty() ? nullptr : s->texels.as<uint2>();
    d.ewaLut = h.mipLevels.empty() ? nullptr : s->ewaLut.as<float>();
    d.envCdfRows = h.envCdfRows.empty() ? nullptr : s->envCdfRows.as<float>();
    d.envCdfCols = h.envCdfRows.empty() ? nullptr : s->envCdfCols.as<float>();
    d.envRowWeights = h.envRowWeights.empty() ? nullptr : s->envRowWeights.as<float>();
    if (s->devSelf.ensure(sizeof(DevScene)) != hipSuccess) { return fail(MTSAMD_EDEVICE, "out of device memory"); }
    d.self = s->devSelf.as<DevScene>();
    if (hipMemcpy(s->devSelf.ptr, &d, sizeof(DevScene), hipMemcpyHostToDevice) != hipSuccess) { return fail(MTSAMD_EDEVICE, "upload of the scene record failed"); }
    s->dev = d;
    const char *ct = getenv("MTSAMD_COUNT_TRAVERSAL");
    s->countTraversal = ct && *ct == '1';
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, s->device));
        if (const char *e = getenv("MTSAMD_REFILL_BELOW")) s->refillBelow = (uint32_t) std::min(64, std::max(1, atoi(e)));
        // The triangle phase runs once this many 64ths of a wave's busy lanes wait for it. Small scenes spend more of a
        // ray on triangle tests than on nodes and gain from waiting for fuller phases (veach_bidir, 3 392 triangles:
        // 8 040 -> 8 360 Mrays/s at 24, Cornell box +2.5 %, a 21 000-triangle field +0.4 %); from about 40 000
        // triangles on 16 is best (a quarter of a million: 24 costs 0.7 %, 32 costs 4 %).
        s->triShare = h.desc.n_triangles <= 32768u ? 24u : 16u;
        if (const char *e = getenv("MTSAMD_TRI_SHARE")) s->triShare = (uint32_t) std::min(64, std::max(1, atoi(e)));
        if (const char *e = getenv("MTSAMD_OVERLAP_BELOW")) s->overlapBelow = (uint64_t) std::max(0ll, atoll(e));
        // One hand-out range per XCD keeps each XCD's L2 on its own part of the queue: +0.3 to 1 % where the structure is
        // small (58 MB: 1M triangles), -3.5 % on a 10M-triangle scene (575 MB), where eight separate working sets in
        // flight evict each other from the memory-side cache. Hence: per-XCD ranges up to 128 MB of nodes and triangles.
        s->xcdRanges = s->nodes.bytes + s->tris.bytes <= ((size_t) 128 << 20) ? 8 : 1;
        if (const char *e = getenv("MTSAMD_XCD_RANGES")) s->xcdRanges = (uint32_t) std::min(8, std::max(1, atoi(e)));
        if (const char *e = getenv("MTSAMD_TAIL_BELOW")) s->tailBelow = (uint32_t) std::max(0ll, atoll(e));
        // on / off only: the library holds the 16-sweep and the 1-sweep build of the shading kernels (any value >= 2 means 16)
        if (const char *e = getenv("MTSAMD_SHADE_ROUNDS")) s->shadeRounds = atoi(e) >= 2 ? 16 : 1;
        if (const char *e = getenv("MTSAMD_SHADE_ROUNDS_ABOVE")) s->shadeRoundsAbove = (uint32_t) std::max(0ll, atoll(e));
        if (const char *e = getenv("MTSAMD_CHUNK_MAX")) s->chunkMax = (uint32_t) std::min(65536, std::max(64, atoi(e)));
        if (const char *e = getenv("MTSAMD_TRACE_BLOCK")) s->traceBlock = (unsigned) std::min(1024, std::max(64, atoi(e) / 64 * 64));
        // Traversal stack layout by the depth of the tree that was built: all of it in LDS where that leaves five
        // workgroups per CU (8 entries: depth <= 10; 10 entries: depth <= 12), else 4 entries in LDS and the rest in scratch.
        const int need = std::max(1, (int) h.bvh.depth - 2);
        s->traceStack = need <= 8 ? 8 : need <= 10 ? 10 : 0;
        if (const char *e = getenv("MTSAMD_TRACE_STACK")) { const int v = atoi(e); if (v == 0 || ((v == 8 || v == 10) && need <= v)) s->traceStack = v; }
        const size_t lds = traceLdsBytes(s->traceBlock, s->traceStack ? s->traceStack : kLdsStack);
        const void *fns[] = { (const void *) k_trace<TRACE_CLOSEST, false, 0>, (const void *) k_trace<TRACE_CLOSEST, true, 0>,
                              (const void *) k_trace<TRACE_OCCLUDED, false, 0>, (const void *) k_trace<TRACE_OCCLUDED, true, 0>,
                              (const void *) k_trace<TRACE_CLOSEST, false, 8>, (const void *) k_trace<TRACE_CLOSEST, true, 8>,
                              (const void *) k_trace<TRACE_OCCLUDED, false, 8>, (const void *) k_trace<TRACE_OCCLUDED, true, 8>,
                              (const void *) k_trace<TRACE_CLOSEST, false, 10>, (const void *) k_trace<TRACE_CLOSEST, true, 10>,
                              (const void *) k_trace<TRACE_OCCLUDED, false, 10>, (const void *) k_trace<TRACE_OCCLUDED, true, 10> };
        if (lds > (size_t) prop.sharedMemPerBlock)
            for (const void *f : fns) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        int perCu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, fns[s->traceStack == 8 ? 4 : s->traceStack == 10 ? 8 : 0], (int) s->traceBlock, lds));
        s->traceGrid = (unsigned) std::max(1, perCu) * (unsigned) std::max(1, prop.multiProcessorCount);
        if (getenv("MTSAMD_VERBOSE")) fprintf(stderr, "mtsamd: trace block %u, %d blocks/CU, %zu B LDS, stack layout %d (tree depth %u)\n", s->traceBlock, perCu, lds, s->traceStack, h.bvh.depth);
    }
    *out = owner.release();
    return MTSAMD_OK;
}

extern "C" void mtsamd_scene_destroy(mtsamd_scene *scene) { delete scene; }

extern "C" int mtsamd_scene_bvh_info(const mtsamd_scene *scene, mtsamd_bvh_info *info) {
    if (!scene || !info) return fail(MTSAMD_EINVAL, "null argument");
    const BVH &b = scene->host.bvh;
    info->n_nodes = b.nodeCount; info->n_triangles = b.triCount; info->depth = b.depth;
    info->node_bytes = (uint64_t) b.nodeCount * sizeof(WideNode); info->triangle_bytes = (uint64_t) b.triCount * sizeof(TriRecord);
    info->build_seconds = b.buildSeconds; info->sah_cost = b.sahCost;
    info->binary_seconds = b.binarySeconds; info->builder = b.builder; info->reserved = 0;
    return MTSAMD_OK;
}

extern "C" int mtsamd_bvh_host_info(const float *positions, uint32_t n_vertices, const uint32_t *indices, uint32_t n_triangles, mtsamd_bvh_info *info) {
    if (!info || (n_triangles && (!positions || !indices))) return fail(MTSAMD_EINVAL, "null argument");
    BVH b;
    const char *err = "";
    try {
        err = buildBVH(positions, n_vertices, indices, n_triangles, b, BVH_BUILDER_HOST);
    } catch (const std::exception &ex) {
        return fail(MTSAMD_EINVAL, "BVH build failed: %s", ex.what());
    }
    if (err && *err) return fail(MTSAMD_EINVAL, "%s", err);
    info->n_nodes = b.nodeCount; info->n_triangles = b.triCount; info->depth = b.depth;
    info->node_bytes = (uint64_t) b.nodeCount * sizeof(WideNode); info->triangle_bytes = (uint64_t) b.triCount * sizeof(TriRecord);
    info->build_seconds = b.buildSeconds; info->sah_cost = b.sahCost;
    info->binary_seconds = b.binarySeconds; info->builder = b.builder; info->reserved = 0;
    return MTSAMD_OK;
}

extern "C" int mtsamd_bvh_max_depth(void) { return kMaxWideDepth; }

extern "C" void mtsamd_cancel(mtsamd_scene *scene) { if (scene) scene->cancel.store(1); }

extern "C" int mtsamd_scene_set_counting(mtsamd_scene *scene, int enabled) {
    if (!scene) return fail(MTSAMD_EINVAL, "null scene");
    scene->countTraversal = enabled != 0;
    return MTSAMD_OK;
}

namespace {

struct EventTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool[6];
    bool enabled = false;
    hipStream_t stream = nullptr;
    void begin(int cls, hipStream_t on = nullptr) {
        if (!enabled) return;
        hipEvent_t a, b;
        (void) hipEventCreate(&a); (void) hipEventCreate(&b);
        (void) hipEventRecord(a, on ? on : stream);
        pool[cls].push_back({ a, b });
    }
    void end(int cls, hipStream_t on = nullptr) {
        if (!enabled) return;
        (void) hipEventRecord(pool[cls].back().second, on ? on : stream);
    }
    double total(int cls) {
        double ms = 0;
        for (auto &p : pool[cls]) {
            float t = 0;
            if (hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) ms += t;
            (void) hipEventDestroy(p.first); (void) hipEventDestroy(p.second);
        }
        pool[cls].clear();
        return ms;
    }
};
enum { T_RAYGEN = 0, T_EXTEND = 1, T_SHADE = 2, T_SHADOW = 3, T_FILM = 4, T_TAIL = 5 };

int checkOpts(const mtsamd_scene *s, const mtsamd_render_opts *o, mtsamd_render_opts &r) {
    if (!s || !o) return fail(MTSAMD_EINVAL, "null argument");
    r = *o;
    const int W = s->dev.cropW, H = s->dev.cropH;
    if (r.x0 == 0 && r.y0 == 0 && r.x1 == 0 && r.y1 == 0) { r.x1 = W; r.y1 = H; }
    if (r.x0 < 0 || r.y0 < 0 || r.x1 > W || r.y1 > H || r.x0 >= r.x1 || r.y0 >= r.y1) return fail(MTSAMD_EINVAL, "invalid pixel rectangle");
    if (r.spp <= 0) return fail(MTSAMD_EINVAL, "spp must be positive");
    if (r.spp_total <= 0) r.spp_total = r.sample_offset + r.spp;
    if (r.sample_offset < 0 || r.sample_offset + r.spp > r.spp_total) return fail(MTSAMD_EINVAL, "sample_offset + spp exceeds spp_total");
    if (r.band_rows < 0 || (r.band_rows > 0 && r.band_period < r.band_rows)) return fail(MTSAMD_EINVAL, "band_period must be at least band_rows");
    if (r.band_rows == 0) r.band_period = 0;
    return MTSAMD_OK;
}

// Rows of the rectangle [y0, y1) that belong to the interleaved bands starting at y0 (all of them without bands)
int bandedRows(int y0, int y1, int bandRows, int bandPeriod) {
    if (bandRows <= 0) return y1 - y0;
    const int span = y1 - y0, full = span / bandPeriod, rest = span - full * bandPeriod;
    return full * bandRows + std::min(rest, bandRows);
}

// Launches the persistent traversal kernel over n rays. The hand-out counter must be zero when the kernel starts.
template <int MODE>
void launchTrace(mtsamd_scene *s, TraceArgs a, hipStream_t st, uint32_t *heads = nullptr) {
    a.refillBelow = s->refillBelow; a.triShare = s->triShare; a.blockThreads = s->traceBlock;
    const unsigned grid = std::min(s->traceGrid, blocksFor(a.n, s->traceBlock));
    // chunk of rays a wave takes per atomic: large enough to make the counter cheap, small enough to balance the tail
    const uint64_t waves = (uint64_t) grid * (s->traceBlock / 64);
    a.chunk = (uint32_t) std::min<uint64_t>(s->chunkMax, std::max<uint64_t>(64, (a.n / (waves * 8)) / 64 * 64));
    // hand-out counters: one range of rays per XCD (MTSAMD_XCD_RANGES=1: a single range)
    a.segments = s->xcdRanges;
    a.segmentLen = (uint32_t) (((uint64_t) a.n + a.segments - 1) / a.segments + 63) / 64 * 64;
    a.fetchHeads = heads ? heads : s->fetchHeads.as<uint32_t>() + (MODE == TRACE_CLOSEST ? 0 : kFetchHeads * kFetchHeadStride);
    (void) hipMemsetAsync(a.fetchHeads, 0, kFetchHeads * kFetchHeadStride * sizeof(uint32_t), st);
    const size_t lds = traceLdsBytes(s->traceBlock, s->traceStack ? s->traceStack : kLdsStack);
#define MTS_LAUNCH_TRACE(STACK) do { \
        if (s->countTraversal) hipLaunchKernelGGL((k_trace<MODE, true, STACK>), dim3(grid), dim3(s->traceBlock), lds, st, s->dev, a); \
        else hipLaunchKernelGGL((k_trace<MODE, false, STACK>), dim3(grid), dim3(s->traceBlock), lds, st, s->dev, a); } while (0)
    if (s->traceStack == 8) MTS_LAUNCH_TRACE(8);
    else if (s->traceStack == 10) MTS_LAUNCH_TRACE(10);
    else MTS_LAUNCH_TRACE(0);
#undef MTS_LAUNCH_TRACE
}

// One pass: `pp.spp` samples for every pixel of the rectangle. When liOut is non-null the per-sample radiance is
// copied out instead of being filtered into a film.
int renderPass(mtsamd_scene *s, const PassParams &pp, float *filmDev, float *liOut, float *liPos, size_t liOffset,
               mtsamd_stats *stats, EventTimer &timer) {
    const DevScene &sc = s->dev;
    const size_t nPaths = (size_t) pp.w * pp.h * pp.spp;
    if (nPaths * s->raysPerSample() >= (1ull << 31)) return fail(MTSAMD_EINVAL, "pass too large");
    int rc = s->ensureWork(nPaths);
    if (rc != MTSAMD_OK) return rc;
    hipStream_t st = s->stream;
    uint32_t *cnt = s->counters.as<uint32_t>();
    unsigned long long *cnt64 = s->counters.as<unsigned long long>();
    float4 *sampleL = s->sampleL.as<float4>();
    float2 *samplePos = s->samplePos.as<float2>();
    float4 *hits = s->hits.as<float4>();
    ShadowQueue sq = s->shadowQueue();

    uint32_t n = (uint32_t) nPaths;
    int cur = 0;
    HIP_TRY(hipMemsetAsync(cnt, 0, CNT_QUEUE_WORDS * sizeof(uint32_t), st));
    timer.begin(T_RAYGEN);
    hipLaunchKernelGGL(k_raygen, dim3(blocksFor(n, 256)), dim3(256), 0, st, sc, pp, s->queue(cur), sampleL, samplePos, n);
    timer.end(T_RAYGEN);

    uint64_t normalRays = 0, shadowRays = 0, iterations = 0;
    bool tailLaunch = false;
    bool shadowPending = false;
    const bool traceIters = getenv("MTSAMD_TRACE_ITERS") != nullptr;
    const mtsamd_integrator &ig = s->host.desc.integrator;
    const bool simple = ig.type == MTSAMD_INTEGRATOR_VOLPATH_SIMPLE;       // its own estimator, in the plain or the every-feature kernels
    if (ig.type == MTSAMD_INTEGRATOR_DIRECT) {
        // camera rays -> shade the camera vertex into nEmitter shadow rays and nBsdf BSDF-sampled rays per sample -> both
        // traversals -> gather. The queues keep a fixed place for every ray of a sample.
        const int nE = ig.emitter_samples, nB = ig.bsdf_samples;
        PathQueue qin = s->queue(0), qout = s->queue(1);
        TraceArgs ta;
        memset(&ta, 0, sizeof(ta));
        ta.rayO = qin.rayO; ta.rayD = qin.rayD; ta.stride = 1; ta.n = n;
        ta.hits = hits; ta.stats = cnt64 + CNT64_EXT_NODES;
        timer.begin(T_EXTEND);
        launchTrace<TRACE_CLOSEST>(s, ta, st);
        timer.end(T_EXTEND);
        unsigned long long *counts = cnt64 + CNT64_TAIL_NORMAL;
        HIP_TRY(hipMemsetAsync(counts, 0, 2 * sizeof(unsigned long long), st));
        timer.begin(T_SHADE);
#define MTS_LAUNCH_DIRECT(...) hipLaunchKernelGGL((k_direct_shade<__VA_ARGS__>), dim3(blocksFor(n, kShadeBlock * kDirectRounds)), dim3(kShadeBlock), 0, st, sc, pp, qin, hits, qout, sq, sampleL, samplePos, counts, n, nE, nB)
        if (s->host.mixture && !s->host.blend) { if (s->host.extEmitters || s->roughDielectric) MTS_LAUNCH_DIRECT(true, true, true, false); else MTS_LAUNCH_DIRECT(false, false, true, false); }
        else if (s->host.mixture) { if (s->host.extEmitters || s->roughDielectric) MTS_LAUNCH_DIRECT(true, true, true); else MTS_LAUNCH_DIRECT(false, false, true); }
        else if (s->host.extEmitters) { if (s->roughDielectric) MTS_LAUNCH_DIRECT(true, true, false); else MTS_LAUNCH_DIRECT(true, false, false); }
        else { if (s->roughDielectric) MTS_LAUNCH_DIRECT(false, true, false); else MTS_LAUNCH_DIRECT(false, false, false); }
#undef MTS_LAUNCH_DIRECT
        timer.end(T_SHADE);
        if (nE > 0) {
            TraceArgs sa;
            memset(&sa, 0, sizeof(sa));
            sa.rayO = sq.rayO; sa.rayD = sq.rayD; sa.stride = 1; sa.n = n * (uint32_t) nE;
            sa.occluded = s->shadowFlags.as<uint8_t>(); sa.stats = cnt64 + CNT64_SHD_NODES;
            timer.begin(T_SHADOW);
            launchTrace<TRACE_OCCLUDED>(s, sa, st);
            timer.end(T_SHADOW);
            if (stats) stats->shadow_launches++;
        }
        if (nB > 0) {
            ta.rayO = qout.rayO; ta.rayD = qout.rayD; ta.n = n * (uint32_t) nB;
            timer.begin(T_EXTEND);
            launchTrace<TRACE_CLOSEST>(s, ta, st);
            timer.end(T_EXTEND);
        }
        timer.begin(T_SHADE);
        hipLaunchKernelGGL(k_direct_gather, dim3(blocksFor(n, 256)), dim3(256), 0, st, sc, qin, qout, hits, sq, s->shadowFlags.as<uint8_t>(), sampleL, n, nE, nB);
        timer.end(T_SHADE);
        HIP_TRY(hipMemcpyAsync(s->hostTail, counts, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        normalRays += n + s->hostTail[0]; shadowRays += s->hostTail[1];
        iterations = 2;
        n = 0;
    }
    while (n > 0) {
        // Few paths left: one launch takes them through all their remaining bounces. (The tail of a pass is a
        // long run of bounces on a handful of paths inside dielectrics; three launches and a host round trip for
        // each of them cost far more than the rays.) Counting runs keep to the wavefront kernels, which count.
        if (n <= s->tailBelow && !s->countTraversal) {
            if (shadowPending) { HIP_TRY(hipStreamWaitEvent(st, s->evShadowed, 0)); shadowPending = false; }
            unsigned long long *tailCounts = cnt64 + CNT64_TAIL_NORMAL;
            HIP_TRY(hipMemsetAsync(tailCounts, 0, 2 * sizeof(unsigned long long), st));
            timer.begin(T_TAIL);
#define MTS_LAUNCH_TAIL(...) hipLaunchKernelGGL((k_tail<__VA_ARGS__>), dim3(blocksFor(n, kTailBlock)), dim3(kTailBlock), 0, st, sc, pp, s->queue(cur), sampleL, samplePos, n, s->triShare, tailCounts)
            // scenes with mixtures, textures or frame adapters run the variants that know them (with or without the other emitter kinds and the rough dielectric)
            if (simple) { if (s->host.mixture || s->host.extEmitters || s->roughDielectric) MTS_LAUNCH_TAIL(true, true, true, true); else MTS_LAUNCH_TAIL(false, false, false, true); }
            else if (s->host.mixture && !s->host.blend) { if (s->host.extEmitters || s->roughDielectric) MTS_LAUNCH_TAIL(true, true, true, false, false); else MTS_LAUNCH_TAIL(false, false, true, false, false); }
            else if (s->host.mixture) { if (s->host.extEmitters || s->roughDielectric) MTS_LAUNCH_TAIL(true, true, true); else MTS_LAUNCH_TAIL(false, false, true); }
            else if (s->host.extEmitters) { if (s->roughDielectric) MTS_LAUNCH_TAIL(true, true, false); else MTS_LAUNCH_TAIL(true, false, false); }
            else { if (s->roughDielectric) MTS_LAUNCH_TAIL(false, true, false); else MTS_LAUNCH_TAIL(false, false, false); }
#undef MTS_LAUNCH_TAIL
            timer.end(T_TAIL);
            HIP_TRY(hipMemcpyAsync(s->hostTail, tailCounts, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            normalRays += s->hostTail[0]; shadowRays += s->hostTail[1];
            if (stats) { stats->tail_normal_rays += s->hostTail[0]; stats->tail_shadow_rays += s->hostTail[1]; }
            ++iterations;
            tailLaunch = true;
            break;
        }
        if (s->cancel.load()) { (void) hipStreamSynchronize(st); (void) hipStreamSynchronize(s->shadowStream); return fail(MTSAMD_ECANCELLED, "render cancelled"); }
        PathQueue qin = s->queue(cur), qout = s->queue(cur ^ 1);
        TraceArgs ta;
        memset(&ta, 0, sizeof(ta));
        ta.rayO = qin.rayO; ta.rayD = qin.rayD; ta.stride = 1; ta.n = n;
        ta.hits = hits; ta.stats = cnt64 + CNT64_EXT_NODES;
        timer.begin(T_EXTEND);
        launchTrace<TRACE_CLOSEST>(s, ta, st);
        timer.end(T_EXTEND);
        normalRays += n;
        // the previous bounce's shadow rays (other stream) must have landed in sampleL and released their counter
        if (shadowPending) { HIP_TRY(hipStreamWaitEvent(st, s->evShadowed, 0)); shadowPending = false; }
        // queue lengths and both ray hand-out counters start from zero for the rest of this bounce and the next extend
        HIP_TRY(hipMemsetAsync(cnt, 0, CNT_QUEUE_WORDS * sizeof(uint32_t), st));
        timer.begin(T_SHADE);
        // large queues: several sweeps per workgroup (purer wavefronts); small ones: one, to keep every CU busy
        const int rounds = n >= s->shadeRoundsAbove ? s->shadeRounds : 1;
#define MTS_LAUNCH_SHADE_AS(...) hipLaunchKernelGGL((k_shade<__VA_ARGS__>), dim3(blocksFor(n, kShadeBlock * rounds16)), dim3(kShadeBlock), 0, st, sc, pp, qin, hits, qout, sq, sampleL, samplePos, cnt, n)
#define MTS_LAUNCH_SHADE(R) do { \
            const int rounds16 = R; \
            if (simple) { if (s->host.mixture || s->host.extEmitters || s->roughDielectric) MTS_LAUNCH_SHADE_AS(R, true, true, true, true); else MTS_LAUNCH_SHADE_AS(R, false, false, false, true); } \
            else if (s->host.mixture && !s->host.blend) { if (s->host.extEmitters || s->roughDielectric) MTS_LAUNCH_SHADE_AS(R, true, true, true, false, false); else MTS_LAUNCH_SHADE_AS(R, false, false, true, false, false); } \
            else if (s->host.mixture) { if (s->host.extEmitters || s->roughDielectric) MTS_LAUNCH_SHADE_AS(R, true, true, true); else MTS_LAUNCH_SHADE_AS(R, false, false, true); } \
            else if (s->host.extEmitters) { if (s->roughDielectric) MTS_LAUNCH_SHADE_AS(R, true, true, false); else MTS_LAUNCH_SHADE_AS(R, true, false, false); } \
            else { if (s->roughDielectric) MTS_LAUNCH_SHADE_AS(R, false, true, false); else MTS_LAUNCH_SHADE_AS(R, false, false, false); } } while (0)
        // (sweeps per workgroup: 16 for large queues, 1 for small ones; MTSAMD_SHADE_ROUNDS=1 switches the sweeps off.
        // Measured on the 1M-triangle scene: 8 sweeps 117 ms per 256-spp step, 16 sweeps 114, 32 sweeps 120.)
        if (rounds >= 2) MTS_LAUNCH_SHADE(16);
        else MTS_LAUNCH_SHADE(1);
#undef MTS_LAUNCH_SHADE_AS
#undef MTS_LAUNCH_SHADE
        timer.end(T_SHADE);
        HIP_TRY(hipEventRecord(s->evShaded, st));
        HIP_TRY(hipMemcpyAsync(s->hostCounters, cnt, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const uint32_t nNext = s->hostCounters[CNT_NEXT], nShadow = s->hostCounters[CNT_SHADOW];
        if (nShadow) {
            TraceArgs sa;
            memset(&sa, 0, sizeof(sa));
            sa.rayO = sq.rayO; sa.rayD = sq.rayD; sa.stride = 1; sa.n = nShadow;
            sa.occluded = s->shadowFlags.as<uint8_t>(); sa.stats = cnt64 + CNT64_SHD_NODES;
            // Launches that cannot fill the device by themselves go to a second stream: the shadow rays only need this
            // bounce's shade and run beside the next bounce's extend, each hiding the other's ramp-up and drain.
            const bool beside = (uint64_t) nShadow + nNext < s->overlapBelow;
            hipStream_t sst = beside ? s->shadowStream : st;
            if (beside) HIP_TRY(hipStreamWaitEvent(sst, s->evShaded, 0));
            timer.begin(T_SHADOW, sst);
            // The traversal only stores one visibility byte per ray; the radiance of the unoccluded ones is added in a
            // streaming pass of its own. (Doing the read-modify-write inside the traversal loop stalls the whole
            // wavefront on dependent loads every time one lane finishes: 120 ms instead of 100 + 7 per 256-spp pass.)
            launchTrace<TRACE_OCCLUDED>(s, sa, sst);
            hipLaunchKernelGGL(k_shadow_accumulate, dim3(blocksFor(nShadow, 256)), dim3(256), 0, sst, nShadow, sa.occluded, sq.contrib, sampleL);
            timer.end(T_SHADOW, sst);
            if (beside) { HIP_TRY(hipEventRecord(s->evShadowed, sst)); shadowPending = true; }
            shadowRays += nShadow;
            if (stats) stats->shadow_launches++;
        }
        if (traceIters) fprintf(stderr, "iter %llu n=%u shadow=%u next=%u\n", (unsigned long long) iterations, n, nShadow, nNext);
        n = nNext;
        cur ^= 1;
        ++iterations;
    }
    if (shadowPending) { HIP_TRY(hipStreamWaitEvent(st, s->evShadowed, 0)); shadowPending = false; }
    if (liOut) {
        hipLaunchKernelGGL(k_copy_samples, dim3(blocksFor(nPaths, 256)), dim3(256), 0, st, sampleL, samplePos, (uint32_t) nPaths, liOut, liPos, liOffset);
    } else {
        // ImageBlock::put for every sample of the pass. Interleaved bands are filtered band by band: each is a rectangle
        // of its own whose samples lie together in the pass's buffers (launches on one stream: bands whose filter borders
        // meet add to the film one after the other).
        const int border = (int) std::ceil(sc.filterRadius - 0.5f);
        timer.begin(T_FILM);
        const int nBands = pp.bandRows ? (pp.h + pp.bandRows - 1) / pp.bandRows : 1;
        for (int b = 0; b < nBands; ++b) {
            PassParams bp = pp;
            bp.bandRows = bp.bandPeriod = 0; bp.bandMagic = 0;
            size_t first = 0;
            if (pp.bandRows) {
                bp.y0 = pp.y0 + b * pp.bandPeriod;
                bp.h = std::min(pp.bandRows, pp.h - b * pp.bandRows);
                first = (size_t) b * pp.bandRows * pp.w * pp.spp;
            }
            const int gx0 = std::max(bp.x0 - border, 0), gy0 = std::max(bp.y0 - border, 0);
            const int gx1 = std::min(bp.x0 + bp.w + border, sc.cropW), gy1 = std::min(bp.y0 + bp.h + border, sc.cropH);
            const size_t npx = (size_t) (gx1 - gx0) * (gy1 - gy0);
            const int rw = kFilmTile + 2 * border;
            if (rw * rw <= kFilmLdsEntries) {
                const int chunk = std::max(1, std::min(bp.spp, kFilmLdsEntries / (rw * rw)));
                const int tilesX = (gx1 - gx0 + kFilmTile - 1) / kFilmTile, tilesY = (gy1 - gy0 + kFilmTile - 1) / kFilmTile;
                hipLaunchKernelGGL(k_film_tiled, dim3((unsigned) (tilesX * tilesY)), dim3(256), 0, st, sc, bp, sampleL + first, samplePos + first, filmDev,
                                   border, gx0, gy0, gx1 - gx0, gy1 - gy0, tilesX, chunk);
            } else {    // very wide filters: the staged region would not fit in LDS
                hipLaunchKernelGGL(k_film, dim3(blocksFor(npx, 256)), dim3(256), 0, st, sc, bp, sampleL + first, samplePos + first, filmDev);
            }
        }
        timer.end(T_FILM);
    }
    HIP_TRY(hipGetLastError());
    if (stats) {
        stats->normal_rays += normalRays; stats->shadow_rays += shadowRays; stats->paths += nPaths;
        stats->bounce_iterations += iterations;
        stats->extend_launches += iterations - (tailLaunch ? 1 : 0);
    }
    return MTSAMD_OK;
}

int renderCommon(mtsamd_scene *s, const mtsamd_render_opts *opts, float *filmDev, float *liOutDev, float *liPosDev, mtsamd_stats *stats) {
    mtsamd_render_opts o;
    int rc = checkOpts(s, opts, o);
    if (rc != MTSAMD_OK) return rc;
    if (stats) memset(stats, 0, sizeof(*stats));
    s->cancel.store(0);
    HIP_TRY(hipSetDevice(s->device));
    const int w = o.x1 - o.x0, h = bandedRows(o.y0, o.y1, o.band_rows, o.band_period);
    if (liOutDev && o.band_rows) return fail(MTSAMD_EINVAL, "per-sample output takes a plain rectangle");
    int sppPass = o.spp_per_pass;
    if (sppPass <= 0) {
        // Paths in flight per pass: large passes keep the later, thinner bounces of the wavefront loop wide enough to
        // fill the device (measured on a 1M-triangle scene: 64 Mi paths 2.7, 128 Mi 3.0, 256 Mi 3.3, 512 Mi 3.6,
        // 768 Mi 3.6 Grays/s), so take up to 512 Mi paths as long as their work buffers (224 B per path) stay within
        // half of the free device memory.
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) freeB = (size_t) 16 << 30;
        const size_t byMemory = std::max<size_t>(freeB / 2 / 224, (size_t) 1 << 20);
        const size_t target = std::min<size_t>((size_t) 512 << 20, byMemory + s->capacity) / s->raysPerSample();
        sppPass = (int) std::max<size_t>(1, std::min<size_t>((size_t) o.spp, target / ((size_t) w * h)));
    }
    sppPass = std::min(sppPass, o.spp);
    if (liOutDev) sppPass = o.spp;      // per-sample output keeps one pixel-major layout
    EventTimer timer;
    timer.stream = s->stream;
    timer.enabled = stats != nullptr;
    if (s->countTraversal) HIP_TRY(hipMemsetAsync(s->counters.ptr, 0, CNT_WORDS * sizeof(uint32_t), s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    // The work buffers are allocated before the clock starts; when the device cannot hold the chosen pass size the pass
    // is halved until it fits (per-sample output needs the whole pass and fails instead).
    while ((rc = s->ensureWork((size_t) w * h * sppPass)) != MTSAMD_OK) {
        if (sppPass == 1 || liOutDev || o.spp_per_pass > 0) return rc;
        (void) hipGetLastError();
        sppPass = std::max(1, sppPass / 2);
    }
    auto t0 = std::chrono::steady_clock::now();
    for (int done = 0; done < o.spp; done += sppPass) {
        PassParams pp;
        pp.x0 = o.x0; pp.y0 = o.y0; pp.w = w; pp.h = h;
        pp.spp = std::min(sppPass, o.spp - done);
        pp.sppTotal = o.spp_total; pp.sampleBase = o.sample_offset + done; pp.seed = o.seed;
        pp.sppMagic = divisionMagic((uint32_t) pp.spp); pp.wMagic = divisionMagic((uint32_t) pp.w);
        pp.bandRows = o.band_rows; pp.bandPeriod = o.band_period; pp.bandMagic = divisionMagic((uint32_t) std::max(1, o.band_rows));
        rc = renderPass(s, pp, filmDev, liOutDev, liPosDev, 0, stats, timer);
        if (rc != MTSAMD_OK) return rc;
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (stats) {
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        stats->raygen_ms = timer.total(T_RAYGEN); stats->extend_ms = timer.total(T_EXTEND); stats->shade_ms = timer.total(T_SHADE);
        stats->shadow_ms = timer.total(T_SHADOW); stats->film_ms = timer.total(T_FILM); stats->tail_ms = timer.total(T_TAIL);
        if (s->countTraversal) {
            unsigned long long c[CNT_WORDS / 2];
            HIP_TRY(hipMemcpy(c, s->counters.ptr, sizeof(c), hipMemcpyDeviceToHost));
            stats->node_visits = c[CNT64_EXT_NODES]; stats->tri_tests = c[CNT64_EXT_TRIS];
            stats->shadow_node_visits = c[CNT64_SHD_NODES]; stats->shadow_tri_tests = c[CNT64_SHD_TRIS];
            if (getenv("MTSAMD_VERBOSE"))
                fprintf(stderr, "mtsamd: extend node visits %llu, of which no child hit %llu (after a hit was found: %llu)\n", c[CNT64_EXT_NODES], c[CNT64_EXT_EMPTY], c[CNT64_EXT_STALE]);
        }
    }
    return MTSAMD_OK;
}

}  // namespace

extern "C" int mtsamd_abi_version(void) { return MTSAMD_ABI_VERSION; }

extern "C" int mtsamd_bsdf_query_uv(mtsamd_scene *s, float u, float v) {
    if (!s) return fail(MTSAMD_EINVAL, "null scene");
    s->queryUv = make_float2(u, v);
    return MTSAMD_OK;
}

extern "C" int64_t mtsamd_sizeof(int record) {
    switch (record) {
        case MTSAMD_SIZEOF_SCENE_DESC: return (int64_t) sizeof(mtsamd_scene_desc);
        case MTSAMD_SIZEOF_MATERIAL: return (int64_t) sizeof(mtsamd_material);
        case MTSAMD_SIZEOF_EMITTER: return (int64_t) sizeof(mtsamd_emitter);
        case MTSAMD_SIZEOF_SHAPE: return (int64_t) sizeof(mtsamd_shape);
        case MTSAMD_SIZEOF_RENDER_OPTS: return (int64_t) sizeof(mtsamd_render_opts);
        case MTSAMD_SIZEOF_STATS: return (int64_t) sizeof(mtsamd_stats);
        case MTSAMD_SIZEOF_LDR_OPTS: return (int64_t) sizeof(mtsamd_ldr_opts);
        case MTSAMD_SIZEOF_TEXTURE: return (int64_t) sizeof(mtsamd_texture);
        default: return -1;
    }
}

extern "C" int mtsamd_render_device(mtsamd_scene *scene, const mtsamd_render_opts *opts, float *film_device, mtsamd_stats *stats) {
    if (!scene || !film_device) return fail(MTSAMD_EINVAL, "null argument");
    return renderCommon(scene, opts, film_device, nullptr, nullptr, stats);
}

extern "C" int mtsamd_render(mtsamd_scene *scene, const mtsamd_render_opts *opts, float *film_host, mtsamd_stats *stats) {
    if (!scene || !film_host) return fail(MTSAMD_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(scene->device));
    const size_t bytes = (size_t) scene->dev.cropW * scene->dev.cropH * 5 * sizeof(float);
    DeviceBuffer film;
    HIP_TRY(film.ensure(bytes));
    HIP_TRY(hipMemcpy(film.ptr, film_host, bytes, hipMemcpyHostToDevice));
    int rc = renderCommon(scene, opts, film.as<float>(), nullptr, nullptr, stats);
    if (rc != MTSAMD_OK) return rc;
    HIP_TRY(hipMemcpy(film_host, film.ptr, bytes, hipMemcpyDeviceToHost));
    return MTSAMD_OK;
}

// One render job dealt out over several GPUs behind the ABI: scenes[i] is the same scene created on device i; GPU i
// renders the i-th of every n full-width tiles of `band_rows` rows (interleaved, so that expensive parts of the image
// spread over all GPUs) in passes of its own on a host thread of its own, and the per-GPU films are added into
// film_host in device order (Film::put(block) under the reference's mutex, renderproc.cpp:142-149; the order is fixed
// here, so the sum is reproducible).
extern "C" int mtsamd_render_multi(mtsamd_scene *const *scenes, int32_t n_scenes, const mtsamd_render_opts *opts, int32_t band_rows,
                                   float *film_host, mtsamd_stats *stats) {
    if (!scenes || n_scenes <= 0 || !opts || !film_host) return fail(MTSAMD_EINVAL, "null argument");
    if (n_scenes > 64) return fail(MTSAMD_EINVAL, "at most 64 devices");
    for (int i = 0; i < n_scenes; ++i) {
        if (!scenes[i]) return fail(MTSAMD_EINVAL, "null scene handle");
        if (scenes[i]->dev.cropW != scenes[0]->dev.cropW || scenes[i]->dev.cropH != scenes[0]->dev.cropH || scenes[i]->dev.nTris != scenes[0]->dev.nTris)
            return fail(MTSAMD_EINVAL, "scene %d is not a replica of scene 0", i);
        for (int j = 0; j < i; ++j) if (scenes[j] == scenes[i]) return fail(MTSAMD_EINVAL, "scene handle %d given twice", i);
    }
    mtsamd_render_opts base;
    int rc = checkOpts(scenes[0], opts, base);
    if (rc != MTSAMD_OK) return rc;
    if (base.band_rows) return fail(MTSAMD_EINVAL, "mtsamd_render_multi deals the bands out itself: leave band_rows / band_period of the options zero");
    if (band_rows <= 0) band_rows = 16;
    const int W = scenes[0]->dev.cropW, H = scenes[0]->dev.cropH;
    const size_t floats = (size_t) W * H * 5, rowFloats = (size_t) W * 5;
    // What a device hands back: only the frame rows its bands and their filter borders touch (a device's film is zero
    // everywhere else), through page-locked memory. 8 GPUs at 3840x2160: 8 x 34 MB instead of 8 x 166 MB.
    struct Segment { int row0, rows; size_t offset; };      // frame rows [row0, row0 + rows) at `offset` floats of the staging buffer
    struct Share { std::vector<Segment> segments; float *staging = nullptr; size_t floats = 0; };
    std::vector<Share> shares((size_t) n_scenes);
    std::vector<int> codes((size_t) n_scenes, MTSAMD_OK);
    std::vector<std::string> errors((size_t) n_scenes);
    auto work = [&](int i) {
        // every error path of this thread leaves its message in the thread's own g_error: carry it to the caller
        auto run = [&]() -> int {
            mtsamd_scene *sc = scenes[i];
            mtsamd_render_opts o = base;
            o.y0 = base.y0 + i * band_rows;
            o.band_rows = band_rows; o.band_period = band_rows * n_scenes;
            if (stats) memset(&stats[i], 0, sizeof(mtsamd_stats));
            if (o.y0 >= o.y1) return MTSAMD_OK;            // fewer bands than GPUs: nothing for this one
            HIP_TRY(hipSetDevice(sc->device));
            DeviceBuffer film;
            HIP_TRY(film.ensure(floats * sizeof(float)));
            HIP_TRY(hipMemsetAsync(film.ptr, 0, floats * sizeof(float), sc->stream));
            int r = renderCommon(sc, &o, film.as<float>(), nullptr, nullptr, stats ? &stats[i] : nullptr);
            if (r != MTSAMD_OK) return r;
            auto t0 = std::chrono::steady_clock::now();
            Share &sh = shares[(size_t) i];
            const int border = (int) std::ceil(sc->dev.filterRadius - 0.5f);
            for (int yb = o.y0; yb < o.y1; yb += o.band_period) {
                const int r0 = std::max(0, yb - border), r1 = std::min(H, std::min(yb + band_rows, o.y1) + border);
                if (!sh.segments.empty() && r0 <= sh.segments.back().row0 + sh.segments.back().rows) {
                    Segment &last = sh.segments.back();     // borders of neighbouring bands meet: one segment
                    last.rows = std::max(last.rows, r1 - last.row0);
                } else sh.segments.push_back({ r0, r1 - r0, 0 });
            }
            for (Segment &g : sh.segments) { g.offset = sh.floats; sh.floats += (size_t) g.rows * rowFloats; }
            HIP_TRY(hipHostMalloc((void **) &sh.staging, sh.floats * sizeof(float), hipHostMallocDefault));
            for (const Segment &g : sh.segments)
                HIP_TRY(hipMemcpyAsync(sh.staging + g.offset, film.as<float>() + (size_t) g.row0 * rowFloats, (size_t) g.rows * rowFloats * sizeof(float),
                                       hipMemcpyDeviceToHost, sc->stream));
            HIP_TRY(hipStreamSynchronize(sc->stream));
            if (stats) stats[i].gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return MTSAMD_OK;
        };
        try { codes[(size_t) i] = run(); } catch (const std::exception &ex) { codes[(size_t) i] = fail(MTSAMD_EDEVICE, "%s", ex.what()); }
        if (codes[(size_t) i] != MTSAMD_OK) errors[(size_t) i] = g_error;
    };
    {
        std::vector<std::thread> threads;
        for (int i = 1; i < n_scenes; ++i) threads.emplace_back(work, i);
        work(0);
        for (std::thread &t : threads) t.join();
    }
    int failed = -1;
    for (int i = 0; i < n_scenes && failed < 0; ++i) if (codes[(size_t) i] != MTSAMD_OK) failed = i;
    if (failed < 0) {
        // film_host += share 0 + share 1 + ... (fixed order per element, so the sum is reproducible; the frame's rows
        // are cut into slabs that are summed side by side)
        auto t0 = std::chrono::steady_clock::now();
        const int nt = std::max(1, std::min(H, 16));
        std::vector<std::thread> threads;
        auto sum = [&](int t) {
            const int ra = (int) ((int64_t) H * t / nt), rb = (int) ((int64_t) H * (t + 1) / nt);
            for (int i = 0; i < n_scenes; ++i)
                for (const Segment &g : shares[(size_t) i].segments) {
                    const int a = std::max(ra, g.row0), b = std::min(rb, g.row0 + g.rows);
                    if (a >= b) continue;
                    const float *src = shares[(size_t) i].staging + g.offset + (size_t) (a - g.row0) * rowFloats;
                    float *dst = film_host + (size_t) a * rowFloats;
                    for (size_t k = 0, n = (size_t) (b - a) * rowFloats; k < n; ++k) dst[k] += src[k];
                }
        };
        for (int t = 1; t < nt; ++t) threads.emplace_back(sum, t);
        sum(0);
        for (std::thread &t : threads) t.join();
        if (stats) stats[0].gather_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    for (Share &sh : shares) if (sh.staging) (void) hipHostFree(sh.staging);
    if (failed >= 0) return fail(codes[(size_t) failed], "device %d: %s", scenes[failed]->device, errors[(size_t) failed].c_str());
    return MTSAMD_OK;
}

extern "C" int mtsamd_li(mtsamd_scene *scene, const mtsamd_render_opts *opts, float *out, float *pos, mtsamd_stats *stats) {
    if (!scene || !out || !pos) return fail(MTSAMD_EINVAL, "null argument");
    mtsamd_render_opts o;
    int rc = checkOpts(scene, opts, o);
    if (rc != MTSAMD_OK) return rc;
    HIP_TRY(hipSetDevice(scene->device));
    const size_t n = (size_t) (o.x1 - o.x0) * (o.y1 - o.y0) * o.spp;
    DeviceBuffer dOut, dPos;
    HIP_TRY(dOut.ensure(n * 16)); HIP_TRY(dPos.ensure(n * 8));
    rc = renderCommon(scene, opts, nullptr, dOut.as<float>(), dPos.as<float>(), stats);
    if (rc != MTSAMD_OK) return rc;
    HIP_TRY(hipMemcpy(out, dOut.ptr, n * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pos, dPos.ptr, n * 8, hipMemcpyDeviceToHost));
    return MTSAMD_OK;
}

extern "C" int mtsamd_film_develop(const mtsamd_scene *scene, const float *film, float *rgba) {
    if (!scene || !film || !rgba) return fail(MTSAMD_EINVAL, "null argument");
    const size_t n = (size_t) scene->dev.cropW * scene->dev.cropH;
    for (size_t i = 0; i < n; ++i) {
        const float w = film[5 * i + 4];
        const float inv = w != 0 ? 1.0f / w : 0.0f;     // fmtconv.cpp:978-990
        for (int c = 0; c < 4; ++c) rgba[4 * i + c] = film[5 * i + c] * inv;
    }
    return MTSAMD_OK;
}

namespace mtsamd_host { const char *developLdr(const float *film, size_t pixels, const mtsamd_ldr_opts &o, uint8_t *out); }

extern "C" int mtsamd_film_develop_ldr(const float *film, int32_t width, int32_t height, const mtsamd_ldr_opts *opts, uint8_t *out) {
    if (!film || !opts || !out) return fail(MTSAMD_EINVAL, "null argument");
    if (width < 0 || height < 0) return fail(MTSAMD_EINVAL, "negative image size");
    const char *err = mtsamd_host::developLdr(film, (size_t) width * (size_t) height, *opts, out);
    if (err && *err) return fail(MTSAMD_EINVAL, "%s", err);
    return MTSAMD_OK;
}

// ---- batch entry points ---------------------------------------------------------------------------------------------------
namespace {

struct Staging {
    std::vector<DeviceBuffer> bufs;
    Staging() { bufs.reserve(16); }
    int in(const void *src, size_t bytes, void **dev) {
        bufs.emplace_back();
        HIP_TRY(bufs.back().ensure(std::max<size_t>(bytes, 16)));
        if (bytes) HIP_TRY(hipMemcpy(bufs.back().ptr, src, bytes, hipMemcpyHostToDevice));
        *dev = bufs.back().ptr;
        return MTSAMD_OK;
    }
    int out(size_t bytes, void **dev) {
        bufs.emplace_back();
        HIP_TRY(bufs.back().ensure(std::max<size_t>(bytes, 16)));
        *dev = bufs.back().ptr;
        return MTSAMD_OK;
    }
};

#define TRY(expr) do { int rc_ = (expr); if (rc_ != MTSAMD_OK) return rc_; } while (0)

int finish(hipStream_t st) {
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return MTSAMD_OK;
}

// Batch query over host-layout rays (8 floats each: origin, mint, direction, maxt) already on the device.
template <int MODE>
int traceBatch(mtsamd_scene *s, const void *raysDev, uint64_t n, float4 *hitsDev, uint8_t *occludedDev, uint32_t *heads = nullptr) {
    TraceArgs a;
    memset(&a, 0, sizeof(a));
    a.rayO = (const float4 *) raysDev; a.rayD = a.rayO + 1; a.stride = 2; a.n = (uint32_t) n;
    a.hits = hitsDev; a.occluded = occludedDev;
    a.stats = s->counters.as<unsigned long long>() + CNT64_EXT_NODES;
    launchTrace<MODE>(s, a, s->stream, heads);
    return MTSAMD_OK;
}

int checkBatch(mtsamd_scene *s, uint64_t n) {
    if (!s) return fail(MTSAMD_EINVAL, "null scene");
    if (n >= (1ull << 31)) return fail(MTSAMD_EINVAL, "batch too large");
    HIP_TRY(hipSetDevice(s->device));
    return MTSAMD_OK;
}

}  // namespace

extern "C" int mtsamd_trace(mtsamd_scene *s, uint64_t n, const float *rays, mtsamd_hit *hits) {
    TRY(checkBatch(s, n));
    if (n == 0) return MTSAMD_OK;
    if (!rays || !hits) return fail(MTSAMD_EINVAL, "null argument");
    Staging sg; void *dR, *dH, *dO;
    TRY(sg.in(rays, n * 32, &dR)); TRY(sg.out(n * 16, &dH)); TRY(sg.out(n * sizeof(mtsamd_hit), &dO));
    TRY(traceBatch<TRACE_CLOSEST>(s, dR, n, (float4 *) dH, nullptr));
    hipLaunchKernelGGL(k_hits_to_records, dim3(blocksFor(n, 256)), dim3(256), 0, s->stream, s->dev, (const float4 *) dH, (uint32_t) n, (mtsamd_hit *) dO);
    TRY(finish(s->stream));
    HIP_TRY(hipMemcpy(hits, dO, n * sizeof(mtsamd_hit), hipMemcpyDeviceToHost));
    return MTSAMD_OK;
}

extern "C" int mtsamd_occluded(mtsamd_scene *s, uint64_t n, const float *rays, uint8_t *occluded) {
    TRY(checkBatch(s, n));
    if (n == 0) return MTSAMD_OK;
    if (!rays || !occluded) return fail(MTSAMD_EINVAL, "null argument");
    Staging sg; void *dR, *dO;
    TRY(sg.in(rays, n * 32, &dR)); TRY(sg.out(n, &dO));
    TRY(traceBatch<TRACE_OCCLUDED>(s, dR, n, nullptr, (uint8_t *) dO));
    TRY(finish(s->stream));
    HIP_TRY(hipMemcpy(occluded, dO, n, hipMemcpyDeviceToHost));
    return MTSAMD_OK;
}

namespace {
// A pair of events around one launch on the library's stream
struct LaunchClock {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    explicit LaunchClock(hipStream_t s) : st(s) { (void) hipEventCreate(&a); (void) hipEventCreate(&b); (void) hipEventRecord(a, st); }
    double stop() {
        (void) hipEventRecord(b, st); (void) hipEventSynchronize(b);
        float ms = 0; (void) hipEventElapsedTime(&ms, a, b);
        return ms * 1e-3;
    }
    ~LaunchClock() { if (a) (void) hipEventDestroy(a); if (b) (void) hipEventDestroy(b); }
};
}  // namespace

extern "C" int mtsamd_trace_device(mtsamd_scene *s, uint64_t n, const float *raysDev, mtsamd_hit *hitsDev, double *kernelSeconds) {
    TRY(checkBatch(s, n));
    if (kernelSeconds) *kernelSeconds = 0;
    if (n == 0) return MTSAMD_OK;
    if (!raysDev || !hitsDev) return fail(MTSAMD_EINVAL, "null argument");
    Staging sg; void *dH;
    TRY(sg.out(n * 16, &dH));
    {
        LaunchClock clock(s->stream);
        TRY(traceBatch<TRACE_CLOSEST>(s, raysDev, n, (float4 *) dH, nullptr));
        if (kernelSeconds) *kernelSeconds = clock.stop();
    }
    hipLaunchKernelGGL(k_hits_to_records, dim3(blocksFor(n, 256)), dim3(256), 0, s->stream, s->dev, (const float4 *) dH, (uint32_t) n, hitsDev);
    return finish(s->stream);
}

extern "C" int mtsamd_trace_device_static(mtsamd_scene *s, uint64_t n, const float *raysDev, mtsamd_hit *hitsDev) {
    TRY(checkBatch(s, n));
    if (n == 0) return MTSAMD_OK;
    if (!raysDev || !hitsDev) return fail(MTSAMD_EINVAL, "null argument");
    if (n > kStaticMaxRays) return fail(MTSAMD_EINVAL, "mtsamd_trace_device_static takes at most %llu rays per call (%llu given)",
                                        (unsigned long long) kStaticMaxRays, (unsigned long long) n);
    unsigned char *base = nullptr;
    TRY(staticScratch(&base));
    // the array keeps the previous call's contents: the hand-out counters are cleared by launchTrace, the raw hits
    // of all n rays are overwritten by the traversal before k_hits_to_records reads them
    float4 *rawHits = (float4 *) (base + kStaticHeadsBytes);
    TRY(traceBatch<TRACE_CLOSEST>(s, raysDev, n, rawHits, nullptr, (uint32_t *) base));
    hipLaunchKernelGGL(k_hits_to_records, dim3(blocksFor(n, 256)), dim3(256), 0, s->stream, s->dev, (const float4 *) rawHits, (uint32_t) n, hitsDev);
    return finish(s->stream);
}

extern "C" uint64_t mtsamd_trace_device_static_max(void) { return kStaticMaxRays; }

extern "C" int mtsamd_occluded_device(mtsamd_scene *s, uint64_t n, const float *raysDev, uint8_t *occludedDev, double *kernelSeconds) {
    TRY(checkBatch(s, n));
    if (kernelSeconds) *kernelSeconds = 0;
    if (n == 0) return MTSAMD_OK;
    if (!raysDev || !occludedDev) return fail(MTSAMD_EINVAL, "null argument");
    {
        LaunchClock clock(s->stream);
        TRY(traceBatch<TRACE_OCCLUDED>(s, raysDev, n, nullptr, occludedDev));
        if (kernelSeconds) *kernelSeconds = clock.stop();
    }
    return finish(s->stream);
}

extern "C" int mtsamd_intersect(mtsamd_scene *s, uint64_t n, const float *rays, mtsamd_intersection *its) {
    TRY(checkBatch(s, n));
    if (n == 0) return MTSAMD_OK;
    if (!rays || !its) return fail(MTSAMD_EINVAL, "null argument");
    Staging sg; void *dR, *dH, *dO;
    TRY(sg.in(rays, n * 32, &dR)); TRY(sg.out(n * 16, &dH)); TRY(sg.out(n * sizeof(mtsamd_intersection), &dO));
    TRY(traceBatch<TRACE_CLOSEST>(s, dR, n, (float4 *) dH, nullptr));
    hipLaunchKernelGGL(k_fill_isect, dim3(blocksFor(n, 256)), dim3(256), 0, s->stream, s->dev, (const float *) dR, (const float4 *) dH, (uint32_t) n, (mtsamd_intersection *) dO);
    TRY(finish(s->stream));
    HIP_TRY(hipMemcpy(its, dO, n * sizeof(mtsamd_intersection), hipMemcpyDeviceToHost));
    return MTSAMD_OK;
}

static int bsdfEvalPdf(mtsamd_scene *s, int material, int mode, uint64_t n, const float *wi, const float *wo, float *out) {
    TRY(checkBatch(s, n));
    if (material < 0 || (uint32_t) material >= s->dev.nMaterials) return fail(MTSAMD_EINVAL, "material index out of range");
    if (n == 0) return MTSAMD_OK;
    if (!wi || !wo || !out) return fail(MTSAMD_EINVAL, "null argument");
    Staging sg; void *dI, *dW, *dO;
    const size_t ob = mode == 0 ? n * 12 : n * 4;
    TRY(sg.in(wi, n * 12, &dI)); TRY(sg.in(wo, n * 12, &dW)); TRY(sg.out(ob, &dO));
    hipLaunchKernelGGL(k_bsdf_evalpdf, dim3(blocksFor(n, 256)), dim3(256), 0, s->stream, s->dev, material, s->queryUv, mode, (uint32_t) n, (const float *) dI, (const float *) dW, (float *) dO);
    TRY(finish(s->stream));
    HIP_TRY(hipMemcpy(out, dO, ob, hipMemcpyDeviceToHost));
    return MTSAMD_OK;
}

extern "C" int mtsamd_texture_eval(mtsamd_scene *s, int texture, uint64_t n, const float *uv, const float *duv, float *out) {
    TRY(checkBatch(s, n));
    if (texture < 1 || (size_t) texture > s->host.plainTextures.size()) return fail(MTSAMD_EINVAL, "texture index out of range");
    if (n == 0) return MTSAMD_OK;
    if (!uv || !out) return fail(MTSAMD_EINVAL, "null argument");
    Staging sg; void *dU, *dD = nullptr, *dO;
    TRY(sg.in(uv, n * 8, &dU));
    if (duv) TRY(sg.in(duv, n * 16, &dD));
    TRY(sg.out(n * 12, &dO));
    hipLaunchKernelGGL(k_texture_eval, dim3(blocksFor(n, 256)), dim3(256), 0, s->stream, s->dev, s->plainTextures.as<DevTexture>(), texture - 1,
                       (uint32_t) n, (const float *) dU, (const float *) dD, (float *) dO);
    TRY(finish(s->stream));
    HIP_TRY(hipMemcpy(out, dO, n * 12, hipMemcpyDeviceToHost));
    return MTSAMD_OK;
}

extern "C" int mtsamd_env_eval(mtsamd_scene *s, uint64_t n, const float *d, const float *refN, float *value, float *pdf) {
    TRY(checkBatch(s, n));
    if (s->dev.envEmitter < 0) return fail(MTSAMD_EINVAL, "the scene has no environment emitter");
    if (n == 0) return MTSAMD_OK;
    if (!d || !refN || !value || !pdf) return fail(MTSAMD_EINVAL, "null argument");
    Staging sg; void *dD, *dN, *dV, *dP;
    TRY(sg.in(d, n * 12, &dD)); TRY(sg.in(refN, n * 12, &dN)); TRY(sg.out(n * 12, &dV)); TRY(sg.out(n * 4, &dP));
    hipLaunchKernelGGL(k_env_eval, dim3(blocksFor(n, 256)), dim3(256), 0, s->stream, s->dev, (uint32_t) n, (const float *) dD, (const float *) dN,
                       (float *) dV, (float *) dP);
    TRY(finish(s->stream));
    HIP_TRY(hipMemcpy(value, dV, n * 12, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pdf, dP, n * 4, hipMemcpyDeviceToHost));
    return MTSAMD_OK;
}

extern "C" int mtsamd_bsdf_eval(mtsamd_scene *s, int material, uint64_t n, const float *wi, const float *wo, float *out) {
    return bsdfEvalPdf(s, material, 0, n, wi, wo, out);
}
extern "C" int mtsamd_bsdf_pdf(mtsamd_scene *s, int material, uint64_t n, const float *wi, const float *wo, float *out) {
    return bsdfEvalPdf(s, material, 1, n, wi, wo, out);
}

extern "C" int mtsamd_bsdf_sample(mtsamd_scene *s, int material, uint64_t n, const float *wi, const float *sample,
                                  float *wo, float *weight, float *pdf, float *eta, uint32_t *sampled_type, const float *sampler_1d) {
    TRY(checkBatch(s, n));
    if (material < 0 || (uint32_t) material >= s->dev.nMaterials) return fail(MTSAMD_EINVAL, "material index out of range");
    if (n == 0) return MTSAMD_OK;
    if (!wi || !sample || !wo || !weight || !pdf || !eta || !sampled_type) return fail(MTSAMD_EINVAL, "null argument");
    Staging sg; void *dI, *dS, *dWo, *dW, *dP, *dE, *dT;
    void *dU = nullptr;
    TRY(sg.in(wi, n * 12, &dI)); TRY(sg.in(sample, n * 8, &dS));
    if (sampler_1d) TRY(sg.in(sampler_1d, n * 4, &dU));
    TRY(sg.out(n * 12, &dWo)); TRY(sg.out(n * 12, &dW)); TRY(sg.out(n * 4, &dP)); TRY(sg.out(n * 4, &dE)); TRY(sg.out(n * 4, &dT));
    hipLaunchKernelGGL(k_bsdf_sample, dim3(blocksFor(n, 256)), dim3(256), 0, s->stream, s->dev, material, s->queryUv, (uint32_t) n, (const float *) dI,
                       (const float *) dS, (const float *) dU, (float *) dWo, (float *) dW, (float *) dP, (float *) dE, (uint32_t *) dT);
    TRY(finish(s->stream));
    HIP_TRY(hipMemcpy(wo, dWo, n * 12, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(weight, dW, n * 12, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pdf, dP, n * 4, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(eta, dE, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sampled_type, dT, n * 4, hipMemcpyDeviceToHost));
    return MTSAMD_OK;
}

extern "C" int mtsamd_sample_emitter_direct(mtsamd_scene *s, uint64_t n, const float *ref, const float *ref_n, const float *sample,
                                            float *d, float *dist, float *pdf, float *value) {
    TRY(checkBatch(s, n));
    if (n == 0) return MTSAMD_OK;
    if (!ref || !ref_n || !sample || !d || !dist || !pdf || !value) return fail(MTSAMD_EINVAL, "null argument");
    Staging sg; void *dR, *dN, *dS, *dD, *dDi, *dP, *dV;
    TRY(sg.in(ref, n * 12, &dR)); TRY(sg.in(ref_n, n * 12, &dN)); TRY(sg.in(sample, n * 8, &dS));
    TRY(sg.out(n * 12, &dD)); TRY(sg.out(n * 4, &dDi)); TRY(sg.out(n * 4, &dP)); TRY(sg.out(n * 12, &dV));
    hipLaunchKernelGGL(k_emitter_sample, dim3(blocksFor(n, 256)), dim3(256), 0, s->stream, s->dev, (uint32_t) n, (const float *) dR,
                       (const float *) dN, (const float *) dS, (float *) dD, (float *) dDi, (float *) dP, (float *) dV);
    TRY(finish(s->stream));
    HIP_TRY(hipMemcpy(d, dD, n * 12, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(dist, dDi, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pdf, dP, n * 4, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(value, dV, n * 12, hipMemcpyDeviceToHost));
    return MTSAMD_OK;
}

extern "C" int mtsamd_camera_rays(mtsamd_scene *s, uint64_t n, const float *pos, float *rays) {
    TRY(checkBatch(s, n));
    if (n == 0) return MTSAMD_OK;
    if (!pos || !rays) return fail(MTSAMD_EINVAL, "null argument");
    Staging sg; void *dP, *dR;
    TRY(sg.in(pos, n * 8, &dP)); TRY(sg.out(n * 32, &dR));
    hipLaunchKernelGGL(k_camera_rays, dim3(blocksFor(n, 256)), dim3(256), 0, s->stream, s->dev, (uint32_t) n, (const float *) dP, (float *) dR);
    TRY(finish(s->stream));
    HIP_TRY(hipMemcpy(rays, dR, n * 32, hipMemcpyDeviceToHost));
    return MTSAMD_OK;
}

extern "C" int mtsamd_film_put(mtsamd_scene *s, uint64_t n, const float *pos, const float *value, float *film_host) {
    TRY(checkBatch(s, n));
    if (!film_host) return fail(MTSAMD_EINVAL, "null argument");
    if (n == 0) return MTSAMD_OK;
    if (!pos || !value) return fail(MTSAMD_EINVAL, "null argument");
    const size_t fb = (size_t) s->dev.cropW * s->dev.cropH * 20;
    Staging sg; void *dP, *dV, *dF;
    TRY(sg.in(pos, n * 8, &dP)); TRY(sg.in(value, n * 16, &dV)); TRY(sg.in(film_host, fb, &dF));
    hipLaunchKernelGGL(k_film_put, dim3(blocksFor((size_t) s->dev.cropW * s->dev.cropH, 256)), dim3(256), 0, s->stream, s->dev, (uint32_t) n,
                       (const float *) dP, (const float *) dV, (float *) dF);
    TRY(finish(s->stream));
    HIP_TRY(hipMemcpy(film_host, dF, fb, hipMemcpyDeviceToHost));
    return MTSAMD_OK;
}

extern "C" int mtsamd_rng_stream(uint32_t seed, uint64_t key, uint32_t n, float *out) {
    if (!out && n) return fail(MTSAMD_EINVAL, "null argument");
    if (n == 0) return MTSAMD_OK;
    DeviceBuffer b;
    HIP_TRY(b.ensure((size_t) n * 4));
    hipLaunchKernelGGL(k_rng_stream, dim3(1), dim3(64), 0, 0, seed, key, n, b.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, b.ptr, (size_t) n * 4, hipMemcpyDeviceToHost));
    return MTSAMD_OK;
}
