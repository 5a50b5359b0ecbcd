// GPU kernels of the wavefront path tracer (gfx950): raygen -> extend -> shade -> shadow-connect -> film.
#pragma once
#include "device_trace.h"
#include "device_shade.h"

namespace mtsamd {

static constexpr uint32_t kMiss = 0xFFFFFFFFu;

MTS_DEV uint64_t sampleKey(const DevScene &sc, const PassParams &pp, uint32_t slot, int &px, int &py, int &s) {
    const uint32_t pix = divideBy(slot, (uint32_t) pp.spp, pp.sppMagic);
    s = (int) (slot - pix * (uint32_t) pp.spp);
    const uint32_t row = divideBy(pix, (uint32_t) pp.w, pp.wMagic);
    py = pp.y0 + (int) row;
    if (pp.bandRows) {      // interleaved bands: the pass's rows are every bandPeriod-th group of bandRows frame rows
        const uint32_t band = divideBy(row, (uint32_t) pp.bandRows, pp.bandMagic);
        py += (int) (band * (uint32_t) (pp.bandPeriod - pp.bandRows));
    }
    px = pp.x0 + (int) (pix - row * (uint32_t) pp.w);
    const uint64_t pixelIndex = (uint64_t) py * (uint64_t) sc.cropW + (uint64_t) px;
    return pixelIndex * (uint64_t) pp.sppTotal + (uint64_t) (pp.sampleBase + s);
}

// ---- K1 raygen: SamplingIntegrator::renderBlock sample loop + sensor->sampleRayDifferential
//      (src/librender/integrator.cpp:162-186, src/sensors/perspective.cpp:271-298) ------------------------------------
__global__ void __launch_bounds__(256) k_raygen(DevScene sc, PassParams pp, PathQueue q, float4 *sampleL, float2 *samplePos, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int px, py, s;
    const uint64_t key = sampleKey(sc, pp, i, px, py, s);
    Rng rng;
    rng.init(pp.seed, key);
    const float jx = rng.next1D(), jy = rng.next1D();
    const float posx = (float) px + jx, posy = (float) py + jy;
    V3 o, d; float mint, maxt;
    cameraRay(sc, posx, posy, o, d, mint, maxt);
    q.rayO[i] = make_float4(o.x, o.y, o.z, mint);
    q.rayD[i] = make_float4(d.x, d.y, d.z, maxt);
    q.thr[i] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    q.misc[i] = make_uint4(i, 0u, (uint32_t) rng.state, (uint32_t) (rng.state >> 32));
    q.bsdfPdf[i] = 0.0f;
    sampleL[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    samplePos[i] = make_float2(posx, posy);
}

// ---- K2 extend / K4 shadow-connect / batch queries: one persistent traversal kernel, two ways of finishing a ray.
//   TRACE_CLOSEST       closest hit for every live path (Scene::rayIntersect(ray, its))
//   TRACE_OCCLUDED      any-hit test, one byte per ray: the visibility test inside Scene::sampleEmitterDirect
//                       (src/librender/scene.cpp:839-843); k_shadow_accumulate then adds the unoccluded samples' values
#ifndef MTS_TRACE_WAVES
#define MTS_TRACE_WAVES 5
#endif
//   STACK               layout of the per-lane traversal stack (StackLayout): 0 for any tree, N for trees of depth <= N + 2
template <int MODE, bool COUNT, int STACK>
__global__ void __attribute__((amdgpu_flat_work_group_size(64, 1024), amdgpu_waves_per_eu(MTS_TRACE_WAVES))) k_trace(DevScene sc, TraceArgs a) {
    extern __shared__ float4 traceLds[];
    LdsUint2 *stacks = (LdsUint2 *) &traceLds[0];
    LdsFloat4 *rayBuf = (LdsFloat4 *) (stacks + StackLayout<STACK>::kLds * a.blockThreads);
    traceLoop<MODE, COUNT, STACK>(sc, a, stacks, rayBuf, (LdsUint *) (rayBuf + 2 * a.blockThreads));
}

// What shading one vertex produces: the extension ray with the path state that travels with it, and the shadow ray of
// the emitter sample with the radiance it carries if unoccluded.
// Shade workgroups: threads, wavefronts per SIMD the register budget is set for, and whether a thread's shadow ray
// waits in workgroup memory while its BSDF is sampled
#ifndef MTS_SHADE_WAVES
#define MTS_SHADE_WAVES 4
#endif
#ifndef MTS_SHADE_BLOCK
#define MTS_SHADE_BLOCK 256
#endif
static constexpr int kShadeBlock = MTS_SHADE_BLOCK;
#ifndef MTS_SHADE_WAVES_MIX
#define MTS_SHADE_WAVES_MIX 3
#endif
#ifndef MTS_SHADE_PARK
#define MTS_SHADE_PARK 1
#endif
static constexpr bool kShadePark = MTS_SHADE_PARK != 0;
#ifndef MTS_SHADE_EMITTER_LDS
#define MTS_SHADE_EMITTER_LDS 1
#endif
#ifndef MTS_ENV_LATE_MATERIAL
#define MTS_ENV_LATE_MATERIAL 1
#endif

struct ShadeOut {
    bool alive = false, wantShadow = false;
    float4 oRayO = make_float4(0, 0, 0, 0), oRayD = oRayO, oThr = oRayO, sRayO = oRayO, sRayD = oRayO, sContrib = oRayO;
    uint4 oMisc = make_uint4(0, 0, 0, 0);
    float oPdf = 0, oRefCos = 0;
};

// Path state as the shading code reads it: from the queue arrays (wavefront kernels) ...
struct QueuedPath {
    PathQueue q;
    uint32_t i;
    MTS_DEV float4 rayO() const { return q.rayO[i]; }
    MTS_DEV float4 rayD() const { return q.rayD[i]; }
    MTS_DEV float4 thr() const { return q.thr[i]; }
    MTS_DEV uint4 misc() const { return q.misc[i]; }
    MTS_DEV float bsdfPdf() const { return q.bsdfPdf[i]; }
    MTS_DEV float refCos() const { return q.refCos[i]; }
};
// ... or from registers (the kernel that follows a path through all its remaining bounces)
struct HeldPath {
    float4 ro, rd, th;
    uint4 mi;
    float pdf, rc;
    MTS_DEV float4 rayO() const { return ro; }
    MTS_DEV float4 rayD() const { return rd; }
    MTS_DEV float4 thr() const { return th; }
    MTS_DEV uint4 misc() const { return mi; }
    MTS_DEV float bsdfPdf() const { return pdf; }
    MTS_DEV float refCos() const { return rc; }
};

// One vertex of MIPathTracer::Li (src/integrators/path/path.cpp:135-287) for the path `in` whose ray produced `hit`.
// Radiance gathered at this vertex goes to sampleL[slot] directly; the rest comes back in `o`.
// `park` (workgroup memory, two float4 per thread of the workgroup) or nullptr: where the shadow ray's direction and
// contribution wait while the BSDF is sampled, instead of in eight registers.
// VS: the estimator of volpath_simple on a scene without media (src/integrators/path/volpath_simple.cpp:88-297) in place
// of path's: no multiple importance sampling; emitters are seen by the camera ray and after Dirac bounces only.
// BLEND: some record of the scene is a mixture (MIX alone: textures and frame adapters only, no mixture code).
template <bool ENV, bool RD, bool MIX, bool PARK, typename Path, bool VS = false, bool BLEND = MIX>
MTS_DEV void shadeVertex(const DevScene &sc, const PassParams &pp, const Path &in, const float4 hit, float4 *sampleL, const float2 *samplePos,
                         ShadeOut &o, float4 *park = nullptr, const EmitterTables *emitterTables = nullptr, bool useEmitterTables = false) {
    // one instance of every real call per register budget of the kernels that share this function
    constexpr int kTag = PARK ? (MIX ? 2 : 1) : 0;
    constexpr bool kLateMaterial = ENV && MTS_ENV_LATE_MATERIAL != 0;
    const uint32_t triWord = __float_as_uint(hit.w);
    if (triWord != kMiss) {
        const uint32_t tri = triWord & kTriIndexMask;
        const float4 rd = in.rayD(), th = in.thr();
        const uint4 misc = in.misc();
        const V3 rayD = mk(rd.x, rd.y, rd.z);
        const uint32_t slot = misc.x;
        const uint32_t prevDepth = misc.y & 0xffffu;
        uint32_t flags = misc.y >> 16;
        Rng rng;
        {
            int px, py, s;
            const uint64_t key = sampleKey(sc, pp, slot, px, py, s);
            rng.inc = (key << 1) | 1u;
            rng.state = (uint64_t) misc.z | ((uint64_t) misc.w << 32);
        }
        V3 throughput = mk(th.x, th.y, th.z);
        float eta = th.w;

        Isect its;
        fillIntersection(sc, tri, hit.y, hit.z, rayD, its);
        const float wiZ = its.wi.z;
        DevMaterial bsdf;
        uint32_t bsdfType;      // what the integrator asks of the BSDF as a whole (for a two-sided record: the adapter's type)
        bool flipped, framed = false;
        // Side storage in memory for what only textured, framed or mixture records use (the uv partials, the perturbed
        // frame): kept here instead of in registers, it costs the other paths nothing.
        SurfaceQuery q;
        bool texResolve = false;        // the call that resolves the textures of the answering record was made
        // Variants that load the answering record late: the record, with the textured colours and roughness the call found
        // (a mixture's textured children look up q.uv)
        auto loadAnswering = [&]() {
            loadMaterial(sc, its.material, bsdf);
            if (BLEND && !texResolve && bsdf.type == 8) q.uv = surfUV(its.uvx, its.uvy);
            if (MIX && texResolve) {
                if (q.textured & 1u) { bsdf.diffuse[0] = q.colour[0]; bsdf.diffuse[1] = q.colour[1]; bsdf.diffuse[2] = q.colour[2]; }
                if (q.textured & 2u) { bsdf.specular[0] = q.colour[3]; bsdf.specular[1] = q.colour[4]; bsdf.specular[2] = q.colour[5]; }
                if (q.textured & 4u) { bsdf.transmittance[0] = q.colour[6]; bsdf.transmittance[1] = q.colour[7]; bsdf.transmittance[2] = q.colour[8]; }
                if (q.textured & 8u) bsdf.alphaU = q.alpha[0];
                if (q.textured & 16u) bsdf.alphaV = q.alpha[1];
            }
        };
        if (MIX && sc.materialTextures) {
            // Scenes with textures. Who answers on this side (twosided.cpp:108-183) is read from three words of the hit's
            // record, so that the texture call below is made BEFORE the answering record is loaded: whatever is live
            // across a real call has to sit in the few registers a callee preserves, or in memory.
            const DevMaterial *rec = sc.materials + its.material;
            bsdfType = rec->typeFlags;
            // its.getBSDF(ray) (path.cpp:145, records.inl:69-75): uv partials for the vertex a camera ray reaches, if its
            // BSDF filters its textures
            const bool first = prevDepth == 0 && (sc.materialTextures[its.material].w & 1u);
            flipped = rec->twoSided && wiZ < 0;
            if (flipped) its.material = (uint32_t) rec->backMaterial;
            // textured colours of the answering record at this hit, and the frame of a bumpmap / normalmap around it, in
            // which it is asked (bumpmap.cpp:161-237): one call, made only for records that have any of it
            const uint4 tw = sc.materialTextures[its.material];
            const bool resolve = first || (tw.x | tw.y | tw.z | (tw.w & 14u)) != 0;
            const bool heavy = first || (tw.w & 6u) != 0;      // partials or a frame: the query goes through memory
            if (resolve && heavy) {
                q.p = its.p; q.geoN = its.geoN; q.dpdu = its.dpdu; q.dpdv = its.dpdv; q.sh = its.sh; q.u = its.uvx; q.v = its.uvy;
                const float4 ro = in.rayO();
                q.rayO = mk(ro.x, ro.y, ro.z); q.rayD = rayD;
                q.pos = first ? samplePos[slot] : make_float2(0.0f, 0.0f);
                q.sppTotal = pp.sppTotal; q.material = its.material; q.first = first ? 1u : 0u;
                resolveSurfaceCall<kTag>(sc.self, &q);
            } else if (resolve) {
                resolveColoursCall<kTag>(sc.self, its.material, its.uvx, its.uvy, &q);
            }
            if (kLateMaterial) {
                texResolve = resolve;
                framed = resolve && heavy && q.framed != 0;
            } else {
                loadMaterial(sc, its.material, bsdf);
                if (BLEND && !resolve && bsdf.type == 8) q.uv = surfUV(its.uvx, its.uvy);       // a mixture's textured children look here
                if (resolve) {
                    if (q.textured & 1u) { bsdf.diffuse[0] = q.colour[0]; bsdf.diffuse[1] = q.colour[1]; bsdf.diffuse[2] = q.colour[2]; }
                    if (q.textured & 2u) { bsdf.specular[0] = q.colour[3]; bsdf.specular[1] = q.colour[4]; bsdf.specular[2] = q.colour[5]; }
                    if (q.textured & 4u) { bsdf.transmittance[0] = q.colour[6]; bsdf.transmittance[1] = q.colour[7]; bsdf.transmittance[2] = q.colour[8]; }
                    if (q.textured & 8u) bsdf.alphaU = q.alpha[0];
                    if (q.textured & 16u) bsdf.alphaV = q.alpha[1];
                    framed = heavy && q.framed != 0;
                }
            }
        } else if (kLateMaterial) {
            // Scenes with emitters that are sampled through a real call: the answering record is loaded after that call
            // (below), so that its 28 words are not live across it; who answers is read from three words of the hit's record.
            const DevMaterial *rec = sc.materials + its.material;
            bsdfType = rec->typeFlags;
            flipped = rec->twoSided && wiZ < 0;
            if (flipped) its.material = (uint32_t) rec->backMaterial;
        } else {
            // the material record comes in as seven 16-byte loads issued together, not field by field
            loadMaterial(sc, its.material, bsdf);
            if (BLEND && bsdf.type == 8) q.uv = surfUV(its.uvx, its.uvy);
            bsdfType = bsdf.typeFlags;
            // ... and the BRDF that answers on this side (twosided.cpp:108-183)
            flipped = resolveTwoSided(sc, its.material, wiZ, bsdf);
        }
        const V3 wiLocal = flipped ? mirrorZ(its.wi) : its.wi;
        const V3 wiB = framed ? toLocal(q.pf, toWorld(its.sh, wiLocal)) : wiLocal;
        // The sample's radiance is only fetched where something is added to it: at the camera vertex it still is
        // what k_raygen wrote (zero), later only paths that hit an emitter touch it.
        float4 L = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        bool Lchanged = false, terminate = false;
        int depth;

        if (prevDepth == 0) {
            // camera vertex: alpha (records.inl:121-137) and directly visible emitters (path.cpp:148-150)
            depth = 1;
            L.w = 1.0f; Lchanged = true;
            if (its.emitter >= 0 && !sc.hideEmitters) {
                if (dot(its.sh.n, -rayD) > 0) {     // AreaLight::eval (area.cpp:104-109)
                    const V3 c = throughput * ld3(sc.emitters[its.emitter].radiance);
                    L.x += c.x; L.y += c.y; L.z += c.z;
                }
            }
        } else if (VS) {
            // roulette closes the previous iteration before anything is gathered here (volpath_simple.cpp:282-292) ...
            if ((int) prevDepth >= sc.rrDepth) {
                const float q = fminf(maxComp(throughput) * eta * eta, 0.95f);
                if (rng.next1D() >= q) terminate = true;
                throughput = throughput / q;
            }
            // ... then the emitter counts only after a Dirac bounce (:243-263, :184-186), at full weight
            if (!terminate && its.emitter >= 0 && (flags & PF_PREV_DELTA) && dot(its.sh.n, -rayD) > 0) {
                const V3 c = throughput * ld3(sc.emitters[its.emitter].radiance);
                if (c.x != 0 || c.y != 0 || c.z != 0) {
                    L = sampleL[slot];
                    L.x += c.x; L.y += c.y; L.z += c.z; Lchanged = true;
                }
            }
            depth = (int) prevDepth + 1;
            if (!(depth <= sc.maxDepth || sc.maxDepth < 0)) terminate = true;      // volpath_simple.cpp:112
        } else {
            // emitter reached by BSDF sampling: MIS against the emitter-sampling pdf (path.cpp:228-232,257-264)
            if (its.emitter >= 0) {
                const DevEmitter em = sc.emitters[its.emitter];
                const V3 value = dot(its.sh.n, -rayD) > 0 ? ld3(em.radiance) : mk(0.0f);
                float lumPdf = 0;
                if (!(flags & PF_PREV_DELTA)) {
                    // pdfEmitterDirect (scene.cpp:949-952) -> AreaLight::pdfDirect (area.cpp:175-183) -> Shape::pdfDirect (shape.cpp:117-121)
                    if ((flags & PF_REFN_OK) && dot(rayD, its.sh.n) < 0)
                        lumPdf = em.invArea * (hit.x * hit.x) / fabsf(dot(rayD, its.sh.n));
                    lumPdf *= em.pdfDiscrete;
                }
                const V3 c = throughput * value * miWeight(in.bsdfPdf(), lumPdf);
                if (c.x != 0 || c.y != 0 || c.z != 0) {
                    L = sampleL[slot];
                    L.x += c.x; L.y += c.y; L.z += c.z; Lchanged = true;
                }
            }
            // Russian roulette (path.cpp:276-286) ...
            if ((int) prevDepth >= sc.rrDepth) {
                const float q = fminf(maxComp(throughput) * eta * eta, 0.95f);
                if (rng.next1D() >= q) terminate = true;
                throughput = throughput / q;
            }
            depth = (int) prevDepth + 1;
            // ... and the loop condition (path.cpp:135)
            if (!(depth <= sc.maxDepth || sc.maxDepth < 0)) terminate = true;
        }

        // written here, not at the end: nothing below adds to it, and L would occupy four registers until then
        if (Lchanged) sampleL[slot] = L;
        // path.cpp:156-165; volpath_simple.cpp:193-196 lets a product of exactly zero pass
        if (!terminate && ((depth >= sc.maxDepth && sc.maxDepth > 0) ||
                           (sc.strictNormals && (VS ? dot(rayD, its.geoN) * wiZ > 0 : dot(rayD, its.geoN) * wiZ >= 0))))
            terminate = true;

        if (!terminate) {
            // DirectSamplingRecord(its) (records.inl:160-164)
            V3 refN = mk(0.0f);
            if ((bsdfType & (ETransmission | EBackSide)) == 0) refN = its.sh.n;

            // direct illumination sampling (path.cpp:172-200)
            const bool smooth = (bsdfType & ESmooth) != 0;
            DirectRec earlyRec;         // used by the variants that load the answering record late only
            V3 earlyValue;
            if (kLateMaterial) {
                // the emitter is sampled first, the answering record loaded after that call
                earlyRec.pdf = 0;
                earlyValue = mk(0.0f);
                if (smooth) {
                    const float sx = rng.next1D(), sy = rng.next1D();
                    earlyValue = sampleEmitterDirect<ENV, kTag>(sc, its.p, refN, sx, sy, earlyRec, emitterTables, useEmitterTables);
                }
                loadAnswering();
            }
            if (smooth) {
                DirectRec dRec;
                V3 value;
                if (kLateMaterial) { dRec = earlyRec; value = earlyValue; }
                else {
                    const float sx = rng.next1D(), sy = rng.next1D();
                    value = sampleEmitterDirect<ENV, kTag>(sc, its.p, refN, sx, sy, dRec, emitterTables, useEmitterTables);
                }
                if (dRec.pdf != 0) {
                    V3 c = mk(0.0f);
                    const V3 wo = toLocal(its.sh, dRec.d);
                    const V3 woSide = flipped ? mirrorZ(wo) : wo;
                    const V3 woB = framed ? toLocal(q.pf, toWorld(its.sh, woSide)) : woSide;
                    const bool sameSide = !framed || woSide.z * woB.z > 0;
                    const V3 bsdfVal = sameSide ? bsdfEvalAny<RD, BLEND, kTag>(sc, bsdf, wiB, woB, &q.uv) : mk(0.0f);
                    if (!isZero(bsdfVal) && (!sc.strictNormals || dot(its.geoN, dRec.d) * wo.z > 0)) {
                        // path.cpp:192-193: only emitters with a surface, sampled in the solid-angle measure, compete with BSDF sampling
                        if (VS) c = throughput * value * bsdfVal;       // volpath_simple.cpp:224: no weighting against BSDF sampling
                        else {
                            const float bPdf = ENV && dRec.delta ? 0.0f : bsdfPdfAny<RD, BLEND, kTag>(sc, bsdf, wiB, woB, &q.uv);
                            c = throughput * value * bsdfVal * miWeight(dRec.pdf, bPdf);
                        }
                    }
                    o.wantShadow = true;
                    o.sRayO = make_float4(its.p.x, its.p.y, its.p.z, kEpsilon);
                    // scene.cpp:840-841; Scene::evalTransmittance (scene.cpp:619-626) shortens the ray only towards a point on a surface
                    const float4 sRayD = make_float4(dRec.d.x, dRec.d.y, dRec.d.z, (VS && ENV && dRec.delta) ? dRec.dist : dRec.dist * (1 - kShadowEpsilon));
                    const float4 sContrib = make_float4(c.x, c.y, c.z, __uint_as_float(slot));
                    if (PARK) { park[threadIdx.x] = sRayD; park[kShadeBlock + threadIdx.x] = sContrib; }
                    else { o.sRayD = sRayD; o.sContrib = sContrib; }
                }
            }

            // BSDF sampling (path.cpp:206-226)
            const float bx = rng.next1D(), by = rng.next1D();
            V3 woLocal; float bPdf, bEta; uint32_t sampledType;
            const float bu = RD && bsdfDrawsFromSampler<BLEND>(bsdf, bx) ? rng.next1D() : 0.0f;     // bRec.sampler->next1D() inside BSDF::sample
            V3 bsdfWeight = bsdfSampleAny<RD, BLEND, kTag>(sc, bsdf, wiB, bx, by, bu, woLocal, bPdf, bEta, sampledType, &q.uv);
            if (framed && !isZero(bsdfWeight)) {        // back into the unperturbed frame (bumpmap.cpp:227-234)
                const V3 pwo = woLocal;
                woLocal = toLocal(its.sh, toWorld(q.pf, pwo));
                if (woLocal.z * pwo.z <= 0) bsdfWeight = mk(0.0f);
            }
            if (flipped && !isZero(bsdfWeight) && bPdf != 0) woLocal.z = -woLocal.z;      // twosided.cpp:172-181
            if (!isZero(bsdfWeight)) {
                if (sampledType != ENull) flags |= PF_SCATTERED;
                const V3 wo = toWorld(its.sh, woLocal);
                const float woDotGeoN = dot(its.geoN, wo);
                // volpath_simple.cpp:243-263: the path goes on only if the next vertex may gather something
                const bool goesOn = !VS || depth + 1 < sc.maxDepth || sc.maxDepth < 0 || ((sampledType & EDelta) && depth < sc.maxDepth);
                if (goesOn && !(sc.strictNormals && woDotGeoN * woLocal.z <= 0)) {
                    o.alive = true;
                    throughput = throughput * bsdfWeight;   // path.cpp:252-253
                    eta *= bEta;
                    uint32_t nf = flags & PF_SCATTERED;
                    if (sampledType & EDelta) nf |= PF_PREV_DELTA;
                    if (dot(wo, refN) >= 0) nf |= PF_REFN_OK;
                    o.oRayO = make_float4(its.p.x, its.p.y, its.p.z, kEpsilon);     // Ray(its.p, wo, time): mint = Epsilon, maxt = inf
                    o.oRayD = make_float4(wo.x, wo.y, wo.z, kInf);
                    o.oThr = make_float4(throughput.x, throughput.y, throughput.z, eta);
                    o.oMisc = make_uint4(slot, (uint32_t) depth | (nf << 16), (uint32_t) rng.state, (uint32_t) (rng.state >> 32));
                    o.oPdf = bPdf;
                    if (ENV) o.oRefCos = isZero(refN) ? -2.0f : dot(wo, refN);
                }
            }
        }
    } else if (ENV && sc.envEmitter >= 0) {
        // the ray left the scene and there is an environment emitter
        const uint4 misc = in.misc();
        const uint32_t slot = misc.x, prevDepth = misc.y & 0xffffu, flags = misc.y >> 16;
        const float4 th = in.thr();
        const V3 throughput = mk(th.x, th.y, th.z);
        const float4 rd = in.rayD();
        const V3 rayD = mk(rd.x, rd.y, rd.z);
        if (prevDepth == 0) {
            // camera ray (path.cpp:136-143): alpha stays 0 (records.inl:121-137); scene->evalEnvironment(ray) sees the
            // camera ray's offset rays, so a map is looked up with the elliptical filter
            if (!sc.hideEmitters) {
                const V3 c = throughput * (sc.envIsMap ? cameraEnvironmentRadiance(sc, pp, samplePos[slot], rayD) : ld3(sc.envRadiance));
                sampleL[slot] = make_float4(c.x, c.y, c.z, 0.0f);
            }
        } else if (VS) {
            // volpath_simple.cpp:282-292, then :170-181: roulette first, then the environment after a Dirac bounce only
            V3 thr = throughput;
            bool dead = false;
            if ((int) prevDepth >= sc.rrDepth) {
                Rng rng;
                int px, py, s;
                const uint64_t key = sampleKey(sc, pp, slot, px, py, s);
                rng.inc = (key << 1) | 1u;
                rng.state = (uint64_t) misc.z | ((uint64_t) misc.w << 32);
                const float q = fminf(maxComp(thr) * th.w * th.w, 0.95f);
                if (rng.next1D() >= q) dead = true;
                thr = thr / q;
            }
            if (!dead && (flags & PF_PREV_DELTA)) {
                const V3 c = thr * environmentRadiance(sc, rayD);
                if (c.x != 0 || c.y != 0 || c.z != 0) {
                    float4 L = sampleL[slot];
                    L.x += c.x; L.y += c.y; L.z += c.z;
                    sampleL[slot] = L;
                }
            }
        } else if (!(sc.hideEmitters && !(flags & PF_SCATTERED))) {
            // ray from BSDF sampling (path.cpp:233-247,257-264): MIS against the environment emitter's pdfDirect
            const float4 ro = in.rayO();
            float farT;
            if (environmentExit(sc, mk(ro.x, ro.y, ro.z), rayD, farT)) {     // fillDirectSamplingRecord (constant.cpp:245-261, envmap.cpp:358-374)
                const float lumPdf = (flags & PF_PREV_DELTA) ? 0.0f : environmentPdf(sc, rayD, in.refCos());
                const V3 c = throughput * environmentRadiance(sc, rayD) * miWeight(in.bsdfPdf(), lumPdf);
                if (c.x != 0 || c.y != 0 || c.z != 0) {
                    float4 L = sampleL[slot];
                    L.x += c.x; L.y += c.y; L.z += c.z;
                    sampleL[slot] = L;
                }
            }
        }
    }

}

// ---- K3 shade: one vertex of MIPathTracer::Li per live path (src/integrators/path/path.cpp:135-287) --------------------
// The loop body of the reference is cut at the extension-ray trace: everything after `scene->rayIntersect(ray, its)`
// (:226) for vertex k runs at the start of the invocation that shades vertex k+1. Queue entries carry the depth of
// the vertex that spawned the ray (0 for camera rays).
// A workgroup first sorts its paths by the BSDF type the hit carries (misses last), so that a wavefront mostly
// evaluates one BSDF model instead of stepping through all of them; whole wavefronts of misses leave at once.

// The paths [first, first + ROUNDS * kShadeBlock) of a workgroup, ordered by the BSDF type their hit carries (misses
// last): sOrder[k] is the offset from `first` of the k-th path in that order. sCount is scratch, [key][round][wave].
static constexpr int kShadeSortKeys = 9;    // BSDF models 0..7, misses
template <int ROUNDS>
MTS_DEV void sortByBsdfType(const float4 *hits, uint32_t first, uint32_t n, uint32_t *sCount, uint32_t *sOrder) {
    constexpr int kWaves = kShadeBlock / 64, kKeys = kShadeSortKeys, kCells = kKeys * ROUNDS * kWaves;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    {
        uint32_t key[ROUNDS], rank[ROUNDS];
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t i = first + r * kShadeBlock + tid;
            key[r] = kKeys - 1;
            if (i < n) {
                const uint32_t w = __float_as_uint(hits[i].w);
                if (w != kMiss) key[r] = w >> 28;
            }
            // One ballot per key bit instead of one per key: the lanes that hold key k are the intersection of the four
            // bit masks (or their complements); every lane forms it for its own key (its rank), lane k for key k (the count).
            static_assert(kKeys <= 16, "keys are four bits");
            const unsigned long long b0 = __ballot(key[r] & 1u), b1 = __ballot(key[r] & 2u), b2 = __ballot(key[r] & 4u),
                                     b3 = __ballot(key[r] & 8u);
            auto lanesWith = [&](uint32_t k) {
                return ((k & 1u) ? b0 : ~b0) & ((k & 2u) ? b1 : ~b1) & ((k & 4u) ? b2 : ~b2) & ((k & 8u) ? b3 : ~b3);
            };
            rank[r] = (uint32_t) __popcll(lanesWith(key[r]) & below);
            if (lane < (uint32_t) kKeys) sCount[(lane * ROUNDS + r) * kWaves + wave] = (uint32_t) __popcll(lanesWith(lane));
        }
        __syncthreads();
        // exclusive prefix over the cells in key-major order
        constexpr int kCellsPerThread = (kCells + kShadeBlock - 1) / kShadeBlock;
        uint32_t start[kCellsPerThread];
#pragma unroll
        for (int j = 0; j < kCellsPerThread; ++j) {
            const uint32_t cell = j * kShadeBlock + tid;
            start[j] = 0;
            if (cell < (uint32_t) kCells)
                for (uint32_t c = 0; c < cell; ++c) start[j] += sCount[c];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kCellsPerThread; ++j)
            if (j * kShadeBlock + tid < (uint32_t) kCells) sCount[j * kShadeBlock + tid] = start[j];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            sOrder[sCount[(key[r] * ROUNDS + r) * kWaves + wave] + rank[r]] = r * kShadeBlock + tid;
        __syncthreads();
    }
}

// ROUNDS: a workgroup sorts ROUNDS x kShadeBlock paths at once and shades them in ROUNDS sweeps of one path per
// thread; the larger sort domain leaves fewer wavefronts that straddle two BSDF models.
// ENV: the scene has emitters other than area lights: an environment emitter (rays that leave the scene are shaded
// too, one more queue stream), point, spot or directional lights.
// RD: the scene has a rough dielectric material.
template <int ROUNDS, bool ENV, bool RD, bool MIX, bool VS = false, bool BLEND = MIX>
__global__ void __launch_bounds__(kShadeBlock, MIX ? MTS_SHADE_WAVES_MIX : MTS_SHADE_WAVES) k_shade(DevScene sc, PassParams pp, PathQueue in, const float4 *hits, PathQueue out,
                                                       ShadowQueue sq, float4 *sampleL, const float2 *samplePos, uint32_t *counters, uint32_t n) {
    constexpr int kWaves = kShadeBlock / 64;
    __shared__ uint32_t sCount[kShadeSortKeys * ROUNDS * kWaves];
    __shared__ uint32_t sOrder[kShadeBlock * ROUNDS];
    const uint32_t first = blockIdx.x * (kShadeBlock * ROUNDS);
    // few area lights: their tables come into workgroup memory (the sort's barriers stand between this and their use)
    __shared__ EmitterTables sEmitters;
    const bool useEmitterTables = MTS_SHADE_EMITTER_LDS && !ENV && sc.emitterTris != 0;
    if (useEmitterTables) sEmitters.fill(sc, threadIdx.x, kShadeBlock);
    sortByBsdfType<ROUNDS>(hits, first, n, sCount, sOrder);
    // reservation cells, one set per sweep parity: a sweep's cells are rewritten two sweeps later, by when every wave has
    // passed the barriers of the sweep in between, so no barrier is needed at the end of a sweep
    __shared__ uint32_t sWaveShadowB[2][kWaves], sWaveNextB[2][kWaves], sBaseB[2][2];
    __shared__ float4 sPark[kShadePark ? 2 * kShadeBlock : 1];
  for (int round = 0; round < ROUNDS; ++round) {
    // The thread's coordinates are made opaque per sweep so that nothing derived from them is carried in registers
    // across the shading code (the register allocator would spill it there).
    uint32_t tidS = threadIdx.x;
    asm volatile("" : "+v"(tidS));
    const uint32_t i = first + sOrder[round * kShadeBlock + tidS];
    ShadeOut o;
    if (i < n) shadeVertex<ENV, RD, MIX, kShadePark, QueuedPath, VS, BLEND>(sc, pp, QueuedPath{ in, i }, hits[i], sampleL, samplePos, o, sPark, &sEmitters, useEmitterTables);
    const bool alive = o.alive, wantShadow = o.wantShadow;
    uint32_t tidL = threadIdx.x;
    asm volatile("" : "+v"(tidL));
    const uint32_t laneL = tidL & 63u, waveL = tidL >> 6;
    uint32_t *sWaveShadow = sWaveShadowB[round & 1], *sWaveNext = sWaveNextB[round & 1], *sBase = sBaseB[round & 1];

    // Queue slots are reserved once per workgroup (two atomics per workgroup, not per wavefront: thousands of
    // wavefronts hammering the same two counters serialise at the L2): wave totals go through LDS, thread 0 reserves
    // the workgroup's ranges, every wave adds the totals of the waves before it.
    const unsigned long long below = (1ull << laneL) - 1ull;
    const unsigned long long mShadow = __ballot(wantShadow), mNext = __ballot(alive);
    if (laneL == 0) { sWaveShadow[waveL] = (uint32_t) __popcll(mShadow); sWaveNext[waveL] = (uint32_t) __popcll(mNext); }
    __syncthreads();
    if (tidL == 0) {
        uint32_t ts = 0, tn = 0;
        for (int w = 0; w < kWaves; ++w) { ts += sWaveShadow[w]; tn += sWaveNext[w]; }
        // both queue lengths live in one 64-bit word (next in the low half, shadow in the high half): one atomic
        static_assert(CNT_NEXT == 0 && CNT_SHADOW == 1, "the two queue counters form one 64-bit word");
        unsigned long long old = 0;
        if (ts | tn) old = atomicAdd(reinterpret_cast<unsigned long long *>(counters), (unsigned long long) tn | ((unsigned long long) ts << 32));
        sBase[0] = (uint32_t) (old >> 32);
        sBase[1] = (uint32_t) old;
    }
    __syncthreads();
    uint32_t si = sBase[0] + (uint32_t) __popcll(mShadow & below), oi = sBase[1] + (uint32_t) __popcll(mNext & below);
    for (uint32_t w = 0; w < waveL; ++w) { si += sWaveShadow[w]; oi += sWaveNext[w]; }
    if (wantShadow) {
        sq.rayO[si] = o.sRayO;
        if (kShadePark) { sq.rayD[si] = sPark[tidL]; sq.contrib[si] = sPark[kShadeBlock + tidL]; }
        else { sq.rayD[si] = o.sRayD; sq.contrib[si] = o.sContrib; }
    }
    if (alive) {
        out.rayO[oi] = o.oRayO; out.rayD[oi] = o.oRayD; out.thr[oi] = o.oThr; out.misc[oi] = o.oMisc; out.bsdfPdf[oi] = o.oPdf;
        if (ENV) out.refCos[oi] = o.oRefCos;
    }
  }
}

// ---- The thin end of a pass: once few paths are left, one launch follows each of them through all its remaining
// bounces (extend, shade, shadow ray, next extend ...) instead of three launches and a host round trip per bounce.
// One lane per path, one workgroup per wavefront; the lanes of a wavefront stay in step bounce by bounce. Same
// functions as the wavefront kernels, hence the same results. counts[0] / counts[1] receive the closest-hit and
// shadow rays traced (the reference's two ray counters).
static constexpr int kTailBlock = 64;
template <bool ENV, bool RD, bool MIX, bool VS = false, bool BLEND = MIX>
__global__ void __launch_bounds__(kTailBlock) k_tail(DevScene sc, PassParams pp, PathQueue in, float4 *sampleL, const float2 *samplePos, uint32_t n,
                                                     uint32_t triShare, unsigned long long *counts) {
    __shared__ uint2 sStack[kLdsStack * kTailBlock];
    LdsUint2 *ldsStack = (LdsUint2 *) sStack;
    const uint32_t i = blockIdx.x * kTailBlock + threadIdx.x;
    bool active = i < n;
    HeldPath path;
    path.ro = path.rd = path.th = make_float4(0, 0, 0, 0); path.mi = make_uint4(0, 0, 0, 0); path.pdf = 0; path.rc = 0;
    if (active) {
        path.ro = in.rayO[i]; path.rd = in.rayD[i]; path.th = in.thr[i]; path.mi = in.misc[i]; path.pdf = in.bsdfPdf[i];
        if (ENV) path.rc = in.refCos[i];
    }
    uint32_t nNormal = 0, nShadow = 0;
    while (__ballot(active)) {
        // extend (Scene::rayIntersect(ray, its))
        LaneRay r;
        r.tg = make_uint2(0u, 0u); r.ng = make_uint2(0u, 0u); r.tv = 0u;
        r.hit.t = kInf; r.hit.u = 0; r.hit.v = 0; r.hit.tri = 0xFFFFFFFFu;
        bool busy = false;
        if (active) {
            ++nNormal;
            float mint, maxt;
            if (rayInterval(sc, mk(path.ro.x, path.ro.y, path.ro.z), mk(path.rd.x, path.rd.y, path.rd.z), path.ro.w, path.rd.w, false, mint, maxt)) {
                laneStart(r, mk(path.ro.x, path.ro.y, path.ro.z), mk(path.rd.x, path.rd.y, path.rd.z), mint, maxt);
                busy = true;
            }
        }
        traceHeldRays<false>(sc, r, busy, ldsStack, kTailBlock, triShare);
        // shade
        ShadeOut o;
        if (active) shadeVertex<ENV, RD, MIX, false, HeldPath, VS, BLEND>(sc, pp, path, make_float4(r.hit.t, r.hit.u, r.hit.v, __uint_as_float(r.hit.tri)), sampleL, samplePos, o);
        // shadow ray of the emitter sample (scene.cpp:839-843), then path.cpp:196-198
        const bool wantShadow = active && o.wantShadow;
        busy = false;
        if (wantShadow) {
            ++nShadow;
            float mint, maxt;
            if (rayInterval(sc, mk(o.sRayO.x, o.sRayO.y, o.sRayO.z), mk(o.sRayD.x, o.sRayD.y, o.sRayD.z), o.sRayO.w, o.sRayD.w, true, mint, maxt)) {
                laneStart(r, mk(o.sRayO.x, o.sRayO.y, o.sRayO.z), mk(o.sRayD.x, o.sRayD.y, o.sRayD.z), mint, maxt);
                busy = true;
            }
        }
        traceHeldRays<true>(sc, r, busy, ldsStack, kTailBlock, triShare);
        if (wantShadow && !busy && (o.sContrib.x != 0 || o.sContrib.y != 0 || o.sContrib.z != 0)) {
            const uint32_t slot = __float_as_uint(o.sContrib.w);
            float4 L = sampleL[slot];
            L.x += o.sContrib.x; L.y += o.sContrib.y; L.z += o.sContrib.z;
            sampleL[slot] = L;
        }
        // next bounce
        active = active && o.alive;
        if (active) { path.ro = o.oRayO; path.rd = o.oRayD; path.th = o.oThr; path.mi = o.oMisc; path.pdf = o.oPdf; path.rc = o.oRefCos; }
    }
    // one pair of atomics per wavefront
    for (int off = 32; off > 0; off >>= 1) { nNormal += __shfl_down(nNormal, off); nShadow += __shfl_down(nShadow, off); }
    if (threadIdx.x == 0) { atomicAdd(&counts[0], (unsigned long long) nNormal); atomicAdd(&counts[1], (unsigned long long) nShadow); }
}

// ---- K4b: the value of an unoccluded emitter sample joins the path's radiance (path.cpp:196-198). One thread per
// shadow ray in queue order; a sample slot occurs at most once per bounce, so plain loads and stores suffice.
__global__ void __launch_bounds__(256) k_shadow_accumulate(uint32_t n, const uint8_t *occluded, const float4 *contrib, float4 *sampleL) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || occluded[i]) return;
    const float4 c = contrib[i];
    if (c.x == 0 && c.y == 0 && c.z == 0) return;
    const uint32_t slot = __float_as_uint(c.w);
    float4 L = sampleL[slot];
    L.x += c.x; L.y += c.y; L.z += c.z;
    sampleL[slot] = L;
}

// ---- MIDirectIntegrator::Li (src/integrators/direct/direct.cpp:146-309) in two kernels around the traversals -------
// k_direct_shade: one thread per camera sample (a workgroup takes 16 x 256 of them, ordered by the BSDF type of their
// hits). It shades the camera vertex, writes `nEmitter` shadow rays to
// entries [i * nEmitter, (i + 1) * nEmitter) of the shadow queue and `nBsdf` BSDF-sampled rays to entries
// [i * nBsdf, ...) of the output queue. Entries keep their place whether or not they carry a ray (an idle entry has an
// empty ray interval), so that k_direct_gather can add a sample's contributions in a fixed order without atomics.
// Sample order per camera sample: the emitter samples' 2-D numbers, then per BSDF sample its 2-D number followed by
// what the BSDF itself takes from the sampler. counts[0] += BSDF rays, counts[1] += shadow rays that the reference traces.
static constexpr uint32_t PF_IDLE = 0x8000u;

static constexpr int kDirectRounds = 16;    // sweeps per workgroup, sorted by BSDF type like k_shade
#ifndef MTS_DIRECT_WAVES
#define MTS_DIRECT_WAVES 1
#endif
#ifndef MTS_DIRECT_WAVES_PLAIN
#define MTS_DIRECT_WAVES_PLAIN 1
#endif
// ENV, RD, MIX, BLEND: the same split as k_shade's variants (the code of those features costs registers whether or not a scene
// uses them); the variant without any of them runs at whatever occupancy its registers allow
template <bool ENV, bool RD, bool MIX, bool BLEND = MIX>
__global__ void __launch_bounds__(kShadeBlock, (ENV || RD || MIX) ? MTS_DIRECT_WAVES : MTS_DIRECT_WAVES_PLAIN) k_direct_shade(DevScene sc, PassParams pp, PathQueue in, const float4 *hits, PathQueue out, ShadowQueue sq,
                                                             float4 *sampleL, const float2 *samplePos, unsigned long long *counts, uint32_t n, int nEmitter, int nBsdf) {
    __shared__ uint32_t sCount[kShadeSortKeys * kDirectRounds * (kShadeBlock / 64)];
    __shared__ uint32_t sOrder[kShadeBlock * kDirectRounds];
    const uint32_t first = blockIdx.x * (kShadeBlock * kDirectRounds);
    sortByBsdfType<kDirectRounds>(hits, first, n, sCount, sOrder);
    uint32_t nNormal = 0, nShadow = 0;
  for (int round = 0; round < kDirectRounds; ++round) {
    const uint32_t i = first + sOrder[round * kShadeBlock + threadIdx.x];
    if (i < n) {
        const float4 idleO = make_float4(0.0f, 0.0f, 0.0f, 1.0f), idleD = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
        const float4 hit = hits[i], rd = in.rayD[i];
        const uint4 misc = in.misc[i];
        const uint32_t slot = misc.x, triWord = __float_as_uint(hit.w);
        const V3 rayD = mk(rd.x, rd.y, rd.z);
        bool shade = false;
        float4 L = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        Isect its;
        DevMaterial bsdf;
        Rng rng;
        if (triWord == kMiss) {
            // direct.cpp:156-163: the environment seen directly, alpha 0
            if (ENV && sc.envEmitter >= 0 && !sc.hideEmitters) {
                const V3 e = sc.envIsMap ? cameraEnvironmentRadiance(sc, pp, samplePos[slot], rayD) : ld3(sc.envRadiance);
                L.x = e.x; L.y = e.y; L.z = e.z;
            }
        } else {
            int px, py, sIdx;
            const uint64_t key = sampleKey(sc, pp, slot, px, py, sIdx);
            rng.inc = (key << 1) | 1u;
            rng.state = (uint64_t) misc.z | ((uint64_t) misc.w << 32);
            fillIntersection(sc, triWord & kTriIndexMask, hit.y, hit.z, rayD, its);
            loadMaterial(sc, its.material, bsdf);
            L.w = 1.0f;
            if (its.emitter >= 0 && !sc.hideEmitters && dot(its.sh.n, -rayD) > 0) {    // direct.cpp:166-167
                const V3 e = ld3(sc.emitters[its.emitter].radiance);
                L.x = e.x; L.y = e.y; L.z = e.z;
            }
            shade = !(sc.strictNormals && dot(rayD, its.geoN) * its.wi.z >= 0);         // direct.cpp:175-187
        }
        sampleL[slot] = L;

        uint32_t bsdfType = 0;
        bool flipped = false;
        V3 wiLocal = mk(0.0f), refN = mk(0.0f), wiB = mk(0.0f);
        SurfUV uv = surfUV(0.0f, 0.0f);
        bool framed = false;
        Frame3 pf;
        if (shade) {
            bsdfType = bsdf.typeFlags;
            uv = surfUV(its.uvx, its.uvy);
            if (MIX && sc.materialTextures && (sc.materialTextures[its.material].w & 1u))     // its.getBSDF(ray) (direct.cpp:176)
                cameraUVPartials(sc, pp, samplePos[slot], its, in.rayO[i], rayD, uv);
            flipped = resolveTwoSided(sc, its.material, its.wi.z, bsdf);
            wiLocal = flipped ? mirrorZ(its.wi) : its.wi;
            if (MIX) applyTextures(sc, its.material, uv, bsdf);
            if (MIX && sc.materialTextures) {      // bumpmap / normalmap around the answering record
                const uint32_t fw = sc.materialTextures[its.material].w;
                if (fw & 6u) { framed = true; pf = perturbedFrame(sc, fw, its.sh, its.geoN, its.dpdu, its.dpdv, its.uvx, its.uvy); }
            }
            wiB = framed ? toLocal(pf, toWorld(its.sh, wiLocal)) : wiLocal;
            if ((bsdfType & (ETransmission | EBackSide)) == 0) refN = its.sh.n;
        }
        // MIDirectIntegrator::configure (direct.cpp:128-136)
        const float weightLum = 1 / (float) nEmitter;
        const float fracBSDF = nBsdf / (float) (nEmitter + nBsdf), fracLum = nEmitter / (float) (nEmitter + nBsdf);

        // emitter sampling (direct.cpp:210-245)
        for (int k = 0; k < nEmitter; ++k) {
            const size_t e = (size_t) i * nEmitter + k;
            float4 sO = idleO, sD = idleD, sC = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(slot));
            if (shade) {
                const float sx = rng.next1D(), sy = rng.next1D();
                if (bsdfType & ESmooth) {
                    DirectRec dRec;
                    const V3 value = sampleEmitterDirect<ENV>(sc, its.p, refN, sx, sy, dRec);
                    if (dRec.pdf != 0) {
                        V3 c = mk(0.0f);
                        const V3 wo = toLocal(its.sh, dRec.d);
                        const V3 woSide = flipped ? mirrorZ(wo) : wo;
                        const V3 woB = framed ? toLocal(pf, toWorld(its.sh, woSide)) : woSide;
                        const bool sameSide = !framed || woSide.z * woB.z > 0;
                        const V3 bsdfVal = sameSide ? bsdfEvalAny<RD, BLEND>(sc, bsdf, wiB, woB, uv) : mk(0.0f);
                        if (!isZero(bsdfVal) && (!sc.strictNormals || dot(its.geoN, dRec.d) * wo.z > 0)) {
                            const float bPdf = ENV && dRec.delta ? 0.0f : bsdfPdfAny<RD, BLEND>(sc, bsdf, wiB, woB, uv);
                            c = value * bsdfVal * (miWeight(dRec.pdf * fracLum, bPdf * fracBSDF) * weightLum);
                        }
                        ++nShadow;
                        sO = make_float4(its.p.x, its.p.y, its.p.z, kEpsilon);
                        sD = make_float4(dRec.d.x, dRec.d.y, dRec.d.z, dRec.dist * (1 - kShadowEpsilon));
                        sC = make_float4(c.x, c.y, c.z, __uint_as_float(slot));
                    }
                }
            }
            sq.rayO[e] = sO; sq.rayD[e] = sD; sq.contrib[e] = sC;
        }

        // BSDF sampling (direct.cpp:251-275): the ray; what it finds is weighted in k_direct_gather
        for (int k = 0; k < nBsdf; ++k) {
            const size_t e = (size_t) i * nBsdf + k;
            float4 oO = idleO, oD = idleD, oT = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            uint32_t flags = PF_IDLE;
            float oPdf = 0.0f, oRefCos = -2.0f;
            if (shade) {
                const float bx = rng.next1D(), by = rng.next1D();
                const float bu = RD && bsdfDrawsFromSampler<BLEND>(bsdf, bx) ? rng.next1D() : 0.0f;
                V3 woLocal; float bPdf, bEta; uint32_t sampledType;
                V3 bsdfVal = bsdfSampleAny<RD, BLEND>(sc, bsdf, wiB, bx, by, bu, woLocal, bPdf, bEta, sampledType, uv);
                if (framed && !isZero(bsdfVal)) {
                    const V3 pwo = woLocal;
                    woLocal = toLocal(its.sh, toWorld(pf, pwo));
                    if (woLocal.z * pwo.z <= 0) bsdfVal = mk(0.0f);
                }
                if (flipped && !isZero(bsdfVal) && bPdf != 0) woLocal.z = -woLocal.z;
                if (!isZero(bsdfVal)) {
                    const V3 wo = toWorld(its.sh, woLocal);
                    if (!(sc.strictNormals && dot(its.geoN, wo) * woLocal.z <= 0)) {
                        flags = 0;
                        if (sampledType & EDelta) flags |= PF_PREV_DELTA;
                        if (dot(wo, refN) >= 0) flags |= PF_REFN_OK;
                        oO = make_float4(its.p.x, its.p.y, its.p.z, kEpsilon);
                        oD = make_float4(wo.x, wo.y, wo.z, kInf);
                        oT = make_float4(bsdfVal.x, bsdfVal.y, bsdfVal.z, 1.0f);
                        oPdf = bPdf;
                        oRefCos = isZero(refN) ? -2.0f : dot(wo, refN);
                        ++nNormal;
                    }
                }
            }
            out.rayO[e] = oO; out.rayD[e] = oD; out.thr[e] = oT; out.bsdfPdf[e] = oPdf; out.refCos[e] = oRefCos;
            out.misc[e] = make_uint4(slot, 1u | (flags << 16), 0u, 0u);
        }
    }
  }
    for (int off = 32; off > 0; off >>= 1) { nNormal += __shfl_down(nNormal, off); nShadow += __shfl_down(nShadow, off); }
    if ((threadIdx.x & 63) == 0 && (nNormal | nShadow)) { atomicAdd(&counts[0], (unsigned long long) nNormal); atomicAdd(&counts[1], (unsigned long long) nShadow); }
}

// k_direct_gather: one thread per camera sample adds, in entry order, the unoccluded emitter samples and what the
// BSDF-sampled rays found (direct.cpp:276-305), weighted against the emitter-sampling density.
__global__ void __launch_bounds__(256) k_direct_gather(DevScene sc, PathQueue in, PathQueue rays, const float4 *hits, ShadowQueue sq, const uint8_t *occluded,
                                                      float4 *sampleL, uint32_t n, int nEmitter, int nBsdf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = in.misc[i].x;
    float4 L = sampleL[slot];
    for (int k = 0; k < nEmitter; ++k) {
        const size_t e = (size_t) i * nEmitter + k;
        if (occluded[e]) continue;
        const float4 c = sq.contrib[e];
        L.x += c.x; L.y += c.y; L.z += c.z;
    }
    const float weightBSDF = 1 / (float) nBsdf;
    const float fracBSDF = nBsdf / (float) (nEmitter + nBsdf), fracLum = nEmitter / (float) (nEmitter + nBsdf);
    for (int k = 0; k < nBsdf; ++k) {
        const size_t e = (size_t) i * nBsdf + k;
        const uint32_t flags = rays.misc[e].y >> 16;
        if (flags & PF_IDLE) continue;
        const float4 hit = hits[e], rd = rays.rayD[e], th = rays.thr[e];
        const V3 rayD = mk(rd.x, rd.y, rd.z);
        const uint32_t triWord = __float_as_uint(hit.w);
        V3 value;
        float lumPdf = 0.0f;
        if (triWord != kMiss) {
            Isect its;
            fillIntersection(sc, triWord & kTriIndexMask, hit.y, hit.z, rayD, its);
            if (its.emitter < 0) continue;
            const DevEmitter em = sc.emitters[its.emitter];
            value = dot(its.sh.n, -rayD) > 0 ? ld3(em.radiance) : mk(0.0f);
            if (!(flags & PF_PREV_DELTA)) {     // pdfEmitterDirect (scene.cpp:949-952), as in the path integrator
                if ((flags & PF_REFN_OK) && dot(rayD, its.sh.n) < 0) lumPdf = em.invArea * (hit.x * hit.x) / fabsf(dot(rayD, its.sh.n));
                lumPdf *= em.pdfDiscrete;
            }
        } else {
            if (sc.envEmitter < 0) continue;    // hideEmitters only hides what a null interaction would show (direct.cpp:288)
            const float4 ro = rays.rayO[e];
            float farT;
            if (!environmentExit(sc, mk(ro.x, ro.y, ro.z), rayD, farT)) continue;
            value = environmentRadiance(sc, rayD);
            if (!(flags & PF_PREV_DELTA)) lumPdf = environmentPdf(sc, rayD, rays.refCos[e]);
        }
        const V3 c = value * mk(th.x, th.y, th.z) * (miWeight(rays.bsdfPdf[e] * fracBSDF, lumPdf * fracLum) * weightBSDF);
        L.x += c.x; L.y += c.y; L.z += c.z;
    }
    sampleL[slot] = L;
}

// ---- K6 film: ImageBlock::put in gather form (include/mitsuba/render/imageblock.h:144-189). One thread per film
// pixel of the pass rectangle grown by the filter border; it visits the samples of the pixels whose footprint can
// reach it, in a fixed order, so the accumulation is deterministic and needs no atomics. A sample with a
// non-finite or negative channel is dropped as a whole (imageblock.h:147-151).
__global__ void __launch_bounds__(256) k_film(DevScene sc, PassParams pp, const float4 *sampleL, const float2 *samplePos, float *film) {
    const int border = (int) ceilf(sc.filterRadius - 0.5f);     // rfilter.cpp:50
    const int gx0 = max(pp.x0 - border, 0), gy0 = max(pp.y0 - border, 0);
    const int gx1 = min(pp.x0 + pp.w + border, sc.cropW), gy1 = min(pp.y0 + pp.h + border, sc.cropH);
    const int gw = gx1 - gx0, gh = gy1 - gy0;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= gw * gh) return;
    const int x = gx0 + idx % gw, y = gy0 + idx / gw;
    const float scaleFactor = 31.0f / sc.filterRadius;          // rfilter.cpp:49 (MTS_FILTER_RESOLUTION = 31)
    float acc[5] = { 0, 0, 0, 0, 0 };
    const int qx0 = max(x - border, pp.x0), qx1 = min(x + border, pp.x0 + pp.w - 1);
    const int qy0 = max(y - border, pp.y0), qy1 = min(y + border, pp.y0 + pp.h - 1);
    for (int qy = qy0; qy <= qy1; ++qy)
        for (int qx = qx0; qx <= qx1; ++qx) {
            const uint32_t base = ((uint32_t) (qy - pp.y0) * (uint32_t) pp.w + (uint32_t) (qx - pp.x0)) * (uint32_t) pp.spp;
            for (int s = 0; s < pp.spp; ++s) {
                const float4 L = sampleL[base + s];
                const float2 pos = samplePos[base + s];
                if (!(isfinite(L.x) && isfinite(L.y) && isfinite(L.z) && isfinite(L.w)) || L.x < 0 || L.y < 0 || L.z < 0 || L.w < 0) continue;
                const float px = pos.x - 0.5f, py = pos.y - 0.5f;
                // pixels outside [ceil(p - r), floor(p + r)] are not touched by the reference
                if ((float) x < ceilf(px - sc.filterRadius) || (float) x > floorf(px + sc.filterRadius) ||
                    (float) y < ceilf(py - sc.filterRadius) || (float) y > floorf(py + sc.filterRadius)) continue;
                const float wx = sc.filterValues[min((int) fabsf(((float) x - px) * scaleFactor), 31)];
                const float wy = sc.filterValues[min((int) fabsf(((float) y - py) * scaleFactor), 31)];
                const float w = wx * wy;
                acc[0] += w * L.x; acc[1] += w * L.y; acc[2] += w * L.z; acc[3] += w * L.w; acc[4] += w * 1.0f;
            }
        }
    float *dst = film + ((size_t) y * sc.cropW + x) * 5;
#pragma unroll
    for (int c = 0; c < 5; ++c) dst[c] += acc[c];
}

// LDS-tiled form of the same gather: a workgroup owns a 16x16 tile of film pixels, stages the samples of the
// (16+2*border)^2 source pixels around it in LDS (a few samples per pixel at a time, sample-major so that lanes with
// consecutive x read consecutive LDS words), and every thread walks its neighbourhood from LDS. Each sample is read
// from memory once per tile that can see it instead of once per film pixel. Dropped samples and source pixels outside
// the pass rectangle are staged with position +inf, which fails the footprint test.
static constexpr int kFilmTile = 16;
#ifndef MTS_FILM_LDS_ENTRIES
#define MTS_FILM_LDS_ENTRIES 2048
#endif
static constexpr int kFilmLdsEntries = MTS_FILM_LDS_ENTRIES;

__global__ void __launch_bounds__(256) k_film_tiled(DevScene sc, PassParams pp, const float4 *sampleL, const float2 *samplePos, float *film,
                                                    int border, int gx0, int gy0, int gw, int gh, int tilesX, int chunk) {
    __shared__ float4 sL[kFilmLdsEntries];
    __shared__ float2 sP[kFilmLdsEntries];
    __shared__ float sFilter[32];
    const int tid = threadIdx.x;
    if (tid < 32) sFilter[tid] = sc.filterValues[tid];
    const int tx = blockIdx.x % tilesX, ty = blockIdx.x / tilesX;
    const int ox = gx0 + tx * kFilmTile, oy = gy0 + ty * kFilmTile;       // first film pixel of the tile
    const int rw = kFilmTile + 2 * border, rs = rw * rw;                  // staged region (side, pixels)
    const int sx0 = ox - border, sy0 = oy - border;
    const int lx = tid % kFilmTile, ly = tid / kFilmTile;
    const int x = ox + lx, y = oy + ly;
    const bool inside = x < gx0 + gw && y < gy0 + gh;
    const float scaleFactor = 31.0f / sc.filterRadius;
    const float fx = (float) x, fy = (float) y, r = sc.filterRadius;
    float acc[5] = { 0, 0, 0, 0, 0 };
    const float kInfPos = __builtin_inff();

    for (int s0 = 0; s0 < pp.spp; s0 += chunk) {
        const int cs = min(chunk, pp.spp - s0);
        __syncthreads();
        for (int e = tid; e < rs * cs; e += 256) {
            const int s = e / rs, p = e - s * rs;
            const int qx = sx0 + p % rw, qy = sy0 + p / rw;
            float4 L = make_float4(0, 0, 0, 0);
            float2 pos = make_float2(kInfPos, kInfPos);
            if (qx >= pp.x0 && qx < pp.x0 + pp.w && qy >= pp.y0 && qy < pp.y0 + pp.h) {
                const uint32_t slot = ((uint32_t) (qy - pp.y0) * (uint32_t) pp.w + (uint32_t) (qx - pp.x0)) * (uint32_t) pp.spp + (uint32_t) (s0 + s);
                L = sampleL[slot];
                const bool ok = isfinite(L.x) && isfinite(L.y) && isfinite(L.z) && isfinite(L.w) && L.x >= 0 && L.y >= 0 && L.z >= 0 && L.w >= 0;
                if (ok) { const float2 q = samplePos[slot]; pos = make_float2(q.x - 0.5f, q.y - 0.5f); }
            }
            sL[e] = L; sP[e] = pos;
        }
        __syncthreads();
        if (inside) {
            for (int dy = 0; dy <= 2 * border; ++dy)
                for (int dx = 0; dx <= 2 * border; ++dx) {
                    const int p = (ly + dy) * rw + (lx + dx);
                    for (int s = 0; s < cs; ++s) {
                        const float2 pos = sP[s * rs + p];
                        // x in [ceil(p - r), floor(p + r)] for an integer x is p - r <= x <= p + r
                        if (fx < pos.x - r || fx > pos.x + r || fy < pos.y - r || fy > pos.y + r) continue;
                        const float4 L = sL[s * rs + p];
                        const float wx = sFilter[min((int) fabsf((fx - pos.x) * scaleFactor), 31)];
                        const float wy = sFilter[min((int) fabsf((fy - pos.y) * scaleFactor), 31)];
                        const float w = wx * wy;
                        acc[0] += w * L.x; acc[1] += w * L.y; acc[2] += w * L.z; acc[3] += w * L.w; acc[4] += w * 1.0f;
                    }
                }
        }
    }
    if (inside) {
        float *dst = film + ((size_t) y * sc.cropW + x) * 5;
#pragma unroll
        for (int c = 0; c < 5; ++c) dst[c] += acc[c];
    }
}

// After a device-side BVH build: every triangle record learns the BSDF type of its triangle's material (bits 28..30 of
// its triangle word), which the traversal hands on with the hit. The host-built path does the same on the host.
__global__ void k_tag_triangle_records(float4 *tris, uint32_t triStride, uint32_t n, const float4 *shadeTris, const DevMaterial *materials) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 *rec = tris + (size_t) triStride * i;
    const uint32_t tri = __float_as_uint(rec[2].z) & kTriIndexMask;
    const uint32_t mat = __float_as_uint(shadeTris[kShadeTriStride * (size_t) tri].w) & 0xffffffu;
    const int type = materials[mat].type;
    rec[2].z = __uint_as_float(tri | ((uint32_t) (type < 7 ? type : 7) << 28));      // mixtures share the last sorting key
}

// ---- batch kernels behind the parity entry points -------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fill_isect(DevScene sc, const float *rays, const float4 *hits, uint32_t n, mtsamd_intersection *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 h = hits[i];
    const uint32_t tri = __float_as_uint(h.w) == kMiss ? kMiss : (__float_as_uint(h.w) & kTriIndexMask);
    mtsamd_intersection r;
    memset(&r, 0, sizeof(r));
    if (tri == kMiss) {
        r.t = kInf; r.shape = r.prim = kMiss;
    } else {
        const float *ry = rays + 8 * (size_t) i;
        Isect its;
        fillIntersection(sc, tri, h.y, h.z, mk(ry[4], ry[5], ry[6]), its);
        r.t = h.x; r.u = h.y; r.v = h.z; r.shape = its.shape; r.prim = tri - sc.shapes[its.shape].firstTri;
        r.p[0] = its.p.x; r.p[1] = its.p.y; r.p[2] = its.p.z;
        r.geo_n[0] = its.geoN.x; r.geo_n[1] = its.geoN.y; r.geo_n[2] = its.geoN.z;
        r.sh_n[0] = its.sh.n.x; r.sh_n[1] = its.sh.n.y; r.sh_n[2] = its.sh.n.z;
        r.sh_s[0] = its.sh.s.x; r.sh_s[1] = its.sh.s.y; r.sh_s[2] = its.sh.s.z;
        r.sh_t[0] = its.sh.t.x; r.sh_t[1] = its.sh.t.y; r.sh_t[2] = its.sh.t.z;
        r.wi[0] = its.wi.x; r.wi[1] = its.wi.y; r.wi[2] = its.wi.z;
        r.uv[0] = its.uvx; r.uv[1] = its.uvy;
    }
    out[i] = r;
}

__global__ void k_hits_to_records(DevScene sc, const float4 *hits, uint32_t n, mtsamd_hit *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 h = hits[i];
    const uint32_t tri = __float_as_uint(h.w) == kMiss ? kMiss : (__float_as_uint(h.w) & kTriIndexMask);
    mtsamd_hit r;
    r.t = h.x; r.u = h.y; r.v = h.z;
    if (tri == kMiss) { r.shape = r.prim = kMiss; }
    else { r.shape = __float_as_uint(sc.shadeTris[kShadeTriStride * (size_t) tri + 2].w); r.prim = tri - sc.shapes[r.shape].firstTri; }
    out[i] = r;
}

// mode 0: eval -> out[3n]; 1: pdf -> out[n]
__global__ void k_bsdf_evalpdf(DevScene sc, int material, float2 uvIn, int mode, uint32_t n, const float *wi, const float *wo, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DevMaterial m;
    uint32_t index = (uint32_t) material;
    loadMaterial(sc, index, m);
    V3 a = ld3(wi, i), b = ld3(wo, i);
    if (resolveTwoSided(sc, index, a.z, m)) { a = mirrorZ(a); b = mirrorZ(b); }
    const SurfUV uv = surfUV(uvIn.x, uvIn.y);
    applyTextures(sc, index, uv, m);
    bool zero = false;
    if (sc.materialTextures && (sc.materialTextures[index].w & 6u)) {       // a query without an intersection: the canonical local frame
        Frame3 sh; sh.s = mk(1, 0, 0); sh.t = mk(0, 1, 0); sh.n = mk(0, 0, 1);
        const Frame3 pf = perturbedFrame(sc, sc.materialTextures[index].w, sh, sh.n, sh.s, sh.t, uv.u, uv.v);
        const V3 pb = toLocal(pf, b);
        zero = b.z * pb.z <= 0;
        a = toLocal(pf, a); b = pb;
    }
    if (mode == 0) { const V3 r = zero ? mk(0.0f) : bsdfEvalAny(sc, m, a, b, uv); out[3 * (size_t) i] = r.x; out[3 * (size_t) i + 1] = r.y; out[3 * (size_t) i + 2] = r.z; }
    else out[i] = zero ? 0.0f : bsdfPdfAny(sc, m, a, b, uv);
}

__global__ void k_bsdf_sample(DevScene sc, int material, float2 uvIn, uint32_t n, const float *wi, const float *sample, const float *sampler1D,
                              float *wo, float *weight, float *pdf, float *eta, uint32_t *sampledType) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DevMaterial m;
    uint32_t index = (uint32_t) material;
    loadMaterial(sc, index, m);
    V3 a = ld3(wi, i);
    const bool flipped = resolveTwoSided(sc, index, a.z, m);
    if (flipped) a = mirrorZ(a);
    const SurfUV uv = surfUV(uvIn.x, uvIn.y);
    applyTextures(sc, index, uv, m);
    V3 o; float p, e; uint32_t t;
    const bool framed = sc.materialTextures && (sc.materialTextures[index].w & 6u);
    Frame3 pf;
    if (framed) {
        Frame3 sh; sh.s = mk(1, 0, 0); sh.t = mk(0, 1, 0); sh.n = mk(0, 0, 1);
        pf = perturbedFrame(sc, sc.materialTextures[index].w, sh, sh.n, sh.s, sh.t, uv.u, uv.v);
        a = toLocal(pf, a);
    }
    V3 w = bsdfSampleAny(sc, m, a, sample[2 * (size_t) i], sample[2 * (size_t) i + 1], sampler1D ? sampler1D[i] : 0.5f, o, p, e, t, uv);
    if (framed && !isZero(w)) {
        const V3 pwo = o;
        o = toWorld(pf, pwo);
        if (o.z * pwo.z <= 0) w = mk(0.0f);
    }
    if (flipped && !isZero(w) && p != 0) o.z = -o.z;
    wo[3 * (size_t) i] = o.x; wo[3 * (size_t) i + 1] = o.y; wo[3 * (size_t) i + 2] = o.z;
    weight[3 * (size_t) i] = w.x; weight[3 * (size_t) i + 1] = w.y; weight[3 * (size_t) i + 2] = w.z;
    pdf[i] = p; eta[i] = e; sampledType[i] = t;
}

// Scene::evalEnvironment and the environment emitter's pdfDirect times its discrete probability for n directions
__global__ void k_env_eval(DevScene sc, uint32_t n, const float *d, const float *refN, float *value, float *pdf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 dir = ld3(d, i), rn = ld3(refN, i);
    const V3 v = environmentRadiance(sc, dir);
    value[3 * (size_t) i] = v.x; value[3 * (size_t) i + 1] = v.y; value[3 * (size_t) i + 2] = v.z;
    pdf[i] = environmentPdf(sc, dir, isZero(rn) ? -2.0f : dot(dir, rn));
}

// Texture::eval(its) for textures on their own: uv n*2, duv n*4 or null
__global__ void k_texture_eval(DevScene sc, const DevTexture *textures, int texture, uint32_t n, const float *uv, const float *duv, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SurfUV q = surfUV(uv[2 * (size_t) i], uv[2 * (size_t) i + 1]);
    if (duv) { q.filtered = 1u; q.dudx = duv[4 * (size_t) i]; q.dvdx = duv[4 * (size_t) i + 1]; q.dudy = duv[4 * (size_t) i + 2]; q.dvdy = duv[4 * (size_t) i + 3]; }
    const V3 v = textureEval(sc, textures[texture], q);
    out[3 * (size_t) i] = v.x; out[3 * (size_t) i + 1] = v.y; out[3 * (size_t) i + 2] = v.z;
}

__global__ void k_emitter_sample(DevScene sc, uint32_t n, const float *ref, const float *refN, const float *sample, float *d, float *dist,
                                 float *pdf, float *value) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DirectRec dRec;
    const V3 v = sampleEmitterDirect(sc, ld3(ref, i), ld3(refN, i), sample[2 * (size_t) i], sample[2 * (size_t) i + 1], dRec);
    d[3 * (size_t) i] = dRec.d.x; d[3 * (size_t) i + 1] = dRec.d.y; d[3 * (size_t) i + 2] = dRec.d.z;
    value[3 * (size_t) i] = v.x; value[3 * (size_t) i + 1] = v.y; value[3 * (size_t) i + 2] = v.z;
    dist[i] = dRec.dist; pdf[i] = dRec.pdf;
}

__global__ void k_camera_rays(DevScene sc, uint32_t n, const float *pos, float *rays) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 o, d; float mint, maxt;
    cameraRay(sc, pos[2 * (size_t) i], pos[2 * (size_t) i + 1], o, d, mint, maxt);
    float *r = rays + 8 * (size_t) i;
    r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = mint; r[4] = d.x; r[5] = d.y; r[6] = d.z; r[7] = maxt;
}

__global__ void k_rng_stream(uint32_t seed, uint64_t key, uint32_t n, float *out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Rng rng;
    rng.init(seed, key);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng.next1D();
}

// ImageBlock::put for explicit samples, gather form: each film pixel scans all samples in order.
__global__ void k_film_put(DevScene sc, uint32_t n, const float *pos, const float *value, float *film) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= sc.cropW * sc.cropH) return;
    const int x = idx % sc.cropW, y = idx / sc.cropW;
    const float scaleFactor = 31.0f / sc.filterRadius;
    float acc[5] = { 0, 0, 0, 0, 0 };
    for (uint32_t s = 0; s < n; ++s) {
        const float *L = value + 4 * (size_t) s;
        if (!(isfinite(L[0]) && isfinite(L[1]) && isfinite(L[2]) && isfinite(L[3])) || L[0] < 0 || L[1] < 0 || L[2] < 0 || L[3] < 0) continue;
        const float px = pos[2 * (size_t) s] - 0.5f, py = pos[2 * (size_t) s + 1] - 0.5f;
        if ((float) x < ceilf(px - sc.filterRadius) || (float) x > floorf(px + sc.filterRadius) ||
            (float) y < ceilf(py - sc.filterRadius) || (float) y > floorf(py + sc.filterRadius)) continue;
        const float wx = sc.filterValues[min((int) fabsf(((float) x - px) * scaleFactor), 31)];
        const float wy = sc.filterValues[min((int) fabsf(((float) y - py) * scaleFactor), 31)];
        const float w = wx * wy;
        acc[0] += w * L[0]; acc[1] += w * L[1]; acc[2] += w * L[2]; acc[3] += w * L[3]; acc[4] += w * 1.0f;
    }
    float *dst = film + (size_t) idx * 5;
    for (int c = 0; c < 5; ++c) dst[c] += acc[c];
}

// copies per-sample radiance and positions out in pixel-major order for mtsamd_li
__global__ void k_copy_samples(const float4 *sampleL, const float2 *samplePos, uint32_t n, float *out, float *pos, size_t outOffset) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 L = sampleL[i];
    const float2 p = samplePos[i];
    float *o = out + 4 * (outOffset + i);
    o[0] = L.x; o[1] = L.y; o[2] = L.z; o[3] = L.w;
    pos[2 * (outOffset + i)] = p.x; pos[2 * (outOffset + i) + 1] = p.y;
}

}  // namespace mtsamd
