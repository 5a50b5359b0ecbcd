// TEST INFRASTRUCTURE ONLY -- CPU oracle: scene set-up, emitter sampling, the MIPathTracer::Li
// loop, perspective camera, film accumulation and a block-parallel renderer, restated from the
// reference files cited per function.
#include "bsdf.h"
#include "rng.h"
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>

namespace orc {

// ---- DiscreteDistribution (include/mitsuba/core/pmf.h:101-188) ----------------------------------
float DiscreteDistribution::normalize() {
    sum = cdf[cdf.size() - 1];
    if (sum > 0) {
        normalization = 1.0f / sum;
        for (size_t i = 1; i < cdf.size(); ++i) cdf[i] *= normalization;
        cdf[cdf.size() - 1] = 1.0f;
    } else {
        normalization = 0.0f;
    }
    return sum;
}

size_t DiscreteDistribution::sample(float v) const {
    std::vector<float>::const_iterator entry = std::lower_bound(cdf.begin(), cdf.end(), v);
    size_t index = std::min(cdf.size() - 2, (size_t) std::max((ptrdiff_t) 0, entry - cdf.begin() - 1));
    while (operator[](index) == 0 && index < cdf.size() - 1) ++index;
    return index;
}

size_t DiscreteDistribution::sampleReuse(float &v) const {
    size_t index = sample(v);
    v = (v - cdf[index]) / (cdf[index + 1] - cdf[index]);
    return index;
}

size_t DiscreteDistribution::sampleReuse(float &v, float &pdf) const {
    size_t index = sample(v);
    pdf = operator[](index);
    v = (v - cdf[index]) / (cdf[index + 1] - cdf[index]);
    return index;
}

// ---- 4x4 helpers for the camera (src/libcore/transform.cpp:99-123, perspective.cpp:126-157) -----
static void matMul(const float *a, const float *b, float *out) {
    float r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0;
            for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
            r[i * 4 + j] = s;
        }
    memcpy(out, r, sizeof(r));
}

static bool matInvert(const float *m, float *out) {
    // Gauss-Jordan elimination with partial pivoting (Matrix4x4::invert, src/libcore/matrix.cpp)
    float a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = m[i * 4 + j]; a[i][j + 4] = (i == j) ? 1.0f : 0.0f; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::abs(a[r][c]) > std::abs(a[piv][c])) piv = r;
        if (a[piv][c] == 0) return false;
        if (piv != c) for (int j = 0; j < 8; ++j) std::swap(a[c][j], a[piv][j]);
        float inv = 1.0f / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            float f = a[r][c];
            if (f != 0) for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[i * 4 + j] = a[i][j + 4];
    return true;
}

static void matScale(float x, float y, float z, float *m) {
    memset(m, 0, 64); m[0] = x; m[5] = y; m[10] = z; m[15] = 1;
}
static void matTranslate(float x, float y, float z, float *m) {
    matScale(1, 1, 1, m); m[3] = x; m[7] = y; m[11] = z;
}

// ---- Scene::build ----------------------------------------------------------------------------------
// EnvironmentMap::EnvironmentMap / configure (src/emitters/envmap.cpp:105-189,260-329)
const char *Scene::buildEnvMap(const orc_emitter &em) {
    if (!em.pixels || em.width <= 0 || em.height <= 0) return "envmap: invalid image";
    if (std::max(em.width, em.height) > 0xFFFF) return "Environment maps images must be smaller than 65536 pixels in width and height";
    auto env = std::make_shared<EnvMap>();
    env->mip = std::make_shared<MIPMap>(em.pixels, em.width, em.height, 3, BC_REPEAT, BC_CLAMP, MIP_EWA, 10.0f, std::numeric_limits<float>::infinity());
    env->w = em.width; env->h = em.height; env->scale = em.scale;
    const int w = env->w, h = env->h;
    env->cdfCols.resize((size_t) (w + 1) * h); env->cdfRows.resize(h + 1); env->rowWeights.resize(h);
    size_t colPos = 0, rowPos = 0;
    float rowSum = 0.0f;
    env->cdfRows[rowPos++] = 0;
    for (int y = 0; y < h; ++y) {
        float colSum = 0;
        env->cdfCols[colPos++] = 0;
        for (int x = 0; x < w; ++x) {
            colSum += luminance(env->mip->evalTexel(0, x, y));     // the stored (half precision) texel
            env->cdfCols[colPos++] = colSum;
        }
        const float normalization = 1.0f / colSum;
        for (int x = 1; x < w; ++x) env->cdfCols[colPos - x - 1] *= normalization;
        env->cdfCols[colPos - 1] = 1.0f;
        const float weight = std::sin((y + 0.5f) * M_PI_F / h);
        env->rowWeights[y] = weight;
        rowSum += colSum * weight;
        env->cdfRows[rowPos++] = rowSum;
    }
    const float normalization = 1.0f / rowSum;
    for (int y = 1; y < h; ++y) env->cdfRows[rowPos - y - 1] *= normalization;
    env->cdfRows[rowPos - 1] = 1.0f;
    if (rowSum == 0) return "The environment map is completely black -- this is not allowed.";
    if (!std::isfinite(rowSum)) return "The environment map contains an invalid floating point value (nan/inf) -- giving up.";
    env->normalization = 1.0f / (rowSum * (2 * M_PI_F / w) * (M_PI_F / h));
    env->pixelSize = Vec2{ 2 * M_PI_F / w, M_PI_F / h };
    // linear part of the transform and of its inverse (Transform keeps both matrices, include/mitsuba/core/transform.h)
    const float *T = em.to_world;
    const double a = T[0], b = T[1], c = T[2], d = T[4], e = T[5], f = T[6], g = T[8], hh = T[9], i = T[10];
    const double det = a * (e * i - f * hh) - b * (d * i - f * g) + c * (d * hh - e * g);
    if (det == 0) return "envmap: singular toWorld transform";
    const double inv[9] = { (e * i - f * hh) / det, (c * hh - b * i) / det, (b * f - c * e) / det,
                            (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                            (d * hh - e * g) / det, (b * g - a * hh) / det, (a * e - b * d) / det };
    for (int k = 0; k < 9; ++k) { env->toWorld[k] = T[(k / 3) * 4 + k % 3]; env->toLocal[k] = (float) inv[k]; }
    envMap = env;
    return nullptr;
}

std::string Scene::build(const orc_scene_desc *d) {
    desc = *d;
    positions.resize(d->n_vertices);
    for (uint32_t i = 0; i < d->n_vertices; ++i)
        positions[i] = Vec3(d->positions[3 * i], d->positions[3 * i + 1], d->positions[3 * i + 2]);
    if (d->normals) {
        normals.resize(d->n_vertices);
        for (uint32_t i = 0; i < d->n_vertices; ++i)
            normals[i] = Vec3(d->normals[3 * i], d->normals[3 * i + 1], d->normals[3 * i + 2]);
    }
    if (d->texcoords) {
        texcoords.resize(d->n_vertices);
        for (uint32_t i = 0; i < d->n_vertices; ++i) texcoords[i] = Vec2{ d->texcoords[2 * i], d->texcoords[2 * i + 1] };
    }
    indices.assign(d->indices, d->indices + (size_t) d->n_triangles * 3);
    for (size_t i = 0; i < indices.size(); ++i)
        if (indices[i] >= d->n_vertices) return "vertex index out of range";
    shapes.assign(d->shapes, d->shapes + d->n_shapes);
    emitters.assign(d->emitters, d->emitters + d->n_emitters);
    materials.resize(d->n_materials);
    try {
        textures.resize(d->textures ? d->n_textures : 0);
        for (size_t i = 0; i < textures.size(); ++i)
            if (const char *err = buildTexture(d->textures[i], textures[i])) throw std::runtime_error(err);
        for (uint32_t i = 0; i < d->n_materials; ++i) { materials[i].m = d->materials[i]; materials[i].configure(textures); }
    } catch (const std::runtime_error &e) {
        static thread_local std::string message;
        message = e.what();
        return message.c_str();
    }
    if (const char *err = linkMixtures(materials)) return err;
    if (const char *err = linkTwoSided(materials)) return err;
    anyRayDifferentials = false;
    for (const Material &m : materials) anyRayDifferentials |= m.usesRayDifferentials;
    for (uint32_t e = 0; e < d->n_emitters; ++e) anyRayDifferentials |= d->emitters[e].type == 4;   // escaping camera rays are filtered

    triShape.assign(d->n_triangles, 0xFFFFFFFFu);
    uvDpdu.assign(d->n_triangles, Vec3(0.0f));
    uvDpdv.assign(d->n_triangles, Vec3(0.0f));
    uvDpduValid.assign(d->n_triangles, 0);
    for (uint32_t s = 0; s < d->n_shapes; ++s) {
        const orc_shape &sh = shapes[s];
        if ((uint64_t) sh.first_triangle + sh.n_triangles > d->n_triangles) return "shape triangle range out of bounds";
        if (sh.material < 0 || (uint32_t) sh.material >= d->n_materials) return "shape material out of range";
        if ((sh.flags & ORC_SHAPE_HAS_NORMALS) && !d->normals) return "shape claims normals but none given";
        if ((sh.flags & ORC_SHAPE_HAS_TEXCOORDS) && !d->texcoords) return "shape claims texcoords but none given";
        for (uint32_t j = 0; j < sh.n_triangles; ++j) {
            uint32_t t = sh.first_triangle + j;
            triShape[t] = s;
            if (sh.flags & ORC_SHAPE_HAS_TEXCOORDS) {
                // TriMesh::computeUVTangents (src/librender/trimesh.cpp:701-735)
                uint32_t i0 = indices[3 * t], i1 = indices[3 * t + 1], i2 = indices[3 * t + 2];
                const Vec3 &v0 = positions[i0], &v1 = positions[i1], &v2 = positions[i2];
                const Vec2 &uv0 = texcoords[i0], &uv1 = texcoords[i1], &uv2 = texcoords[i2];
                Vec3 dP1 = v1 - v0, dP2 = v2 - v0;
                Vec2 dUV1{ uv1.x - uv0.x, uv1.y - uv0.y }, dUV2{ uv2.x - uv0.x, uv2.y - uv0.y };
                Vec3 n = cross(dP1, dP2);
                float length = n.length();
                if (length == 0) continue;
                float determinant = dUV1.x * dUV2.y - dUV1.y * dUV2.x;
                if (determinant == 0) {
                    Vec3 dpdu, dpdv;
                    coordinateSystem(n / length, dpdu, dpdv);
                    uvDpdu[t] = dpdu;
                    uvDpdv[t] = dpdv;
                } else {
                    float invDet = 1.0f / determinant;
                    uvDpdu[t] = (dUV2.y * dP1 - dUV1.y * dP2) * invDet;
                    uvDpdv[t] = (-dUV2.x * dP1 + dUV1.x * dP2) * invDet;
                }
                uvDpduValid[t] = 1;
            }
        }
    }
    for (uint32_t t = 0; t < d->n_triangles; ++t)
        if (triShape[t] == 0xFFFFFFFFu) return "triangle not owned by any shape";

    // Per-emitter-shape area distribution (TriMesh::prepareSamplingTable, trimesh.cpp:388-403)
    areaDistr.assign(d->n_shapes, DiscreteDistribution());
    invSurfaceArea.assign(d->n_shapes, 0.0f);
    envEmitter = -1;
    for (uint32_t e = 0; e < d->n_emitters; ++e) {
        int s = emitters[e].shape;
        if (emitters[e].type == 4) {    // environment map
            if (envEmitter >= 0) return "The scene may only contain one environment emitter";
            envEmitter = (int) e;
            if (const char *err = buildEnvMap(emitters[e])) return err;
            emitterPDF.append(emitters[e].sampling_weight);
            continue;
        }
        if (emitters[e].type != 0) {    // point, spot, directional: no surface
            if (emitters[e].type < 1 || emitters[e].type > 3) return "unknown emitter type";
            emitterPDF.append(emitters[e].sampling_weight);
            continue;
        }
        if (s < 0) {    // constant environment emitter; only one is allowed (scene.cpp:509-513)
            if (envEmitter >= 0) return "The scene may only contain one environment emitter";
            envEmitter = (int) e;
            emitterPDF.append(emitters[e].sampling_weight);
            continue;
        }
        if ((uint32_t) s >= d->n_shapes) return "emitter shape out of range";
        if (shapes[s].emitter != (int32_t) e) return "shape/emitter cross reference mismatch";
        DiscreteDistribution &ad = areaDistr[s];
        for (uint32_t j = 0; j < shapes[s].n_triangles; ++j) {
            uint32_t t = shapes[s].first_triangle + j;
            const Vec3 &p0 = positions[indices[3 * t]], &p1 = positions[indices[3 * t + 1]], &p2 = positions[indices[3 * t + 2]];
            ad.append(0.5f * cross(p1 - p0, p2 - p0).length());     // Triangle::surfaceArea, triangle.cpp:61-67
        }
        float area = ad.normalize();
        invSurfaceArea[s] = 1.0f / area;
        emitterPDF.append(emitters[e].sampling_weight);             // scene.cpp:375-380
    }
    if (d->n_emitters) emitterPDF.normalize();

    buildKDTree();

    if (envEmitter >= 0) {
        // Scene::initializeBidirectional (scene.cpp:386-413): kd-tree bounds grown by the sensor's position;
        // AABB::getBSphere (aabb.cpp:44-47); ConstantBackgroundEmitter::createShape (constant.cpp:68-72)
        AABB box = aabb;
        box.expandBy(Vec3(d->camera.to_world[3], d->camera.to_world[7], d->camera.to_world[11]));
        envCenter = (box.max + box.min) * 0.5f;
        envRadius = std::max(Epsilon, (envCenter - box.max).length() * 1.5f);
    }

    if (d->integrator.type == 1) {      // MIDirectIntegrator (direct.cpp:96-107)
        if (d->integrator.emitter_samples < 0 || d->integrator.bsdf_samples < 0 || d->integrator.emitter_samples + d->integrator.bsdf_samples <= 0)
            return "direct: the numbers of emitter and BSDF samples must not be negative and not both zero";
    } else if (d->integrator.type != 0 && d->integrator.type != 2) {
        return "unknown integrator type";
    }
    deltaEmitters.assign(d->n_emitters, DeltaEmitter());
    {
        // AABB::getBSphere (aabb.cpp:44-47) of the kd-tree bounds, DirectionalEmitter::createShape (directional.cpp:88-94)
        dirCenter = (aabb.max + aabb.min) * 0.5f;
        dirRadius = (dirCenter - aabb.max).length() * 1.1f;
    }
    for (uint32_t e = 0; e < d->n_emitters; ++e) {
        const orc_emitter &em = emitters[e];
        if (em.type == 0) continue;
        DeltaEmitter &de = deltaEmitters[e];
        const float *M = em.to_world;
        de.position = Vec3(M[3], M[7], M[11]);
        de.direction = Vec3(M[2], M[6], M[10]);
        // linear part of the inverse transform (cofactors over the determinant)
        const double a = M[0], b = M[1], c = M[2], dd = M[4], ee = M[5], ff = M[6], g = M[8], h = M[9], i = M[10];
        const double det = a * (ee * i - ff * h) - b * (dd * i - ff * g) + c * (dd * h - ee * g);
        if (det == 0) return "singular emitter transform";
        const double inv[9] = { (ee * i - ff * h), (c * h - b * i), (b * ff - c * ee),
                                (ff * g - dd * i), (a * i - c * g), (c * dd - a * ff),
                                (dd * h - ee * g), (b * g - a * h), (a * ee - b * dd) };
        for (int k = 0; k < 9; ++k) de.toLocal[k] = (float) (inv[k] / det);
        if (em.type == 2) {     // SpotEmitter (spot.cpp:70-74,89-94)
            const float cutoff = em.cutoff_angle * (M_PI_F / 180.0f), beam = em.beam_width * (M_PI_F / 180.0f);
            if (!(cutoff >= beam)) return "spot emitter: the cutoff angle must not be smaller than the beam width";
            de.cutoffAngle = cutoff;
            de.cosBeam = std::cos(beam);
            de.cosCutoff = std::cos(cutoff);
            de.invTransitionWidth = 1.0f / (cutoff - beam);
        }
    }

    // Camera (perspective.cpp:126-157)
    const orc_film &f = d->film;
    aspect = (float) f.width / (float) f.height;
    invResolution = Vec2{ 1.0f / (float) f.crop_w, 1.0f / (float) f.crop_h };
    float relSizeX = (float) f.crop_w / (float) f.width, relSizeY = (float) f.crop_h / (float) f.height;
    float relOffX = (float) f.crop_x / (float) f.width, relOffY = (float) f.crop_y / (float) f.height;
    float persp[16];
    {
        float nearC = d->camera.near_clip, farC = d->camera.far_clip;
        float recip = 1.0f / (farC - nearC);
        float cot = 1.0f / std::tan((d->camera.x_fov / 2.0f) * (M_PI_F / 180.0f));
        memset(persp, 0, 64);
        persp[0] = cot; persp[5] = cot; persp[10] = farC * recip; persp[11] = -nearC * farC * recip; persp[14] = 1;
    }
    float m[16], tmp[16];
    matScale(1.0f / relSizeX, 1.0f / relSizeY, 1.0f, m);
    matTranslate(-relOffX, -relOffY, 0.0f, tmp); matMul(m, tmp, m);
    matScale(-0.5f, -0.5f * aspect, 1.0f, tmp); matMul(m, tmp, m);
    matTranslate(-1.0f, -1.0f / aspect, 0.0f, tmp); matMul(m, tmp, m);
    matMul(m, persp, m);
    if (!matInvert(m, sampleToCamera)) return "singular camera matrix";
    memcpy(camToWorld, d->camera.to_world, 64);
    return "";
}

// solveQuadratic (src/libcore/util.cpp:447-485) and BSphere::rayIntersect (include/mitsuba/core/bsphere.h:88-95)
static bool solveQuadratic(float a, float b, float c, float &x0, float &x1) {
    if (a == 0) {
        if (b != 0) { x0 = x1 = -c / b; return true; }
        return false;
    }
    float discrim = b * b - 4.0f * a * c;
    if (discrim < 0) return false;
    float temp, sqrtDiscrim = std::sqrt(discrim);
    if (b < 0) temp = -0.5f * (b - sqrtDiscrim);
    else temp = -0.5f * (b + sqrtDiscrim);
    x0 = temp / a;
    x1 = c / temp;
    if (x0 > x1) std::swap(x0, x1);
    return true;
}

static bool sphereIntersect(const Vec3 &center, float radius, const Vec3 &ro, const Vec3 &rd, float &nearHit, float &farHit) {
    Vec3 o = ro - center;
    float A = rd.lengthSquared();
    float B = 2 * dot(o, rd);
    float C = o.lengthSquared() - radius * radius;
    return solveQuadratic(A, B, C, nearHit, farHit);
}

// ---- EnvironmentMap (src/emitters/envmap.cpp) ----------------------------------------------------------------------
// The offset rays of a RayDifferential (include/mitsuba/core/ray.h:129-175); the perspective camera's start at the
// ray origin (perspective.cpp:289)
struct RayDifferentials { bool has = false; Vec3 rxDirection, ryDirection; };

static float intervalToTent(float sample) {     // src/libcore/warp.cpp:151-163
    float sign;
    if (sample < 0.5f) { sign = 1; sample *= 2; }
    else { sign = -1; sample = 2 * (sample - 0.5f); }
    return sign * (1 - std::sqrt(sample));
}

static uint32_t envSampleReuse(const float *cdf, uint32_t size, float &sample) {        // envmap.cpp:657-662
    const float *entry = std::lower_bound(cdf, cdf + size + 1, sample);
    const uint32_t index = std::min((uint32_t) std::max((ptrdiff_t) 0, entry - cdf - 1), size - 1);
    sample = (sample - cdf[index]) / (cdf[index + 1] - cdf[index]);
    return index;
}

// the four texels around a fractional pixel position, interpolated row by row (envmap.cpp:577-591, :618-632)
static void envRows(const Scene::EnvMap &env, int xPos, int yPos, float dx1, float dy1, Spectrum &value1, Spectrum &value2) {
    const float dx2 = 1.0f - dx1, dy2 = 1.0f - dy1;
    value1 = env.mip->evalTexel(0, xPos, yPos) * dx2 * dy2 + env.mip->evalTexel(0, xPos + 1, yPos) * dx1 * dy2;
    value2 = env.mip->evalTexel(0, xPos, yPos + 1) * dx2 * dy1 + env.mip->evalTexel(0, xPos + 1, yPos + 1) * dx1 * dy1;
}

// EnvironmentMap::internalSampleDirection (envmap.cpp:567-600)
static void envSampleDirection(const Scene::EnvMap &env, Vec2 sample, Vec3 &d, Spectrum &value, float &pdf) {
    const uint32_t row = envSampleReuse(env.cdfRows.data(), (uint32_t) env.h, sample.y),
                   col = envSampleReuse(env.cdfCols.data() + (size_t) row * (env.w + 1), (uint32_t) env.w, sample.x);
    const Vec2 pos{ (float) col + intervalToTent(sample.x), (float) row + intervalToTent(sample.y) };
    const int xPos = (int) std::floor(pos.x), yPos = (int) std::floor(pos.y);
    Spectrum value1, value2;
    envRows(env, xPos, yPos, pos.x - xPos, pos.y - yPos, value1, value2);
    value = (value1 + value2) * env.scale;
    pdf = (luminance(value1) * env.rowWeights[std::min(env.h - 1, std::max(0, yPos))] +
           luminance(value2) * env.rowWeights[std::min(env.h - 1, std::max(0, yPos + 1))]) * env.normalization;
    float sinPhi, cosPhi, sinTheta, cosTheta;
    ::sincosf(env.pixelSize.x * (pos.x + 0.5f), &sinPhi, &cosPhi);
    ::sincosf(env.pixelSize.y * (pos.y + 0.5f), &sinTheta, &cosTheta);
    d = Vec3(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
    pdf /= std::max(std::abs(sinTheta), Epsilon);
}

// EnvironmentMap::internalPdfDirection (envmap.cpp:603-633), d in the emitter's frame
static float envPdfDirection(const Scene::EnvMap &env, const Vec3 &d) {
    const Vec2 uv{ std::atan2(d.x, -d.z) * INV_TWOPI_F, std::acos(std::min(1.0f, std::max(-1.0f, d.y))) * INV_PI_F };
    if (!std::isfinite(uv.x) || !std::isfinite(uv.y)) return 0.0f;
    const float u = uv.x * env.w - 0.5f, v = uv.y * env.h - 0.5f;
    const int xPos = (int) std::floor(u), yPos = (int) std::floor(v);
    Spectrum value1, value2;
    envRows(env, xPos, yPos, u - xPos, v - yPos, value1, value2);
    const float sinTheta = safe_sqrt(1 - d.y * d.y);
    return (luminance(value1) * env.rowWeights[std::min(env.h - 1, std::max(0, yPos))] +
            luminance(value2) * env.rowWeights[std::min(env.h - 1, std::max(0, yPos + 1))]) * env.normalization / std::max(std::abs(sinTheta), Epsilon);
}

// EnvironmentMap::evalEnvironment (envmap.cpp:380-410): the radiance arriving along a ray that leaves the scene;
// a ray with offset rays (a camera ray) gets a filtered lookup
static Spectrum envEvalMap(const Scene::EnvMap &env, const Vec3 &rayD, const RayDifferentials &rd) {
    const Vec3 v = env.local(rayD);
    const Vec2 uv{ std::atan2(v.x, -v.z) * INV_TWOPI_F, std::acos(std::min(1.0f, std::max(-1.0f, v.y))) * INV_PI_F };
    Spectrum value;
    if (!rd.has) value = env.mip->evalBilinear(0, uv);
    else {
        const Vec3 dvdx = env.local(rd.rxDirection) - v, dvdy = env.local(rd.ryDirection) - v;
        const float t1 = INV_TWOPI_F / (v.x * v.x + v.z * v.z), t2 = -INV_PI_F / std::max(safe_sqrt(1.0f - v.y * v.y), Epsilon);
        const Vec2 dudx{ t1 * (dvdx.z * v.x - dvdx.x * v.z), t2 * dvdx.y }, dudy{ t1 * (dvdy.z * v.x - dvdy.x * v.z), t2 * dvdy.y };
        value = env.mip->eval(uv, dudx, dudy);
    }
    return value * env.scale;
}

// ---- Emitter sampling --------------------------------------------------------------------------------
struct DirectSample {
    Vec3 p, n, d;
    float dist = 0, pdf = 0;
    int emitter = -1;
    bool delta = false;     // discrete measure: the emitter cannot be reached by BSDF sampling
};

// Scene::sampleEmitterDirect sans visibility (scene.cpp:828-852), AreaLight::sampleDirect
// (area.cpp:158-173), Shape::sampleDirect (shape.cpp:102-115), TriMesh::samplePosition
// (trimesh.cpp:412-423), Triangle::sample (triangle.cpp:24-59)
static Spectrum sampleEmitterDirect(const Scene &sc, const Vec3 &ref, const Vec3 &refN, Vec2 sample, DirectSample &dRec) {
    if (sc.emitters.empty()) { dRec.pdf = 0; return Spectrum(0.0f); }
    float emPdf;
    size_t index = sc.emitterPDF.sampleReuse(sample.x, emPdf);
    const orc_emitter &em = sc.emitters[index];
    if (em.type == 4) {     // EnvironmentMap::sampleDirect (envmap.cpp:516-543)
        const Scene::EnvMap &env = *sc.envMap;
        Spectrum value; Vec3 d; float pdf;
        envSampleDirection(env, sample, d, value, pdf);
        const Vec3 dw = env.world(d);
        float nearT, farT;
        dRec.pdf = 0.0f;
        if (value.isZero() || pdf == 0 || !sphereIntersect(sc.envCenter, sc.envRadius, ref, dw, nearT, farT) || nearT >= 0 || farT <= 0)
            return Spectrum(0.0f);
        dRec.pdf = pdf;
        dRec.p = ref + dw * farT;
        dRec.n = normalize(sc.envCenter - dRec.p);
        dRec.dist = farT;
        dRec.d = dw;
        dRec.emitter = (int) index;
        value = value / pdf;
        dRec.pdf *= emPdf;
        value /= emPdf;
        return value;
    }
    if (em.type != 0) {
        // PointEmitter::sampleDirect (point.cpp:131-147), SpotEmitter::sampleDirect (spot.cpp:184-200),
        // DirectionalEmitter::sampleDirect (directional.cpp:159-180): one position or direction, discrete measure
        const Scene::DeltaEmitter &de = sc.deltaEmitters[index];
        Spectrum value(em.radiance[0], em.radiance[1], em.radiance[2]);
        dRec.pdf = 0.0f;
        if (em.type == 3) {
            const Vec3 d = de.direction;
            const Vec3 diskCenter = sc.dirCenter - d * sc.dirRadius;
            const float distance = dot(ref - diskCenter, d);
            if (distance < 0) return Spectrum(0.0f);
            dRec.p = ref - d * distance;
            dRec.d = -d;
            dRec.n = d;
            dRec.dist = distance;
        } else {
            dRec.p = de.position;
            dRec.d = dRec.p - ref;
            dRec.dist = dRec.d.length();
            const float invDist = 1.0f / dRec.dist;
            dRec.d *= invDist;
            dRec.n = Vec3(0.0f);
            if (em.type == 2) {     // SpotEmitter::falloffCurve (spot.cpp:105-125), constant texture
                const Vec3 w = -dRec.d;
                const float *T = de.toLocal;
                const float cosTheta = T[6] * w.x + T[7] * w.y + T[8] * w.z;
                float falloff = 1.0f;
                if (cosTheta <= de.cosCutoff) falloff = 0.0f;
                else if (cosTheta < de.cosBeam) falloff = (de.cutoffAngle - std::acos(cosTheta)) * de.invTransitionWidth;
                value *= falloff;
            }
            value *= invDist * invDist;
        }
        dRec.pdf = 1.0f;
        dRec.delta = true;
        dRec.emitter = (int) index;
        dRec.pdf *= emPdf;
        value /= emPdf;
        return value;
    }
    if (em.type == 0 && em.shape < 0 && sc.envMap) return Spectrum(0.0f);      // (cannot happen: a map is of type 4)
    if (em.type == 0 && em.shape < 0) {     // ConstantBackgroundEmitter::sampleDirect (constant.cpp:178-216)
        Vec3 d;
        float pdf;
        if (!refN.isZero()) {
            d = squareToCosineHemisphere(sample);
            pdf = squareToCosineHemispherePdf(d);
            d = Frame(refN).toWorld(d);
        } else {
            d = squareToUniformSphere(sample);
            pdf = INV_FOURPI_F;
        }
        float nearT, farT;
        dRec.pdf = 0.0f;
        if (!sphereIntersect(sc.envCenter, sc.envRadius, ref, d, nearT, farT)) return Spectrum(0.0f);
        if (!(nearT < 0 && farT > 0)) return Spectrum(0.0f);
        dRec.p = ref + d * farT;
        dRec.n = normalize(sc.envCenter - dRec.p);
        dRec.d = d;
        dRec.dist = farT;
        dRec.pdf = pdf;
        dRec.emitter = (int) index;
        Spectrum value(0.0f);
        if (!(!refN.isZero() && dot(dRec.d, refN) <= 0))
            value = Spectrum(em.radiance[0], em.radiance[1], em.radiance[2]) / pdf;
        // Scene::sampleEmitterDirect (scene.cpp:838-847): the pdf is non-zero here, so the sample counts
        dRec.pdf *= emPdf;
        value /= emPdf;
        return value;
    }
    const orc_shape &sh = sc.shapes[em.shape];
    size_t triIdx = sc.areaDistr[em.shape].sampleReuse(sample.y);
    uint32_t t = sh.first_triangle + (uint32_t) triIdx;
    uint32_t i0 = sc.indices[3 * t], i1 = sc.indices[3 * t + 1], i2 = sc.indices[3 * t + 2];
    const Vec3 &p0 = sc.positions[i0], &p1 = sc.positions[i1], &p2 = sc.positions[i2];
    Vec2 bary = squareToUniformTriangle(sample);
    Vec3 sideA = p1 - p0, sideB = p2 - p0;
    dRec.p = p0 + (sideA * bary.x) + (sideB * bary.y);
    if (sh.flags & ORC_SHAPE_HAS_NORMALS) {
        const Vec3 &n0 = sc.normals[i0], &n1 = sc.normals[i1], &n2 = sc.normals[i2];
        dRec.n = normalize(n0 * (1.0f - bary.x - bary.y) + n1 * bary.x + n2 * bary.y);
    } else {
        dRec.n = normalize(cross(sideA, sideB));
    }
    dRec.pdf = sc.invSurfaceArea[em.shape];
    dRec.d = dRec.p - ref;
    float distSquared = dRec.d.lengthSquared();
    dRec.dist = std::sqrt(distSquared);
    dRec.d /= dRec.dist;
    float dp = absDot(dRec.d, dRec.n);
    dRec.pdf *= dp != 0 ? (distSquared / dp) : 0.0f;
    Spectrum value;
    if (dot(dRec.d, refN) >= 0 && dot(dRec.d, dRec.n) < 0 && dRec.pdf != 0) {
        value = Spectrum(em.radiance[0], em.radiance[1], em.radiance[2]) / dRec.pdf;
    } else {
        dRec.pdf = 0.0f;
        return Spectrum(0.0f);
    }
    dRec.emitter = (int) index;
    dRec.pdf *= emPdf;
    value /= emPdf;
    return value;
}

// ConstantBackgroundEmitter::pdfDirect in the solid-angle measure (constant.cpp:218-233) times pdfEmitterDiscrete
static float envPdfDirect(const Scene &sc, const Vec3 &d, const Vec3 &refN) {
    if (sc.envMap)      // EnvironmentMap::pdfDirect, solid-angle measure (envmap.cpp:545-551)
        return envPdfDirection(*sc.envMap, sc.envMap->local(d)) * sc.emitters[sc.envEmitter].sampling_weight * sc.emitterPDF.normalization;
    float pdfSA = !refN.isZero() ? INV_PI_F * std::max(0.0f, dot(d, refN)) : INV_FOURPI_F;
    return pdfSA * sc.emitters[sc.envEmitter].sampling_weight * sc.emitterPDF.normalization;
}

// Scene::evalEnvironment: the constant emitter's radiance or the map's value along the ray
static Spectrum envRadiance(const Scene &sc, const Vec3 &rayD, const RayDifferentials &rd) {
    if (sc.envMap) return envEvalMap(*sc.envMap, rayD, rd);
    const orc_emitter &em = sc.emitters[sc.envEmitter];
    return Spectrum(em.radiance[0], em.radiance[1], em.radiance[2]);
}

static inline float miWeight(float pdfA, float pdfB) {
    pdfA *= pdfA; pdfB *= pdfB;
    return pdfA / (pdfA + pdfB);
}

struct RayCounters { uint64_t normal = 0, shadow = 0; TraversalCounters trav; };

// Intersection::computePartials (src/librender/intersection.cpp:5-76) for offset rays that start at the ray origin
static UVPartials computePartials(const Intersection &its, const Ray &ray, const RayDifferentials &rd) {
    UVPartials out;
    if (!rd.has) return out;
    out.has = true;
    if (its.dpdu.isZero() && its.dpdv.isZero()) return out;
    const Vec3 &n = its.geoFrame.n;
    const float pp = dot(n, its.p), pox = dot(n, ray.o), poy = dot(n, ray.o), prx = dot(n, rd.rxDirection), pry = dot(n, rd.ryDirection);
    if (prx == 0 || pry == 0) return out;
    const float tx = (pp - pox) / prx, ty = (pp - poy) / pry;
    const float absX = std::abs(n.x), absY = std::abs(n.y), absZ = std::abs(n.z);
    int axes[2];
    if (absX > absY && absX > absZ) { axes[0] = 1; axes[1] = 2; }
    else if (absY > absZ) { axes[0] = 0; axes[1] = 2; }
    else { axes[0] = 0; axes[1] = 1; }
    const float A[2][2] = { { its.dpdu[axes[0]], its.dpdv[axes[0]] }, { its.dpdu[axes[1]], its.dpdv[axes[1]] } };
    const Vec3 px = ray.o + rd.rxDirection * tx, py = ray.o + rd.ryDirection * ty;
    const float Bx[2] = { px[axes[0]] - its.p[axes[0]], px[axes[1]] - its.p[axes[1]] };
    const float By[2] = { py[axes[0]] - its.p[axes[0]], py[axes[1]] - its.p[axes[1]] };
    // solveLinearSystem2x2 (src/libcore/util.cpp:527-539)
    const float det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
    if (std::abs(det) <= RCPOVERFLOW_F) {
        out.dudx = 1; out.dvdx = 0;
        out.dudy = 1; out.dvdy = 0;     // intersection.cpp:74 assigns dudy twice: 0, then 1; dvdy keeps its initial value
        return out;
    }
    const float inverse = 1.0f / det;
    out.dudx = (A[1][1] * Bx[0] - A[0][1] * Bx[1]) * inverse; out.dvdx = (A[0][0] * Bx[1] - A[1][0] * Bx[0]) * inverse;
    out.dudy = (A[1][1] * By[0] - A[0][1] * By[1]) * inverse; out.dvdy = (A[0][0] * By[1] - A[1][0] * By[0]) * inverse;
    return out;
}
// Intersection::getBSDF(ray) (records.inl:69-75): the partials exist only if the BSDF asks for them
static UVPartials partialsFor(const Material &bsdf, const Intersection &its, const Ray &ray, const RayDifferentials &rd) {
    const bool wants = bsdf.m.two_sided ? bsdf.twoSidedRayDifferentials : bsdf.usesRayDifferentials;
    return wants ? computePartials(its, ray, rd) : UVPartials();
}

// ---- MIPathTracer::Li (src/integrators/path/path.cpp:119-294) -------------------------------------------
static Spectrum Li(const Scene &sc, const Ray &r, const RayDifferentials &rd0, Pcg32 &rng, float &alpha, RayCounters &cnt) {
    const orc_integrator &ig = sc.desc.integrator;
    Intersection its;
    Ray ray(r);
    RayDifferentials rd = rd0;
    Spectrum Li(0.0f);
    bool scattered = false;
    bool emittedRadiance = true;   // rRec.type & EEmittedRadiance
    int depth = 1;

    cnt.normal++;
    sc.rayIntersect(ray, its, &cnt.trav);
    alpha = its.isValid() ? 1.0f : 0.0f;   // records.inl:121-137
    ray.mint = Epsilon;

    Spectrum throughput(1.0f);
    float eta = 1.0f;

    while (depth <= ig.max_depth || ig.max_depth < 0) {
        if (!its.isValid()) {          // path.cpp:136-143
            if (sc.envEmitter >= 0 && emittedRadiance && (!ig.hide_emitters || scattered))
                Li += throughput * envRadiance(sc, ray.d, rd);      // scene->evalEnvironment(ray): a camera ray keeps its offset rays
            break;
        }
        const orc_shape &shape = sc.shapes[its.shape];
        const Material &bsdf = sc.materials[shape.material];
        g_textureUV = its.uv;       // the intersection the BSDF queries of this vertex refer to
        g_bsdfIts = &its;
        g_texturePartials = partialsFor(bsdf, its, ray, rd);        // its.getBSDF(ray) (path.cpp:145)
        rd.has = false;             // the rays that follow are plain Rays (path.cpp:222: ray = Ray(its.p, wo, ray.time))

        if (shape.emitter >= 0 && emittedRadiance && (!ig.hide_emitters || scattered)) {
            // AreaLight::eval (area.cpp:104-109)
            const orc_emitter &em = sc.emitters[shape.emitter];
            if (dot(its.shFrame.n, -ray.d) > 0)
                Li += throughput * Spectrum(em.radiance[0], em.radiance[1], em.radiance[2]);
        }

        if ((depth >= ig.max_depth && ig.max_depth > 0) ||
            (ig.strict_normals && dot(ray.d, its.geoFrame.n) * Frame::cosTheta(its.wi) >= 0))
            break;

        // DirectSamplingRecord(its) (records.inl:160-164)
        Vec3 refN(0.0f);
        if ((bsdf.type_flags & (ETransmission | EBackSide)) == 0) refN = its.shFrame.n;

        if (bsdf.type_flags & ESmooth) {
            DirectSample dRec;
            Vec2 s = rng.next2D();
            Spectrum value = sampleEmitterDirect(sc, its.p, refN, s, dRec);
            if (dRec.pdf != 0) {
                cnt.shadow++;
                Ray shadowRay(its.p, dRec.d, Epsilon, dRec.dist * (1 - ShadowEpsilon));
                if (sc.rayIntersectShadow(shadowRay, &cnt.trav)) value = Spectrum(0.0f);
            }
            if (!value.isZero()) {
                Vec3 wo = its.toLocal(dRec.d);
                const Spectrum bsdfVal = bsdfEval(bsdf, its.wi, wo);
                if (!bsdfVal.isZero() && (!ig.strict_normals || dot(its.geoFrame.n, dRec.d) * Frame::cosTheta(wo) > 0)) {
                    // path.cpp:192-193: only emitters with a surface, sampled in the solid-angle measure, compete with BSDF sampling
                    float bsdfPdf_ = dRec.delta ? 0.0f : bsdfPdf(bsdf, its.wi, wo);
                    float weight = miWeight(dRec.pdf, bsdfPdf_);
                    Li += throughput * value * bsdfVal * weight;
                }
            }
        }

        float bsdfPdf_;
        BSDFSample bs;
        struct StreamTap : Sampler1D { Pcg32 &r; explicit StreamTap(Pcg32 &r_) : r(r_) {} float next1D() override { return r.next1D(); } } tap(rng);
        const Vec2 bsdfSample2D = rng.next2D();     // drawn before anything the BSDF takes from the sampler itself
        Spectrum bsdfWeight = bsdfSample(bsdf, its.wi, bsdfSample2D, bs, bsdfPdf_, &tap);
        if (bsdfWeight.isZero()) break;
        scattered |= bs.sampledType != ENull;

        const Vec3 wo = its.toWorld(bs.wo);
        float woDotGeoN = dot(its.geoFrame.n, wo);
        if (ig.strict_normals && woDotGeoN * Frame::cosTheta(bs.wo) <= 0) break;

        bool hitEmitter = false, hitEnvironment = false;
        Spectrum value;
        Vec3 refP = its.p;
        ray = Ray(its.p, wo);
        cnt.normal++;
        if (sc.rayIntersect(ray, its, &cnt.trav)) {
            const orc_shape &hs = sc.shapes[its.shape];
            if (hs.emitter >= 0) {
                const orc_emitter &em = sc.emitters[hs.emitter];
                value = dot(its.shFrame.n, -ray.d) > 0 ? Spectrum(em.radiance[0], em.radiance[1], em.radiance[2]) : Spectrum(0.0f);
                hitEmitter = true;
            }
        } else {
            // path.cpp:233-247: the environment emitter, if there is one
            if (sc.envEmitter < 0) break;
            if (ig.hide_emitters && !scattered) break;
            value = envRadiance(sc, ray.d, RayDifferentials());    // evalEnvironment(RayDifferential(ray)): no offset rays
            float nearT, farT;     // fillDirectSamplingRecord (constant.cpp:245-261, envmap.cpp:358-374)
            if (!sphereIntersect(sc.envCenter, sc.envRadius, ray.o, ray.d, nearT, farT) || nearT > 0 || farT < 0) break;
            hitEnvironment = true;
        }
        (void) refP;

        throughput *= bsdfWeight;
        eta *= bs.eta;

        if (hitEmitter) {
            // pdfEmitterDirect (scene.cpp:949-952) -> AreaLight::pdfDirect (area.cpp:175-183)
            // -> Shape::pdfDirect (shape.cpp:117-121) with dRec.setQuery(ray, its)
            float lumPdf = 0;
            if (!(bs.sampledType & EDelta)) {
                const orc_shape &hs = sc.shapes[its.shape];
                if (dot(ray.d, refN) >= 0 && dot(ray.d, its.shFrame.n) < 0)
                    lumPdf = sc.invSurfaceArea[its.shape] * (its.t * its.t) / absDot(ray.d, its.shFrame.n);
                lumPdf *= sc.emitters[hs.emitter].sampling_weight * sc.emitterPDF.normalization;
            }
            Li += throughput * value * miWeight(bsdfPdf_, lumPdf);
        } else if (hitEnvironment) {
            const float lumPdf = !(bs.sampledType & EDelta) ? envPdfDirect(sc, ray.d, refN) : 0.0f;
            Li += throughput * value * miWeight(bsdfPdf_, lumPdf);
        }

        emittedRadiance = false;   // rRec.type = ERadianceNoEmission
        if (depth++ >= ig.rr_depth) {
            float q = std::min(throughput.max() * eta * eta, 0.95f);
            if (rng.next1D() >= q) break;
            throughput /= q;
        }
    }
    return Li;
}

// ---- SimpleVolumetricPathTracer::Li (src/integrators/path/volpath_simple.cpp:88-297) for scenes without participating
// media (rRec.medium is always NULL, no BSDF has a null component): a path tracer without multiple importance sampling.
// Emitters are seen by the camera ray and after Dirac bounces only; every other vertex gathers direct light by emitter
// sampling at full weight, with the visibility test of Scene::sampleAttenuatedEmitterDirect (scene.cpp:876-898,
// evalTransmittance :619-676: a ray from the surface to the sampled point, shortened by ShadowEpsilon only where that
// point lies on a surface).
static Spectrum LiSimple(const Scene &sc, const Ray &r, const RayDifferentials &rd0, Pcg32 &rng, float &alpha, RayCounters &cnt) {
    const orc_integrator &ig = sc.desc.integrator;
    Intersection its;
    Ray ray(r);
    RayDifferentials rd = rd0;
    Spectrum Li(0.0f);
    bool scattered = false;
    bool emitted = true, direct = true;         // rRec.type & EEmittedRadiance, & EDirectSurfaceRadiance
    float eta = 1.0f;
    int depth = 1;

    cnt.normal++;
    sc.rayIntersect(ray, its, &cnt.trav);
    alpha = its.isValid() ? 1.0f : 0.0f;
    ray.mint = Epsilon;
    Spectrum throughput(1.0f);
    if (ig.max_depth == 1) direct = false;      // :103-104

    while (depth <= ig.max_depth || ig.max_depth < 0) {
        if (!its.isValid()) {                   // :170-181
            if (sc.envEmitter >= 0 && emitted && (!ig.hide_emitters || scattered))
                Li += throughput * envRadiance(sc, ray.d, rd);
            break;
        }
        const orc_shape &shape = sc.shapes[its.shape];
        const Material &bsdf = sc.materials[shape.material];
        g_textureUV = its.uv;
        g_bsdfIts = &its;
        g_texturePartials = partialsFor(bsdf, its, ray, rd);
        rd.has = false;

        if (shape.emitter >= 0 && emitted && (!ig.hide_emitters || scattered)) {     // :184-186
            const orc_emitter &em = sc.emitters[shape.emitter];
            if (dot(its.shFrame.n, -ray.d) > 0)
                Li += throughput * Spectrum(em.radiance[0], em.radiance[1], em.radiance[2]);
        }
        // :193-196: strictly negative, unlike path.cpp's test
        if (ig.strict_normals && (-dot(its.geoFrame.n, ray.d)) * Frame::cosTheta(its.wi) < 0) break;

        Vec3 refN(0.0f);
        if ((bsdf.type_flags & (ETransmission | EBackSide)) == 0) refN = its.shFrame.n;

        if (direct && (bsdf.type_flags & ESmooth)) {        // :205-226
            DirectSample dRec;
            Vec2 s = rng.next2D();
            Spectrum value = sampleEmitterDirect(sc, its.p, refN, s, dRec);
            if (dRec.pdf != 0) {
                cnt.shadow++;
                Ray shadowRay(its.p, dRec.d, Epsilon, dRec.delta ? dRec.dist : dRec.dist * (1 - ShadowEpsilon));
                if (sc.rayIntersectShadow(shadowRay, &cnt.trav)) value = Spectrum(0.0f);
            }
            if (!value.isZero()) {
                Vec3 wo = its.toLocal(dRec.d);
                if (!ig.strict_normals || dot(its.geoFrame.n, dRec.d) * Frame::cosTheta(wo) > 0)
                    Li += throughput * value * bsdfEval(bsdf, its.wi, wo);
            }
        }

        float bsdfPdf_;
        BSDFSample bs;
        struct StreamTap : Sampler1D { Pcg32 &r; explicit StreamTap(Pcg32 &r_) : r(r_) {} float next1D() override { return r.next1D(); } } tap(rng);
        const Vec2 bsdfSample2D = rng.next2D();
        Spectrum bsdfVal = bsdfSample(bsdf, its.wi, bsdfSample2D, bs, bsdfPdf_, &tap);
        if (bsdfVal.isZero()) break;

        // :243-263 (no null components: nullChain plays no part)
        bool nextDirect = false, nextEmitted = false;
        if (depth + 1 < ig.max_depth || ig.max_depth < 0) nextDirect = true;
        if ((depth < ig.max_depth || ig.max_depth < 0) && (bs.sampledType & EDelta)) nextEmitted = true;
        if (!nextDirect && !nextEmitted) break;
        direct = nextDirect; emitted = nextEmitted;

        const Vec3 wo = its.toWorld(bs.wo);
        if (dot(its.geoFrame.n, wo) * Frame::cosTheta(bs.wo) <= 0 && ig.strict_normals) break;

        throughput *= bsdfVal;
        eta *= bs.eta;
        ray = Ray(its.p, wo);
        cnt.normal++;
        sc.rayIntersect(ray, its, &cnt.trav);
        scattered |= bs.sampledType != ENull;

        if (depth++ >= ig.rr_depth) {           // :282-292
            float q = std::min(throughput.max() * eta * eta, 0.95f);
            if (rng.next1D() >= q) break;
            throughput /= q;
        }
    }
    return Li;
}

// ---- MIDirectIntegrator::Li (src/integrators/direct/direct.cpp:146-309) -----------------------------------
// Sample order per camera sample: the emitter samples' 2-D numbers, then for every BSDF sample its 2-D number
// followed by what the BSDF itself takes from the sampler.
static Spectrum LiDirect(const Scene &sc, const Ray &r, const RayDifferentials &rd, Pcg32 &rng, float &alpha, RayCounters &cnt) {
    const orc_integrator &ig = sc.desc.integrator;
    Intersection its;
    Ray ray(r);
    Spectrum Li(0.0f);
    cnt.normal++;
    const bool hit = sc.rayIntersect(ray, its, &cnt.trav);
    alpha = hit ? 1.0f : 0.0f;
    if (!hit) {     // direct.cpp:156-163
        if (sc.envEmitter >= 0 && !ig.hide_emitters) return envRadiance(sc, ray.d, rd);
        return Spectrum(0.0f);
    }
    const orc_shape &shape = sc.shapes[its.shape];
    const Material &bsdf = sc.materials[shape.material];
    g_textureUV = its.uv;
    g_bsdfIts = &its;
    g_texturePartials = partialsFor(bsdf, its, ray, rd);    // its.getBSDF(ray) (direct.cpp:176)
    if (shape.emitter >= 0 && !ig.hide_emitters) {      // direct.cpp:166-167, AreaLight::eval (area.cpp:104-109)
        const orc_emitter &em = sc.emitters[shape.emitter];
        if (dot(its.shFrame.n, -ray.d) > 0) Li += Spectrum(em.radiance[0], em.radiance[1], em.radiance[2]);
    }
    if (ig.strict_normals && dot(ray.d, its.geoFrame.n) * Frame::cosTheta(its.wi) >= 0) return Li;      // direct.cpp:175-187

    const int nDirect = ig.emitter_samples, nBsdf = ig.bsdf_samples;
    const int sum = nDirect + nBsdf;
    // MIDirectIntegrator::configure (direct.cpp:128-136)
    const float weightBSDF = 1 / (float) nBsdf, weightLum = 1 / (float) nDirect;
    const float fracBSDF = nBsdf / (float) sum, fracLum = nDirect / (float) sum;

    Vec3 refN(0.0f);    // DirectSamplingRecord(its) (records.inl:160-164)
    if ((bsdf.type_flags & (ETransmission | EBackSide)) == 0) refN = its.shFrame.n;

    // emitter sampling (direct.cpp:210-245); the numbers are drawn whether or not the BSDF is smooth
    for (int i = 0; i < nDirect; ++i) {
        const Vec2 s = rng.next2D();
        if (!(bsdf.type_flags & ESmooth)) continue;
        DirectSample dRec;
        Spectrum value = sampleEmitterDirect(sc, its.p, refN, s, dRec);
        if (dRec.pdf != 0) {
            cnt.shadow++;
            Ray shadowRay(its.p, dRec.d, Epsilon, dRec.dist * (1 - ShadowEpsilon));
            if (sc.rayIntersectShadow(shadowRay, &cnt.trav)) value = Spectrum(0.0f);
        }
        if (value.isZero()) continue;
        const Vec3 wo = its.toLocal(dRec.d);
        const Spectrum bsdfVal = bsdfEval(bsdf, its.wi, wo);
        if (!bsdfVal.isZero() && (!ig.strict_normals || dot(its.geoFrame.n, dRec.d) * Frame::cosTheta(wo) > 0)) {
            const float bsdfPdf_ = dRec.delta ? 0.0f : bsdfPdf(bsdf, its.wi, wo);      // emitter->isOnSurface()
            const float weight = miWeight(dRec.pdf * fracLum, bsdfPdf_ * fracBSDF) * weightLum;
            Li += value * bsdfVal * weight;
        }
    }

    // BSDF sampling (direct.cpp:251-306)
    struct StreamTap : Sampler1D { Pcg32 &r; explicit StreamTap(Pcg32 &r_) : r(r_) {} float next1D() override { return r.next1D(); } } tap(rng);
    for (int i = 0; i < nBsdf; ++i) {
        float bsdfPdf_;
        BSDFSample bs;
        const Vec2 s = rng.next2D();
        const Spectrum bsdfVal = bsdfSample(bsdf, its.wi, s, bs, bsdfPdf_, &tap);
        if (bsdfVal.isZero()) continue;
        const Vec3 wo = its.toWorld(bs.wo);
        if (ig.strict_normals && dot(its.geoFrame.n, wo) * Frame::cosTheta(bs.wo) <= 0) continue;
        Ray bsdfRay(its.p, wo);
        Intersection bsdfIts;
        Spectrum value;
        float lumPdf = 0;
        cnt.normal++;
        if (sc.rayIntersect(bsdfRay, bsdfIts, &cnt.trav)) {
            const orc_shape &hs = sc.shapes[bsdfIts.shape];
            if (hs.emitter < 0) continue;
            const orc_emitter &em = sc.emitters[hs.emitter];
            value = dot(bsdfIts.shFrame.n, -bsdfRay.d) > 0 ? Spectrum(em.radiance[0], em.radiance[1], em.radiance[2]) : Spectrum(0.0f);
            if (!(bs.sampledType & EDelta)) {   // pdfEmitterDirect (scene.cpp:949-952), as in the path integrator
                if (dot(bsdfRay.d, refN) >= 0 && dot(bsdfRay.d, bsdfIts.shFrame.n) < 0)
                    lumPdf = sc.invSurfaceArea[bsdfIts.shape] * (bsdfIts.t * bsdfIts.t) / absDot(bsdfRay.d, bsdfIts.shFrame.n);
                lumPdf *= em.sampling_weight * sc.emitterPDF.normalization;
            }
        } else {
            if (sc.envEmitter < 0) continue;    // (hideEmitters only hides what a null interaction would show, direct.cpp:288)
            value = envRadiance(sc, bsdfRay.d, RayDifferentials());
            float nearT, farT;
            if (!sphereIntersect(sc.envCenter, sc.envRadius, bsdfRay.o, bsdfRay.d, nearT, farT) || nearT > 0 || farT < 0) continue;
            if (!(bs.sampledType & EDelta)) lumPdf = envPdfDirect(sc, bsdfRay.d, refN);
        }
        const float weight = miWeight(bsdfPdf_ * fracBSDF, lumPdf * fracLum) * weightBSDF;
        Li += value * bsdfVal * weight;
    }
    return Li;
}

// ---- PerspectiveCameraImpl::sampleRayDifferential (perspective.cpp:271-298), ray part ---------------------
// Transform::operator()(Point) of the sample-to-camera matrix
static Vec3 sampleToCameraPoint(const Scene &sc, float sx, float sy) {
    const float *M = sc.sampleToCamera;
    Vec3 p(M[0] * sx + M[1] * sy + M[3], M[4] * sx + M[5] * sy + M[7], M[8] * sx + M[9] * sy + M[11]);
    const float w = M[12] * sx + M[13] * sy + M[15];
    if (w != 1.0f) p = p / w;
    return p;
}

// `diff`, if given, receives the offset rays scaled by `diffScale` (RayDifferential::scaleDifferential, ray.h:163-168;
// renderBlock scales by 1 / sqrt(sampleCount), integrator.cpp:150,177)
static Ray cameraRay(const Scene &sc, float px, float py, RayDifferentials *diff = nullptr, float diffScale = 1.0f) {
    const float *M = sc.sampleToCamera;
    float sx = px * sc.invResolution.x, sy = py * sc.invResolution.y;
    // Transform::operator()(Point) with homogeneous divide (transform.h)
    float x = M[0] * sx + M[1] * sy + M[3];
    float y = M[4] * sx + M[5] * sy + M[7];
    float z = M[8] * sx + M[9] * sy + M[11];
    float w = M[12] * sx + M[13] * sy + M[15];
    Vec3 nearP(x, y, z);
    if (w != 1.0f) nearP = nearP / w;
    Vec3 d = normalize(nearP);
    float invZ = 1.0f / d.z;
    const float *W = sc.camToWorld;
    Vec3 o(W[3], W[7], W[11]);   // transformAffine(Point(0))
    Vec3 dw(W[0] * d.x + W[1] * d.y + W[2] * d.z,
            W[4] * d.x + W[5] * d.y + W[6] * d.z,
            W[8] * d.x + W[9] * d.y + W[10] * d.z);
    if (diff) {
        // position differentials on the near plane (perspective.cpp:158-162)
        const Vec3 origin = sampleToCameraPoint(sc, 0.0f, 0.0f);
        const Vec3 dxv = sampleToCameraPoint(sc, sc.invResolution.x, 0.0f) - origin, dyv = sampleToCameraPoint(sc, 0.0f, sc.invResolution.y) - origin;
        auto toWorld = [&](const Vec3 &v) {
            return Vec3(W[0] * v.x + W[1] * v.y + W[2] * v.z, W[4] * v.x + W[5] * v.y + W[6] * v.z, W[8] * v.x + W[9] * v.y + W[10] * v.z);
        };
        const Vec3 rx = toWorld(normalize(nearP + dxv)), ry = toWorld(normalize(nearP + dyv));
        diff->has = true;
        diff->rxDirection = dw + (rx - dw) * diffScale;
        diff->ryDirection = dw + (ry - dw) * diffScale;
    }
    return Ray(o, dw, sc.desc.camera.near_clip * invZ, sc.desc.camera.far_clip * invZ);
}

// ---- ImageBlock::put (include/mitsuba/render/imageblock.h:144-204) onto a border-less film -----------------
static bool filmPut(const Scene &sc, float *film, float posx, float posy, const float *value /* R,G,B,alpha */) {
    const orc_film &f = sc.desc.film;
    float temp[5] = { value[0], value[1], value[2], value[3], 1.0f };
    for (int i = 0; i < 5; ++i)
        if (!std::isfinite(temp[i]) || temp[i] < 0) return false;
    const float filterRadius = f.filter_radius;
    const float scaleFactor = 31.0f / filterRadius;    // rfilter.cpp:49
    const float px = posx - 0.5f, py = posy - 0.5f;
    const int minx = std::max((int) std::ceil(px - filterRadius), 0), miny = std::max((int) std::ceil(py - filterRadius), 0);
    const int maxx = std::min((int) std::floor(px + filterRadius), f.crop_w - 1), maxy = std::min((int) std::floor(py + filterRadius), f.crop_h - 1);
    float wx[16], wy[16];
    for (int x = minx, idx = 0; x <= maxx; ++x) wx[idx++] = f.filter_values[std::min((int) std::abs((x - px) * scaleFactor), 31)];
    for (int y = miny, idx = 0; y <= maxy; ++y) wy[idx++] = f.filter_values[std::min((int) std::abs((y - py) * scaleFactor), 31)];
    for (int y = miny, yr = 0; y <= maxy; ++y, ++yr) {
        const float weightY = wy[yr];
        float *dest = film + ((size_t) y * f.crop_w + minx) * 5;
        for (int x = minx, xr = 0; x <= maxx; ++x, ++xr) {
            const float weight = wx[xr] * weightY;
            for (int k = 0; k < 5; ++k) *dest++ += weight * temp[k];
        }
    }
    return true;
}

}  // namespace orc

using namespace orc;

struct orc_scene { Scene s; };

static thread_local std::string g_err;
extern "C" const char *orc_last_error(void) { return g_err.c_str(); }

extern "C" orc_scene *orc_scene_create(const orc_scene_desc *desc) {
    orc_scene *h = new orc_scene();
    std::string err = h->s.build(desc);
    if (!err.empty()) { g_err = err; delete h; return nullptr; }
    return h;
}
extern "C" void orc_scene_destroy(orc_scene *s) { delete s; }

static void fillHit(const Intersection &its, bool hit, orc_hit &h) {
    memset(&h, 0, sizeof(h));
    if (!hit) { h.t = std::numeric_limits<float>::infinity(); h.shape = h.prim = 0xFFFFFFFFu; return; }
    h.t = its.t; h.u = its.u; h.v = its.v; h.shape = its.shape; h.prim = its.prim;
    for (int i = 0; i < 3; ++i) {
        h.p[i] = its.p[i]; h.geo_n[i] = its.geoFrame.n[i]; h.sh_n[i] = its.shFrame.n[i];
        h.sh_s[i] = its.shFrame.s[i]; h.sh_t[i] = its.shFrame.t[i]; h.wi[i] = its.wi[i];
    }
    h.uv[0] = its.uv.x; h.uv[1] = its.uv.y;
}

static Ray rayFrom(const float *r) {
    return Ray(Vec3(r[0], r[1], r[2]), Vec3(r[4], r[5], r[6]), r[3], r[7]);
}

template <typename F> static void parallelFor(uint64_t n, int nThreads, F fn) {
    unsigned nt = nThreads > 0 ? (unsigned) nThreads : std::max(1u, std::thread::hardware_concurrency());
    if (n < 4096) nt = 1;
    std::atomic<uint64_t> next(0);
    const uint64_t chunk = 1024;
    auto worker = [&]() {
        for (;;) {
            uint64_t b = next.fetch_add(chunk);
            if (b >= n) break;
            uint64_t e = std::min(n, b + chunk);
            for (uint64_t i = b; i < e; ++i) fn(i);
        }
    };
    if (nt == 1) { worker(); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(worker);
    for (auto &t : th) t.join();
}

extern "C" int orc_trace(const orc_scene *s, uint64_t n, const float *rays, orc_hit *hits, int brute) {
    parallelFor(n, 0, [&](uint64_t i) {
        Ray ray = rayFrom(rays + 8 * i);
        Intersection its;
        bool hit = brute ? s->s.rayIntersectBrute(ray, its, false) : s->s.rayIntersect(ray, its);
        fillHit(its, hit, hits[i]);
    });
    return 0;
}

extern "C" int orc_occluded(const orc_scene *s, uint64_t n, const float *rays, uint8_t *out, int brute) {
    parallelFor(n, 0, [&](uint64_t i) {
        Ray ray = rayFrom(rays + 8 * i);
        Intersection its;
        out[i] = brute ? s->s.rayIntersectBrute(ray, its, true) : s->s.rayIntersectShadow(ray);
    });
    return 0;
}

static Vec2 g_queryUV{ 0, 0 };
extern "C" void orc_bsdf_query_uv(float u, float v) { g_queryUV = Vec2{ u, v }; }

extern "C" int orc_bsdf_eval(const orc_scene *s, int material, uint64_t n, const float *wi, const float *wo, float *out) {
    if (material < 0 || (size_t) material >= s->s.materials.size()) return -1;
    const Material &m = s->s.materials[material];
    g_textureUV = g_queryUV;
    g_texturePartials = UVPartials();
    g_bsdfIts = nullptr;
    for (uint64_t i = 0; i < n; ++i) {
        Spectrum r = bsdfEval(m, Vec3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), Vec3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]));
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
    return 0;
}

extern "C" int orc_bsdf_pdf(const orc_scene *s, int material, uint64_t n, const float *wi, const float *wo, float *out) {
    if (material < 0 || (size_t) material >= s->s.materials.size()) return -1;
    const Material &m = s->s.materials[material];
    g_textureUV = g_queryUV;
    g_texturePartials = UVPartials();
    g_bsdfIts = nullptr;
    for (uint64_t i = 0; i < n; ++i)
        out[i] = bsdfPdf(m, Vec3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), Vec3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]));
    return 0;
}

extern "C" int orc_bsdf_sample(const orc_scene *s, int material, uint64_t n, const float *wi, const float *sample,
                               float *wo, float *weight, float *pdf, float *eta, uint32_t *sampled_type, const float *sampler_1d) {
    if (material < 0 || (size_t) material >= s->s.materials.size()) return -1;
    const Material &m = s->s.materials[material];
    g_textureUV = g_queryUV;
    g_texturePartials = UVPartials();
    g_bsdfIts = nullptr;
    struct Fixed : Sampler1D { float v = 0.5f; float next1D() override { return v; } } fixed;
    for (uint64_t i = 0; i < n; ++i) {
        BSDFSample bs;
        float p;
        fixed.v = sampler_1d ? sampler_1d[i] : 0.5f;
        Spectrum w = bsdfSample(m, Vec3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), Vec2{ sample[2 * i], sample[2 * i + 1] }, bs, p, &fixed);
        wo[3 * i] = bs.wo.x; wo[3 * i + 1] = bs.wo.y; wo[3 * i + 2] = bs.wo.z;
        weight[3 * i] = w.x; weight[3 * i + 1] = w.y; weight[3 * i + 2] = w.z;
        pdf[i] = p; eta[i] = bs.eta; sampled_type[i] = bs.sampledType;
    }
    return 0;
}

extern "C" int orc_sample_emitter_direct(const orc_scene *s, uint64_t n, const float *ref, const float *ref_n,
                                         const float *sample, float *d, float *dist, float *pdf, float *value) {
    for (uint64_t i = 0; i < n; ++i) {
        DirectSample dRec;
        Spectrum v = sampleEmitterDirect(s->s, Vec3(ref[3 * i], ref[3 * i + 1], ref[3 * i + 2]),
                                         Vec3(ref_n[3 * i], ref_n[3 * i + 1], ref_n[3 * i + 2]),
                                         Vec2{ sample[2 * i], sample[2 * i + 1] }, dRec);
        for (int k = 0; k < 3; ++k) { d[3 * i + k] = dRec.d[k]; value[3 * i + k] = v[k]; }
        dist[i] = dRec.dist; pdf[i] = dRec.pdf;
    }
    return 0;
}

extern "C" int orc_texture_eval(const orc_scene *s, int texture, uint64_t n, const float *uv, const float *duv, int gradient, float *out) {
    if (texture < 1 || (size_t) texture > s->s.textures.size()) return -1;
    const Texture &t = s->s.textures[texture - 1];
    for (uint64_t i = 0; i < n; ++i) {
        g_texturePartials = UVPartials();
        if (duv) { g_texturePartials.has = true; g_texturePartials.dudx = duv[4 * i]; g_texturePartials.dvdx = duv[4 * i + 1];
                   g_texturePartials.dudy = duv[4 * i + 2]; g_texturePartials.dvdy = duv[4 * i + 3]; }
        const Vec2 p{ uv[2 * i], uv[2 * i + 1] };
        if (gradient) {
            Spectrum g[2];
            t.evalGradient(p, g);
            for (int k = 0; k < 3; ++k) { out[6 * i + k] = g[0][k]; out[6 * i + 3 + k] = g[1][k]; }
        } else {
            const Spectrum v = t.eval(p);
            for (int k = 0; k < 3; ++k) out[3 * i + k] = v[k];
        }
    }
    g_texturePartials = UVPartials();
    return 0;
}

extern "C" int orc_env_eval(const orc_scene *s, uint64_t n, const float *d, const float *refN, float *value, float *pdf) {
    if (s->s.envEmitter < 0) return -1;
    for (uint64_t i = 0; i < n; ++i) {
        const Vec3 dir(d[3 * i], d[3 * i + 1], d[3 * i + 2]), rn(refN[3 * i], refN[3 * i + 1], refN[3 * i + 2]);
        const Spectrum v = envRadiance(s->s, dir, RayDifferentials());
        for (int k = 0; k < 3; ++k) value[3 * i + k] = v[k];
        pdf[i] = envPdfDirect(s->s, dir, rn);
    }
    return 0;
}

extern "C" int orc_camera_rays(const orc_scene *s, uint64_t n, const float *pos, float *rays) {
    for (uint64_t i = 0; i < n; ++i) {
        Ray r = cameraRay(s->s, pos[2 * i], pos[2 * i + 1]);
        float *o = rays + 8 * i;
        o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.mint;
        o[4] = r.d.x; o[5] = r.d.y; o[6] = r.d.z; o[7] = r.maxt;
    }
    return 0;
}

static bool checkOpts(const Scene &sc, const orc_render_opts *o) {
    const orc_film &f = sc.desc.film;
    if (o->x0 < 0 || o->y0 < 0 || o->x1 > f.crop_w || o->y1 > f.crop_h || o->x0 >= o->x1 || o->y0 >= o->y1) { g_err = "bad rectangle"; return false; }
    if (o->spp <= 0 || o->spp_total < o->sample_offset + o->spp) { g_err = "bad sample counts"; return false; }
    return true;
}

// One camera sample: SamplingIntegrator::renderBlock inner loop (integrator.cpp:169-186)
static inline void renderSample(const Scene &sc, const orc_render_opts *o, int x, int y, int sIdx, float *pos, float *rgba, RayCounters &cnt) {
    uint64_t pixelIndex = (uint64_t) y * sc.desc.film.crop_w + x;
    Pcg32 rng(o->seed, pixelIndex * (uint64_t) o->spp_total + (uint64_t) sIdx);
    Vec2 j = rng.next2D();
    pos[0] = (float) x + j.x; pos[1] = (float) y + j.y;
    RayDifferentials rd;
    Ray ray = sc.anyRayDifferentials ? cameraRay(sc, pos[0], pos[1], &rd, 1.0f / std::sqrt((float) o->spp_total)) : cameraRay(sc, pos[0], pos[1]);
    float alpha;
    const int integrator = sc.desc.integrator.type;
    Spectrum L = integrator == 1 ? LiDirect(sc, ray, rd, rng, alpha, cnt) : integrator == 2 ? LiSimple(sc, ray, rd, rng, alpha, cnt) : Li(sc, ray, rd, rng, alpha, cnt);
    rgba[0] = L.x; rgba[1] = L.y; rgba[2] = L.z; rgba[3] = alpha;
}

extern "C" int orc_li(const orc_scene *s, const orc_render_opts *o, float *out, float *pos, orc_stats *stats) {
    const Scene &sc = s->s;
    if (!checkOpts(sc, o)) return -1;
    int w = o->x1 - o->x0, h = o->y1 - o->y0;
    std::mutex mtx;
    RayCounters total;
    auto t0 = std::chrono::steady_clock::now();
    std::atomic<int> nextRow(0);
    unsigned nt = o->n_threads > 0 ? (unsigned) o->n_threads : std::max(1u, std::thread::hardware_concurrency());
    auto worker = [&]() {
        RayCounters cnt;
        for (;;) {
            int row = nextRow.fetch_add(1);
            if (row >= h) break;
            for (int xx = 0; xx < w; ++xx)
                for (int k = 0; k < o->spp; ++k) {
                    size_t idx = ((size_t) row * w + xx) * o->spp + k;
                    renderSample(sc, o, o->x0 + xx, o->y0 + row, o->sample_offset + k, pos + 2 * idx, out + 4 * idx, cnt);
                }
        }
        std::lock_guard<std::mutex> g(mtx);
        total.normal += cnt.normal; total.shadow += cnt.shadow;
        total.trav.steps += cnt.trav.steps; total.trav.tests += cnt.trav.tests;
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(worker);
    for (auto &t : th) t.join();
    if (stats) {
        stats->normal_rays = total.normal; stats->shadow_rays = total.shadow;
        stats->paths = (uint64_t) w * h * o->spp;
        stats->kd_traversal_steps = total.trav.steps; stats->kd_prim_tests = total.trav.tests;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        stats->threads = (int) nt;
    }
    return 0;
}

// Block-parallel render: 32x32 blocks (scene.cpp:24) handed to a thread pool, each rendered into a
// private bordered block and merged into the film under a mutex (renderproc.cpp:68-86,142-149).
extern "C" int orc_render(const orc_scene *s, const orc_render_opts *o, float *film, orc_stats *stats) {
    const Scene &sc = s->s;
    if (!checkOpts(sc, o)) return -1;
    const orc_film &f = sc.desc.film;
    const int BS = 32;
    int bw = (o->x1 - o->x0 + BS - 1) / BS, bh = (o->y1 - o->y0 + BS - 1) / BS;
    std::mutex mtx;
    RayCounters total;
    std::atomic<int> nextBlock(0);
    unsigned nt = o->n_threads > 0 ? (unsigned) o->n_threads : std::max(1u, std::thread::hardware_concurrency());
    auto t0 = std::chrono::steady_clock::now();
    // The block-local accumulation followed by a merge is what the reference does; summation order
    // therefore differs from a direct film splat only by float re-association.
    auto worker = [&]() {
        RayCounters cnt;
        const int border = (int) std::ceil(f.filter_radius - 0.5f);
        const int tw = BS + 2 * border;
        std::vector<float> tile((size_t) tw * tw * 5);
        for (;;) {
            int b = nextBlock.fetch_add(1);
            if (b >= bw * bh) break;
            int bx0 = o->x0 + (b % bw) * BS, by0 = o->y0 + (b / bw) * BS;
            int bx1 = std::min(bx0 + BS, o->x1), by1 = std::min(by0 + BS, o->y1);
            std::fill(tile.begin(), tile.end(), 0.0f);
            const float scaleFactor = 31.0f / f.filter_radius;
            for (int y = by0; y < by1; ++y)
                for (int x = bx0; x < bx1; ++x)
                    for (int k = 0; k < o->spp; ++k) {
                        float pos[2], rgba[4];
                        renderSample(sc, o, x, y, o->sample_offset + k, pos, rgba, cnt);
                        // ImageBlock::put into the bordered block (imageblock.h:144-189)
                        float temp[5] = { rgba[0], rgba[1], rgba[2], rgba[3], 1.0f };
                        bool bad = false;
                        for (int i = 0; i < 5; ++i) if (!std::isfinite(temp[i]) || temp[i] < 0) bad = true;
                        if (bad) continue;
                        const float px = pos[0] - 0.5f - (float) (bx0 - border), py = pos[1] - 0.5f - (float) (by0 - border);
                        const int minx = std::max((int) std::ceil(px - f.filter_radius), 0), miny = std::max((int) std::ceil(py - f.filter_radius), 0);
                        const int maxx = std::min((int) std::floor(px + f.filter_radius), tw - 1), maxy = std::min((int) std::floor(py + f.filter_radius), tw - 1);
                        float wx[16], wy[16];
                        for (int xx = minx, idx = 0; xx <= maxx; ++xx) wx[idx++] = f.filter_values[std::min((int) std::abs((xx - px) * scaleFactor), 31)];
                        for (int yy = miny, idx = 0; yy <= maxy; ++yy) wy[idx++] = f.filter_values[std::min((int) std::abs((yy - py) * scaleFactor), 31)];
                        for (int yy = miny, yr = 0; yy <= maxy; ++yy, ++yr) {
                            float *dest = tile.data() + ((size_t) yy * tw + minx) * 5;
                            for (int xx = minx, xr = 0; xx <= maxx; ++xx, ++xr) {
                                const float weight = wx[xr] * wy[yr];
                                for (int c = 0; c < 5; ++c) *dest++ += weight * temp[c];
                            }
                        }
                    }
            std::lock_guard<std::mutex> g(mtx);
            for (int ty = 0; ty < tw; ++ty) {
                int fy = by0 - border + ty;
                if (fy < 0 || fy >= f.crop_h) continue;
                for (int tx = 0; tx < tw; ++tx) {
                    int fx = bx0 - border + tx;
                    if (fx < 0 || fx >= f.crop_w) continue;
                    const float *src = tile.data() + ((size_t) ty * tw + tx) * 5;
                    float *dst = film + ((size_t) fy * f.crop_w + fx) * 5;
                    for (int c = 0; c < 5; ++c) dst[c] += src[c];
                }
            }
        }
        std::lock_guard<std::mutex> g(mtx);
        total.normal += cnt.normal; total.shadow += cnt.shadow;
        total.trav.steps += cnt.trav.steps; total.trav.tests += cnt.trav.tests;
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(worker);
    for (auto &t : th) t.join();
    if (stats) {
        stats->normal_rays = total.normal; stats->shadow_rays = total.shadow;
        stats->paths = (uint64_t) (o->x1 - o->x0) * (o->y1 - o->y0) * o->spp;
        stats->kd_traversal_steps = total.trav.steps; stats->kd_prim_tests = total.trav.tests;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        stats->threads = (int) nt;
    }
    return 0;
}

extern "C" int orc_film_put(const orc_scene *s, uint64_t n, const float *pos, const float *value, float *film) {
    for (uint64_t i = 0; i < n; ++i) filmPut(s->s, film, pos[2 * i], pos[2 * i + 1], value + 4 * i);
    return 0;
}

// Bitmap::convert ESpectrumAlphaWeight -> RGBA (src/libcore/fmtconv.cpp:978-990)
extern "C" int orc_film_develop(const orc_scene *s, const float *film, float *rgba) {
    const orc_film &f = s->s.desc.film;
    for (size_t i = 0; i < (size_t) f.crop_w * f.crop_h; ++i) {
        float w = film[5 * i + 4];
        float inv = w != 0 ? 1.0f / w : 0.0f;
        for (int c = 0; c < 4; ++c) rgba[4 * i + c] = film[5 * i + c] * inv;
    }
    return 0;
}

extern "C" int orc_clipped_aabb(const float *v, const float *box, float *out) {
    AABB b;
    b.min = Vec3(box[0], box[1], box[2]); b.max = Vec3(box[3], box[4], box[5]);
    AABB r = triangleClippedAABB(Vec3(v[0], v[1], v[2]), Vec3(v[3], v[4], v[5]), Vec3(v[6], v[7], v[8]), b);
    for (int i = 0; i < 3; ++i) { out[i] = r.min[i]; out[3 + i] = r.max[i]; }
    return 0;
}

extern "C" int orc_microfacet_eval(int distribution, float au, float av, int sv, int kind, uint64_t n,
                                   const float *wi, const float *m, float *out) {
    MicrofacetDistribution d(distribution, au, av, sv != 0);
    for (uint64_t i = 0; i < n; ++i) {
        Vec3 w(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), mm(m[3 * i], m[3 * i + 1], m[3 * i + 2]);
        out[i] = kind == 0 ? d.eval(mm) : kind == 1 ? d.pdf(w, mm) : d.smithG1(w, mm);
    }
    return 0;
}

extern "C" int orc_microfacet_sample(int distribution, float au, float av, int sv, uint64_t n,
                                     const float *wi, const float *sample, float *m, float *pdf) {
    MicrofacetDistribution d(distribution, au, av, sv != 0);
    for (uint64_t i = 0; i < n; ++i) {
        float p;
        Vec3 r = d.sample(Vec3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), Vec2{ sample[2 * i], sample[2 * i + 1] }, p);
        m[3 * i] = r.x; m[3 * i + 1] = r.y; m[3 * i + 2] = r.z; pdf[i] = p;
    }
    return 0;
}

extern "C" int orc_fresnel_dielectric(uint64_t n, const float *cos_i, float eta, float *F, float *cos_t) {
    for (uint64_t i = 0; i < n; ++i) F[i] = fresnelDielectricExt(cos_i[i], cos_t[i], eta);
    return 0;
}

extern "C" int orc_fresnel_conductor(uint64_t n, const float *cos_i, const float *eta3, const float *k3, float *out3) {
    Spectrum e(eta3[0], eta3[1], eta3[2]), k(k3[0], k3[1], k3[2]);
    for (uint64_t i = 0; i < n; ++i) {
        Spectrum r = fresnelConductorExact(cos_i[i], e, k);
        out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
    }
    return 0;
}

extern "C" int orc_rng_stream(uint32_t seed, uint64_t key, uint32_t n, float *out) {
    Pcg32 rng(seed, key);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng.next1D();
    return 0;
}
