#include "host_scene.h"
#include "mipmap.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace mtsamd {

namespace {

// spectrum luminance (include/mitsuba/core/spectrum.h:754-756)
float luminance(const float *s) { return s[0] * 0.212671f + s[1] * 0.715160f + s[2] * 0.072169f; }
float max3(const float *s) { return std::max(std::max(s[0], s[1]), s[2]); }

// BSDF::ensureEnergyConservation (src/librender/bsdf.cpp:88-146): a "scale" texture of 0.99 * max / actualMax
// A colour parameter of a BSDF: a constant or a two-colour procedural texture, with the factors the reference wraps
// around it as `scale` textures (the scene's own, then BSDF::ensureEnergyConservation's, src/librender/bsdf.cpp:88-113)
struct Colour {
    const mtsamd_texture *tex = nullptr;
    int32_t texIndex = 0;               // 1-based, as the material names it
    const MipPyramid *mip = nullptr;    // bitmap textures
    float base0[3], base1[3];
    float userScale = 1.0f, energyScale = 1.0f;

    void value(int which, float out[3]) const {
        const float *b = which ? base1 : base0;
        for (int i = 0; i < 3; ++i) out[i] = (b[i] * userScale) * energyScale;
    }
    void factor(float out[3]) const {   // what a looked-up bitmap value is multiplied by
        for (int i = 0; i < 3; ++i) out[i] = userScale * energyScale;
    }
    void maximum(float out[3]) const {      // Checkerboard / GridTexture / BitmapTexture::getMaximum, ScaleTexture::getMaximum
        for (int i = 0; i < 3; ++i) out[i] = (mip ? mip->maximum[i] : tex ? std::max(base0[i], base1[i]) : base0[i]) * userScale;
    }
    void average(float out[3]) const {      // ::getAverage (checkerboard.cpp:96-98, gridtexture.cpp:113-118, bitmap.cpp:501-511, scale.cpp)
        for (int i = 0; i < 3; ++i) {
            float a = base0[i];
            if (mip) a = mip->average[i];
            else if (tex && tex->type == MTSAMD_TEX_CHECKERBOARD) a = (base0[i] + base1[i]) * 0.5f;
            else if (tex) {
                const float interiorWidth = std::max(0.0f, 1 - 2 * tex->line_width), interiorArea = interiorWidth * interiorWidth,
                            lineArea = 1 - interiorArea;
                a = base1[i] * lineArea + base0[i] * interiorArea;
            }
            out[i] = (a * userScale) * energyScale;
        }
    }
    bool usesRayDifferentials() const { return mip != nullptr; }     // bitmap.cpp:541-543
};
std::string makeColour(const mtsamd_scene_desc *d, const std::vector<MipPyramid> &pyramids, const float *constant, int32_t texture, Colour &c) {
    memcpy(c.base0, constant, 12); memcpy(c.base1, constant, 12);
    if (texture == 0) return "";
    if (texture < 0 || (uint32_t) texture > d->n_textures || !d->textures) return "material texture index out of range";
    c.tex = &d->textures[texture - 1];
    c.texIndex = texture;
    if (c.tex->type < MTSAMD_TEX_CHECKERBOARD || c.tex->type > MTSAMD_TEX_BITMAP) return "unsupported texture type";
    memcpy(c.base0, c.tex->color0, 12); memcpy(c.base1, c.tex->color1, 12);
    if (c.tex->type == MTSAMD_TEX_BITMAP) c.mip = &pyramids[texture - 1];
    c.userScale = c.tex->scale;
    return "";
}
void ensureEnergy(bool on, Colour &a) {
    if (!on) return;
    float m[3]; a.maximum(m);
    const float actualMax = max3(m);
    if (actualMax > 1.0f) a.energyScale = 0.99f * (1.0f / actualMax);
}
void ensureEnergy2(bool on, Colour &a, Colour &b) {
    if (!on) return;
    float ma[3], mb[3]; a.maximum(ma); b.maximum(mb);
    const float sum[3] = { ma[0] + mb[0], ma[1] + mb[1], ma[2] + mb[2] };
    const float actualMax = max3(sum);
    if (actualMax > 1.0f) a.energyScale = b.energyScale = 0.99f * (1.0f / actualMax);
}
float averageLuminance(const Colour &c) { float a[3]; c.average(a); return luminance(a); }

void mul4(const float *a, const float *b, float *out) {
    float r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0;
            for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
            r[i * 4 + j] = s;
        }
    memcpy(out, r, sizeof(r));
}

bool invert4(const float *m, float *out) {
    float a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = m[i * 4 + j]; a[i][j + 4] = i == j ? 1.0f : 0.0f; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
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

// fresnelDielectricExt for a ray arriving from outside (src/libcore/util.cpp:651-681), float like the reference
float fresnelDielectricOutside(float cosThetaI, float eta) {
    if (eta == 1) return 0.0f;
    const float scale = 1 / eta;
    const float cosThetaTSqr = 1 - (1 - cosThetaI * cosThetaI) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) return 1.0f;
    const float cosThetaT = std::sqrt(cosThetaTSqr);
    const float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    const float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    return 0.5f * (Rs * Rs + Rp * Rp);
}

// fresnelDiffuseReflectance(eta, fast = false) (src/libcore/util.cpp:806-859): integral over xi in [0,1] of the
// Fresnel reflectance at cos(theta) = sqrt(xi). The reference integrates adaptively to a relative error of 1e-5;
// composite Simpson over 2^16 panels in double is well inside that.
float fresnelDiffuseReflectance(float eta) {
    const int n = 1 << 16;
    double sum = 0;
    for (int i = 0; i <= n; ++i) {
        const double xi = (double) i / n;
        const double w = (i == 0 || i == n) ? 1.0 : ((i & 1) ? 4.0 : 2.0);
        sum += w * (double) fresnelDielectricOutside((float) std::sqrt(xi), eta);
    }
    return (float) (sum / (3.0 * n));
}

void scale4(float x, float y, float z, float *m) { memset(m, 0, 64); m[0] = x; m[5] = y; m[10] = z; m[15] = 1; }
void translate4(float x, float y, float z, float *m) { scale4(1, 1, 1, m); m[3] = x; m[7] = y; m[11] = z; }

// DiscreteDistribution::append/normalize (include/mitsuba/core/pmf.h:56-114): running float sum, then a multiply by
// the reciprocal of the total with the last entry forced to 1
float normalizeCdf(float *cdf, size_t n /* entries */, float *normalization) {
    float sum = cdf[n];
    if (sum > 0) {
        float norm = 1.0f / sum;
        for (size_t i = 1; i <= n; ++i) cdf[i] *= norm;
        cdf[n] = 1.0f;
        if (normalization) *normalization = norm;
    } else if (normalization) {
        *normalization = 0.0f;
    }
    return sum;
}

}  // namespace

std::string prepareScene(const mtsamd_scene_desc *d, HostScene &hs) {
    if (!d) return "null scene description";
    if (d->n_triangles && (!d->positions || !d->indices)) return "positions/indices missing";
    if (!d->shapes && d->n_shapes) return "shapes missing";
    if (!d->materials || d->n_materials == 0) return "at least one material is required";
    const mtsamd_film &f = d->film;
    if (f.width <= 0 || f.height <= 0 || f.crop_w <= 0 || f.crop_h <= 0 || f.crop_x < 0 || f.crop_y < 0 ||
        f.crop_x + f.crop_w > f.width || f.crop_y + f.crop_h > f.height) return "invalid film/crop size";
    if (!(f.filter_radius > 0) || f.filter_radius > 8.0f) return "reconstruction filter radius must be in (0, 8]";
    if (d->integrator.rr_depth <= 0) return "'rrDepth' must be set to a value greater than zero!";                 // integrator.cpp:220-221
    if (d->integrator.max_depth <= 0 && d->integrator.max_depth != -1)
        return "'maxDepth' must be set to -1 (infinite) or a value greater than zero!";                            // integrator.cpp:223-224
    if (!(d->camera.near_clip > 0)) return "The 'nearClip' parameter must be greater than zero!";                   // sensor.cpp:164-165
    if (!(d->camera.near_clip < d->camera.far_clip)) return "The 'nearClip' parameter must be smaller than 'farClip'.";

    hs.desc = *d;
    hs.positions.assign(d->positions, d->positions + (size_t) d->n_vertices * 3);
    hs.indices.assign(d->indices, d->indices + (size_t) d->n_triangles * 3);
    hs.hasNormals = d->normals != nullptr;
    hs.hasTexcoords = d->texcoords != nullptr;
    if (hs.hasNormals) hs.normals.assign(d->normals, d->normals + (size_t) d->n_vertices * 3);
    if (hs.hasTexcoords) hs.texcoords.assign(d->texcoords, d->texcoords + (size_t) d->n_vertices * 2);
    for (size_t i = 0; i < hs.indices.size(); ++i)
        if (hs.indices[i] >= d->n_vertices) return "vertex index out of range";

    // ---- bitmap textures: one MIP pyramid per texture of the description (TMIPMap's constructor) ----------
    std::vector<MipPyramid> pyramids(d->textures ? d->n_textures : 0);
    std::vector<uint32_t> pyramidFirstLevel(pyramids.size(), 0);
    for (size_t i = 0; i < pyramids.size(); ++i) {
        const mtsamd_texture &t = d->textures[i];
        if (t.type != MTSAMD_TEX_BITMAP) continue;
        const std::string err = buildMipPyramid(t, 1.0f, pyramids[i]);
        if (!err.empty()) return err;
        pyramidFirstLevel[i] = (uint32_t) hs.mipLevels.size();
        const uint32_t base = (uint32_t) (hs.texels.size() / 4);
        for (DevMipLevel L : pyramids[i].levels) { L.offset += base; hs.mipLevels.push_back(L); }
        hs.texels.insert(hs.texels.end(), pyramids[i].texels.begin(), pyramids[i].texels.end());
        pyramids[i].texels.clear(); pyramids[i].texels.shrink_to_fit();
        if (hs.texels.size() / 4 > 0xffffffffull) return "bitmap textures: more than 2^32 texels";
    }
    // a texture of the description as the device evaluates it, with the factors a BSDF wraps around it
    auto deviceTexture = [&](const mtsamd_texture &t, uint32_t index /* 0-based */, const float c0[3], const float c1[3]) {
        DevTexture dt;
        memset(&dt, 0, sizeof(dt));
        dt.scale = 1.0f;
        memcpy(dt.color0, c0, 12); memcpy(dt.color1, c1, 12);
        dt.type = t.type; dt.lineWidth = t.line_width;
        dt.uOffset = t.u_offset; dt.vOffset = t.v_offset; dt.uScale = t.u_scale; dt.vScale = t.v_scale;
        if (t.type == MTSAMD_TEX_BITMAP) {
            const uint32_t words[3] = { pyramidFirstLevel[index], (uint32_t) pyramids[index].levels.size(),
                                        (uint32_t) t.filter_type | ((uint32_t) t.wrap_u << 8) | ((uint32_t) t.wrap_v << 16) };
            memcpy(dt.color1, words, 12);
            dt.lineWidth = t.filter_type == MTSAMD_MIP_EWA ? t.max_anisotropy : 1.0f;     // bitmap.cpp:232-235
        }
        return dt;
    };
    // the textures on their own (mtsamd_texture_eval): the scene's scale factor only
    for (size_t i = 0; i < pyramids.size(); ++i) {
        const mtsamd_texture &t = d->textures[i];
        if (t.type < MTSAMD_TEX_CHECKERBOARD || t.type > MTSAMD_TEX_BITMAP) return "unsupported texture type";
        const float one[3] = { 1.0f, 1.0f, 1.0f };
        hs.plainTextures.push_back(deviceTexture(t, (uint32_t) i, t.type == MTSAMD_TEX_BITMAP ? one : t.color0, t.color1));
        hs.plainTextures.back().scale = t.scale;
    }
    std::vector<uint8_t> wantsPartials(d->n_materials, 0);      // BSDF::usesRayDifferentials of the record on its own

    // ---- materials ------------------------------------------------------------------------------------
    hs.materials.resize(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const mtsamd_material &m = d->materials[i];
        DevMaterial dm;
        memset(&dm, 0, sizeof(dm));
        dm.type = m.type; dm.distribution = m.distribution; dm.sampleVisible = m.sample_visible; dm.variant = m.variant;
        dm.alphaU = m.alpha_u; dm.alphaV = m.alpha_v;
        dm.eta = 1; dm.invEta = 1; dm.invEta2 = 1;
        const bool ee = m.ensure_energy_conservation != 0;
        Colour diff, spec, trans;
        for (const std::string &err : { makeColour(d, pyramids, m.diffuse_reflectance, m.diffuse_texture, diff),
                                        makeColour(d, pyramids, m.specular_reflectance, m.specular_texture, spec),
                                        makeColour(d, pyramids, m.specular_transmittance, m.transmittance_texture, trans) })
            if (!err.empty()) return err;
        {   // m_usesRayDifferentials as the plugins' configure() sets it (e.g. diffuse.cpp:101, roughplastic.cpp:301-304)
            const bool dd = diff.usesRayDifferentials(), ss = spec.usesRayDifferentials(), tt = trans.usesRayDifferentials();
            switch (m.type) {
                case MTSAMD_BSDF_DIFFUSE: wantsPartials[i] = dd; break;
                case MTSAMD_BSDF_ROUGHCONDUCTOR: case MTSAMD_BSDF_CONDUCTOR: wantsPartials[i] = ss; break;
                case MTSAMD_BSDF_ROUGHPLASTIC: case MTSAMD_BSDF_PLASTIC: case MTSAMD_BSDF_WARD: wantsPartials[i] = dd || ss; break;
                case MTSAMD_BSDF_DIELECTRIC: case MTSAMD_BSDF_ROUGHDIELECTRIC: wantsPartials[i] = ss || tt; break;
                default: break;
            }
        }
        // roughness textures (roughconductor.cpp:228-250, roughdielectric.cpp:245-274): evaluated per query; they count
        // for the anisotropy flag and for usesRayDifferentials like the colours
        if (m.alpha_u_texture || m.alpha_v_texture) {
            if (m.type != MTSAMD_BSDF_ROUGHCONDUCTOR && m.type != MTSAMD_BSDF_ROUGHDIELECTRIC && m.type != MTSAMD_BSDF_ROUGHPLASTIC)
                return "a textured roughness is supported for roughconductor, roughdielectric and roughplastic only";
            if (m.type == MTSAMD_BSDF_ROUGHPLASTIC && m.alpha_u_texture != m.alpha_v_texture)
                return "The 'roughplastic' plugin currently does not support anisotropic microfacet distributions!";
            for (int32_t t : { m.alpha_u_texture, m.alpha_v_texture }) {
                if (t < 0 || (uint32_t) t > d->n_textures || (t && !d->textures)) return "roughness texture index out of range";
                if (t && d->textures[t - 1].type == MTSAMD_TEX_BITMAP) wantsPartials[i] = 1;       // bitmap.cpp:541-543
            }
            if (hs.materialTextures.empty()) hs.materialTextures.assign((size_t) d->n_materials * 4, 0u);
            if (hs.materialAlphaTextures.empty()) hs.materialAlphaTextures.assign((size_t) d->n_materials * 2, 0u);
            hs.materialAlphaTextures[(size_t) i * 2] = (uint32_t) m.alpha_u_texture;
            hs.materialAlphaTextures[(size_t) i * 2 + 1] = (uint32_t) m.alpha_v_texture;
            hs.materialTextures[(size_t) i * 4 + 3] |= 8u;
            hs.mixture = true;
        }
        const unsigned aniso = (m.alpha_u != m.alpha_v || m.alpha_u_texture != m.alpha_v_texture) ? EAnisotropic : 0u;
        if (m.type == MTSAMD_BSDF_ROUGHCONDUCTOR || m.type == MTSAMD_BSDF_ROUGHPLASTIC || m.type == MTSAMD_BSDF_ROUGHDIELECTRIC) {
            if (m.distribution != MTSAMD_DISTR_BECKMANN && m.distribution != MTSAMD_DISTR_GGX)
                return "Specified an invalid distribution, must be \"beckmann\" or \"ggx\"";
        }
        switch (m.type) {
            case MTSAMD_BSDF_DIFFUSE:           // diffuse.cpp:93-104
                ensureEnergy(ee, diff);
                { float mx[3]; diff.maximum(mx); dm.typeFlags = max3(mx) > 0 ? (EDiffuseReflection | EFrontSide) : 0u; }
                break;
            case MTSAMD_BSDF_ROUGHCONDUCTOR:    // roughconductor.cpp:187-190,228-250
                ensureEnergy(ee, spec);
                for (int c = 0; c < 3; ++c) {
                    float recip = 1.0f / m.ext_eta;     // Spectrum / Float multiplies by the reciprocal
                    dm.condEta[c] = m.eta[c] * recip;
                    dm.condK[c] = m.k[c] * recip;
                }
                dm.typeFlags = EGlossyReflection | EFrontSide | aniso;
                break;
            case MTSAMD_BSDF_ROUGHPLASTIC: {    // roughplastic.cpp:209-213,255-307
                if (m.int_ior < 0 || m.ext_ior < 0 || m.int_ior == m.ext_ior)
                    return "The interior and exterior indices of refraction must be positive and differ!";
                if (m.alpha_u != m.alpha_v)
                    return "The 'roughplastic' plugin currently does not support anisotropic microfacet distributions!";
                if (!m.rtrans || m.rtrans_n < 2) return "roughplastic needs the reduced rough transmittance table";
                if (m.alpha_u_texture) {
                    // textured roughness: the table stays two-dimensional (the block of mtsamd_rtrans_reduce_eta)
                    if (m.rtrans_n < 4) return "roughplastic with a textured roughness needs the transmittance table reduced to its index of refraction only";
                    const float nTheta = m.rtrans[0], nAlpha = m.rtrans[1], aMin = m.rtrans[2], aMax = m.rtrans[3];
                    if (!(nTheta >= 2 && nAlpha >= 2 && nTheta <= 4096 && nAlpha <= 4096 && aMax > aMin) ||
                        (uint64_t) m.rtrans_n != 4 + (uint64_t) nTheta * (uint64_t) nAlpha + 2 * (uint64_t) nAlpha)
                        return "roughplastic with a textured roughness: malformed two-dimensional transmittance table";
                    // RoughTransmittance::checkAlpha on the texture's extremes (roughplastic.cpp:287-288, rtrans.h:400-407)
                    const mtsamd_texture &tx = d->textures[m.alpha_u_texture - 1];
                    if (tx.type != MTSAMD_TEX_BITMAP) {
                        for (const float *c : { tx.color0, tx.color1 }) {
                            const float avg = ((c[0] + c[1]) + c[2]) * (1.0f / 3.0f) * tx.scale;
                            if (avg < aMin || avg > aMax) return "roughplastic: the roughness texture leaves the range of the transmittance table";
                        }
                    }
                }
                ensureEnergy(ee, spec);
                ensureEnergy(ee, diff);
                dm.eta = m.int_ior / m.ext_ior;
                dm.invEta2 = 1.0f / (dm.eta * dm.eta);
                float dAvg = averageLuminance(diff), sAvg = averageLuminance(spec);
                dm.specSamplingWeight = sAvg / (dAvg + sAvg);
                dm.fdrInt = m.rtrans_diffuse_int;
                dm.variant = (m.variant ? 1 : 0) | (m.alpha_u_texture ? 2 : 0);     // nonlinear; two-dimensional table
                dm.rtransOffset = (uint32_t) hs.rtrans.size();
                dm.rtransN = m.rtrans_n;
                hs.rtrans.insert(hs.rtrans.end(), m.rtrans, m.rtrans + m.rtrans_n);
                dm.typeFlags = EGlossyReflection | EDiffuseReflection | EFrontSide;
                break;
            }
            case MTSAMD_BSDF_DIELECTRIC:        // dielectric.cpp:154-159,184-202
                if (m.int_ior < 0 || m.ext_ior < 0) return "The interior and exterior indices of refraction must be positive!";
                ensureEnergy(ee, spec);
                ensureEnergy(ee, trans);
                dm.eta = m.int_ior / m.ext_ior;
                dm.invEta = 1 / dm.eta;
                dm.typeFlags = EDeltaReflection | EDeltaTransmission | EFrontSide | EBackSide | ENonSymmetric;
                break;
            case MTSAMD_BSDF_WARD: {            // ward.cpp:139-171
                if (m.variant < 0 || m.variant > 2) return "Specified an invalid model type, must be \"ward\", \"ward-duer\", or \"balanced\"!";
                ensureEnergy2(ee, spec, diff);
                float dAvg = averageLuminance(diff), sAvg = averageLuminance(spec);
                dm.specSamplingWeight = sAvg / (dAvg + sAvg);
                dm.typeFlags = EGlossyReflection | EDiffuseReflection | EFrontSide | aniso;
                break;
            }
            case MTSAMD_BSDF_ROUGHDIELECTRIC:   // roughdielectric.cpp:183-211,245-274
                if (m.int_ior < 0 || m.ext_ior < 0 || m.int_ior == m.ext_ior)
                    return "The interior and exterior indices of refraction must be positive and differ!";
                ensureEnergy(ee, spec);
                ensureEnergy(ee, trans);
                dm.eta = m.int_ior / m.ext_ior;
                dm.invEta = 1 / dm.eta;
                dm.typeFlags = EGlossyReflection | EGlossyTransmission | EFrontSide | EBackSide | ENonSymmetric | aniso;
                break;
            case MTSAMD_BSDF_CONDUCTOR:         // conductor.cpp:172-175,187-200
                ensureEnergy(ee, spec);
                for (int c = 0; c < 3; ++c) {
                    float recip = 1.0f / m.ext_eta;
                    dm.condEta[c] = m.eta[c] * recip;
                    dm.condK[c] = m.k[c] * recip;
                }
                dm.typeFlags = EDeltaReflection | EFrontSide;
                break;
            case MTSAMD_BSDF_PLASTIC: {         // plastic.cpp:148-156,187-216
                if (m.int_ior < 0 || m.ext_ior < 0) return "The interior and exterior indices of refraction must be positive!";
                ensureEnergy(ee, spec);
                ensureEnergy(ee, diff);
                dm.eta = m.int_ior / m.ext_ior;
                dm.fdrInt = fresnelDiffuseReflectance(1 / dm.eta);
                float dAvg = averageLuminance(diff), sAvg = averageLuminance(spec);
                dm.specSamplingWeight = sAvg / (dAvg + sAvg);
                dm.invEta2 = 1 / (dm.eta * dm.eta);
                dm.typeFlags = EDeltaReflection | EDiffuseReflection | EFrontSide;
                break;
            }
            case MTSAMD_BSDF_MIXTURE:           // filled in below, once the nested records are known
                break;
            default:
                return "unsupported BSDF type";
        }
        diff.value(0, dm.diffuse); spec.value(0, dm.specular); trans.value(0, dm.transmittance);
        if (m.frame_kind != MTSAMD_FRAME_NONE) {        // bumpmap.cpp:86-99, normalmap.cpp:58-71
            if (m.frame_kind != MTSAMD_FRAME_BUMPMAP && m.frame_kind != MTSAMD_FRAME_NORMALMAP) return "unknown frame adapter";
            if (m.type == MTSAMD_BSDF_MIXTURE) return "bumpmap / normalmap around a mixturebsdf is not supported";
            if (m.frame_texture < 1 || (uint32_t) m.frame_texture > d->n_textures || !d->textures) return "bumpmap / normalmap: texture index out of range";
            if (dm.typeFlags) dm.typeFlags |= EAnisotropic;
            wantsPartials[i] = 1;
            if (hs.materialTextures.empty()) hs.materialTextures.assign((size_t) d->n_materials * 4, 0u);
            hs.materialTextures[(size_t) i * 4 + 3] |= ((uint32_t) m.frame_kind << 1) | ((uint32_t) m.frame_texture << 8);
            hs.mixture = true;
        }
        if (m.type != MTSAMD_BSDF_MIXTURE && (diff.tex || spec.tex || trans.tex)) {
            if (hs.materialTextures.empty()) hs.materialTextures.assign((size_t) d->n_materials * 4, 0u);
            const Colour *slots[3] = { &diff, &spec, &trans };
            for (int k = 0; k < 3; ++k) {
                if (!slots[k]->tex) continue;
                const mtsamd_texture &t = *slots[k]->tex;
                float c0[3], c1[3];
                slots[k]->value(0, c0); slots[k]->value(1, c1);
                if (t.type == MTSAMD_TEX_BITMAP) slots[k]->factor(c0);
                hs.textures.push_back(deviceTexture(t, (uint32_t) slots[k]->texIndex - 1, c0, c1));
                hs.materialTextures[(size_t) i * 4 + k] = (uint32_t) hs.textures.size();
            }
            hs.mixture = true;      // textured scenes run the kernels that know every feature
        }
        dm.ownFlags = dm.typeFlags;
        dm.twoSided = m.two_sided != 0;
        dm.backMaterial = (int32_t) i;
        hs.materials[i] = dm;
    }
    // MixtureBSDF::configure (mixturebsdf.cpp:115-172). The record keeps the count in `variant`, the weights in
    // diffuse[0..2] + rtransOffset, the upper ends of the normalised choice intervals in specular[0..2] + rtransN and
    // the nested records' indices in condEta[0..2] + condK[0] (the last of each group as raw bits) and their BSDF types,
    // a byte each, in `distribution`.
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const mtsamd_material &m = d->materials[i];
        if (m.type != MTSAMD_BSDF_MIXTURE) continue;
        DevMaterial &dm = hs.materials[i];
        const int n = m.mix_count;
        if (n < 1 || n > 4) return "mixturebsdf: between one and four nested BSDFs are supported";
        float total = 0;
        for (int k = 0; k < n; ++k) {
            if (m.mix_weight[k] < 0) return "Invalid BSDF weight!";
            total += m.mix_weight[k];
        }
        if (total <= 0) return "The weights must sum to a value greater than zero!";
        const float scale = (m.ensure_energy_conservation && total > 1) ? 1.0f / total : 1.0f;
        float w[4] = { 0, 0, 0, 0 }, cdf[5] = { 0, 0, 0, 0, 0 };
        int32_t child[4] = { 0, 0, 0, 0 };
        unsigned flags = 0;
        uint32_t childTypes = 0;        // one byte per nested record: its BSDF type
        for (int k = 0; k < n; ++k) {
            const int32_t c = m.mix_material[k];
            if (c < 0 || c >= (int32_t) d->n_materials) return "mixturebsdf: nested material index out of range";
            const int t = d->materials[c].type;
            if (d->materials[c].two_sided || t == MTSAMD_BSDF_MIXTURE)
                return "mixturebsdf: twosided and mixturebsdf cannot be nested";
            if (d->materials[c].frame_kind) return "mixturebsdf: bumpmap and normalmap cannot be nested";
            childTypes |= (uint32_t) t << (8 * k);
            child[k] = c;
            w[k] = scale != 1.0f ? m.mix_weight[k] * scale : m.mix_weight[k];
            cdf[k + 1] = cdf[k] + w[k];
            flags |= hs.materials[c].ownFlags;
            if (wantsPartials[c]) wantsPartials[i] = 1;     // mixturebsdf.cpp:167
        }
        normalizeCdf(cdf, (size_t) n, nullptr);
        for (int k = n + 1; k <= 4; ++k) cdf[k] = cdf[n];
        dm.variant = n;
        dm.distribution = (int32_t) childTypes;
        memcpy(dm.diffuse, w, 12); memcpy(&dm.rtransOffset, &w[3], 4);
        memcpy(dm.specular, &cdf[1], 12); memcpy(&dm.rtransN, &cdf[4], 4);
        memcpy(dm.condEta, child, 12); memcpy(&dm.condK[0], &child[3], 4);
        dm.typeFlags = dm.ownFlags = flags;
        hs.mixture = true;
        hs.blend = true;
    }
    // TwoSidedBRDF::configure (twosided.cpp:84-106): the adapter shows the front BRDF's components on the front side
    // and the back BRDF's on the back side
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        DevMaterial &dm = hs.materials[i];
        if (!dm.twoSided) continue;
        const int32_t b = d->materials[i].back_material;
        if (b >= (int32_t) d->n_materials) return "twosided: back material index out of range";
        if (b >= 0) dm.backMaterial = b;
        const unsigned f = dm.ownFlags, k = hs.materials[dm.backMaterial].ownFlags;
        if ((f | k) & ETransmission) return "Only materials without a transmission component can be nested!";
        dm.typeFlags = (f ? ((f & ~EBackSide) | EFrontSide) : 0u) | (k ? ((k & ~EFrontSide) | EBackSide) : 0u);
    }
    // Which records make Intersection::getBSDF(ray) compute uv partials (records.inl:69-75): BSDF::usesRayDifferentials
    // of what the shape holds, i.e. of the twosided adapter where there is one (twosided.cpp:91-92)
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const DevMaterial &dm = hs.materials[i];
        const bool wants = wantsPartials[i] || (dm.twoSided && wantsPartials[dm.backMaterial]);
        if (wants && !hs.materialTextures.empty()) hs.materialTextures[(size_t) i * 4 + 3] |= 1u;
    }

    // ---- shapes ----------------------------------------------------------------------------------------
    hs.shapes.resize(d->n_shapes);
    hs.triShape.assign(d->n_triangles, 0xFFFFFFFFu);
    for (uint32_t s = 0; s < d->n_shapes; ++s) {
        const mtsamd_shape &sh = d->shapes[s];
        if ((uint64_t) sh.first_triangle + sh.n_triangles > d->n_triangles) return "shape triangle range out of bounds";
        if (sh.material < 0 || (uint32_t) sh.material >= d->n_materials) return "shape material out of range";
        if (sh.emitter >= (int32_t) d->n_emitters) return "shape emitter out of range";
        if ((sh.flags & MTSAMD_SHAPE_HAS_NORMALS) && !hs.hasNormals) return "shape claims normals but none given";
        if ((sh.flags & MTSAMD_SHAPE_HAS_TEXCOORDS) && !hs.hasTexcoords) return "shape claims texcoords but none given";
        DevShape ds;
        memset(&ds, 0, sizeof(ds));
        ds.firstTri = sh.first_triangle; ds.nTris = sh.n_triangles; ds.material = sh.material; ds.emitter = sh.emitter; ds.flags = sh.flags;
        hs.shapes[s] = ds;
        for (uint32_t j = 0; j < sh.n_triangles; ++j) {
            if (hs.triShape[sh.first_triangle + j] != 0xFFFFFFFFu) return "triangle owned by two shapes";
            hs.triShape[sh.first_triangle + j] = s;
        }
    }
    for (uint32_t t = 0; t < d->n_triangles; ++t)
        if (hs.triShape[t] == 0xFFFFFFFFu) return "triangle not owned by any shape";


    if (d->integrator.type == MTSAMD_INTEGRATOR_DIRECT) {   // direct.cpp:96-107
        if (d->integrator.emitter_samples < 0 || d->integrator.bsdf_samples < 0 || d->integrator.emitter_samples + d->integrator.bsdf_samples <= 0)
            return "direct: the numbers of emitter and BSDF samples must not be negative and not both zero";
        if (d->integrator.emitter_samples > 1024 || d->integrator.bsdf_samples > 1024) return "direct: at most 1024 emitter and BSDF samples";
    } else if (d->integrator.type != MTSAMD_INTEGRATOR_PATH && d->integrator.type != MTSAMD_INTEGRATOR_VOLPATH_SIMPLE) {
        return "unknown integrator type";
    }

    // ---- emitters: per-mesh area CDF (TriMesh::prepareSamplingTable, trimesh.cpp:388-403) and the scene's
    //      emitter CDF (scene.cpp:375-380) ------------------------------------------------------------------
    hs.emitters.resize(d->n_emitters);
    hs.emitterCdf.assign(1, 0.0f);
    DevScene envParams;         // the environment map's scalars, copied into the scene's parameters further down
    memset(&envParams, 0, sizeof(envParams));
    for (uint32_t e = 0; e < d->n_emitters; ++e) {
        const mtsamd_emitter &em = d->emitters[e];
        if (em.type != MTSAMD_EMITTER_SURFACE && em.type != MTSAMD_EMITTER_ENVMAP) {
            // PointEmitter, SpotEmitter, DirectionalEmitter: no surface, reached by emitter sampling only
            if (em.type < MTSAMD_EMITTER_POINT || em.type > MTSAMD_EMITTER_DIRECTIONAL) return "unknown emitter type";
            DevEmitter de;
            memset(&de, 0, sizeof(de));
            memcpy(de.radiance, em.radiance, 12);
            de.shape = 0xffffffffu;
            de.kind = (uint32_t) em.type + 1;
            de.delta = (uint32_t) hs.deltaEmitters.size();
            DevDeltaEmitter dd;
            memset(&dd, 0, sizeof(dd));
            const float *M = em.to_world;
            dd.position[0] = M[3]; dd.position[1] = M[7]; dd.position[2] = M[11];
            dd.direction[0] = M[2]; dd.direction[1] = M[6]; dd.direction[2] = M[10];
            // third row of the inverse of the linear part (cofactors over the determinant)
            const double a = M[0], b = M[1], c = M[2], p = M[4], q = M[5], r = M[6], g = M[8], h = M[9], i = M[10];
            const double det = a * (q * i - r * h) - b * (p * i - r * g) + c * (p * h - q * g);
            if (det == 0) return "singular emitter transform";
            dd.toLocalZ[0] = (float) ((p * h - q * g) / det);
            dd.toLocalZ[1] = (float) ((b * g - a * h) / det);
            dd.toLocalZ[2] = (float) ((a * q - b * p) / det);
            if (em.type == MTSAMD_EMITTER_SPOT) {   // spot.cpp:70-74,89-94
                const float cutoff = em.cutoff_angle * (3.14159265358979323846f / 180.0f), beam = em.beam_width * (3.14159265358979323846f / 180.0f);
                if (!(cutoff >= beam)) return "spot emitter: the cutoff angle must not be smaller than the beam width";
                dd.cutoffAngle = cutoff;
                dd.cosBeam = std::cos(beam);
                dd.cosCutoff = std::cos(cutoff);
                dd.invTransitionWidth = 1.0f / (cutoff - beam);
            }
            hs.deltaEmitters.push_back(dd);
            hs.emitters[e] = de;
            hs.emitterCdf.push_back(hs.emitterCdf.back() + em.sampling_weight);
            hs.extEmitters = true;
            continue;
        }
        if (em.type == MTSAMD_EMITTER_ENVMAP) {     // EnvironmentMap::EnvironmentMap / configure (src/emitters/envmap.cpp:105-189,260-329)
            if (hs.envEmitter >= 0) return "The scene may only contain one environment emitter";
            if (!em.pixels || em.width <= 0 || em.height <= 0) return "envmap: invalid image";
            if (std::max(em.width, em.height) > 0xFFFF) return "Environment maps images must be smaller than 65536 pixels in width and height";
            hs.envEmitter = (int32_t) e;
            mtsamd_texture image;
            memset(&image, 0, sizeof(image));
            image.type = MTSAMD_TEX_BITMAP; image.pixels = em.pixels; image.width = em.width; image.height = em.height; image.channels = 3;
            image.filter_type = MTSAMD_MIP_EWA; image.wrap_u = MTSAMD_WRAP_REPEAT; image.wrap_v = MTSAMD_WRAP_CLAMP; image.max_anisotropy = 10.0f;
            MipPyramid pyr;
            const std::string err = buildMipPyramid(image, std::numeric_limits<float>::infinity(), pyr);
            if (!err.empty()) return "envmap: " + err;
            DevScene &p = envParams;
            p.envIsMap = 1;
            p.envFirstLevel = (uint32_t) hs.mipLevels.size(); p.envLevels = (uint32_t) pyr.levels.size();
            const uint32_t base = (uint32_t) (hs.texels.size() / 4);
            for (DevMipLevel L : pyr.levels) { L.offset += base; hs.mipLevels.push_back(L); }
            hs.texels.insert(hs.texels.end(), pyr.texels.begin(), pyr.texels.end());
            const int w = em.width, h = em.height;
            p.envW = (uint32_t) w; p.envH = (uint32_t) h; p.envScale = em.scale;
            hs.envCdfCols.resize((size_t) (w + 1) * h); hs.envCdfRows.resize(h + 1); hs.envRowWeights.resize(h);
            size_t colPos = 0, rowPos = 0;
            float rowSum = 0.0f;
            hs.envCdfRows[rowPos++] = 0;
            for (int y = 0; y < h; ++y) {
                float colSum = 0;
                hs.envCdfCols[colPos++] = 0;
                for (int x = 0; x < w; ++x) {
                    const uint16_t *t = &pyr.texels[((size_t) y * w + x) * 4];      // the stored texel, as the reference reads it back
                    const float rgb[3] = { fromHalf(t[0]), fromHalf(t[1]), fromHalf(t[2]) };
                    colSum += luminance(rgb);
                    hs.envCdfCols[colPos++] = colSum;
                }
                const float normalization = 1.0f / colSum;
                for (int x = 1; x < w; ++x) hs.envCdfCols[colPos - x - 1] *= normalization;
                hs.envCdfCols[colPos - 1] = 1.0f;
                const float weight = std::sin((y + 0.5f) * 3.14159265358979323846f / h);
                hs.envRowWeights[y] = weight;
                rowSum += colSum * weight;
                hs.envCdfRows[rowPos++] = rowSum;
            }
            const float normalization = 1.0f / rowSum;
            for (int y = 1; y < h; ++y) hs.envCdfRows[rowPos - y - 1] *= normalization;
            hs.envCdfRows[rowPos - 1] = 1.0f;
            if (rowSum == 0) return "The environment map is completely black -- this is not allowed.";
            if (!std::isfinite(rowSum)) return "The environment map contains an invalid floating point value (nan/inf) -- giving up.";
            const float pi = 3.14159265358979323846f;
            p.envNormalization = 1.0f / (rowSum * (2 * pi / w) * (pi / h));
            p.envPixelX = 2 * pi / w; p.envPixelY = pi / h;
            const float *T = em.to_world;
            const double a = T[0], b = T[1], c = T[2], dd = T[4], ee = T[5], f = T[6], g = T[8], hh = T[9], i = T[10];
            const double det = a * (ee * i - f * hh) - b * (dd * i - f * g) + c * (dd * hh - ee * g);
            if (det == 0) return "envmap: singular toWorld transform";
            const double inv[9] = { (ee * i - f * hh) / det, (c * hh - b * i) / det, (b * f - c * ee) / det,
                                    (f * g - dd * i) / det, (a * i - c * g) / det, (c * dd - a * f) / det,
                                    (dd * hh - ee * g) / det, (b * g - a * hh) / det, (a * ee - b * dd) / det };
            for (int k = 0; k < 9; ++k) { p.envToWorld[k] = T[(k / 3) * 4 + k % 3]; p.envToLocal[k] = (float) inv[k]; }
            DevEmitter de;
            memset(&de, 0, sizeof(de));
            de.shape = 0xffffffffu;
            de.kind = 1;            // the environment emitter; DevScene::envIsMap tells the map from the constant
            hs.emitters[e] = de;
            hs.emitterCdf.push_back(hs.emitterCdf.back() + em.sampling_weight);
            hs.extEmitters = true;
            continue;
        }
        if (em.shape < 0) {     // constant environment emitter (src/emitters/constant.cpp); one at most (scene.cpp:509-513)
            if (hs.envEmitter >= 0) return "The scene may only contain one environment emitter";
            hs.envEmitter = (int32_t) e;
            DevEmitter de;
            memset(&de, 0, sizeof(de));
            memcpy(de.radiance, em.radiance, 12);
            de.shape = 0xffffffffu;
            de.kind = 1;
            hs.emitters[e] = de;
            hs.emitterCdf.push_back(hs.emitterCdf.back() + em.sampling_weight);
            hs.extEmitters = true;
            continue;
        }
        if ((uint32_t) em.shape >= d->n_shapes) return "emitter shape out of range";
        if (hs.shapes[em.shape].emitter != (int32_t) e) return "shape/emitter cross reference mismatch";
        const DevShape &sh = hs.shapes[em.shape];
        if (sh.nTris == 0) return "Encountered an empty triangle mesh!";
        DevEmitter de;
        memset(&de, 0, sizeof(de));
        memcpy(de.radiance, em.radiance, 12);
        de.shape = em.shape; de.firstTri = sh.firstTri; de.nTris = sh.nTris;
        de.cdfOffset = (uint32_t) hs.areaCdf.size();
        size_t base = hs.areaCdf.size();
        hs.areaCdf.push_back(0.0f);
        for (uint32_t j = 0; j < sh.nTris; ++j) {
            const uint32_t t = sh.firstTri + j;
            const float *p0 = &hs.positions[3 * (size_t) hs.indices[3 * (size_t) t]], *p1 = &hs.positions[3 * (size_t) hs.indices[3 * (size_t) t + 1]],
                        *p2 = &hs.positions[3 * (size_t) hs.indices[3 * (size_t) t + 2]];
            float a[3] = { p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2] }, b[3] = { p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2] };
            float c[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
            float area = 0.5f * std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);     // Triangle::surfaceArea (triangle.cpp:61-67)
            hs.areaCdf.push_back(hs.areaCdf.back() + area);
        }
        float total = normalizeCdf(&hs.areaCdf[base], sh.nTris, nullptr);
        de.invArea = 1.0f / total;
        hs.emitters[e] = de;
        hs.emitterCdf.push_back(hs.emitterCdf.back() + em.sampling_weight);
    }
    if (d->n_emitters) {
        float norm;
        normalizeCdf(hs.emitterCdf.data(), d->n_emitters, &norm);
        for (uint32_t e = 0; e < d->n_emitters; ++e) hs.emitters[e].pdfDiscrete = d->emitters[e].sampling_weight * norm;   // scene.h:848-850
    }

    // ---- acceleration structure ---------------------------------------------------------------------------
    // Default: the GPU builds its own structure (tens of milliseconds for millions of triangles). MTSAMD_BVH_BUILDER=host
    // selects the binned-SAH sweep on the host cores instead: a few percent fewer node visits per ray, a build that
    // takes 25x longer.
    int builder = BVH_BUILDER_DEVICE;
    if (const char *e = getenv("MTSAMD_BVH_BUILDER"))
        builder = strcmp(e, "device") == 0 ? BVH_BUILDER_DEVICE : strcmp(e, "device-binary") == 0 ? BVH_BUILDER_DEVICE_BINARY : BVH_BUILDER_HOST;
    hs.bvh.builder = builder;
    if (builder == BVH_BUILDER_DEVICE && d->n_triangles > 0) {
        // the structure itself is built after the upload, straight into device memory; only the tight bounds are needed here
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (size_t i = 0; i < hs.indices.size(); ++i) {
            const float *v = &hs.positions[3 * (size_t) hs.indices[i]];
            if (!(std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]))) return "non-finite vertex position";
            for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], v[k]); hi[k] = std::max(hi[k], v[k]); }
        }
        for (int k = 0; k < 3; ++k) { hs.bvh.aabbMin[k] = lo[k]; hs.bvh.aabbMax[k] = hi[k]; }
        hs.bvh.nodes.clear(); hs.bvh.tris.clear();
    } else {
        hs.bvh.builder = builder == BVH_BUILDER_DEVICE ? BVH_BUILDER_HOST : builder;
        const char *err = buildBVH(hs.positions.data(), d->n_vertices, hs.indices.data(), d->n_triangles, hs.bvh, hs.bvh.builder);
        if (err && *err) return err;
    }

    // ---- packed per-triangle shading records; the BVH's triangle records get the BSDF type in their spare word so
    //      that a hit carries it (bits 28..30 of the triangle word of a hit) -----------------------------------------
    if (d->n_triangles >= (1u << 27)) return "too many triangles";
    hs.shadeTris.assign((size_t) d->n_triangles * 4 * kShadeTriStride, 0.0f);
    for (uint32_t t = 0; t < d->n_triangles; ++t) {
        const DevShape &sh = hs.shapes[hs.triShape[t]];
        float *g = &hs.shadeTris[(size_t) t * 4 * kShadeTriStride];
        const uint32_t w[3] = { (uint32_t) sh.material | (sh.flags << 24), (uint32_t) sh.emitter, hs.triShape[t] };
        for (int k = 0; k < 3; ++k) {
            const uint32_t vi = hs.indices[3 * (size_t) t + k];
            memcpy(g + 4 * k, &hs.positions[3 * (size_t) vi], 12);
            memcpy(g + 4 * k + 3, &w[k], 4);
            if (sh.flags & MTSAMD_SHAPE_HAS_NORMALS) memcpy(g + 12 + 4 * k, &hs.normals[3 * (size_t) vi], 12);
        }
        // the three texture coordinates ride in the record's own 128-byte line (the position partials of
        // TriMesh::computeUVTangents are worked out from them at the hit, uv_tangents.h)
        if (sh.flags & MTSAMD_SHAPE_HAS_TEXCOORDS)
            for (int k = 0; k < 3; ++k) memcpy(g + 24 + 2 * k, &hs.texcoords[2 * (size_t) hs.indices[3 * (size_t) t + k]], 8);
    }
    if (d->n_materials >= (1u << 24)) return "too many materials";
    // the BSDF type of a triangle's material travels in the top bits of the record's triangle word (the traversal hands it on with the hit)
    for (TriRecord &r : hs.bvh.tris) r.tri |= (uint32_t) std::min(hs.materials[hs.shapes[hs.triShape[r.tri]].material].type, 7) << 28;

    // ---- scalar parameters -------------------------------------------------------------------------------------
    DevScene &p = hs.params;
    memset(&p, 0, sizeof(p));
    p.nEmitters = d->n_emitters; p.nTris = d->n_triangles; p.nShapes = d->n_shapes; p.nMaterials = d->n_materials;
    {   // few area lights and nothing else: their tables fit the shading kernel's workgroup memory
        uint32_t tris = 0;
        bool areaOnly = !hs.extEmitters && !hs.emitters.empty() && hs.emitters.size() <= kLdsEmitters;
        for (const DevEmitter &e : hs.emitters) { areaOnly = areaOnly && e.kind == 0; tris += e.nTris; }
        p.areaCdfLen = (uint32_t) hs.areaCdf.size();
        p.emitterTris = areaOnly && tris <= kLdsEmitterTris && p.areaCdfLen <= kLdsEmitterTris + kLdsEmitters ? tris : 0u;
    }
    // enlarged scene bounds (gkdtree.h:1213-1220, MTS_KD_AABB_EPSILON = 1e-3f; note max grows by the already grown extent)
    {
        float lo[3], hi[3];
        const float eps = 1e-3f;
        for (int i = 0; i < 3; ++i) { lo[i] = hs.bvh.aabbMin[i]; hi[i] = hs.bvh.aabbMax[i]; }
        if (d->n_triangles) {
            for (int i = 0; i < 3; ++i) lo[i] = lo[i] - ((hi[i] - lo[i]) * eps + eps);
            for (int i = 0; i < 3; ++i) hi[i] = hi[i] + ((hi[i] - lo[i]) * eps + eps);
        }
        for (int i = 0; i < 3; ++i) { p.aabbMin[i] = lo[i]; p.aabbMax[i] = hi[i]; }
    }
    p.envEmitter = hs.envEmitter;
    if (hs.envEmitter >= 0) {
        // Scene::initializeBidirectional (scene.cpp:386-413): kd-tree bounds grown by the sensor position; AABB::getBSphere
        // (aabb.cpp:44-47); ConstantBackgroundEmitter::createShape (constant.cpp:68-72) takes 1.5 x that radius
        float lo[3], hi[3], c[3], r2 = 0;
        const float cam[3] = { d->camera.to_world[3], d->camera.to_world[7], d->camera.to_world[11] };
        for (int i = 0; i < 3; ++i) {
            lo[i] = std::min(p.aabbMin[i], cam[i]); hi[i] = std::max(p.aabbMax[i], cam[i]);
            c[i] = (hi[i] + lo[i]) * 0.5f;
        }
        for (int i = 0; i < 3; ++i) r2 += (c[i] - hi[i]) * (c[i] - hi[i]);
        memcpy(p.envCenter, c, 12);
        p.envRadius = std::max(1e-4f, std::sqrt(r2) * 1.5f);
        memcpy(p.envRadiance, hs.emitters[hs.envEmitter].radiance, 12);
        p.envPdfDiscrete = hs.emitters[hs.envEmitter].pdfDiscrete;
        p.envIsMap = envParams.envIsMap;
        p.envFirstLevel = envParams.envFirstLevel; p.envLevels = envParams.envLevels; p.envW = envParams.envW; p.envH = envParams.envH;
        p.envNormalization = envParams.envNormalization; p.envScale = envParams.envScale;
        p.envPixelX = envParams.envPixelX; p.envPixelY = envParams.envPixelY;
        memcpy(p.envToWorld, envParams.envToWorld, sizeof(p.envToWorld)); memcpy(p.envToLocal, envParams.envToLocal, sizeof(p.envToLocal));
    }
    if (!hs.deltaEmitters.empty()) {
        // DirectionalEmitter::createShape (directional.cpp:88-94): 1.1 x the bounding sphere (aabb.cpp:44-47) of the
        // kd-tree bounds; the light starts from the disk at center - d * radius (directional.cpp:161-162)
        float c[3], r2 = 0;
        for (int i = 0; i < 3; ++i) c[i] = (p.aabbMax[i] + p.aabbMin[i]) * 0.5f;
        for (int i = 0; i < 3; ++i) r2 += (c[i] - p.aabbMax[i]) * (c[i] - p.aabbMax[i]);
        const float radius = std::sqrt(r2) * 1.1f;
        for (uint32_t e = 0; e < d->n_emitters; ++e) {
            if (hs.emitters[e].kind != 4) continue;
            DevDeltaEmitter &dd = hs.deltaEmitters[hs.emitters[e].delta];
            for (int i = 0; i < 3; ++i) dd.position[i] = c[i] - dd.direction[i] * radius;
        }
    }
    // camera (perspective.cpp:126-157, transform.cpp:99-123)
    {
        const float aspect = (float) f.width / (float) f.height;
        p.invResX = 1.0f / (float) f.crop_w; p.invResY = 1.0f / (float) f.crop_h;
        const float relSizeX = (float) f.crop_w / (float) f.width, relSizeY = (float) f.crop_h / (float) f.height;
        const float relOffX = (float) f.crop_x / (float) f.width, relOffY = (float) f.crop_y / (float) f.height;
        float persp[16];
        const float nearC = d->camera.near_clip, farC = d->camera.far_clip;
        const float recip = 1.0f / (farC - nearC);
        const float cot = 1.0f / std::tan((d->camera.x_fov / 2.0f) * (3.14159265358979323846f / 180.0f));
        memset(persp, 0, 64);
        persp[0] = cot; persp[5] = cot; persp[10] = farC * recip; persp[11] = -nearC * farC * recip; persp[14] = 1;
        float m[16], tmp[16];
        scale4(1.0f / relSizeX, 1.0f / relSizeY, 1.0f, m);
        translate4(-relOffX, -relOffY, 0.0f, tmp); mul4(m, tmp, m);
        scale4(-0.5f, -0.5f * aspect, 1.0f, tmp); mul4(m, tmp, m);
        translate4(-1.0f, -1.0f / aspect, 0.0f, tmp); mul4(m, tmp, m);
        mul4(m, persp, m);
        if (!invert4(m, p.sampleToCamera)) return "singular camera projection";
        memcpy(p.camToWorld, d->camera.to_world, 48);
        p.nearClip = nearC; p.farClip = farC;
    }
    p.cropW = f.crop_w; p.cropH = f.crop_h;
    p.filterRadius = f.filter_radius;
    memcpy(p.filterValues, f.filter_values, sizeof(p.filterValues));
    p.maxDepth = d->integrator.max_depth; p.rrDepth = d->integrator.rr_depth;
    p.strictNormals = d->integrator.strict_normals; p.hideEmitters = d->integrator.hide_emitters;
    return "";
}

}  // namespace mtsamd
