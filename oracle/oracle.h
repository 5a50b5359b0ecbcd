// TEST INFRASTRUCTURE ONLY -- C interface of the CPU oracle (restatement of the Mitsuba 0.5
// `path` integrator hot path). Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
// bench.py may load this library; the GPU product never calls into it.
//
// The scene description structs below have the same memory layout as the ones the GPU
// library takes (include/mtsamd.h) so that one flattened scene can be handed to both sides.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_BSDF_DIFFUSE = 0, ORC_BSDF_ROUGHCONDUCTOR = 1, ORC_BSDF_ROUGHPLASTIC = 2,
       ORC_BSDF_DIELECTRIC = 3, ORC_BSDF_WARD = 4, ORC_BSDF_CONDUCTOR = 5, ORC_BSDF_PLASTIC = 6,
       ORC_BSDF_ROUGHDIELECTRIC = 7, ORC_BSDF_MIXTURE = 8 };
enum { ORC_DISTR_BECKMANN = 0, ORC_DISTR_GGX = 1 };
enum { ORC_SHAPE_HAS_NORMALS = 1, ORC_SHAPE_HAS_TEXCOORDS = 2 };

typedef struct {
    uint32_t first_triangle;
    uint32_t n_triangles;
    int32_t material;
    int32_t emitter;   /* -1: not an emitter */
    uint32_t flags;
} orc_shape;

typedef struct {
    int32_t type;
    int32_t distribution;
    int32_t sample_visible;
    int32_t variant;            /* ward: 0 ward, 1 ward-duer, 2 balanced; roughplastic: nonlinear */
    float alpha_u, alpha_v;
    float int_ior, ext_ior;     /* dielectric / roughplastic */
    float diffuse_reflectance[3];
    float specular_reflectance[3];
    float specular_transmittance[3];
    float eta[3], k[3];         /* conductor eta, k (before division by ext_eta) */
    float ext_eta;              /* conductor exterior IOR */
    int32_t ensure_energy_conservation;
    const float *rtrans;        /* roughplastic: reduced external transmittance table T(mu) */
    uint32_t rtrans_n;
    float rtrans_diffuse_ext;
    float rtrans_diffuse_int;
    int32_t two_sided;          /* nested in a twosided adapter (src/bsdfs/twosided.cpp) */
    int32_t back_material;      /* record seen from behind, -1 = the same one */
    /* MixtureBSDF (src/bsdfs/mixturebsdf.cpp), type ORC_BSDF_MIXTURE: up to four nested BSDF records and their weights */
    int32_t mix_count;
    int32_t mix_material[4];
    float mix_weight[4];
    /* a <texture> in place of the constant colours above: 1-based index into orc_scene_desc.textures, 0 = none */
    int32_t diffuse_texture, specular_texture, transmittance_texture;
    /* the record is the BSDF nested in a bumpmap (1, src/bsdfs/bumpmap.cpp) or normalmap (2, src/bsdfs/normalmap.cpp)
       adapter with that displacement / normal texture (1-based) */
    int32_t frame_kind, frame_texture;
    int32_t alpha_u_texture, alpha_v_texture;   /* textured roughness (roughconductor, roughdielectric; roughplastic with the table of orc_rtrans_reduce_eta), 1-based, 0 = constant */
} orc_material;

/* type 1: checkerboard (src/textures/checkerboard.cpp), 2: gridtexture (src/textures/gridtexture.cpp), 3: bitmap
   (src/textures/bitmap.cpp: `pixels` = width * height * channels linear floats, rows from the top, channels 1 or 3;
   filter_type = EMIPFilterType, wrap_u / wrap_v = ReconstructionFilter::EBoundaryCondition); uv offset and
   scale of Texture2D (src/librender/texture.cpp:81-95); `scale`: factor of a ScaleTexture around it (src/textures/scale.cpp) */
typedef struct {
    int32_t type;
    float color0[3], color1[3];
    float line_width;
    float u_offset, v_offset, u_scale, v_scale;
    float scale;
    const float *pixels;
    int32_t width, height, channels;
    int32_t filter_type, wrap_u, wrap_v;
    float max_anisotropy;
} orc_texture;

/* type 0: attached to `shape` (area light) or, with shape = -1, the constant environment emitter;
   1: point (src/emitters/point.cpp), 2: spot (src/emitters/spot.cpp), 3: directional (src/emitters/directional.cpp),
   4: environment map (src/emitters/envmap.cpp).
   `radiance` holds the intensity of a point/spot light and the irradiance of a directional one. */
typedef struct {
    int32_t shape;
    float radiance[3];
    float sampling_weight;
    int32_t type;
    float to_world[16];         /* row-major emitter-to-world; the emitter looks along +z */
    float cutoff_angle, beam_width;     /* spot, degrees */
    /* type 4: environment map (src/emitters/envmap.cpp): latitude-longitude image of width * height * 3 linear floats,
       first row = top (+y of the emitter's frame); `scale` multiplies the radiance; shape is -1 */
    const float *pixels;
    int32_t width, height;
    float scale;
} orc_emitter;

typedef struct {
    float to_world[16];         /* row-major camera-to-world */
    float x_fov;                /* degrees */
    float near_clip, far_clip;
} orc_camera;

typedef struct {
    int32_t width, height;
    int32_t crop_x, crop_y, crop_w, crop_h;
    float filter_radius;
    float filter_values[32];    /* ReconstructionFilter::m_values */
} orc_film;

/* type 0: `path` (src/integrators/path/path.cpp); 1: `direct` (src/integrators/direct/direct.cpp) with its numbers
   of emitter and BSDF samples per camera sample (max_depth and rr_depth are not used by it) */
typedef struct {
    int32_t max_depth, rr_depth, strict_normals, hide_emitters;
    int32_t type, emitter_samples, bsdf_samples;
} orc_integrator;

typedef struct {
    uint32_t n_vertices, n_triangles, n_shapes, n_materials, n_emitters;
    const float *positions;     /* n_vertices * 3 */
    const float *normals;       /* n_vertices * 3 or NULL */
    const float *texcoords;     /* n_vertices * 2 or NULL */
    const uint32_t *indices;    /* n_triangles * 3, global vertex indices */
    const orc_shape *shapes;
    const orc_material *materials;
    const orc_emitter *emitters;
    orc_camera camera;
    orc_film film;
    orc_integrator integrator;
    uint32_t n_textures;
    const orc_texture *textures;
} orc_scene_desc;

typedef struct {
    int32_t spp;                /* samples per pixel in this call */
    int32_t spp_total;          /* stride of the per-pixel sample index in the RNG key */
    int32_t sample_offset;      /* first sample index */
    uint32_t seed;
    int32_t x0, y0, x1, y1;     /* pixel rectangle [x0,x1) x [y0,y1) inside the crop window */
    int32_t n_threads;          /* 0 = hardware_concurrency */
} orc_render_opts;

typedef struct {
    uint64_t normal_rays, shadow_rays, paths;
    uint64_t kd_traversal_steps, kd_prim_tests;
    double seconds;
    int32_t threads;
} orc_stats;

/* Per-hit record mirroring the Intersection fields the integrator consumes. */
typedef struct {
    float t, u, v;
    uint32_t shape, prim;       /* 0xFFFFFFFF = miss */
    float p[3], geo_n[3], sh_n[3], sh_s[3], sh_t[3], wi[3], uv[2];
} orc_hit;

typedef struct orc_scene orc_scene;

orc_scene *orc_scene_create(const orc_scene_desc *desc);
void orc_scene_destroy(orc_scene *s);
const char *orc_last_error(void);

/* rays: n * 8 floats (ox,oy,oz,mint, dx,dy,dz,maxt) */
int orc_trace(const orc_scene *s, uint64_t n, const float *rays, orc_hit *hits, int brute_force);
int orc_occluded(const orc_scene *s, uint64_t n, const float *rays, uint8_t *out, int brute_force);

/* BSDF batch queries in the local shading frame. wi/wo: n*3, sample: n*2.
   eval -> out[n*3] (f*cos), pdf -> out[n], sample -> wo[n*3], weight[n*3], pdf[n], eta[n], type[n] */
/* its.uv for the three BSDF queries below (where a material's textures are evaluated); (0, 0) until set */
void orc_bsdf_query_uv(float u, float v);
int orc_bsdf_eval(const orc_scene *s, int material, uint64_t n, const float *wi, const float *wo, float *out);
int orc_bsdf_pdf(const orc_scene *s, int material, uint64_t n, const float *wi, const float *wo, float *out);
/* sampler_1d (n values, may be NULL = 0.5): the number a BSDF flagged EUsesSampler takes from the sampler itself */
int orc_bsdf_sample(const orc_scene *s, int material, uint64_t n, const float *wi, const float *sample,
                    float *wo, float *weight, float *pdf, float *eta, uint32_t *sampled_type, const float *sampler_1d);

/* Texture::eval(its) of texture `texture` (1-based, as the materials name it) without any BSDF around it: uv n*2;
   duv n*4 (dudx, dvdx, dudy, dvdy: Intersection::hasUVPartials set) or NULL (unfiltered lookup) -> out n*3.
   gradient != 0: Texture::evalGradient (d/du, d/dv) instead, out n*6 */
int orc_texture_eval(const orc_scene *s, int texture, uint64_t n, const float *uv, const float *duv, int gradient, float *out);

/* The environment emitter for n world-space directions: value n*3 (Scene::evalEnvironment, no offset rays) and pdf n
   (pdfDirect in the solid-angle measure times the emitter's discrete probability; ref_n n*3, zero = no normal) */
int orc_env_eval(const orc_scene *s, uint64_t n, const float *d, const float *ref_n, float *value, float *pdf);

/* Emitter sampling without the visibility test: ref[n*3], ref_n[n*3], sample[n*2] ->
   d[n*3], dist[n], pdf[n], value[n*3] (value = radiance/pdf/emPdf, 0 if rejected) */
int orc_sample_emitter_direct(const orc_scene *s, uint64_t n, const float *ref, const float *ref_n,
                              const float *sample, float *d, float *dist, float *pdf, float *value);

/* Camera rays for given sample positions: pos[n*2] (fractional pixel coords) -> rays[n*8] */
int orc_camera_rays(const orc_scene *s, uint64_t n, const float *pos, float *rays);

/* Radiance per camera sample (no film): for each pixel in the rectangle and each of spp samples,
   out[(pixel*spp + s)*4] = (R,G,B,alpha); pos[(pixel*spp+s)*2] = sample position. */
int orc_li(const orc_scene *s, const orc_render_opts *opts, float *out, float *pos, orc_stats *stats);

/* Full render into a (crop_w * crop_h * 5) float film (R,G,B,alpha,weight), accumulated. */
int orc_render(const orc_scene *s, const orc_render_opts *opts, float *film, orc_stats *stats);

/* ImageBlock::put for explicit samples: pos[n*2], value[n*4] (R,G,B,alpha) -> film += */
int orc_film_put(const orc_scene *s, uint64_t n, const float *pos, const float *value, float *film);
/* HDRFilm::develop weight division: film (w*h*5) -> rgba (w*h*4) */
int orc_film_develop(const orc_scene *s, const float *film, float *rgba);

/* Triangle::getClippedAABB (double precision Sutherland-Hodgman), v: 9 floats, box: 6 floats -> out 6 */
int orc_clipped_aabb(const float *v, const float *box, float *out);

/* RoughTransmittance: load a data/microfacet/<name>.dat image and reduce it exactly like
   RoughPlastic::configure (setEta on the external and internal copies, setAlpha on the external).
   out_T must hold theta_samples (returned) floats. */
int orc_rtrans_reduce(const void *dat, uint64_t size, float eta, float alpha, float *out_T,
                      uint32_t out_cap, uint32_t *theta_samples, float *diffuse_ext, float *diffuse_int);
/* Full 3-D lookup RoughTransmittance::eval(cosTheta, alpha, eta) on the raw table */
int orc_rtrans_reduce_eta(const void *dat, uint64_t size, float eta, float *out, uint32_t out_cap, uint32_t *n);
int orc_rtrans_eval3d(const void *dat, uint64_t size, uint64_t n, const float *cos_theta,
                      const float *alpha, const float *eta, float *out);

/* Microfacet helpers exposed for the property tests (chi^2, pdf consistency):
   kind 0: D(m), 1: pdfVisible/pdfAll(wi, m), 2: smithG1(wi, m) */
int orc_microfacet_eval(int distribution, float alpha_u, float alpha_v, int sample_visible, int kind,
                        uint64_t n, const float *wi, const float *m, float *out);
int orc_microfacet_sample(int distribution, float alpha_u, float alpha_v, int sample_visible,
                          uint64_t n, const float *wi, const float *sample, float *m, float *pdf);
/* fresnelDielectricExt / fresnelConductorExact */
int orc_fresnel_dielectric(uint64_t n, const float *cos_i, float eta, float *F, float *cos_t);
int orc_fresnel_conductor(uint64_t n, const float *cos_i, const float *eta3, const float *k3, float *out3);

/* The shared counter-based sample stream: n floats of stream (seed, key) */
int orc_rng_stream(uint32_t seed, uint64_t key, uint32_t n, float *out);

#ifdef __cplusplus
}
#endif
