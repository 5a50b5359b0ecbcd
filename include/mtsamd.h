/*
 * mtsamd -- C ABI of the MI355X path-tracing core (libmtsamd.so).
 *
 * The library replaces what runs below SamplingIntegrator::render for the `path` integrator of
 * Mitsuba 0.5 (src/librender/integrator.cpp:95-188 -> src/integrators/path/path.cpp:119-294 and
 * everything that loop calls). A host adapter (an Integrator plugin inside a Mitsuba build, or the
 * standalone driver in this repository) flattens the scene graph into the plain arrays below, calls
 * mtsamd_render*, and hands the resulting (R,G,B,alpha,weight) film to Film::setBitmap / develop.
 *
 * Conventions: every function returns 0 on success or a negative MTSAMD_E* code; the message of the
 * last failure on the calling thread is available from mtsamd_last_error(). Nothing throws, nothing
 * aborts. The caller owns every input array (they are copied during mtsamd_scene_create) and every
 * output buffer. One render per scene handle at a time; mtsamd_cancel is the only call that may be
 * made from another thread while a render is running.
 */
#ifndef MTSAMD_H
#define MTSAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    MTSAMD_OK = 0,
    MTSAMD_EINVAL = -1,     /* malformed description / argument */
    MTSAMD_EDEVICE = -2,    /* HIP runtime error, no device, out of memory */
    MTSAMD_ECANCELLED = -3, /* mtsamd_cancel was called */
    MTSAMD_EUNSUPPORTED = -4
};

/* BSDF plugins on the hot path (src/bsdfs/{diffuse,roughconductor,roughplastic,dielectric,ward}.cpp) and their
   neighbours (src/bsdfs/{conductor,plastic,roughdielectric}.cpp). The `twosided` adapter (src/bsdfs/twosided.cpp) is not a
   type of its own: it is the two_sided / back_material pair of the record of the BRDF it wraps. */
enum { MTSAMD_BSDF_DIFFUSE = 0, MTSAMD_BSDF_ROUGHCONDUCTOR = 1, MTSAMD_BSDF_ROUGHPLASTIC = 2,
       MTSAMD_BSDF_DIELECTRIC = 3, MTSAMD_BSDF_WARD = 4, MTSAMD_BSDF_CONDUCTOR = 5, MTSAMD_BSDF_PLASTIC = 6,
       MTSAMD_BSDF_ROUGHDIELECTRIC = 7, MTSAMD_BSDF_MIXTURE = 8 };
/* MicrofacetDistribution::EType (src/bsdfs/microfacet.h:48-57); Phong is not supported */
enum { MTSAMD_DISTR_BECKMANN = 0, MTSAMD_DISTR_GGX = 1 };
enum { MTSAMD_SHAPE_HAS_NORMALS = 1, MTSAMD_SHAPE_HAS_TEXCOORDS = 2 };
/* Who builds the binary tree that is collapsed into the 8-wide BVH (ShapeKDTree::build in the reference,
   src/librender/skdtree.cpp:68-110): Morton sort + locally-ordered clustering on the GPU with a SAH sweep over the
   resulting clusters for the top of the tree (default), or a binned-SAH sweep over all triangles on the host cores.
   Selected with the environment variable MTSAMD_BVH_BUILDER=device|host when the scene is created. */
enum { MTSAMD_BVH_HOST = 0, MTSAMD_BVH_DEVICE = 1,
       MTSAMD_BVH_DEVICE_BINARY = 2 /* MTSAMD_BVH_BUILDER=device-binary: clustering on the GPU, collapse on the host (for comparisons) */ };

/* One TriMesh (include/mitsuba/render/trimesh.h). Its triangles occupy
   [first_triangle, first_triangle + n_triangles) of the global index array; the primitive index the
   reference reports (Intersection::primIndex) is the offset inside that range. */
typedef struct {
    uint32_t first_triangle;
    uint32_t n_triangles;
    int32_t material;       /* Shape::getBSDF() */
    int32_t emitter;        /* Shape::getEmitter(), -1 if none */
    uint32_t flags;         /* MTSAMD_SHAPE_* : which per-vertex arrays are valid for this mesh */
} mtsamd_shape;

/* Constructor-level BSDF parameters (what the XML properties give). Derived quantities
   (energy-conservation scaling, sampling weights, eta ratios) are computed by the library exactly as
   the plugins' configure() methods do. */
typedef struct {
    int32_t type;               /* MTSAMD_BSDF_* */
    int32_t distribution;       /* MTSAMD_DISTR_* */
    int32_t sample_visible;     /* microfacet.h m_sampleVisible (default true) */
    int32_t variant;            /* ward: 0 ward, 1 ward-duer, 2 balanced; roughplastic, plastic: nonlinear flag */
    float alpha_u, alpha_v;
    float int_ior, ext_ior;     /* dielectric, roughplastic, plastic, roughdielectric */
    float diffuse_reflectance[3];   /* diffuse: reflectance; roughplastic, ward: diffuseReflectance */
    float specular_reflectance[3];
    float specular_transmittance[3];
    float eta[3], k[3];         /* roughconductor, conductor: before the division by ext_eta (roughconductor.cpp:189-190) */
    float ext_eta;
    int32_t ensure_energy_conservation;   /* BSDF::m_ensureEnergyConservation (default true) */
    /* roughplastic: external rough transmittance reduced to 1-D, i.e. the state RoughPlastic::configure
       leaves m_externalRoughTransmittance in (roughplastic.cpp:281-299); see mtsamd_rtrans_reduce */
    const float *rtrans;
    uint32_t rtrans_n;
    float rtrans_diffuse_ext;   /* m_externalRoughTransmittance->evalDiffuse(alpha) */
    float rtrans_diffuse_int;   /* m_internalRoughTransmittance->evalDiffuse(alpha) */
    /* TwoSidedBRDF (src/bsdfs/twosided.cpp:84-159): non-zero = this record is the BRDF nested in a `twosided`
       adapter; back_material is the record used when the surface is seen from behind (second nested BRDF),
       -1 = the same record on both sides. Only reflection-only BRDFs may be nested (twosided.cpp:103-105). */
    int32_t two_sided;
    int32_t back_material;
    /* MixtureBSDF (src/bsdfs/mixturebsdf.cpp:66-85,115-172), type MTSAMD_BSDF_MIXTURE: the records of up to four nested
       BSDFs and their linear weights (rescaled when they sum to more than one and ensure_energy_conservation is set).
       Nested records must be one-sided and not mixtures themselves. */
    int32_t mix_count;
    int32_t mix_material[4];
    float mix_weight[4];
    /* Textures in place of the three constant spectra above (a <texture> child named reflectance / diffuseReflectance,
       specularReflectance, specularTransmittance): 1-based index into mtsamd_scene_desc.textures, 0 = the constant.
       The BSDF evaluates them at the intersection's uv like `m_reflectance->eval(bRec.its)` (e.g. diffuse.cpp:110-118). */
    int32_t diffuse_texture, specular_texture, transmittance_texture;
    /* BumpMap / NormalMap (src/bsdfs/bumpmap.cpp:135-237, src/bsdfs/normalmap.cpp:107-260): frame_kind != 0 = this record is
       the BSDF nested in such an adapter, which evaluates it in a shading frame perturbed by the gradient of the
       displacement texture (MTSAMD_FRAME_BUMPMAP) or turned towards the normal the texture stores as a colour
       (MTSAMD_FRAME_NORMALMAP); frame_texture is that texture (1-based). Any record but a mixture may be wrapped, a wrapped
       record may in turn sit in a `twosided` adapter; wrapped records cannot be nested in a mixture. */
    int32_t frame_kind, frame_texture;
    /* Textured roughness of roughconductor and roughdielectric (a <texture> child named alpha, or alphaU / alphaV): 1-based
       texture indices, 0 = the constants alpha_u / alpha_v. The BSDF takes `m_alphaU->eval(bRec.its).average()` at every
       query (src/bsdfs/roughconductor.cpp:257-270, src/bsdfs/roughdielectric.cpp:296-316), with the texture's own
       colours and scale (no energy-conservation factor). roughplastic (one texture: alpha_u_texture == alpha_v_texture,
       roughplastic.cpp:336-341) then needs `rtrans` to be the block of mtsamd_rtrans_reduce_eta instead of the
       one-dimensional table: transmittances are looked up at the textured roughness (rtrans.h:195-205, 254-259);
       rtrans_diffuse_ext / rtrans_diffuse_int are not used. The texture's values must lie inside the table's roughness
       range (RoughTransmittance::checkAlpha, rtrans.h:400-407): checked for procedural textures at scene creation,
       not for bitmaps (a value outside gives a transmittance of 0, as the reference's lookup does). */
    int32_t alpha_u_texture, alpha_v_texture;
} mtsamd_material;
enum { MTSAMD_FRAME_NONE = 0, MTSAMD_FRAME_BUMPMAP = 1, MTSAMD_FRAME_NORMALMAP = 2 };

/* 2-D textures (Texture2D, src/librender/texture.cpp:81-121: uv * (uscale, vscale) + (uoffset, voffset)), then
   src/textures/checkerboard.cpp:66-75, src/textures/gridtexture.cpp:75-88 or, MTSAMD_TEX_BITMAP, the image lookup of
   src/textures/bitmap.cpp:431-499 over the MIP map of include/mitsuba/render/mipmap.h (levels resampled with the
   two-lobed Lanczos filter and kept in half precision like the reference's; nearest, bilinear, trilinear and
   elliptically-weighted-average lookups; the filtered ones use the uv partials Intersection::computePartials derives
   from the camera ray's offset rays, i.e. at the first vertex of a path, src/librender/intersection.cpp:5-76). */
enum { MTSAMD_TEX_CHECKERBOARD = 1, MTSAMD_TEX_GRID = 2, MTSAMD_TEX_BITMAP = 3 };
/* EMIPFilterType (mipmap.h:52-61) and ReconstructionFilter::EBoundaryCondition (include/mitsuba/core/rfilter.h:53-64) */
enum { MTSAMD_MIP_NEAREST = 0, MTSAMD_MIP_BILINEAR = 1, MTSAMD_MIP_TRILINEAR = 2, MTSAMD_MIP_EWA = 3 };
enum { MTSAMD_WRAP_CLAMP = 0, MTSAMD_WRAP_REPEAT = 1, MTSAMD_WRAP_MIRROR = 2, MTSAMD_WRAP_ZERO = 3, MTSAMD_WRAP_ONE = 4 };
typedef struct {
    int32_t type;
    float color0[3], color1[3];
    float line_width;           /* gridtexture (default 0.01) */
    float u_offset, v_offset, u_scale, v_scale;
    float scale;                /* factor of a `scale` texture wrapped around it (src/textures/scale.cpp); 1 = none */
    /* MTSAMD_TEX_BITMAP: the decoded image as BitmapTexture hands it to its MIP map (bitmap.cpp:246-300): width * height *
       channels linear floats (gamma already undone), first row = top of the image, channels 1 (luminance) or 3 (RGB);
       see mtsamd_image_load. filterType (default ewa), wrapModeU / wrapModeV (default repeat), maxAnisotropy (default 20). */
    const float *pixels;
    int32_t width, height, channels;
    int32_t filter_type, wrap_u, wrap_v;
    float max_anisotropy;
} mtsamd_texture;

/* Emitter record. type MTSAMD_EMITTER_SURFACE: an AreaLight attached to a mesh (src/emitters/area.cpp; the shape's
   `emitter` field must point back at this record) or, with shape = -1, the constant environment emitter
   (src/emitters/constant.cpp; at most one per scene, src/librender/scene.cpp:509-513).
   The other types have no surface and are only reached by emitter sampling: MTSAMD_EMITTER_POINT
   (src/emitters/point.cpp, `radiance` = intensity), MTSAMD_EMITTER_SPOT (src/emitters/spot.cpp, `radiance` = intensity,
   constant texture) and MTSAMD_EMITTER_DIRECTIONAL (src/emitters/directional.cpp, `radiance` = irradiance); they take
   their position and direction (+z) from to_world, and `shape` is ignored. */
enum { MTSAMD_EMITTER_SURFACE = 0, MTSAMD_EMITTER_POINT = 1, MTSAMD_EMITTER_SPOT = 2, MTSAMD_EMITTER_DIRECTIONAL = 3,
       MTSAMD_EMITTER_ENVMAP = 4 };
typedef struct {
    int32_t shape;
    float radiance[3];
    float sampling_weight;      /* Emitter::getSamplingWeight() */
    int32_t type;
    float to_world[16];         /* row-major emitter-to-world (Emitter::getWorldTransform) */
    float cutoff_angle;         /* spot: degrees (spot.cpp:70) */
    float beam_width;           /* spot: degrees (spot.cpp:71) */
    /* MTSAMD_EMITTER_ENVMAP (src/emitters/envmap.cpp): the environment emitter as a latitude-longitude image, width * height
       * 3 linear floats (see mtsamd_image_load; a luminance image is replicated by the caller), first row = +y of the
       emitter's frame (to_world), radiance multiplied by `scale` (default 1). Sampled by luminance * sin(theta) with
       bilinear interpolation (envmap.cpp:567-633); rays that leave the scene look it up bilinearly, camera rays through the
       elliptical filter of its MIP map (envmap.cpp:380-410). `shape` is -1, `radiance` is ignored. At most one
       environment emitter (this or the constant one) per scene. */
    const float *pixels;
    int32_t width, height;
    float scale;
} mtsamd_emitter;

/* PerspectiveCamera (src/sensors/perspective.cpp) */
typedef struct {
    float to_world[16];         /* row-major camera-to-world, no scale */
    float x_fov;                /* degrees along x (PerspectiveCamera::getXFov) */
    float near_clip, far_clip;
} mtsamd_camera;

/* Film size/crop (src/librender/film.cpp:25-55) and the discretised reconstruction filter
   (ReconstructionFilter::m_values, include/mitsuba/core/rfilter.h:76-97) */
typedef struct {
    int32_t width, height;
    int32_t crop_x, crop_y, crop_w, crop_h;
    float filter_radius;
    float filter_values[32];
} mtsamd_film;

/* MonteCarloIntegrator parameters (src/librender/integrator.cpp:190-225). type MTSAMD_INTEGRATOR_PATH is the `path`
   integrator (src/integrators/path/path.cpp); MTSAMD_INTEGRATOR_DIRECT is `direct` (src/integrators/direct/direct.cpp:
   emitter_samples / bsdf_samples per camera sample, not both zero; max_depth and rr_depth are not used by it);
   MTSAMD_INTEGRATOR_VOLPATH_SIMPLE is `volpath_simple` (src/integrators/path/volpath_simple.cpp:88-297) on a scene
   without participating media, which is the only kind this library takes: a path tracer without multiple importance
   sampling (emitters are seen by the camera ray and after Dirac bounces; every other vertex samples an emitter at full
   weight, with the visibility ray of Scene::sampleAttenuatedEmitterDirect, src/librender/scene.cpp:876-898). */
enum { MTSAMD_INTEGRATOR_PATH = 0, MTSAMD_INTEGRATOR_DIRECT = 1, MTSAMD_INTEGRATOR_VOLPATH_SIMPLE = 2 };
typedef struct {
    int32_t max_depth, rr_depth, strict_normals, hide_emitters;
    int32_t type, emitter_samples, bsdf_samples;
} mtsamd_integrator;

typedef struct {
    uint32_t n_vertices, n_triangles, n_shapes, n_materials, n_emitters;
    const float *positions;     /* n_vertices * 3 (world space) */
    const float *normals;       /* n_vertices * 3 or NULL */
    const float *texcoords;     /* n_vertices * 2 or NULL */
    const uint32_t *indices;    /* n_triangles * 3, indices into the global vertex arrays */
    const mtsamd_shape *shapes;
    const mtsamd_material *materials;
    const mtsamd_emitter *emitters;
    mtsamd_camera camera;
    mtsamd_film film;
    mtsamd_integrator integrator;
    uint32_t n_textures;
    const mtsamd_texture *textures;     /* what the materials' *_texture fields index (1-based); NULL if none */
} mtsamd_scene_desc;

typedef struct {
    int32_t spp;            /* samples per pixel rendered by this call */
    int32_t spp_total;      /* stride of the per-pixel sample index in the sample-stream key */
    int32_t sample_offset;  /* index of the first sample (progressive / multi-pass rendering) */
    uint32_t seed;
    int32_t x0, y0, x1, y1; /* pixel rectangle [x0,x1) x [y0,y1) of the crop window rendered by this call
                               (the unit a multi-GPU driver deals out); all zero = whole crop window */
    int32_t spp_per_pass;   /* samples per pixel traced together in one wavefront pass; 0 = automatic */
    /* Interleaved tile partition (one GPU's share of the frame, the counterpart of the blocks one worker gets from
       BlockedImageProcess, src/librender/imageproc.cpp:43-78): with band_rows = h > 0 the call renders, of the
       rectangle, only the rows y0 + k * band_period + j, j < h, i.e. full-width tiles of h rows, every
       (band_period / h)-th of them, in ONE wavefront pass. GPU r of n passes y0 = r * h, band_period = n * h.
       band_rows = 0: the whole rectangle. */
    int32_t band_rows, band_period;
} mtsamd_render_opts;

/* Counters matching the reference's statistics: "Normal rays traced" / "Shadow rays traced"
   (src/librender/skdtree.cpp:46-47,123,210) */
typedef struct {
    uint64_t normal_rays, shadow_rays, paths;
    uint64_t bounce_iterations;
    double seconds;             /* wall time of the render loop, device-synchronised */
    double extend_ms, shade_ms, shadow_ms, film_ms, raygen_ms;   /* per-kernel-class device time (HIP events) */
    uint64_t extend_launches, shadow_launches;
    /* interior-node visits and triangle tests of the closest-hit and of the shadow queries; filled only while
       traversal counting is enabled (mtsamd_scene_set_counting) */
    uint64_t node_visits, tri_tests, shadow_node_visits, shadow_tri_tests;
    /* the launch that follows the last few paths of a pass through all their remaining bounces: its device time and
       the closest-hit / shadow rays it traced (already contained in normal_rays / shadow_rays, not in extend_ms) */
    double tail_ms;
    uint64_t tail_normal_rays, tail_shadow_rays;
    /* mtsamd_render_multi only: the time this device's share of the film took to reach the host (its bands and their
       filter borders, through page-locked memory); entry 0 also carries the host-side sum of the shares */
    double gather_ms;
} mtsamd_stats;

typedef struct {
    uint32_t n_nodes, n_triangles, depth;
    uint64_t node_bytes, triangle_bytes;
    double build_seconds;       /* whole acceleration-structure build */
    double sah_cost;
    double binary_seconds;      /* of which: the binary tree under the wide BVH (host SAH sweep or device clustering) */
    int32_t builder;            /* MTSAMD_BVH_*: who built the tree in the end (a scene whose GPU-built tree would be deeper than
                                   the traversal stack is rebuilt by the host builder, which bounds the depth) */
    int32_t reserved;
} mtsamd_bvh_info;

/* Closest-hit record, the fields of ShapeKDTree::IntersectionCache + t
   (include/mitsuba/render/skdtree.h:237-241) */
typedef struct {
    float t, u, v;
    uint32_t shape, prim;   /* 0xFFFFFFFF = miss */
} mtsamd_hit;

/* Intersection fields the integrator consumes (include/mitsuba/render/shape.h:36-172) */
typedef struct {
    float t, u, v;
    uint32_t shape, prim;
    float p[3], geo_n[3], sh_n[3], sh_s[3], sh_t[3], wi[3], uv[2];
} mtsamd_intersection;

typedef struct mtsamd_scene mtsamd_scene;

const char *mtsamd_last_error(void);

/* The records of this header change between versions of the library (fields are added at the end of a record, records
   are never reordered). A caller compiled against another version must not pass them in: compare MTSAMD_ABI_VERSION with
   mtsamd_abi_version(), and where records are mirrored by hand (FFI bindings) their sizes with mtsamd_sizeof(). No
   counterpart in the reference, whose plugins are compiled against its own headers. */
#define MTSAMD_ABI_VERSION 14
enum { MTSAMD_SIZEOF_SCENE_DESC = 0, MTSAMD_SIZEOF_MATERIAL = 1, MTSAMD_SIZEOF_EMITTER = 2, MTSAMD_SIZEOF_SHAPE = 3,
       MTSAMD_SIZEOF_RENDER_OPTS = 4, MTSAMD_SIZEOF_STATS = 5, MTSAMD_SIZEOF_LDR_OPTS = 6, MTSAMD_SIZEOF_TEXTURE = 7 };
int mtsamd_abi_version(void);
int64_t mtsamd_sizeof(int record);      /* -1 for an unknown record */

/* Select the HIP device used by subsequent calls on this thread (one process per GPU: pass LOCAL_RANK).
   Replaces nothing in the reference (it has no device notion); mirrors Scheduler worker registration
   in src/mitsuba/mitsuba.cpp:278-329. */
int mtsamd_init(int device);

/* Scene::initialize + ShapeKDTree::build (src/librender/scene.cpp:322-384, src/librender/skdtree.cpp:68-110):
   validates and copies the description, configures materials, builds the wide BVH and uploads everything. */
int mtsamd_scene_create(const mtsamd_scene_desc *desc, mtsamd_scene **out);
void mtsamd_scene_destroy(mtsamd_scene *scene);
int mtsamd_scene_bvh_info(const mtsamd_scene *scene, mtsamd_bvh_info *info);

/* SamplingIntegrator::render (src/librender/integrator.cpp:95-129) for the `path` integrator:
   renders opts->spp samples per pixel of the rectangle and ACCUMULATES into film, a
   crop_w*crop_h*5 float array of (R,G,B,alpha,weight) -- the ESpectrumAlphaWeight storage of
   HDRFilm (src/films/hdrfilm.cpp:351-356). mtsamd_render takes a host film, mtsamd_render_device a
   device pointer on the current HIP device (e.g. a torch tensor that is then reduced over RCCL). */
int mtsamd_render(mtsamd_scene *scene, const mtsamd_render_opts *opts, float *film_host, mtsamd_stats *stats);
int mtsamd_render_device(mtsamd_scene *scene, const mtsamd_render_opts *opts, float *film_device, mtsamd_stats *stats);

/* The same job over several GPUs of one node (the role of the Scheduler's workers, src/librender/integrator.cpp:95-129
   and src/librender/renderproc.cpp:142-149): scenes[i] is the same description created on device i (mtsamd_init(i)
   followed by mtsamd_scene_create, per device). The rectangle of opts is cut into full-width tiles of band_rows rows
   (0 = 16); GPU i renders tiles i, i + n, i + 2n, ... on a host thread of its own, and the n films are added into
   film_host (host memory, ACCUMULATED like mtsamd_render) in device order. stats, if not NULL, has n_scenes entries,
   one per GPU (their `seconds` show the balance of the partition, `gather_ms` what the film hand-over cost). A GPU hands
   back only the rows of its tiles and their filter borders, through page-locked memory. opts->band_rows / band_period
   must be zero. */
int mtsamd_render_multi(mtsamd_scene *const *scenes, int32_t n_scenes, const mtsamd_render_opts *opts, int32_t band_rows,
                        float *film_host, mtsamd_stats *stats);

/* Per-ray traversal statistics, the counterpart of rayIntersectHavranCollectStatistics
   (include/mitsuba/render/sahkdtree3.h:330-429): when enabled, renders run instrumented traversal kernels and
   report node visits / triangle tests in mtsamd_stats. Slower; off by default. */
int mtsamd_scene_set_counting(mtsamd_scene *scene, int enabled);

/* Integrator::cancel (include/mitsuba/render/integrator.h:84): polled between wavefront iterations */
void mtsamd_cancel(mtsamd_scene *scene);

/* HDRFilm::develop weight division (src/libcore/fmtconv.cpp:978-990): film (w*h*5) -> rgba (w*h*4), host */
int mtsamd_film_develop(const mtsamd_scene *scene, const float *film_host, float *rgba_host);

/* LDRFilm::develop (src/films/ldrfilm.cpp:300-321): weight division, tone mapping (exposure + gamma, or Reinhard's
   operator, src/libcore/bitmap.cpp:1707-1854), gamma curve and 8-bit quantisation (src/libcore/fmtconv.cpp:1104-1160).
   film (width*height*5) -> out (width*height*channels bytes; 1, 2, 3 or 4 channels by pixel format), host memory;
   needs no scene and no device. The banner the reference
   stamps into the corner (ldrfilm.cpp:323-336) is not drawn. */
enum { MTSAMD_PIXEL_LUMINANCE = 0, MTSAMD_PIXEL_LUMINANCE_ALPHA = 1, MTSAMD_PIXEL_RGB = 2, MTSAMD_PIXEL_RGBA = 3 };
enum { MTSAMD_TONEMAP_GAMMA = 0, MTSAMD_TONEMAP_REINHARD = 1 };
typedef struct {
    int32_t pixel_format;   /* ldrfilm.cpp:159-170, default rgb */
    int32_t tonemap;        /* ldrfilm.cpp:150-157, default gamma */
    float gamma;            /* -1 = sRGB curve (default) */
    float exposure;         /* f-stops, gamma method (default 0) */
    float key, burn;        /* Reinhard (defaults 0.18, 0) */
} mtsamd_ldr_opts;
int mtsamd_film_develop_ldr(const float *film_host, int32_t width, int32_t height, const mtsamd_ldr_opts *opts, uint8_t *out_host);

/* ---- batch entry points (host arrays in, host arrays out) used by the parity harness --------------- */

/* Scene::rayIntersect(ray, its) (include/mitsuba/render/scene.h:187-189): rays = n*8 floats
   (ox,oy,oz,mint, dx,dy,dz,maxt); mint == 1e-4f triggers the adaptive epsilon of skdtree.cpp:126-129 */
int mtsamd_trace(mtsamd_scene *scene, uint64_t n, const float *rays, mtsamd_hit *hits);
/* Scene::rayIntersect(ray) shadow query (src/librender/skdtree.cpp:207-226) */
int mtsamd_occluded(mtsamd_scene *scene, uint64_t n, const float *rays, uint8_t *occluded);
/* The same two queries for callers whose rays already live in device memory (same layouts, device pointers of the
   device the scene was created on). kernel_seconds, if not NULL, receives the duration of the traversal launch
   measured with HIP events on the library's stream. */
int mtsamd_trace_device(mtsamd_scene *scene, uint64_t n, const float *rays_device, mtsamd_hit *hits_device, double *kernel_seconds);
int mtsamd_occluded_device(mtsamd_scene *scene, uint64_t n, const float *rays_device, uint8_t *occluded_device, double *kernel_seconds);
/* mtsamd_trace_device for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
   hipMalloc). No counterpart in the reference. The call still sets the device, launches on the library's own stream and
   waits for it, so it is not something to record into a caller's HIP graph; what it guarantees is that it allocates nothing. Same query, same results; its work
   memory (ray hand-out counters and the raw hits) is a static device array inside the library instead of buffers
   allocated per call, so one call takes at most mtsamd_trace_device_static_max() rays (2^20; 16 MiB of device memory
   per GPU the library runs on, held until the process ends) and fails with MTSAMD_EINVAL beyond that. The call
   allocates and frees nothing. Every call on a GPU uses that GPU's one array: calls that overlap on the same GPU
   (other threads, other scenes) race; calls in sequence are independent of each other. */
int mtsamd_trace_device_static(mtsamd_scene *scene, uint64_t n, const float *rays_device, mtsamd_hit *hits_device);
uint64_t mtsamd_trace_device_static_max(void);
/* rayIntersect + fillIntersectionRecord (include/mitsuba/render/skdtree.h:343-428) */
int mtsamd_intersect(mtsamd_scene *scene, uint64_t n, const float *rays, mtsamd_intersection *its);

/* BSDF::eval / pdf / sample(bRec, pdf, sample) in the local shading frame, ERadiance mode, all components.
   wi, wo: n*3; sample: n*2; eval -> n*3 (value includes the cosine factor); sample -> wo n*3, weight n*3,
   pdf n, eta n, sampled_type n (BSDF::EBSDFType bits). sampler_1d (n values, may be NULL = 0.5): the number a BSDF
   flagged EUsesSampler (roughdielectric) takes from the sampler itself, bRec.sampler->next1D() */
/* Where on the surface (its.uv) the three queries below evaluate a material's textures; (0, 0) until set */
int mtsamd_bsdf_query_uv(mtsamd_scene *scene, float u, float v);
int mtsamd_bsdf_eval(mtsamd_scene *scene, int material, uint64_t n, const float *wi, const float *wo, float *out);
int mtsamd_bsdf_pdf(mtsamd_scene *scene, int material, uint64_t n, const float *wi, const float *wo, float *out);
int mtsamd_bsdf_sample(mtsamd_scene *scene, int material, uint64_t n, const float *wi, const float *sample,
                       float *wo, float *weight, float *pdf, float *eta, uint32_t *sampled_type, const float *sampler_1d);

/* Texture::eval(its) of texture `texture` (1-based, as the materials name it) on its own, with the `scale` factor but
   without a BSDF's energy-conservation factor: uv n*2; duv n*4 (dudx, dvdx, dudy, dvdy: an intersection with uv
   partials, Texture2D::eval -> eval(uv, d0, d1), texture.cpp:112-121) or NULL (no partials: the full-resolution lookup)
   -> out n*3 */
int mtsamd_texture_eval(mtsamd_scene *scene, int texture, uint64_t n, const float *uv, const float *duv, float *out);

/* The environment emitter for n world-space directions d (n*3): Scene::evalEnvironment for a ray without offset rays
   (value n*3; src/emitters/constant.cpp, src/emitters/envmap.cpp:380-410) and Emitter::pdfDirect in the solid-angle
   measure times the probability of picking the emitter (pdf n; for the constant emitter relative to the reference
   normals ref_n n*3, a zero normal standing for none; constant.cpp:218-233, envmap.cpp:545-551, :603-633).
   MTSAMD_EINVAL without an environment emitter. */
int mtsamd_env_eval(mtsamd_scene *scene, uint64_t n, const float *d, const float *ref_n, float *value, float *pdf);

/* Scene::sampleEmitterDirect without the visibility test (src/librender/scene.cpp:828-852):
   ref n*3, ref_n n*3 (DirectSamplingRecord::refN), sample n*2 -> d n*3, dist n, pdf n, value n*3 */
int mtsamd_sample_emitter_direct(mtsamd_scene *scene, uint64_t n, const float *ref, const float *ref_n,
                                 const float *sample, float *d, float *dist, float *pdf, float *value);

/* PerspectiveCameraImpl::sampleRayDifferential (src/sensors/perspective.cpp:271-298), ray part only:
   pos n*2 fractional pixel coordinates relative to the crop window -> rays n*8 */
int mtsamd_camera_rays(mtsamd_scene *scene, uint64_t n, const float *pos, float *rays);

/* MIPathTracer::Li per camera sample (no film): out[(pixel*spp+s)*4] = (R,G,B,alpha) for the pixels of the
   rectangle in row-major order, pos[(pixel*spp+s)*2] = sample position */
int mtsamd_li(mtsamd_scene *scene, const mtsamd_render_opts *opts, float *out, float *pos, mtsamd_stats *stats);

/* ImageBlock::put (include/mitsuba/render/imageblock.h:124-204) for explicit samples: pos n*2,
   value n*4 (R,G,B,alpha); film host array, accumulated */
int mtsamd_film_put(mtsamd_scene *scene, uint64_t n, const float *pos, const float *value, float *film_host);

/* The sample stream that stands in for Sampler::next1D (src/samplers/independent.cpp:95-103):
   n consecutive floats of the stream keyed by (seed, key), generated on the device */
int mtsamd_rng_stream(uint32_t seed, uint64_t key, uint32_t n, float *out);

/* ---- host-side helpers -------------------------------------------------------------------------- */

/* RoughTransmittance (src/bsdfs/rtrans.h:81-149 loader, :292-388 setEta/setAlpha): reduce the image of a
   data/microfacet/<distribution>.dat file to what RoughPlastic::configure keeps. out_T receives
   *theta_samples floats (capacity out_cap). */
int mtsamd_rtrans_reduce(const void *dat, uint64_t size, float eta, float alpha, float *out_T, uint32_t out_cap,
                         uint32_t *theta_samples, float *diffuse_ext, float *diffuse_int);
/* The same file reduced only to the material's relative index of refraction (setEta, rtrans.h:292-345), for a
   roughplastic whose roughness is a texture (RoughPlastic::configure with a non-constant alpha, roughplastic.cpp:281-299:
   the exterior copy stays two-dimensional, and both copies answer evalDiffuse(alpha) from a table over alpha). out
   receives *n floats (capacity out_cap; *n is set even when the capacity is too small):
   thetaSamples, alphaSamples, alphaMin, alphaMax, then T[alpha][theta] of the exterior copy, then the diffuse
   transmittance over alpha of the exterior and of the interior (1 / eta) copy. This block is what
   mtsamd_material.rtrans points to for such a material. */
int mtsamd_rtrans_reduce_eta(const void *dat, uint64_t size, float eta, float *out, uint32_t out_cap, uint32_t *n);

/* A measured spectrum (n wavelength [nm] / value pairs, wavelengths increasing) to linear Rec. 709 RGB, as the
   three-sample build of the reference converts the data/ior/<material>.{eta,k}.spd presets: product with the CIE 1931
   matching functions averaged over 360-830 nm by adaptive Gauss-Lobatto quadrature in 50 nm pieces, in single precision
   (Spectrum::fromContinuousSpectrum, src/libcore/spectrum.cpp:172-184; ContinuousSpectrum::average, :546-568;
   InterpolatedSpectrum::eval, :688-714; GaussLobattoIntegrator, src/libcore/quad.cpp:305-409). The file variant
   reads the text format of InterpolatedSpectrum(path) (spectrum.cpp:575-602). MTSAMD_EINVAL for an unreadable or
   empty file and for wavelengths that do not increase. */
int mtsamd_spd_to_rgb(const float *wavelengths, const float *values, uint32_t n, float rgb[3]);
int mtsamd_spd_file_to_rgb(const char *path, float rgb[3]);

/* The spectral values of a scene file, as SceneHandler converts them in the three-sample build
   (src/librender/scenehandler.cpp:534-620): <spectrum value="l1:v1, l2:v2, ..."> and <spectrum filename=...> are
   InterpolatedSpectrum::zeroExtend (src/libcore/spectrum.cpp:630-648; needs two entries), the conversion above, and a
   clamp of negative channels to zero; <blackbody temperature= scale=> integrates Planck's law
   (BlackBodySpectrum::eval, spectrum.cpp:483-495) the same way, clamps, then scales. MTSAMD_EINVAL as above, for
   fewer than two entries and for a temperature that is not positive. Need no device. */
int mtsamd_spectrum_value_to_rgb(const float *wavelengths, const float *values, uint32_t n, float rgb[3]);
int mtsamd_spectrum_file_to_rgb(const char *path, float rgb[3]);
int mtsamd_blackbody_to_rgb(float temperature, float scale, float rgb[3]);

/* Bitmap(EAuto, stream) followed by the conversion BitmapTexture's MIP map applies (src/libcore/bitmap.cpp:2534-2541
   and :284-286 for the gamma of a file, src/libcore/fmtconv.cpp:1093-1160 for undoing it; src/textures/bitmap.cpp:246-278):
   reads an image file into linear floats, 1 channel (luminance and luminance-alpha files) or 3 (RGB and RGBA files;
   alpha is dropped), first row = top. `gamma`: 0 = the file's own (sRGB for 8- and 16-bit files unless a PNG says
   otherwise, linear for floating-point ones), else the value of the texture's `gamma` property (-1 = sRGB). Formats: PNG
   (not interlaced), PFM, OpenEXR scan-line files with no, ZIPS, ZIP or PIZ compression (R, G, B or Y channels). The pixels
   belong to the library until mtsamd_image_free. Needs no device. */
typedef struct { int32_t width, height, channels; float *pixels; } mtsamd_image;
int mtsamd_image_load(const char *path, float gamma, mtsamd_image *out);
void mtsamd_image_free(mtsamd_image *image);

/* The host builder of the acceleration structure alone (binned-SAH sweep, collapse to 8-wide nodes, depth bounded to
   what the traversal stack holds): fills `info`, keeps nothing, needs no device. Counterpart of building a ShapeKDTree and
   reading its statistics (src/librender/skdtree.cpp:68-110; include/mitsuba/render/gkdtree.h:1213-1263). */
int mtsamd_bvh_host_info(const float *positions, uint32_t n_vertices, const uint32_t *indices, uint32_t n_triangles, mtsamd_bvh_info *info);
/* Deepest tree (levels of wide nodes, root = 1) the traversal kernels accept; the builders never return a deeper one */
int mtsamd_bvh_max_depth(void);

/* GaussianFilter / BoxFilter tables as ReconstructionFilter::configure builds them
   (src/rfilters/gaussian.cpp:32-57, src/libcore/rfilter.cpp:37-55) */
int mtsamd_filter_gaussian(float stddev, float *radius, float values[32]);
int mtsamd_filter_box(float radius_in, float *radius, float values[32]);
/* Any of the reference's reconstruction filters by kind: gaussian (p0 = stddev), box (p0 = radius), tent
   (src/rfilters/tent.cpp), mitchell (p0 = B, p1 = C; src/rfilters/mitchell.cpp), catmullrom (src/rfilters/catmullrom.cpp),
   lanczos (p0 = lobes; src/rfilters/lanczos.cpp) */
enum { MTSAMD_FILTER_GAUSSIAN = 0, MTSAMD_FILTER_BOX = 1, MTSAMD_FILTER_TENT = 2, MTSAMD_FILTER_MITCHELL = 3,
       MTSAMD_FILTER_CATMULLROM = 4, MTSAMD_FILTER_LANCZOS = 5 };
int mtsamd_filter_table(int kind, float p0, float p1, float *radius, float values[32]);

#ifdef __cplusplus
}
#endif
#endif /* MTSAMD_H */
