"""ctypes binding of libmtsamd.so (include/mtsamd.h) -- the only compute path of this package.

There is no CPU fallback: if the HIP library is missing or no device is present, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MTSAMD_LIB selects another build of the same library (kernel tuning experiments); the default is the in-tree build
_LIB_PATH = os.environ.get("MTSAMD_LIB") or os.path.join(_HERE, "libmtsamd.so")
_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


class Shape(C.Structure):
    _fields_ = [("first_triangle", C.c_uint32), ("n_triangles", C.c_uint32), ("material", C.c_int32),
                ("emitter", C.c_int32), ("flags", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("distribution", C.c_int32), ("sample_visible", C.c_int32), ("variant", C.c_int32),
                ("alpha_u", C.c_float), ("alpha_v", C.c_float), ("int_ior", C.c_float), ("ext_ior", C.c_float),
                ("diffuse_reflectance", C.c_float * 3), ("specular_reflectance", C.c_float * 3),
                ("specular_transmittance", C.c_float * 3), ("eta", C.c_float * 3), ("k", C.c_float * 3),
                ("ext_eta", C.c_float), ("ensure_energy_conservation", C.c_int32),
                ("rtrans", _f32p), ("rtrans_n", C.c_uint32), ("rtrans_diffuse_ext", C.c_float),
                ("rtrans_diffuse_int", C.c_float), ("two_sided", C.c_int32), ("back_material", C.c_int32),
                ("mix_count", C.c_int32), ("mix_material", C.c_int32 * 4), ("mix_weight", C.c_float * 4),
                ("diffuse_texture", C.c_int32), ("specular_texture", C.c_int32), ("transmittance_texture", C.c_int32),
                ("frame_kind", C.c_int32), ("frame_texture", C.c_int32),
                ("alpha_u_texture", C.c_int32), ("alpha_v_texture", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_int32), ("color0", C.c_float * 3), ("color1", C.c_float * 3), ("line_width", C.c_float),
                ("u_offset", C.c_float), ("v_offset", C.c_float), ("u_scale", C.c_float), ("v_scale", C.c_float),
                ("scale", C.c_float), ("pixels", _f32p), ("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32),
                ("filter_type", C.c_int32), ("wrap_u", C.c_int32), ("wrap_v", C.c_int32), ("max_anisotropy", C.c_float)]


class Emitter(C.Structure):
    _fields_ = [("shape", C.c_int32), ("radiance", C.c_float * 3), ("sampling_weight", C.c_float), ("type", C.c_int32),
                ("to_world", C.c_float * 16), ("cutoff_angle", C.c_float), ("beam_width", C.c_float),
                ("pixels", _f32p), ("width", C.c_int32), ("height", C.c_int32), ("scale", C.c_float)]


class LdrOpts(C.Structure):
    _fields_ = [("pixel_format", C.c_int32), ("tonemap", C.c_int32), ("gamma", C.c_float), ("exposure", C.c_float),
                ("key", C.c_float), ("burn", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("to_world", C.c_float * 16), ("x_fov", C.c_float), ("near_clip", C.c_float), ("far_clip", C.c_float)]


class Film(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("crop_x", C.c_int32), ("crop_y", C.c_int32),
                ("crop_w", C.c_int32), ("crop_h", C.c_int32), ("filter_radius", C.c_float),
                ("filter_values", C.c_float * 32)]


class Integrator(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("rr_depth", C.c_int32), ("strict_normals", C.c_int32),
                ("hide_emitters", C.c_int32), ("type", C.c_int32), ("emitter_samples", C.c_int32), ("bsdf_samples", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32), ("n_shapes", C.c_uint32),
                ("n_materials", C.c_uint32), ("n_emitters", C.c_uint32),
                ("positions", _f32p), ("normals", _f32p), ("texcoords", _f32p), ("indices", _u32p),
                ("shapes", C.POINTER(Shape)), ("materials", C.POINTER(Material)), ("emitters", C.POINTER(Emitter)),
                ("camera", Camera), ("film", Film), ("integrator", Integrator),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(Texture))]


class RenderOpts(C.Structure):
    _fields_ = [("spp", C.c_int32), ("spp_total", C.c_int32), ("sample_offset", C.c_int32), ("seed", C.c_uint32),
                ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32), ("spp_per_pass", C.c_int32),
                ("band_rows", C.c_int32), ("band_period", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("normal_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("paths", C.c_uint64),
                ("bounce_iterations", C.c_uint64), ("seconds", C.c_double),
                ("extend_ms", C.c_double), ("shade_ms", C.c_double), ("shadow_ms", C.c_double), ("film_ms", C.c_double),
                ("raygen_ms", C.c_double), ("extend_launches", C.c_uint64), ("shadow_launches", C.c_uint64),
                ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("shadow_node_visits", C.c_uint64), ("shadow_tri_tests", C.c_uint64),
                ("tail_ms", C.c_double), ("tail_normal_rays", C.c_uint64), ("tail_shadow_rays", C.c_uint64),
                ("gather_ms", C.c_double)]


class BvhInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_triangles", C.c_uint32), ("depth", C.c_uint32),
                ("node_bytes", C.c_uint64), ("triangle_bytes", C.c_uint64), ("build_seconds", C.c_double),
                ("sah_cost", C.c_double), ("binary_seconds", C.c_double), ("builder", C.c_int32), ("reserved", C.c_int32)]


HIT_DTYPE = np.dtype([("t", "f4"), ("u", "f4"), ("v", "f4"), ("shape", "u4"), ("prim", "u4")])
ISECT_DTYPE = np.dtype([("t", "f4"), ("u", "f4"), ("v", "f4"), ("shape", "u4"), ("prim", "u4"), ("p", "f4", 3),
                        ("geo_n", "f4", 3), ("sh_n", "f4", 3), ("sh_s", "f4", 3), ("sh_t", "f4", 3), ("wi", "f4", 3),
                        ("uv", "f4", 2)])

# every symbol include/mtsamd.h declares
EXPORTS = ["mtsamd_last_error", "mtsamd_init", "mtsamd_scene_create", "mtsamd_scene_destroy", "mtsamd_scene_bvh_info",
           "mtsamd_render", "mtsamd_render_device", "mtsamd_cancel", "mtsamd_scene_set_counting", "mtsamd_film_develop", "mtsamd_trace", "mtsamd_trace_device", "mtsamd_occluded_device",
           "mtsamd_occluded", "mtsamd_intersect", "mtsamd_bsdf_eval", "mtsamd_bsdf_pdf", "mtsamd_bsdf_sample",
           "mtsamd_sample_emitter_direct", "mtsamd_camera_rays", "mtsamd_li", "mtsamd_film_put", "mtsamd_film_develop_ldr", "mtsamd_abi_version", "mtsamd_sizeof", "mtsamd_filter_table", "mtsamd_rng_stream",
           "mtsamd_rtrans_reduce", "mtsamd_rtrans_reduce_eta", "mtsamd_filter_gaussian", "mtsamd_filter_box", "mtsamd_bvh_host_info", "mtsamd_bvh_max_depth",
           "mtsamd_trace_device_static", "mtsamd_trace_device_static_max", "mtsamd_render_multi", "mtsamd_spd_to_rgb",
           "mtsamd_spd_file_to_rgb", "mtsamd_bsdf_query_uv", "mtsamd_texture_eval", "mtsamd_image_load", "mtsamd_image_free", "mtsamd_env_eval",
           "mtsamd_spectrum_value_to_rgb", "mtsamd_spectrum_file_to_rgb", "mtsamd_blackbody_to_rgb"]

_lib = None
ABI_VERSION = 14    # MTSAMD_ABI_VERSION of include/mtsamd.h


class MtsamdError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise MtsamdError("%s is missing: build it with `make -C mitsuba_amd/csrc` (hipcc, gfx950); "
                              "there is no CPU fallback" % _LIB_PATH)
        # PyTorch-ROCm ships its own libamdhip64.so.7; importing it first lets this library bind to that same
        # runtime instance, so device pointers (e.g. a film tensor handed to render_device) are shared.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_LIB_PATH)
        L.mtsamd_last_error.restype = C.c_char_p
        # the records above are mirrored by hand: refuse a library built from another version of the header
        L.mtsamd_sizeof.restype = C.c_int64
        if L.mtsamd_abi_version() != ABI_VERSION:
            raise MtsamdError("%s implements ABI version %d, this binding %d: rebuild with `make -C mitsuba_amd/csrc`"
                              % (_LIB_PATH, L.mtsamd_abi_version(), ABI_VERSION))
        for record, cls in enumerate((SceneDesc, Material, Emitter, Shape, RenderOpts, Stats, LdrOpts, Texture)):
            if L.mtsamd_sizeof(record) != C.sizeof(cls):
                raise MtsamdError("record %s is %d bytes in %s and %d bytes in this binding"
                                  % (cls.__name__, L.mtsamd_sizeof(record), _LIB_PATH, C.sizeof(cls)))
        L.mtsamd_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        L.mtsamd_scene_destroy.argtypes = [C.c_void_p]
        L.mtsamd_cancel.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise MtsamdError("mtsamd error %d: %s" % (rc, lib().mtsamd_last_error().decode()))


def _fp(a):
    return a.ctypes.data_as(_f32p)


def init(device: int = 0):
    _check(lib().mtsamd_init(int(device)))


def fill_desc(sd):
    """SceneData -> (SceneDesc, keep-alive list)."""
    keep = []
    d = SceneDesc()
    pos = np.ascontiguousarray(sd.positions, np.float32)
    idx = np.ascontiguousarray(sd.indices, np.uint32)
    keep += [pos, idx]
    d.n_vertices, d.n_triangles = pos.shape[0], idx.shape[0]
    d.positions, d.indices = _fp(pos), idx.ctypes.data_as(_u32p)
    if sd.normals is not None:
        nrm = np.ascontiguousarray(sd.normals, np.float32)
        keep.append(nrm)
        d.normals = _fp(nrm)
    if sd.texcoords is not None:
        uv = np.ascontiguousarray(sd.texcoords, np.float32)
        keep.append(uv)
        d.texcoords = _fp(uv)
    shapes = (Shape * max(1, len(sd.shapes)))()
    for i, s in enumerate(sd.shapes):
        shapes[i].first_triangle, shapes[i].n_triangles = s["first_triangle"], s["n_triangles"]
        shapes[i].material, shapes[i].emitter, shapes[i].flags = s["material"], s["emitter"], s["flags"]
    mats = (Material * max(1, len(sd.materials)))()
    for i, m in enumerate(sd.materials):
        c = mats[i]
        for k in ("type", "distribution", "sample_visible", "variant", "alpha_u", "alpha_v", "int_ior", "ext_ior",
                  "ext_eta", "ensure_energy_conservation", "rtrans_diffuse_ext", "rtrans_diffuse_int", "two_sided",
                  "back_material"):
            setattr(c, k, m[k])
        for k in ("diffuse_reflectance", "specular_reflectance", "specular_transmittance", "eta", "k"):
            setattr(c, k, (C.c_float * 3)(*[float(x) for x in m[k]]))
        for k in ("diffuse_texture", "specular_texture", "transmittance_texture", "frame_kind", "frame_texture", "alpha_u_texture", "alpha_v_texture"):
            setattr(c, k, int(m.get(k, 0)))
        mix = m.get("mix", ())
        c.mix_count = len(mix)
        for j, (child, weight) in enumerate(mix[:4]):
            c.mix_material[j], c.mix_weight[j] = int(child), float(weight)
        if m.get("rtrans") is not None:
            t = np.ascontiguousarray(m["rtrans"], np.float32)
            keep.append(t)
            c.rtrans, c.rtrans_n = _fp(t), t.shape[0]
    ems = (Emitter * max(1, len(sd.emitters)))()
    for i, e in enumerate(sd.emitters):
        ems[i].shape = e["shape"]
        ems[i].radiance = (C.c_float * 3)(*e["radiance"])
        ems[i].sampling_weight = e["sampling_weight"]
        ems[i].type = e.get("type", 0)
        ems[i].to_world = (C.c_float * 16)(*np.asarray(e.get("to_world", np.eye(4)), np.float32).reshape(-1))
        ems[i].cutoff_angle, ems[i].beam_width = e.get("cutoff_angle", 20.0), e.get("beam_width", 15.0)
        if e.get("pixels") is not None:     # environment map: height x width x 3 linear floats
            px = np.ascontiguousarray(e["pixels"], np.float32)
            keep.append(px)
            ems[i].pixels, ems[i].height, ems[i].width, ems[i].scale = px.ctypes.data_as(_f32p), px.shape[0], px.shape[1], e.get("scale", 1.0)
    keep += [shapes, mats, ems]
    d.n_shapes, d.n_materials, d.n_emitters = len(sd.shapes), len(sd.materials), len(sd.emitters)
    d.shapes, d.materials, d.emitters = shapes, mats, ems
    cam = sd.camera
    d.camera.to_world = (C.c_float * 16)(*np.asarray(cam["to_world"], np.float32).reshape(-1))
    d.camera.x_fov, d.camera.near_clip, d.camera.far_clip = cam["x_fov"], cam["near_clip"], cam["far_clip"]
    f = sd.film
    for k in ("width", "height", "crop_x", "crop_y", "crop_w", "crop_h", "filter_radius"):
        setattr(d.film, k, f[k])
    d.film.filter_values = (C.c_float * 32)(*np.asarray(f["filter_values"], np.float32))
    for k in ("max_depth", "rr_depth", "strict_normals", "hide_emitters"):
        setattr(d.integrator, k, sd.integrator[k])
    tex_list = getattr(sd, "textures", None) or []
    if tex_list:
        texs = (Texture * len(tex_list))()
        for i, t in enumerate(tex_list):
            texs[i].type = t["type"]
            texs[i].color0, texs[i].color1 = (C.c_float * 3)(*t["color0"]), (C.c_float * 3)(*t["color1"])
            for k in ("line_width", "u_offset", "v_offset", "u_scale", "v_scale", "scale"):
                setattr(texs[i], k, t[k])
            if t.get("pixels") is not None:     # bitmap: height x width (x 3) linear floats, rows from the top
                px = np.ascontiguousarray(t["pixels"], np.float32)
                keep.append(px)
                texs[i].pixels = px.ctypes.data_as(_f32p)
                texs[i].height, texs[i].width = px.shape[0], px.shape[1]
                texs[i].channels = 1 if px.ndim == 2 else px.shape[2]
                for k in ("filter_type", "wrap_u", "wrap_v", "max_anisotropy"):
                    setattr(texs[i], k, t[k])
        keep.append(texs)
        d.n_textures, d.textures = len(tex_list), texs
    d.integrator.type = {"path": 0, "direct": 1, "volpath_simple": 2}[sd.integrator.get("type", "path")]
    d.integrator.emitter_samples = sd.integrator.get("emitter_samples", 1)
    d.integrator.bsdf_samples = sd.integrator.get("bsdf_samples", 1)
    return d, keep


def develop_ldr(film, pixel_format="rgb", tonemap="gamma", gamma=-1.0, exposure=0.0, key=0.18, burn=0.0):
    """LDRFilm::develop (src/films/ldrfilm.cpp:300-321): film (h, w, 5) float32 -> (h, w, channels) uint8. Host work."""
    film = np.ascontiguousarray(film, np.float32)
    h, w = film.shape[:2]
    fmt = ("luminance", "luminancealpha", "rgb", "rgba").index(pixel_format.lower())
    o = LdrOpts(fmt, ("gamma", "reinhard").index(tonemap.lower()), gamma, exposure, key, burn)
    out = np.zeros((h, w, fmt + 1), np.uint8)
    _check(lib().mtsamd_film_develop_ldr(_fp(film), C.c_int32(w), C.c_int32(h), C.byref(o), out.ctypes.data_as(C.c_void_p)))
    return out


class Scene:
    """An uploaded scene (mtsamd_scene handle)."""

    def __init__(self, sd):
        self.L = lib()
        self.sd = sd
        desc, self._keep = fill_desc(sd)
        h = C.c_void_p()
        _check(self.L.mtsamd_scene_create(C.byref(desc), C.byref(h)))
        self.h = h
        self.w, self.hgt = sd.film["crop_w"], sd.film["crop_h"]

    def close(self):
        if getattr(self, "h", None):
            self.L.mtsamd_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def bvh_info(self) -> BvhInfo:
        info = BvhInfo()
        _check(self.L.mtsamd_scene_bvh_info(self.h, C.byref(info)))
        return info

    def _opts(self, spp, spp_total=None, sample_offset=0, seed=1, rect=None, spp_per_pass=0, bands=None):
        """bands = (rows per band, period in rows): render only every (period / rows)-th full-width band of the
        rectangle, starting at its first row (one GPU's share of an interleaved tile partition)."""
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, self.w, self.hgt)
        band_rows, band_period = bands if bands is not None else (0, 0)
        return RenderOpts(spp, spp_total if spp_total is not None else sample_offset + spp, sample_offset, seed,
                          x0, y0, x1, y1, spp_per_pass, band_rows, band_period)

    def render(self, spp, film=None, **kw):
        o = self._opts(spp, **kw)
        if film is None:
            film = np.zeros((self.hgt, self.w, 5), np.float32)
        st = Stats()
        _check(self.L.mtsamd_render(self.h, C.byref(o), _fp(film), C.byref(st)))
        return film, st

    def render_device(self, spp, film_ptr: int, **kw):
        """Accumulate into a device film (crop_h*crop_w*5 floats) given by its raw pointer."""
        o = self._opts(spp, **kw)
        st = Stats()
        _check(self.L.mtsamd_render_device(self.h, C.byref(o), C.cast(C.c_void_p(film_ptr), _f32p), C.byref(st)))
        return st

    def li(self, spp, **kw):
        o = self._opts(spp, **kw)
        npx = (o.x1 - o.x0) * (o.y1 - o.y0)
        out = np.zeros((npx, spp, 4), np.float32)
        pos = np.zeros((npx, spp, 2), np.float32)
        st = Stats()
        _check(self.L.mtsamd_li(self.h, C.byref(o), _fp(out), _fp(pos), C.byref(st)))
        return out, pos, st

    def develop(self, film):
        film = np.ascontiguousarray(film, np.float32)
        out = np.zeros((self.hgt, self.w, 4), np.float32)
        _check(self.L.mtsamd_film_develop(self.h, _fp(film), _fp(out)))
        return out

    def develop_ldr(self, film, **kw):
        return develop_ldr(film, **kw)

    def cancel(self):
        self.L.mtsamd_cancel(self.h)

    def set_counting(self, enabled: bool):
        _check(self.L.mtsamd_scene_set_counting(self.h, int(enabled)))

    def trace(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros(rays.shape[0], HIT_DTYPE)
        _check(self.L.mtsamd_trace(self.h, C.c_uint64(rays.shape[0]), _fp(rays), hits.ctypes.data_as(C.c_void_p)))
        return hits

    def occluded(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], np.uint8)
        _check(self.L.mtsamd_occluded(self.h, C.c_uint64(rays.shape[0]), _fp(rays), out.ctypes.data_as(C.c_void_p)))
        return out

    def trace_device(self, rays_ptr: int, n: int, hits_ptr: int) -> float:
        """Closest-hit query on n rays in device memory (8 floats each) into n device hit records (20 bytes each);
        returns the traversal launch's duration in seconds."""
        sec = C.c_double(0)
        _check(self.L.mtsamd_trace_device(self.h, C.c_uint64(n), C.c_void_p(rays_ptr), C.c_void_p(hits_ptr), C.byref(sec)))
        return sec.value

    def trace_device_static(self, rays_ptr: int, n: int, hits_ptr: int):
        """mtsamd_trace_device_static: the same query through the library's static device scratch (no allocation)."""
        _check(self.L.mtsamd_trace_device_static(self.h, C.c_uint64(n), C.c_void_p(rays_ptr), C.c_void_p(hits_ptr)))

    def occluded_device(self, rays_ptr: int, n: int, out_ptr: int) -> float:
        sec = C.c_double(0)
        _check(self.L.mtsamd_occluded_device(self.h, C.c_uint64(n), C.c_void_p(rays_ptr), C.c_void_p(out_ptr), C.byref(sec)))
        return sec.value

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        its = np.zeros(rays.shape[0], ISECT_DTYPE)
        _check(self.L.mtsamd_intersect(self.h, C.c_uint64(rays.shape[0]), _fp(rays), its.ctypes.data_as(C.c_void_p)))
        return its

    def bsdf_query_uv(self, u, v):
        """Where on the surface bsdf_eval / bsdf_pdf / bsdf_sample evaluate a material's textures."""
        _check(self.L.mtsamd_bsdf_query_uv(self.h, C.c_float(u), C.c_float(v)))

    def env_eval(self, d, ref_n=None):
        """The environment emitter along directions d (n x 3): (radiance n x 3, emitter-sampling density n)."""
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        rn = np.zeros_like(d) if ref_n is None else np.ascontiguousarray(ref_n, np.float32).reshape(-1, 3)
        value, pdf = np.zeros_like(d), np.zeros(d.shape[0], np.float32)
        _check(self.L.mtsamd_env_eval(self.h, C.c_uint64(d.shape[0]), _fp(d), _fp(rn), _fp(value), _fp(pdf)))
        return value, pdf

    def texture_eval(self, texture, uv, duv=None):
        """Texture `texture` (1-based) of the scene at uv (n x 2); duv (n x 4: dudx, dvdx, dudy, dvdy) makes it a filtered lookup."""
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        d = None if duv is None else np.ascontiguousarray(duv, np.float32).reshape(-1, 4)
        out = np.zeros((uv.shape[0], 3), np.float32)
        _check(self.L.mtsamd_texture_eval(self.h, int(texture), C.c_uint64(uv.shape[0]), _fp(uv), None if d is None else _fp(d), _fp(out)))
        return out

    def bsdf_eval(self, mat, wi, wo):
        wi, wo = np.ascontiguousarray(wi, np.float32), np.ascontiguousarray(wo, np.float32)
        out = np.zeros_like(wi)
        _check(self.L.mtsamd_bsdf_eval(self.h, int(mat), C.c_uint64(wi.shape[0]), _fp(wi), _fp(wo), _fp(out)))
        return out

    def bsdf_pdf(self, mat, wi, wo):
        wi, wo = np.ascontiguousarray(wi, np.float32), np.ascontiguousarray(wo, np.float32)
        out = np.zeros(wi.shape[0], np.float32)
        _check(self.L.mtsamd_bsdf_pdf(self.h, int(mat), C.c_uint64(wi.shape[0]), _fp(wi), _fp(wo), _fp(out)))
        return out

    def bsdf_sample(self, mat, wi, sample, sampler_1d=None):
        wi, sample = np.ascontiguousarray(wi, np.float32), np.ascontiguousarray(sample, np.float32)
        n = wi.shape[0]
        wo, w = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        pdf, eta, typ = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint32)
        extra = None if sampler_1d is None else np.ascontiguousarray(sampler_1d, np.float32)
        _check(self.L.mtsamd_bsdf_sample(self.h, int(mat), C.c_uint64(n), _fp(wi), _fp(sample), _fp(wo), _fp(w), _fp(pdf),
                                         _fp(eta), typ.ctypes.data_as(_u32p), None if extra is None else _fp(extra)))
        return wo, w, pdf, eta, typ

    def sample_emitter_direct(self, ref, ref_n, sample):
        ref, ref_n, sample = (np.ascontiguousarray(a, np.float32) for a in (ref, ref_n, sample))
        n = ref.shape[0]
        d, value = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        dist, pdf = np.zeros(n, np.float32), np.zeros(n, np.float32)
        _check(self.L.mtsamd_sample_emitter_direct(self.h, C.c_uint64(n), _fp(ref), _fp(ref_n), _fp(sample), _fp(d),
                                                   _fp(dist), _fp(pdf), _fp(value)))
        return d, dist, pdf, value

    def camera_rays(self, pos):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        rays = np.zeros((pos.shape[0], 8), np.float32)
        _check(self.L.mtsamd_camera_rays(self.h, C.c_uint64(pos.shape[0]), _fp(pos), _fp(rays)))
        return rays

    def film_put(self, pos, value, film=None):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        value = np.ascontiguousarray(value, np.float32).reshape(-1, 4)
        if film is None:
            film = np.zeros((self.hgt, self.w, 5), np.float32)
        _check(self.L.mtsamd_film_put(self.h, C.c_uint64(pos.shape[0]), _fp(pos), _fp(value), _fp(film)))
        return film


def render_multi(scenes, spp, film=None, band_rows=0, **kw):
    """mtsamd_render_multi: one job over the devices the scenes were created on; returns (film, [Stats per device])."""
    first = scenes[0]
    o = first._opts(spp, **kw)
    if film is None:
        film = np.zeros((first.hgt, first.w, 5), np.float32)
    handles = (C.c_void_p * len(scenes))(*[s.h for s in scenes])
    stats = (Stats * len(scenes))()
    _check(lib().mtsamd_render_multi(handles, C.c_int32(len(scenes)), C.byref(o), C.c_int32(band_rows), _fp(film), stats))
    return film, list(stats)


def bvh_host_info(positions, indices) -> BvhInfo:
    """The host builder alone (no device): statistics of the wide BVH it makes of these triangles."""
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
    info = BvhInfo()
    _check(lib().mtsamd_bvh_host_info(_fp(pos), C.c_uint32(pos.shape[0]), idx.ctypes.data_as(_u32p), C.c_uint32(idx.shape[0]), C.byref(info)))
    return info


def trace_device_static_max() -> int:
    f = lib().mtsamd_trace_device_static_max
    f.restype = C.c_uint64
    return int(f())


def bvh_max_depth() -> int:
    return int(lib().mtsamd_bvh_max_depth())


def spd_to_rgb(wavelengths, values, scene_value=False):
    """Linear RGB of a measured spectrum (see mtsamd_spd_to_rgb); host code only, no device needed. `scene_value`:
    convert as a <spectrum> element of a scene file is converted (mtsamd_spectrum_value_to_rgb)."""
    w = np.ascontiguousarray(wavelengths, np.float32)
    v = np.ascontiguousarray(values, np.float32)
    if w.shape != v.shape or w.ndim != 1:
        raise ValueError("wavelengths and values must be 1-D arrays of one length")
    out = np.zeros(3, np.float32)
    fn = lib().mtsamd_spectrum_value_to_rgb if scene_value else lib().mtsamd_spd_to_rgb
    if fn(_fp(w), _fp(v), C.c_uint32(len(w)), _fp(out)) != 0:
        raise MtsamdError("spectrum: wavelengths must increase%s" % (", and at least two entries are needed" if scene_value else ""))
    return out


def spd_file_to_rgb(path, scene_value=False):
    out = np.zeros(3, np.float32)
    fn = lib().mtsamd_spectrum_file_to_rgb if scene_value else lib().mtsamd_spd_file_to_rgb
    if fn(os.fsencode(path), _fp(out)) != 0:
        raise MtsamdError('InterpolatedSpectrum: could not read "%s"' % path)
    return out


def blackbody_to_rgb(temperature, scale=1.0):
    """mtsamd_blackbody_to_rgb: a <blackbody> element as linear RGB."""
    out = np.zeros(3, np.float32)
    if lib().mtsamd_blackbody_to_rgb(C.c_float(temperature), C.c_float(scale), _fp(out)) != 0:
        raise MtsamdError("blackbody: the temperature must be positive")
    return out


class Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("pixels", _f32p)]


def image_load(path, gamma=0.0):
    """mtsamd_image_load: an image file as linear floats, height x width (luminance) or height x width x 3."""
    img = Image()
    L = lib()
    L.mtsamd_image_load.argtypes = [C.c_char_p, C.c_float, C.POINTER(Image)]
    L.mtsamd_image_free.argtypes = [C.POINTER(Image)]
    _check(L.mtsamd_image_load(os.fsencode(path), C.c_float(gamma), C.byref(img)))
    try:
        n = img.width * img.height * img.channels
        out = np.ctypeslib.as_array(img.pixels, shape=(n,)).copy()
    finally:
        L.mtsamd_image_free(C.byref(img))
    return out.reshape(img.height, img.width) if img.channels == 1 else out.reshape(img.height, img.width, img.channels)


def rng_stream(seed, key, n):
    out = np.zeros(n, np.float32)
    _check(lib().mtsamd_rng_stream(C.c_uint32(seed), C.c_uint64(key), C.c_uint32(n), _fp(out)))
    return out


def rtrans_reduce(dat: bytes, eta, alpha):
    """RoughPlastic::configure's table reduction (mtsamd_rtrans_reduce): -> (T[theta], diffuse_ext, diffuse_int)."""
    out = np.zeros(256, np.float32)
    n = C.c_uint32()
    de, di = C.c_float(), C.c_float()
    _check(lib().mtsamd_rtrans_reduce(dat, C.c_uint64(len(dat)), C.c_float(eta), C.c_float(alpha), _fp(out), 256,
                                      C.byref(n), C.byref(de), C.byref(di)))
    return out[:n.value].copy(), de.value, di.value


def rtrans_reduce_eta(dat: bytes, eta):
    """The table a roughplastic with a TEXTURED roughness needs (mtsamd_rtrans_reduce_eta): one float32 block
    [thetaSamples, alphaSamples, alphaMin, alphaMax, T[alpha][theta], diffuse exterior [alpha], diffuse interior [alpha]]."""
    n = C.c_uint32()
    out = np.zeros(1 << 16, np.float32)
    _check(lib().mtsamd_rtrans_reduce_eta(dat, C.c_uint64(len(dat)), C.c_float(eta), _fp(out), C.c_uint32(out.size), C.byref(n)))
    return out[:n.value].copy()


def filter_gaussian(stddev=0.5):
    r = C.c_float()
    v = (C.c_float * 32)()
    _check(lib().mtsamd_filter_gaussian(C.c_float(stddev), C.byref(r), v))
    return r.value, np.array(v, np.float32)


def filter_box(radius=0.5):
    r = C.c_float()
    v = (C.c_float * 32)()
    _check(lib().mtsamd_filter_box(C.c_float(radius), C.byref(r), v))
    return r.value, np.array(v, np.float32)


def filter_table(kind, p0=0.0, p1=0.0):
    """kind: 'gaussian' (p0 = stddev), 'box' (radius), 'tent', 'mitchell' (B, C), 'catmullrom', 'lanczos' (lobes)."""
    k = ("gaussian", "box", "tent", "mitchell", "catmullrom", "lanczos").index(kind)
    r = C.c_float()
    v = (C.c_float * 32)()
    _check(lib().mtsamd_filter_table(C.c_int(k), C.c_float(p0), C.c_float(p1), C.byref(r), v))
    return r.value, np.array(v, np.float32)
