"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the CPU oracle (oracle/liboracle*.so).

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py import this module.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
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
                ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32), ("n_threads", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("normal_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("paths", C.c_uint64),
                ("kd_traversal_steps", C.c_uint64), ("kd_prim_tests", C.c_uint64), ("seconds", C.c_double),
                ("threads", C.c_int32)]


HIT_DTYPE = np.dtype([("t", "f4"), ("u", "f4"), ("v", "f4"), ("shape", "u4"), ("prim", "u4"), ("p", "f4", 3),
                      ("geo_n", "f4", 3), ("sh_n", "f4", 3), ("sh_s", "f4", 3), ("sh_t", "f4", 3), ("wi", "f4", 3),
                      ("uv", "f4", 2)])


def _fp(a):
    return a.ctypes.data_as(_f32p)


def fill_desc(sd, desc_cls=SceneDesc, shape_cls=Shape, mat_cls=Material, em_cls=Emitter, tex_cls=Texture):
    """SceneData -> (ctypes desc, keep-alive list). Shared by the oracle and (via subclass layout
    equality) usable for any library that takes the same struct layout."""
    keep = []
    d = desc_cls()
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
    shapes = (shape_cls * max(1, len(sd.shapes)))()
    for i, s in enumerate(sd.shapes):
        shapes[i].first_triangle, shapes[i].n_triangles = s["first_triangle"], s["n_triangles"]
        shapes[i].material, shapes[i].emitter, shapes[i].flags = s["material"], s["emitter"], s["flags"]
    mats = (mat_cls * max(1, len(sd.materials)))()
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
    ems = (em_cls * max(1, len(sd.emitters)))()
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
        texs = (tex_cls * len(tex_list))()
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


def build():
    """Compile the oracle with its Makefile (g++ only)."""
    import subprocess
    subprocess.check_call(["make", "-C", _HERE, "-j4"], stdout=subprocess.DEVNULL)


_libs = {}


def lib(fast=False):
    name = "liboracle_fast.so" if fast else "liboracle.so"
    if name not in _libs:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.POINTER(SceneDesc)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        _libs[name] = L
    return _libs[name]


class OracleScene:
    def __init__(self, sd, fast=False):
        self.L = lib(fast)
        self.sd = sd
        desc, self._keep = fill_desc(sd)
        self.h = self.L.orc_scene_create(C.byref(desc))
        if not self.h:
            raise RuntimeError("oracle: " + self.L.orc_last_error().decode())
        self.w, self.hgt = sd.film["crop_w"], sd.film["crop_h"]

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_scene_destroy(C.c_void_p(self.h))
            self.h = None

    def _opts(self, spp, spp_total=None, sample_offset=0, seed=1, rect=None, n_threads=0):
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, self.w, self.hgt)
        return RenderOpts(spp, spp_total if spp_total is not None else spp, sample_offset, seed, x0, y0, x1, y1, n_threads)

    def trace(self, rays, brute=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros(rays.shape[0], HIT_DTYPE)
        self.L.orc_trace(C.c_void_p(self.h), C.c_uint64(rays.shape[0]), _fp(rays), hits.ctypes.data_as(C.c_void_p), int(brute))
        return hits

    def occluded(self, rays, brute=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], np.uint8)
        self.L.orc_occluded(C.c_void_p(self.h), C.c_uint64(rays.shape[0]), _fp(rays), out.ctypes.data_as(C.c_void_p), int(brute))
        return out

    def bsdf_query_uv(self, u, v):
        self.L.orc_bsdf_query_uv(C.c_float(u), C.c_float(v))

    def env_eval(self, d, ref_n=None):
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        rn = np.zeros_like(d) if ref_n is None else np.ascontiguousarray(ref_n, np.float32).reshape(-1, 3)
        value, pdf = np.zeros_like(d), np.zeros(d.shape[0], np.float32)
        if self.L.orc_env_eval(C.c_void_p(self.h), C.c_uint64(d.shape[0]), _fp(d), _fp(rn), _fp(value), _fp(pdf)) != 0:
            raise RuntimeError("oracle: the scene has no environment emitter")
        return value, pdf

    def texture_eval(self, texture, uv, duv=None, gradient=False):
        """Texture `texture` (1-based) at uv (n x 2); duv (n x 4: dudx, dvdx, dudy, dvdy) makes the lookup a filtered one;
        gradient=True returns n x 2 x 3 (d/du, d/dv)."""
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        d = None if duv is None else np.ascontiguousarray(duv, np.float32).reshape(-1, 4)
        out = np.zeros((uv.shape[0], 6 if gradient else 3), np.float32)
        rc = self.L.orc_texture_eval(C.c_void_p(self.h), int(texture), C.c_uint64(uv.shape[0]), _fp(uv), None if d is None else _fp(d),
                                     int(bool(gradient)), _fp(out))
        if rc != 0:
            raise RuntimeError("oracle: texture index out of range")
        return out.reshape(-1, 2, 3) if gradient else out

    def bsdf_eval(self, mat, wi, wo):
        wi, wo = np.ascontiguousarray(wi, np.float32), np.ascontiguousarray(wo, np.float32)
        out = np.zeros_like(wi)
        self.L.orc_bsdf_eval(C.c_void_p(self.h), mat, C.c_uint64(wi.shape[0]), _fp(wi), _fp(wo), _fp(out))
        return out

    def bsdf_pdf(self, mat, wi, wo):
        wi, wo = np.ascontiguousarray(wi, np.float32), np.ascontiguousarray(wo, np.float32)
        out = np.zeros(wi.shape[0], np.float32)
        self.L.orc_bsdf_pdf(C.c_void_p(self.h), mat, C.c_uint64(wi.shape[0]), _fp(wi), _fp(wo), _fp(out))
        return out

    def bsdf_sample(self, mat, wi, sample, sampler_1d=None):
        wi, sample = np.ascontiguousarray(wi, np.float32), np.ascontiguousarray(sample, np.float32)
        n = wi.shape[0]
        wo, w = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        pdf, eta, typ = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint32)
        extra = None if sampler_1d is None else np.ascontiguousarray(sampler_1d, np.float32)
        self.L.orc_bsdf_sample(C.c_void_p(self.h), mat, C.c_uint64(n), _fp(wi), _fp(sample), _fp(wo), _fp(w), _fp(pdf),
                               _fp(eta), typ.ctypes.data_as(_u32p), None if extra is None else _fp(extra))
        return wo, w, pdf, eta, typ

    def sample_emitter_direct(self, ref, ref_n, sample):
        ref, ref_n, sample = (np.ascontiguousarray(a, np.float32) for a in (ref, ref_n, sample))
        n = ref.shape[0]
        d, value = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        dist, pdf = np.zeros(n, np.float32), np.zeros(n, np.float32)
        self.L.orc_sample_emitter_direct(C.c_void_p(self.h), C.c_uint64(n), _fp(ref), _fp(ref_n), _fp(sample), _fp(d),
                                         _fp(dist), _fp(pdf), _fp(value))
        return d, dist, pdf, value

    def camera_rays(self, pos):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        rays = np.zeros((pos.shape[0], 8), np.float32)
        self.L.orc_camera_rays(C.c_void_p(self.h), C.c_uint64(pos.shape[0]), _fp(pos), _fp(rays))
        return rays

    def li(self, spp, **kw):
        o = self._opts(spp, **kw)
        npx = (o.x1 - o.x0) * (o.y1 - o.y0)
        out = np.zeros((npx, spp, 4), np.float32)
        pos = np.zeros((npx, spp, 2), np.float32)
        st = Stats()
        r = self.L.orc_li(C.c_void_p(self.h), C.byref(o), _fp(out), _fp(pos), C.byref(st))
        if r != 0:
            raise RuntimeError("oracle: " + self.L.orc_last_error().decode())
        return out, pos, st

    def render(self, spp, film=None, **kw):
        o = self._opts(spp, **kw)
        if film is None:
            film = np.zeros((self.hgt, self.w, 5), np.float32)
        st = Stats()
        r = self.L.orc_render(C.c_void_p(self.h), C.byref(o), _fp(film), C.byref(st))
        if r != 0:
            raise RuntimeError("oracle: " + self.L.orc_last_error().decode())
        return film, st

    def film_put(self, pos, value, film=None):
        pos, value = np.ascontiguousarray(pos, np.float32).reshape(-1, 2), np.ascontiguousarray(value, np.float32).reshape(-1, 4)
        if film is None:
            film = np.zeros((self.hgt, self.w, 5), np.float32)
        self.L.orc_film_put(C.c_void_p(self.h), C.c_uint64(pos.shape[0]), _fp(pos), _fp(value), _fp(film))
        return film

    def develop(self, film):
        film = np.ascontiguousarray(film, np.float32)
        out = np.zeros((self.hgt, self.w, 4), np.float32)
        self.L.orc_film_develop(C.c_void_p(self.h), _fp(film), _fp(out))
        return out


# ---- scene-free helpers -------------------------------------------------------------------------------
def develop_ldr(film, pixel_format="rgb", tonemap="gamma", gamma=-1.0, exposure=0.0, key=0.18, burn=0.0):
    """LDRFilm::develop (src/films/ldrfilm.cpp:300-321) restated in numpy, single precision: the weight division of
    Bitmap::convert (src/libcore/fmtconv.cpp:955-1005), tonemapReinhard (src/libcore/bitmap.cpp:1707-1854) or the
    exposure multiplier, applyGamma and the 8-bit rounding of convertScalar (fmtconv.cpp:1104-1160).
    film: (h, w, 5) float32 -> (h, w, channels) uint8."""
    f32 = np.float32
    film = np.asarray(film, f32)
    h, w = film.shape[:2]
    px = film.reshape(-1, 5)
    wgt = px[:, 4]
    with np.errstate(all="ignore"):
        inv = np.where(wgt != 0, f32(1) / wgt, wgt).astype(f32)
        colour = pixel_format.lower() in ("rgb", "rgba")
        alpha = pixel_format.lower() in ("luminancealpha", "rgba")
        lum_w = (f32(0.212671), f32(0.715160), f32(0.072169))

        def luminance(c):
            return (c[:, 0] * lum_w[0] + c[:, 1] * lum_w[1]) + c[:, 2] * lum_w[2]

        img = (px[:, :3] * inv[:, None]) if colour else (luminance(px) * inv)[:, None]
        img = img.astype(f32)
        multiplier = f32(1)
        if tonemap.lower() == "reinhard":
            lum = luminance(img) if colour else img[:, 0]
            # the running maximum restarts at a pixel whose luminance is exactly 1024 (bitmap.cpp:1745-1746)
            restart = np.nonzero(lum == 1024)[0]
            max_lum = f32(max(0.0, lum[restart[-1]:].max() if len(restart) else lum.max())) if lum.size else f32(0)
            if not max_lum >= 0:
                max_lum = f32(0)
            log_avg = np.exp(np.cumsum(np.log(f32(1e-3) + lum), dtype=f32)[-1] / f32(lum.size), dtype=f32)
            if max_lum != 0:
                b = f32(min(1.0, max(1e-8, 1 - burn)))
                scale = f32(key) / log_avg
                l_white = max_lum * scale
                inv_wp2 = f32(1) / (l_white * l_white * np.power(b, f32(4)))
                if colour:
                    X = (img[:, 0] * f32(0.412453) + img[:, 1] * f32(0.357580)) + img[:, 2] * f32(0.180423)
                    Y = (img[:, 0] * f32(0.212671) + img[:, 1] * f32(0.715160)) + img[:, 2] * f32(0.072169)
                    Z = (img[:, 0] * f32(0.019334) + img[:, 1] * f32(0.119193)) + img[:, 2] * f32(0.950227)
                    norm = f32(1) / (X + Y + Z)
                    x, y, Lp = X * norm, Y * norm, Y * scale
                    Y = Lp * (f32(1) + Lp * inv_wp2) / (f32(1) + Lp)
                    ratio = Y / y
                    X, Z = ratio * x, ratio * (f32(1) - x - y)
                    img = np.stack([(f32(3.240479) * X + f32(-1.537150) * Y) + f32(-0.498535) * Z,
                                    (f32(-0.969256) * X + f32(1.875991) * Y) + f32(0.041556) * Z,
                                    (f32(0.055648) * X + f32(-0.204043) * Y) + f32(1.057311) * Z], 1).astype(f32)
                else:
                    Lp = img[:, 0] * scale
                    img = (Lp * (f32(1) + Lp * inv_wp2) / (f32(1) + Lp))[:, None].astype(f32)
        else:
            multiplier = np.power(f32(2), f32(exposure))

        def quantise(v, mult, inv_gamma):
            v = (v * mult).astype(f32)
            if inv_gamma == -1:
                v = np.where(v <= f32(0.0031308), f32(12.92) * v, f32(1.055) * np.power(v, f32(1.0 / 2.4)) - f32(0.055)).astype(f32)
            elif inv_gamma != 1:
                v = np.power(v, f32(inv_gamma)).astype(f32)
            s = v * f32(255) + f32(0.5)
            s = np.where(s > 0, s, f32(0))          # std::max(0, NaN) is 0
            return np.minimum(f32(255), s).astype(np.uint8)

        out = [quantise(img, multiplier, f32(1) / f32(gamma))]
        if alpha:
            out.append(quantise((px[:, 3] * inv)[:, None], f32(1), f32(1)))
    return np.concatenate(out, 1).reshape(h, w, -1)


def clipped_aabb(v, box):
    v, box = np.ascontiguousarray(v, np.float32), np.ascontiguousarray(box, np.float32)
    out = np.zeros(6, np.float32)
    lib().orc_clipped_aabb(_fp(v), _fp(box), _fp(out))
    return out


def rtrans_reduce(dat: bytes, eta, alpha):
    out = np.zeros(256, np.float32)
    n = C.c_uint32()
    de, di = C.c_float(), C.c_float()
    r = lib().orc_rtrans_reduce(dat, C.c_uint64(len(dat)), C.c_float(eta), C.c_float(alpha), _fp(out), 256, C.byref(n),
                                C.byref(de), C.byref(di))
    if r != 0:
        raise RuntimeError("orc_rtrans_reduce failed: %d" % r)
    return out[:n.value].copy(), de.value, di.value


def rtrans_reduce_eta(dat: bytes, eta):
    """orc_rtrans_reduce_eta: the two-dimensional table of a roughplastic with a textured roughness, one float32 block."""
    n = C.c_uint32()
    out = np.zeros(1 << 16, np.float32)
    r = lib().orc_rtrans_reduce_eta(dat, C.c_uint64(len(dat)), C.c_float(eta), _fp(out), C.c_uint32(out.size), C.byref(n))
    if r != 0:
        raise RuntimeError("orc_rtrans_reduce_eta failed: %d" % r)
    return out[:n.value].copy()


def rtrans_eval3d(dat: bytes, cos_theta, alpha, eta):
    ct, al, et = (np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float32), np.shape(cos_theta)), np.float32)
                  for a in (cos_theta, alpha, eta))
    out = np.zeros(ct.shape, np.float32)
    r = lib().orc_rtrans_eval3d(dat, C.c_uint64(len(dat)), C.c_uint64(ct.size), _fp(ct), _fp(al), _fp(et), _fp(out))
    if r != 0:
        raise RuntimeError("orc_rtrans_eval3d failed")
    return out


def microfacet_eval(distr, alpha_u, alpha_v, sample_visible, kind, wi, m):
    wi, m = np.ascontiguousarray(wi, np.float32), np.ascontiguousarray(m, np.float32)
    out = np.zeros(wi.shape[0], np.float32)
    lib().orc_microfacet_eval(distr, C.c_float(alpha_u), C.c_float(alpha_v), int(sample_visible), kind,
                              C.c_uint64(wi.shape[0]), _fp(wi), _fp(m), _fp(out))
    return out


def microfacet_sample(distr, alpha_u, alpha_v, sample_visible, wi, sample):
    wi, sample = np.ascontiguousarray(wi, np.float32), np.ascontiguousarray(sample, np.float32)
    m, pdf = np.zeros_like(wi), np.zeros(wi.shape[0], np.float32)
    lib().orc_microfacet_sample(distr, C.c_float(alpha_u), C.c_float(alpha_v), int(sample_visible),
                                C.c_uint64(wi.shape[0]), _fp(wi), _fp(sample), _fp(m), _fp(pdf))
    return m, pdf


def fresnel_dielectric(cos_i, eta):
    cos_i = np.ascontiguousarray(cos_i, np.float32)
    F, ct = np.zeros_like(cos_i), np.zeros_like(cos_i)
    lib().orc_fresnel_dielectric(C.c_uint64(cos_i.size), _fp(cos_i), C.c_float(eta), _fp(F), _fp(ct))
    return F, ct


def fresnel_conductor(cos_i, eta3, k3):
    cos_i = np.ascontiguousarray(cos_i, np.float32)
    e, k = np.ascontiguousarray(eta3, np.float32), np.ascontiguousarray(k3, np.float32)
    out = np.zeros((cos_i.size, 3), np.float32)
    lib().orc_fresnel_conductor(C.c_uint64(cos_i.size), _fp(cos_i), _fp(e), _fp(k), _fp(out))
    return out


def rng_stream(seed, key, n):
    out = np.zeros(n, np.float32)
    lib().orc_rng_stream(C.c_uint32(seed), C.c_uint64(key), C.c_uint32(n), _fp(out))
    return out
