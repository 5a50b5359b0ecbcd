"""Flattened scene description handed to the HIP core through the C ABI (include/mtsamd.h).

The reference builds this information as an object graph (Scene/TriMesh/BSDF/Emitter/Sensor/Film,
src/librender/scene.cpp:477-556); the GPU core wants plain arrays. `SceneData` is the host-side
container: numpy arrays for geometry plus small dict records for shapes, materials, emitters,
camera, film and integrator parameters, named after the reference's XML properties.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

BSDF_DIFFUSE, BSDF_ROUGHCONDUCTOR, BSDF_ROUGHPLASTIC, BSDF_DIELECTRIC, BSDF_WARD, BSDF_CONDUCTOR, BSDF_PLASTIC, \
    BSDF_ROUGHDIELECTRIC, BSDF_MIXTURE = range(9)
DISTR_BECKMANN, DISTR_GGX = 0, 1
SHAPE_HAS_NORMALS, SHAPE_HAS_TEXCOORDS = 1, 2
EMITTER_SURFACE, EMITTER_POINT, EMITTER_SPOT, EMITTER_DIRECTIONAL, EMITTER_ENVMAP = range(5)

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# lookupIOR presets used by the configs (src/bsdfs/ior.h:39-111)
IOR = {"vacuum": 1.0, "air": 1.000277, "water": 1.3330, "bk7": 1.5046, "polypropylene": 1.49,
       "diamond": 2.419, "acrylic glass": 1.49, "pyrex": 1.470}


def look_at(origin, target, up) -> np.ndarray:
    """Camera-to-world matrix of Transform::lookAt (src/libcore/transform.cpp:191-214)."""
    o = np.asarray(origin, np.float32)
    t = np.asarray(target, np.float32)
    u = np.asarray(up, np.float32)
    d = t - o
    d = d / np.float32(np.sqrt(np.float32(np.dot(d, d))))
    left = np.cross(u, d).astype(np.float32)
    left = left / np.float32(np.sqrt(np.float32(np.dot(left, left))))
    new_up = np.cross(d, left).astype(np.float32)
    m = np.eye(4, dtype=np.float32)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, o
    return m


def coordinate_system(a):
    """(b, c) of coordinateSystem(a, b, c) (src/libcore/util.cpp:592-601)."""
    a = np.asarray(a, np.float32)
    if abs(a[0]) > abs(a[1]):
        inv = np.float32(1.0) / np.float32(np.sqrt(a[0] * a[0] + a[2] * a[2]))
        c = np.array([a[2] * inv, 0.0, -a[0] * inv], np.float32)
    else:
        inv = np.float32(1.0) / np.float32(np.sqrt(a[1] * a[1] + a[2] * a[2]))
        c = np.array([0.0, a[2] * inv, -a[1] * inv], np.float32)
    return np.cross(c, a).astype(np.float32), c


def gaussian_filter(stddev: float = 0.5):
    """(radius, 32 table values) of GaussianFilter + ReconstructionFilter::configure
    (src/rfilters/gaussian.cpp:32-57, src/libcore/rfilter.cpp:37-55), in single precision."""
    f32 = np.float32
    stddev = f32(stddev)
    radius = f32(4) * stddev
    alpha = f32(-1.0) / (f32(2.0) * stddev * stddev)

    def ev(x):
        a = f32(np.exp(np.float64(alpha * x * x)))
        b = f32(np.exp(np.float64(alpha * radius * radius)))
        return max(f32(0.0), f32(a - b))

    vals = np.zeros(32, np.float32)
    s = f32(0)
    for i in range(31):
        v = ev(f32(radius * f32(i)) / f32(31))
        vals[i] = v
        s = f32(s + v)
    s = f32(s * (f32(2) * radius / f32(31)))
    norm = f32(1.0) / s
    for i in range(31):
        vals[i] = f32(vals[i] * norm)
    return float(radius), vals


def box_filter(radius: float = 0.5):
    """BoxFilter: eval = 1 inside the radius, which is the given one plus 1e-5 (src/rfilters/box.cpp:38-48), same normalisation."""
    f32 = np.float32
    radius = f32(f32(radius) + f32(1e-5))
    vals = np.zeros(32, np.float32)
    s = f32(31.0)
    s = f32(s * (f32(2) * radius / f32(31)))
    vals[:31] = f32(1.0) / s
    return float(radius), vals


def _tabulate(radius, ev):
    """ReconstructionFilter::configure (src/libcore/rfilter.cpp:37-55) in single precision."""
    f32 = np.float32
    radius = f32(radius)
    vals = np.zeros(32, np.float32)
    s = f32(0)
    for i in range(31):
        vals[i] = f32(ev(f32(f32(radius * f32(i)) / f32(31))))
        s = f32(s + vals[i])
    s = f32(s * (f32(2) * radius / f32(31)))
    vals[:31] = (vals[:31] * (f32(1.0) / s)).astype(np.float32)
    return float(radius), vals


def tent_filter():
    """src/rfilters/tent.cpp:34-44"""
    return _tabulate(1.0, lambda x: max(np.float32(0), np.float32(1) - abs(x / np.float32(1.0))))


def mitchell_filter(B=1.0 / 3.0, C=1.0 / 3.0):
    """src/rfilters/mitchell.cpp:35-68 (B = 0, C = 1/2 is catmullrom.cpp:32-55)"""
    f32 = np.float32
    B, C = f32(B), f32(C)

    def ev(x):
        x = abs(x)
        x2 = f32(x * x)
        x3 = f32(x2 * x)
        if x < 1:
            return f32(1.0 / 6.0) * ((12 - 9 * B - 6 * C) * x3 + (-18 + 12 * B + 6 * C) * x2 + (6 - 2 * B))
        if x < 2:
            return f32(1.0 / 6.0) * ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * x + (8 * B + 24 * C))
        return f32(0)
    return _tabulate(2.0, ev)


def catmullrom_filter():
    return mitchell_filter(0.0, 0.5)


def lanczos_filter(lobes=3):
    """src/rfilters/lanczos.cpp:35-55"""
    f32 = np.float32
    r = f32(int(lobes))

    def ev(x):
        x = abs(x)
        if x < 1e-4:
            return f32(1)
        if x > r:
            return f32(0)
        x1 = f32(f32(np.pi) * x)
        x2 = f32(x1 / r)
        return f32(np.sin(x1, dtype=f32) * np.sin(x2, dtype=f32)) / f32(x1 * x2)
    return _tabulate(r, ev)


@dataclass
class SceneData:
    positions: np.ndarray
    indices: np.ndarray
    normals: Optional[np.ndarray] = None
    texcoords: Optional[np.ndarray] = None
    shapes: List[Dict] = field(default_factory=list)
    materials: List[Dict] = field(default_factory=list)
    emitters: List[Dict] = field(default_factory=list)
    camera: Dict = field(default_factory=dict)
    film: Dict = field(default_factory=dict)
    integrator: Dict = field(default_factory=lambda: dict(max_depth=-1, rr_depth=5, strict_normals=0, hide_emitters=0))
    textures: List[Dict] = field(default_factory=list)      # what the materials' *_texture fields index, 1-based

    @property
    def n_triangles(self) -> int:
        return int(self.indices.shape[0])


# ------------------------------------------------------------------------------------------------
# Material records (defaults follow the reference constructors)
# ------------------------------------------------------------------------------------------------
def _mat(**kw) -> Dict:
    m = dict(type=BSDF_DIFFUSE, distribution=DISTR_BECKMANN, sample_visible=1, variant=0,
             alpha_u=0.1, alpha_v=0.1, int_ior=1.5046, ext_ior=1.000277,
             diffuse_reflectance=(0.5, 0.5, 0.5), specular_reflectance=(1.0, 1.0, 1.0),
             specular_transmittance=(1.0, 1.0, 1.0), eta=(0.0, 0.0, 0.0), k=(1.0, 1.0, 1.0),
             ext_eta=1.000277, ensure_energy_conservation=1,
             rtrans=None, rtrans_diffuse_ext=0.0, rtrans_diffuse_int=0.0, two_sided=0, back_material=-1,
             diffuse_texture=0, specular_texture=0, transmittance_texture=0, frame_kind=0, frame_texture=0,
             alpha_u_texture=0, alpha_v_texture=0)
    m.update(kw)
    return m


FRAME_BUMPMAP, FRAME_NORMALMAP = 1, 2


def bumpmap(nested: Dict, texture: int) -> Dict:
    """src/bsdfs/bumpmap.cpp: `nested` evaluated in a shading frame displaced by texture `texture` (1-based)."""
    return dict(nested, frame_kind=FRAME_BUMPMAP, frame_texture=int(texture))


def normalmap(nested: Dict, texture: int) -> Dict:
    """src/bsdfs/normalmap.cpp: `nested` evaluated in the frame whose normal texture `texture` stores as (n + 1) / 2."""
    return dict(nested, frame_kind=FRAME_NORMALMAP, frame_texture=int(texture))


TEX_CHECKERBOARD, TEX_GRID, TEX_BITMAP = 1, 2, 3
# EMIPFilterType (include/mitsuba/render/mipmap.h:52-61), ReconstructionFilter::EBoundaryCondition (rfilter.h:53-64)
MIP_FILTERS = dict(nearest=0, bilinear=1, trilinear=2, ewa=3)
WRAP_MODES = dict(clamp=0, repeat=1, mirror=2, zero=3, black=3, one=4, white=4)


def _tex(kind, color0, color1, line_width, uoffset, voffset, uscale, vscale, scale) -> Dict:
    return dict(type=kind, color0=tuple(float(x) for x in color0), color1=tuple(float(x) for x in color1), line_width=float(line_width),
                u_offset=float(uoffset), v_offset=float(voffset), u_scale=float(uscale), v_scale=float(vscale), scale=float(scale))


def checkerboard(color0=(0.4, 0.4, 0.4), color1=(0.2, 0.2, 0.2), uoffset=0.0, voffset=0.0, uscale=1.0, vscale=1.0, scale=1.0) -> Dict:
    """src/textures/checkerboard.cpp:48-51 (defaults) on top of Texture2D (src/librender/texture.cpp:81-95); `scale` is the
    factor of a `scale` texture around it (src/textures/scale.cpp)."""
    return _tex(TEX_CHECKERBOARD, color0, color1, 0.0, uoffset, voffset, uscale, vscale, scale)


def gridtexture(color0=(0.2, 0.2, 0.2), color1=(0.4, 0.4, 0.4), line_width=0.01, uoffset=0.0, voffset=0.0, uscale=1.0, vscale=1.0,
                scale=1.0) -> Dict:
    """src/textures/gridtexture.cpp:55-59"""
    return _tex(TEX_GRID, color0, color1, line_width, uoffset, voffset, uscale, vscale, scale)


def bitmap(pixels, filter_type="ewa", wrap_mode="repeat", wrap_mode_u=None, wrap_mode_v=None, max_anisotropy=20.0,
           uoffset=0.0, voffset=0.0, uscale=1.0, vscale=1.0, scale=1.0) -> Dict:
    """src/textures/bitmap.cpp:179-235: `pixels` is the decoded image, height x width (luminance) or height x width x 3
    (RGB), linear floats, first row = top of the image."""
    t = _tex(TEX_BITMAP, (0, 0, 0), (0, 0, 0), 0.0, uoffset, voffset, uscale, vscale, scale)
    px = np.ascontiguousarray(pixels, np.float32)
    if px.ndim not in (2, 3) or (px.ndim == 3 and px.shape[2] not in (1, 3)):
        raise ValueError("bitmap texture: pixels must be height x width or height x width x 3")
    if px.ndim == 3 and px.shape[2] == 1:
        px = px[:, :, 0]
    t.update(pixels=px, filter_type=MIP_FILTERS[filter_type.lower()], wrap_u=WRAP_MODES[(wrap_mode_u or wrap_mode)],
             wrap_v=WRAP_MODES[(wrap_mode_v or wrap_mode)], max_anisotropy=float(max_anisotropy))
    return t


def diffuse(reflectance=(0.5, 0.5, 0.5)) -> Dict:
    """src/bsdfs/diffuse.cpp:77-84"""
    return _mat(type=BSDF_DIFFUSE, diffuse_reflectance=tuple(reflectance))


def roughconductor(eta, k, alpha=0.1, distribution="beckmann", specular_reflectance=(1, 1, 1),
                   ext_eta=IOR["air"], sample_visible=True, alpha_v=None) -> Dict:
    """src/bsdfs/roughconductor.cpp:168-201 with explicit RGB eta/k (material presets are spectral data
    the host adapter resolves before filling this record)."""
    return _mat(type=BSDF_ROUGHCONDUCTOR, distribution=_distr(distribution), alpha_u=alpha,
                alpha_v=alpha if alpha_v is None else alpha_v, eta=tuple(eta), k=tuple(k), ext_eta=ext_eta,
                specular_reflectance=tuple(specular_reflectance), sample_visible=int(sample_visible))


def dielectric(int_ior=IOR["bk7"], ext_ior=IOR["air"], specular_reflectance=(1, 1, 1),
               specular_transmittance=(1, 1, 1)) -> Dict:
    """src/bsdfs/dielectric.cpp:147-165"""
    return _mat(type=BSDF_DIELECTRIC, int_ior=int_ior, ext_ior=ext_ior,
                specular_reflectance=tuple(specular_reflectance),
                specular_transmittance=tuple(specular_transmittance))


def ward(alpha=0.1, diffuse_reflectance=(0.5, 0.5, 0.5), specular_reflectance=(0.2, 0.2, 0.2),
         variant="balanced", alpha_v=None) -> Dict:
    """src/bsdfs/ward.cpp:97-126"""
    v = {"ward": 0, "ward-duer": 1, "balanced": 2}[variant]
    return _mat(type=BSDF_WARD, alpha_u=alpha, alpha_v=alpha if alpha_v is None else alpha_v, variant=v,
                diffuse_reflectance=tuple(diffuse_reflectance), specular_reflectance=tuple(specular_reflectance))


def roughplastic(alpha=0.1, distribution="beckmann", int_ior=IOR["polypropylene"], ext_ior=IOR["air"],
                 diffuse_reflectance=(0.5, 0.5, 0.5), specular_reflectance=(1, 1, 1), nonlinear=False,
                 sample_visible=True, rtrans=None, alpha_texture=0) -> Dict:
    """src/bsdfs/roughplastic.cpp:197-228. The reduced rough-transmittance table (what
    RoughPlastic::configure derives from data/microfacet/<distr>.dat, roughplastic.cpp:281-299) is
    looked up in the packaged slices unless given. With `alpha_texture` (what SceneBuilder.add_texture returned) the
    roughness is that texture's average at every intersection and the table is the two-dimensional block of
    capi.rtrans_reduce_eta (given as `rtrans`, or looked up in the packaged ones)."""
    common = dict(type=BSDF_ROUGHPLASTIC, distribution=_distr(distribution), alpha_u=alpha, alpha_v=alpha,
                  int_ior=int_ior, ext_ior=ext_ior, diffuse_reflectance=tuple(diffuse_reflectance),
                  specular_reflectance=tuple(specular_reflectance), variant=int(nonlinear), sample_visible=int(sample_visible))
    if alpha_texture:
        block = load_rtrans_eta_block(distribution, int_ior / ext_ior) if rtrans is None else rtrans
        return _mat(rtrans=np.asarray(block, np.float32), alpha_u_texture=int(alpha_texture), alpha_v_texture=int(alpha_texture), **common)
    if rtrans is None:
        rtrans = load_rtrans_slice(distribution, int_ior / ext_ior, alpha)
    T, d_ext, d_int = rtrans
    return _mat(rtrans=np.asarray(T, np.float32), rtrans_diffuse_ext=float(d_ext), rtrans_diffuse_int=float(d_int), **common)


def conductor(eta, k, specular_reflectance=(1, 1, 1), ext_eta=IOR["air"]) -> Dict:
    """src/bsdfs/conductor.cpp:153-176 with explicit RGB eta/k."""
    return _mat(type=BSDF_CONDUCTOR, eta=tuple(eta), k=tuple(k), ext_eta=ext_eta,
                specular_reflectance=tuple(specular_reflectance))


def plastic(int_ior=IOR["polypropylene"], ext_ior=IOR["air"], diffuse_reflectance=(0.5, 0.5, 0.5),
            specular_reflectance=(1, 1, 1), nonlinear=False) -> Dict:
    """src/bsdfs/plastic.cpp:146-166"""
    return _mat(type=BSDF_PLASTIC, int_ior=int_ior, ext_ior=ext_ior, diffuse_reflectance=tuple(diffuse_reflectance),
                specular_reflectance=tuple(specular_reflectance), variant=int(nonlinear))


def roughdielectric(alpha=0.1, distribution="beckmann", int_ior=IOR["bk7"], ext_ior=IOR["air"], specular_reflectance=(1, 1, 1),
                    specular_transmittance=(1, 1, 1), sample_visible=True, alpha_v=None) -> Dict:
    """src/bsdfs/roughdielectric.cpp:183-211"""
    return _mat(type=BSDF_ROUGHDIELECTRIC, distribution=_distr(distribution), alpha_u=alpha,
                alpha_v=alpha if alpha_v is None else alpha_v, int_ior=int_ior, ext_ior=ext_ior,
                specular_reflectance=tuple(specular_reflectance), specular_transmittance=tuple(specular_transmittance),
                sample_visible=int(sample_visible))


def twosided(front: Dict, back_material: int = -1) -> Dict:
    """src/bsdfs/twosided.cpp:84-106: the nested BRDF's record marked two-sided; back_material is the index of
    the record of a second nested BRDF (the side seen from behind), -1 = the same BRDF on both sides."""
    m = dict(front)
    m["two_sided"], m["back_material"] = 1, int(back_material)
    return m


def mixture(children, weights, ensure_energy_conservation=True) -> Dict:
    """src/bsdfs/mixturebsdf.cpp:66-85,115-172: `children` are the indices of the nested BSDFs' records (at most four),
    `weights` their linear weights."""
    if len(children) != len(weights):
        raise ValueError("BSDF count mismatch: %d bsdfs, but specified %d weights" % (len(children), len(weights)))
    if not 1 <= len(children) <= 4:
        raise ValueError("mixturebsdf: between one and four nested BSDFs are supported")
    return _mat(type=BSDF_MIXTURE, mix=[(int(c), float(w)) for c, w in zip(children, weights)],
                ensure_energy_conservation=int(ensure_energy_conservation))


def _distr(name) -> int:
    if isinstance(name, int):
        return name
    return {"beckmann": DISTR_BECKMANN, "ggx": DISTR_GGX}[name.lower()]


_slices = None


def rtrans_key(distribution, eta, alpha) -> str:
    name = distribution if isinstance(distribution, str) else ("beckmann", "ggx")[distribution]
    return "%s:%.6f:%.6f" % (name.lower(), eta, alpha)


def load_rtrans_slice(distribution, eta, alpha):
    """Reduced tables for the (distribution, eta, alpha) combinations the bundled scenes use, produced by
    tools/make_rtrans_slices.py from the reference's data/microfacet/*.dat with mtsamd_rtrans_reduce."""
    global _slices
    if _slices is None:
        with open(os.path.join(_DATA_DIR, "rtrans_slices.json")) as f:
            _slices = json.load(f)
    key = rtrans_key(distribution, eta, alpha)
    if key not in _slices:
        raise KeyError("no packaged rough-transmittance slice for %s; reduce it from the Mitsuba data file "
                       "with mitsuba_amd.capi.rtrans_reduce()" % key)
    e = _slices[key]
    return np.asarray(e["T"], np.float32), e["diffuse_ext"], e["diffuse_int"]


_eta_blocks = None


def load_rtrans_eta_block(distribution, eta):
    """Rough-transmittance tables reduced to an index of refraction only (capi.rtrans_reduce_eta), for the
    (distribution, eta) combinations the bundled scenes use with a textured roughness (tools/make_golden.py)."""
    global _eta_blocks
    if _eta_blocks is None:
        _eta_blocks = np.load(os.path.join(_DATA_DIR, "rtrans_eta_blocks.npz"))
    key = "%s_%.6f" % (distribution, eta)
    if key not in _eta_blocks.files:
        raise KeyError("no packaged two-dimensional rough-transmittance table for %s; reduce it from the Mitsuba data file "
                       "with mitsuba_amd.capi.rtrans_reduce_eta()" % key)
    return _eta_blocks[key]


# ------------------------------------------------------------------------------------------------
# Mesh assembly
# ------------------------------------------------------------------------------------------------
class SceneBuilder:
    def __init__(self):
        self.P, self.N, self.UV, self.I = [], [], [], []
        self.shapes, self.materials, self.emitters, self.textures = [], [], [], []
        self.nv = 0
        self.nt = 0
        self.any_normals = False
        self.any_uvs = False

    def add_material(self, m: Dict) -> int:
        self.materials.append(m)
        return len(self.materials) - 1

    def add_texture(self, t: Dict) -> int:
        """Returns what goes into a material's diffuse_texture / specular_texture / transmittance_texture (1-based)."""
        self.textures.append(t)
        return len(self.textures)

    def add_constant_emitter(self, radiance=(1.0, 1.0, 1.0), sampling_weight=1.0) -> int:
        """Constant environment emitter (src/emitters/constant.cpp): an emitter record without a shape."""
        self.emitters.append(dict(shape=-1, radiance=tuple(float(x) for x in radiance), sampling_weight=float(sampling_weight)))
        return len(self.emitters) - 1

    def add_envmap_emitter(self, pixels, to_world=None, scale=1.0, sampling_weight=1.0) -> int:
        """Environment map (src/emitters/envmap.cpp): a latitude-longitude image, height x width x 3 linear floats (a
        luminance image is replicated), first row = +y of the emitter's frame."""
        px = np.ascontiguousarray(pixels, np.float32)
        if px.ndim == 2:
            px = np.repeat(px[:, :, None], 3, 2)
        if px.ndim != 3 or px.shape[2] != 3:
            raise ValueError("envmap: pixels must be height x width x 3")
        self.emitters.append(dict(shape=-1, type=EMITTER_ENVMAP, radiance=(0.0, 0.0, 0.0), sampling_weight=float(sampling_weight), pixels=px,
                                  scale=float(scale), to_world=np.asarray(np.eye(4) if to_world is None else to_world, np.float32).reshape(4, 4)))
        return len(self.emitters) - 1

    def _add_delta_emitter(self, kind, value, to_world, sampling_weight, **extra) -> int:
        self.emitters.append(dict(shape=-1, type=kind, radiance=tuple(float(x) for x in value),
                                  to_world=np.asarray(to_world, np.float32).reshape(4, 4), sampling_weight=float(sampling_weight), **extra))
        return len(self.emitters) - 1

    def add_point_emitter(self, position, intensity=(1.0, 1.0, 1.0), sampling_weight=1.0) -> int:
        """Point light (src/emitters/point.cpp)."""
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = position
        return self._add_delta_emitter(EMITTER_POINT, intensity, m, sampling_weight)

    def add_spot_emitter(self, to_world, intensity=(1.0, 1.0, 1.0), cutoff_angle=20.0, beam_width=None, sampling_weight=1.0) -> int:
        """Spot light looking along +z of `to_world` (src/emitters/spot.cpp); beam width defaults to 3/4 of the cutoff angle."""
        cutoff_angle = np.float32(cutoff_angle)
        if beam_width is None:
            beam_width = cutoff_angle * np.float32(3.0) / np.float32(4.0)
        if not cutoff_angle >= beam_width:      # spot.cpp:74
            raise ValueError("spot emitter: the cutoff angle must not be smaller than the beam width")
        return self._add_delta_emitter(EMITTER_SPOT, intensity, to_world, sampling_weight,
                                       cutoff_angle=float(cutoff_angle), beam_width=float(np.float32(beam_width)))

    def add_directional_emitter(self, direction=None, irradiance=(1.0, 1.0, 1.0), to_world=None, sampling_weight=1.0) -> int:
        """Directional light (src/emitters/directional.cpp): either a direction (turned into a look-at transform as
        directional.cpp:64-67 does) or an emitter-to-world rotation whose +z is the direction."""
        if to_world is None:
            d = np.asarray(direction, np.float32)
            d = d / np.float32(np.linalg.norm(d))
            to_world = look_at((0.0, 0.0, 0.0), d, coordinate_system(d)[0])
        return self._add_delta_emitter(EMITTER_DIRECTIONAL, irradiance, to_world, sampling_weight)

    def add_mesh(self, positions, indices, material: int, normals=None, texcoords=None, radiance=None,
                 sampling_weight=1.0) -> int:
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        indices = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
        flags = 0
        n = positions.shape[0]
        if normals is not None:
            flags |= SHAPE_HAS_NORMALS
            self.any_normals = True
            self.N.append(np.ascontiguousarray(normals, np.float32).reshape(-1, 3))
        else:
            self.N.append(np.zeros((n, 3), np.float32))
        if texcoords is not None:
            flags |= SHAPE_HAS_TEXCOORDS
            self.any_uvs = True
            self.UV.append(np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2))
        else:
            self.UV.append(np.zeros((n, 2), np.float32))
        self.P.append(positions)
        self.I.append(indices + np.uint32(self.nv))
        sid = len(self.shapes)
        em = -1
        if radiance is not None:
            em = len(self.emitters)
            self.emitters.append(dict(shape=sid, radiance=tuple(float(x) for x in radiance),
                                      sampling_weight=float(sampling_weight)))
        self.shapes.append(dict(first_triangle=self.nt, n_triangles=int(indices.shape[0]), material=material,
                                emitter=em, flags=flags))
        self.nv += n
        self.nt += int(indices.shape[0])
        return sid

    def add_quad(self, p0, p1, p2, p3, material, **kw) -> int:
        return self.add_mesh(np.array([p0, p1, p2, p3], np.float32), np.array([[0, 1, 2], [0, 2, 3]]), material, **kw)

    def add_box(self, lo, hi, material, to_world=None, **kw) -> int:
        lo = np.asarray(lo, np.float32)
        hi = np.asarray(hi, np.float32)
        c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                      [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]], np.float32)
        if to_world is not None:
            m = np.asarray(to_world, np.float32)
            c = (c @ m[:3, :3].T + m[:3, 3]).astype(np.float32)
        # outward-facing faces
        quads = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 4, 7, 3)]
        idx = []
        for a, b, cc, d in quads:
            idx += [[a, b, cc], [a, cc, d]]
        return self.add_mesh(c, np.array(idx), material, **kw)

    def finish(self, camera: Dict, film: Dict, integrator: Optional[Dict] = None) -> SceneData:
        sd = SceneData(
            positions=np.ascontiguousarray(np.concatenate(self.P), np.float32),
            indices=np.ascontiguousarray(np.concatenate(self.I), np.uint32),
            normals=np.ascontiguousarray(np.concatenate(self.N), np.float32) if self.any_normals else None,
            texcoords=np.ascontiguousarray(np.concatenate(self.UV), np.float32) if self.any_uvs else None,
            shapes=self.shapes, materials=self.materials, emitters=self.emitters, camera=camera, film=film, textures=self.textures)
        if integrator:
            sd.integrator.update(integrator)
        return sd


def make_film(width, height, rfilter=None, crop=None) -> Dict:
    radius, vals = rfilter if rfilter is not None else gaussian_filter()
    cx, cy, cw, ch = crop if crop is not None else (0, 0, width, height)
    return dict(width=width, height=height, crop_x=cx, crop_y=cy, crop_w=cw, crop_h=ch,
                filter_radius=radius, filter_values=vals)


def make_camera(origin, target, up, x_fov, near_clip=1e-2, far_clip=1e4) -> Dict:
    return dict(to_world=look_at(origin, target, up), x_fov=float(x_fov), near_clip=float(near_clip),
                far_clip=float(far_clip))


# ------------------------------------------------------------------------------------------------
# Procedural meshes
# ------------------------------------------------------------------------------------------------
def icosphere(subdiv: int):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    for _ in range(subdiv):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        e.sort(axis=1)
        ue, inv = np.unique(e, axis=0, return_inverse=True)
        mid = v[ue[:, 0]] + v[ue[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        n0 = v.shape[0]
        v = np.concatenate([v, mid])
        nf = f.shape[0]
        a, b, c = inv[:nf] + n0, inv[nf:2 * nf] + n0, inv[2 * nf:] + n0
        f = np.concatenate([np.stack([f[:, 0], a, c], 1), np.stack([f[:, 1], b, a], 1),
                            np.stack([f[:, 2], c, b], 1), np.stack([a, b, c], 1)])
    return v, f


def vertex_normals(p, f):
    """Angle-weighted vertex normals (TriMesh::computeNormals, src/librender/trimesh.cpp:636-672)."""
    p = p.astype(np.float64)
    n = np.zeros_like(p)
    for i in range(3):
        v0, v1, v2 = p[f[:, i]], p[f[:, (i + 1) % 3]], p[f[:, (i + 2) % 3]]
        a, b = v1 - v0, v2 - v0
        fn = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
        ln = np.linalg.norm(fn, axis=1, keepdims=True)
        fn = fn / np.maximum(ln, 1e-30)
        an = a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-30)
        bn = b / np.maximum(np.linalg.norm(b, axis=1, keepdims=True), 1e-30)
        ang = np.arccos(np.clip(np.sum(an * bn, axis=1), -1, 1))
        np.add.at(n, f[:, i], fn * ang[:, None])
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where(ln > 0, n / np.maximum(ln, 1e-30), np.array([1.0, 0, 0]))
    return n.astype(np.float32)


def blob(subdiv: int, rng: np.random.Generator, radius=1.0, bump=0.12):
    """Displaced icosphere: low-frequency sinusoidal lobes plus a small per-vertex jitter."""
    v, f = icosphere(subdiv)
    k = rng.uniform(1.5, 4.0, size=(3, 3))
    ph = rng.uniform(0, 2 * math.pi, size=3)
    disp = sum(np.sin(v @ k[i] + ph[i]) for i in range(3)) / 3.0
    jitter = rng.uniform(-1, 1, size=v.shape[0]) * 0.004
    r = radius * (1.0 + bump * disp + jitter)
    p = (v * r[:, None]).astype(np.float32)
    return p, f.astype(np.uint32), vertex_normals(p, f)


def grid(nx: int, ny: int, lo, hi, z=0.0):
    xs = np.linspace(lo[0], hi[0], nx + 1, dtype=np.float32)
    ys = np.linspace(lo[1], hi[1], ny + 1, dtype=np.float32)
    X, Y = np.meshgrid(xs, ys)
    p = np.stack([X.ravel(), Y.ravel(), np.full(X.size, z, np.float32)], 1).astype(np.float32)
    i = (np.arange(ny)[:, None] * (nx + 1) + np.arange(nx)[None, :]).ravel()
    f = np.concatenate([np.stack([i, i + 1, i + nx + 2], 1), np.stack([i, i + nx + 2, i + nx + 1], 1)])
    return p, f.astype(np.uint32)


# ------------------------------------------------------------------------------------------------
# Benchmark / test scenes
# ------------------------------------------------------------------------------------------------
def cornell_box(width=512, height=512, max_depth=-1) -> SceneData:
    """Standard 555-unit Cornell box authored as triangle meshes with diffuse reflectances and one quad
    area light (config 1: `path`, diffuse BSDF)."""
    b = SceneBuilder()
    white = b.add_material(diffuse((0.725, 0.71, 0.68)))
    red = b.add_material(diffuse((0.63, 0.065, 0.05)))
    green = b.add_material(diffuse((0.14, 0.45, 0.091)))
    light = b.add_material(diffuse((0.78, 0.78, 0.78)))
    S = 555.0
    b.add_quad([0, 0, 0], [0, 0, S], [S, 0, S], [S, 0, 0], white)          # floor (y=0), normal +y
    b.add_quad([0, S, 0], [S, S, 0], [S, S, S], [0, S, S], white)          # ceiling, normal -y
    b.add_quad([0, 0, S], [0, S, S], [S, S, S], [S, 0, S], white)          # back wall (z=S), normal -z
    b.add_quad([S, 0, 0], [S, 0, S], [S, S, S], [S, S, 0], red)            # x=S wall, normal -x
    b.add_quad([0, 0, 0], [0, S, 0], [0, S, S], [0, 0, S], green)          # x=0 wall, normal +x

    def rot_y(deg, t):
        a = math.radians(deg)
        m = np.eye(4, dtype=np.float32)
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
        m[:3, 3] = t
        return m

    b.add_box([-82.5, 0, -82.5], [82.5, 165, 82.5], white, to_world=rot_y(-17, [185, 0, 169]))
    b.add_box([-82.5, 0, -82.5], [82.5, 330, 82.5], white, to_world=rot_y(20, [368, 0, 351]))
    y = S - 0.5
    b.add_quad([213, y, 227], [343, y, 227], [343, y, 332], [213, y, 332], light, radiance=(17.0, 12.0, 4.0))
    cam = make_camera([278, 273, -800], [278, 273, 0], [0, 1, 0], 39.3077, near_clip=10, far_clip=2800)
    return b.finish(cam, make_film(width, height), dict(max_depth=max_depth))


def cornell_mixed(width=256, height=256, max_depth=-1) -> SceneData:
    """The Cornell box with the neighbours of the hot-path BSDFs: walls behind a `twosided` adapter, a smooth
    `plastic` short box, a smooth `conductor` tall box, a `roughdielectric` blob, and a free-standing card whose two sides carry different
    BRDFs (rough plastic towards the back wall, red diffuse towards the camera) and which the camera sees from behind."""
    b = SceneBuilder()
    white = b.add_material(twosided(diffuse((0.725, 0.71, 0.68))))
    red = b.add_material(twosided(diffuse((0.63, 0.065, 0.05))))
    green = b.add_material(diffuse((0.14, 0.45, 0.091)))
    light = b.add_material(diffuse((0.78, 0.78, 0.78)))
    shiny = b.add_material(plastic(int_ior=1.49, diffuse_reflectance=(0.2, 0.3, 0.6)))
    metal = b.add_material(conductor(eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14)))
    card_back = b.add_material(diffuse((0.7, 0.1, 0.1)))
    card = b.add_material(twosided(roughplastic(alpha=0.2, distribution="ggx", diffuse_reflectance=(0.1, 0.5, 0.2)), card_back))
    S = 555.0
    b.add_quad([0, 0, 0], [0, 0, S], [S, 0, S], [S, 0, 0], white)
    b.add_quad([0, S, 0], [S, S, 0], [S, S, S], [0, S, S], white)
    b.add_quad([0, 0, S], [0, S, S], [S, S, S], [S, 0, S], white)
    b.add_quad([S, 0, 0], [S, 0, S], [S, S, S], [S, S, 0], red)
    b.add_quad([0, 0, 0], [0, S, 0], [0, S, S], [0, 0, S], green)

    def rot_y(deg, t):
        a = math.radians(deg)
        m = np.eye(4, dtype=np.float32)
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
        m[:3, 3] = t
        return m

    b.add_box([-82.5, 0, -82.5], [82.5, 165, 82.5], shiny, to_world=rot_y(-17, [185, 0, 169]))
    b.add_box([-82.5, 0, -82.5], [82.5, 330, 82.5], metal, to_world=rot_y(20, [368, 0, 351]))
    # a frosted-glass blob in front of the tall box
    frosted = b.add_material(roughdielectric(alpha=0.15, distribution="ggx", int_ior=1.5, ext_ior=1.0))
    bp, bf, bn = blob(2, np.random.default_rng(11), radius=60.0)
    b.add_mesh(bp + np.float32([420, 70, 120]), bf, frosted, normals=bn)
    # the card: its front normal points at the back wall (+z), the camera looks at its back
    b.add_quad([60, 250, 120], [230, 250, 150], [230, 420, 150], [60, 420, 120], card)
    y = S - 0.5
    b.add_quad([213, y, 227], [343, y, 227], [343, y, 332], [213, y, 332], light, radiance=(17.0, 12.0, 4.0))
    cam = make_camera([278, 273, -800], [278, 273, 0], [0, 1, 0], 39.3077, near_clip=10, far_clip=2800)
    return b.finish(cam, make_film(width, height), dict(max_depth=max_depth))


def microfacet_field(n_objects=64, subdiv=5, ground_res=90, width=1920, height=1080, seed=42,
                     max_depth=-1, n_lights=4, light_radiance=20.0, emitters_grid=0, textured=None) -> SceneData:
    """Configs 3-5: a field of displaced icosphere "blobs" over a tessellated ground plane.
    Objects cycle through roughconductor (GGX), roughplastic (GGX) and dielectric; the ground is
    diffuse; quad area lights hang above. subdiv=5 gives 20480 triangles per object.
    `textured` (a MIP filter name: "ewa", "trilinear", "bilinear", "nearest"): the same geometry with texture
    coordinates, 1024 x 1024 bitmap textures on the ground's and the plastic's diffuse colour and on the conductor's
    specular colour, and a bump map on the ground."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    m_cond = b.add_material(roughconductor(eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), alpha=0.15, distribution="ggx"))
    m_plast = b.add_material(roughplastic(alpha=0.1, distribution="ggx", int_ior=1.49, ext_ior=IOR["air"],
                                          diffuse_reflectance=(0.5, 0.2, 0.2)))
    m_glass = b.add_material(dielectric(1.5, 1.0))
    m_ground = b.add_material(diffuse((0.5, 0.5, 0.5)))
    m_light = b.add_material(diffuse((0.5, 0.5, 0.5)))
    side = int(math.ceil(math.sqrt(n_objects)))
    spacing = 2.6
    ext = side * spacing * 0.5
    k = 0
    for gy in range(side):
        for gx in range(side):
            if k >= n_objects:
                break
            p, f, n = blob(subdiv, rng)
            c = np.array([(gx + 0.5) * spacing - ext + rng.uniform(-0.2, 0.2),
                          (gy + 0.5) * spacing - ext + rng.uniform(-0.2, 0.2), 1.25], np.float32)
            uv = None
            if textured:        # latitude-longitude coordinates of the undisplaced direction
                d = p / np.linalg.norm(p, axis=1, keepdims=True)
                uv = np.stack([np.arctan2(d[:, 1], d[:, 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(d[:, 2], -1, 1)) / np.pi], -1)
            b.add_mesh(p + c, f, (m_cond, m_plast, m_glass)[k % 3], normals=n, texcoords=uv)
            k += 1
    gp, gf = grid(ground_res, ground_res, (-ext - 4, -ext - 4), (ext + 4, ext + 4), 0.0)
    b.add_mesh(gp, gf, m_ground, texcoords=(gp[:, :2] + (ext + 4)) / (2 * (ext + 4)) if textured else None)
    if textured:
        trng = np.random.default_rng(seed + 1)
        yy, xx = np.mgrid[0:1024, 0:1024]

        def image(freq, lo, hi):       # band-limited pattern plus texel noise, three channels
            base = 0.5 + 0.5 * np.sin(xx * (2 * np.pi * freq / 1024))[..., None] * np.cos(yy * (2 * np.pi * freq / 1024))[..., None]
            img = lo + (hi - lo) * (0.75 * base + 0.25 * trng.random((1024, 1024, 3)))
            return img.astype(np.float32)

        M = b.materials
        M[m_ground]["diffuse_texture"] = b.add_texture(bitmap(image(24, 0.15, 0.75), filter_type=textured, uscale=6.0, vscale=6.0))
        M[m_plast]["diffuse_texture"] = b.add_texture(bitmap(image(7, 0.1, 0.7), filter_type=textured, uscale=2.0))
        M[m_cond]["specular_texture"] = b.add_texture(bitmap(image(5, 0.6, 1.0), filter_type=textured))
        bump_tex = b.add_texture(bitmap(image(40, 0.0, 1.0)[..., 0], filter_type=textured, uscale=6.0, vscale=6.0, scale=0.02))
        M[m_ground].update(frame_kind=FRAME_BUMPMAP, frame_texture=bump_tex)
    H = 9.0
    if emitters_grid:
        g = emitters_grid
        s = 2 * (ext + 2) / g
        q = 0.18 * s
        for j in range(g):
            for i in range(g):
                cx, cy = -ext - 2 + (i + 0.5) * s, -ext - 2 + (j + 0.5) * s
                rad = light_radiance * (s * s) / (4 * q * q) * 0.25
                b.add_quad([cx - q, cy - q, H], [cx - q, cy + q, H], [cx + q, cy + q, H], [cx + q, cy - q, H],
                           m_light, radiance=(rad, rad, rad))
    else:
        L = 2.0
        offs = [(-0.5, -0.5), (0.5, -0.5), (-0.5, 0.5), (0.5, 0.5)][:n_lights]
        for ox, oy in offs:
            cx, cy = ox * ext, oy * ext
            b.add_quad([cx - L, cy - L, H], [cx - L, cy + L, H], [cx + L, cy + L, H], [cx + L, cy - L, H],
                       m_light, radiance=(light_radiance,) * 3)
    cam = make_camera([0.0, -2.3 * ext - 6, 0.9 * ext + 4], [0, 0, 0.8], [0, 0, 1], 35.0, near_clip=0.1, far_clip=1000)
    return b.finish(cam, make_film(width, height), dict(max_depth=max_depth))


def config3(width=1920, height=1080) -> SceneData:
    """~1.0 M triangles: 48 objects x 20480 + 2 x 90^2 ground."""
    return microfacet_field(n_objects=48, subdiv=5, ground_res=90, width=width, height=height)


def config3_textured(width=1920, height=1080, filter_type="ewa") -> SceneData:
    """config3 with bitmap textures on three of its four surface materials and a bump-mapped ground."""
    return microfacet_field(n_objects=48, subdiv=5, ground_res=90, width=width, height=height, textured=filter_type)


def config5(width=3840, height=2160) -> SceneData:
    """~10 M triangles, 1024 quad emitters, max depth 16."""
    return microfacet_field(n_objects=120, subdiv=6, ground_res=300, width=width, height=height,
                            max_depth=16, emitters_grid=32)


def rotation_about(axis, degrees) -> np.ndarray:
    """Transform::rotate (src/libcore/transform.cpp) as a 4 x 4 matrix."""
    a = np.asarray(axis, np.float64)
    x, y, z = a / np.linalg.norm(a)
    t = np.radians(degrees)
    c, s_ = np.cos(t), np.sin(t)
    m = np.eye(4)
    m[:3, :3] = [[c + x * x * (1 - c), x * y * (1 - c) - z * s_, x * z * (1 - c) + y * s_],
                 [y * x * (1 - c) + z * s_, c + y * y * (1 - c), y * z * (1 - c) - x * s_],
                 [z * x * (1 - c) - y * s_, z * y * (1 - c) + x * s_, c + z * z * (1 - c)]]
    return m.astype(np.float32)


def test_environment_map(width=64, height=32, seed=5) -> np.ndarray:
    """A synthetic high-dynamic-range latitude-longitude image: a smooth sky gradient, a dim ground, noise at the texel
    scale and a small very bright sun, so that importance sampling, the tent offsets and the filtered lookups all matter."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width]
    theta = (yy + 0.5) / height * np.pi
    sky = np.where(theta < np.pi / 2, 0.3 + 0.7 * np.cos(theta), 0.05)
    img = np.stack([sky * 0.6, sky * 0.8, sky * 1.2], -1) * (0.8 + 0.4 * rng.random((height, width, 1)))
    sun = np.exp(-(((xx - width * 0.3) / 1.5) ** 2 + ((yy - height * 0.25) / 1.5) ** 2))
    img += sun[..., None] * np.array([400.0, 350.0, 250.0])
    return img.astype(np.float32)


def small_test_scene(width=64, height=48, subdiv=2, max_depth=-1, sky=None, sky_weight=1.0, hide_emitters=0, lamps=False, area_lights=True,
                     blend=False, textured=False, bumps=False, rough_maps=False) -> SceneData:
    """A few hundred triangles with every BSDF type present; used by the parity tests. `sky` adds a constant
    environment emitter of that radiance next to the two area lights; `lamps` adds a point, a spot and a directional light;
    `blend` makes the ground and two of the blobs mixtures of the other materials; `textured` puts checkerboard and grid
    textures on colours of most materials (meshes with and without texture coordinates, colours above one, a scale
    texture, two textures in one energy check); `rough_maps` textures the roughness of the rough conductors, the rough plastics
    and a rough dielectric."""
    rng = np.random.default_rng(7)
    b = SceneBuilder()
    mats = [b.add_material(diffuse((0.6, 0.5, 0.4))),
            b.add_material(roughconductor(eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), alpha=0.15, distribution="ggx")),
            b.add_material(roughplastic(alpha=0.1, distribution="ggx", int_ior=1.49, diffuse_reflectance=(0.5, 0.2, 0.2))),
            b.add_material(dielectric(1.5, 1.0)),
            b.add_material(ward(alpha=0.4, diffuse_reflectance=(0.1, 0.1, 0.1), specular_reflectance=(0.616, 0.4752, 0.352))),
            b.add_material(roughconductor(eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), alpha=0.3, distribution="beckmann")),
            b.add_material(roughplastic(alpha=0.3, distribution="beckmann", int_ior=1.49, diffuse_reflectance=(0.2, 0.4, 0.6)))]
    m_ground = b.add_material(diffuse((0.5, 0.5, 0.5)))
    m_light = m_ground
    if textured:
        M = b.materials
        m_ground = b.add_material(diffuse())
        M[m_ground]["diffuse_texture"] = b.add_texture(checkerboard((0.7, 0.7, 0.6), (0.15, 0.2, 0.3), uscale=7.0, vscale=6.0, uoffset=0.25))
        M[mats[0]]["diffuse_texture"] = b.add_texture(gridtexture((0.3, 0.5, 0.2), (1.5, 0.3, 0.3), line_width=0.1))       # barycentric uv, > 1
        M[mats[1]]["specular_texture"] = b.add_texture(checkerboard((1.0, 0.9, 0.8), (0.5, 0.5, 1.0), uscale=3.0, vscale=3.0, scale=0.8))
        M[mats[2]]["diffuse_texture"] = b.add_texture(checkerboard((0.5, 0.2, 0.2), (0.1, 0.3, 0.6), uscale=2.0, vscale=4.0))
        M[mats[2]]["specular_texture"] = b.add_texture(gridtexture((1.0, 1.0, 1.0), (0.2, 0.2, 0.2), line_width=0.2, voffset=0.3))
        M[mats[3]]["transmittance_texture"] = b.add_texture(checkerboard((1.0, 0.6, 0.6), (0.6, 1.0, 0.7), uscale=2.0, vscale=2.0))
        M[mats[4]]["diffuse_texture"] = b.add_texture(checkerboard((0.6, 0.5, 0.1), (0.1, 0.1, 0.5), uscale=4.0, vscale=4.0))
        M[mats[4]]["specular_texture"] = b.add_texture(gridtexture((0.7, 0.6, 0.5), (0.2, 0.9, 0.2), line_width=0.15, uscale=3.0, vscale=3.0))
    if textured == "bitmap":
        # image textures in place of some of the procedural ones: every lookup mode, wrap modes, RGB and luminance
        # images of odd sizes, values above one, a scale factor
        trng = np.random.default_rng(11)

        def image(h, w, channels, lo=0.05, hi=0.95, smooth=3):
            img = trng.uniform(lo, hi, (h, w, channels)).astype(np.float32)
            for _ in range(smooth):     # some structure at several scales, so that the MIP levels differ
                img = (img + np.roll(img, 1, 0) + np.roll(img, 1, 1) + np.roll(np.roll(img, 1, 0), 1, 1)) * np.float32(0.25)
            img = img * np.float32(2.0) - np.float32(0.5) + (np.indices((h, w)).sum(0)[..., None] % 8 < 4) * np.float32(0.3)
            img = np.clip(img, lo, hi + 0.3).astype(np.float32)
            return img[:, :, 0] if channels == 1 else img

        M[m_ground]["diffuse_texture"] = b.add_texture(bitmap(image(61, 47, 3), "ewa", uscale=3.0, vscale=2.0, uoffset=0.25))
        M[mats[0]]["diffuse_texture"] = b.add_texture(bitmap(image(32, 32, 1), "trilinear", wrap_mode="mirror", uscale=2.0, vscale=2.0))
        M[mats[1]]["specular_texture"] = b.add_texture(bitmap(image(16, 24, 3), "bilinear", wrap_mode="clamp", scale=0.8))
        M[mats[2]]["diffuse_texture"] = b.add_texture(bitmap(image(9, 13, 3), "nearest", uscale=2.0, vscale=4.0))
        M[mats[3]]["transmittance_texture"] = b.add_texture(bitmap(image(40, 20, 3, hi=1.2), "ewa", wrap_mode_u="zero", wrap_mode_v="one", max_anisotropy=4.0))
        M[mats[4]]["diffuse_texture"] = b.add_texture(bitmap(image(128, 128, 3, smooth=1), "ewa", uscale=1.5, vscale=1.5))
        if bumps:
            # bump and normal maps: a height image (bitmap gradient), a procedural height (forward differences), a
            # normal image; around plain, textured, two-sided and transmissive records
            bump_img = image(48, 48, 1, smooth=4)
            yy, xx = np.mgrid[0:32, 0:32]
            nrm = np.stack([0.5 + 0.25 * np.sin(xx / 3.0), 0.5 + 0.25 * np.cos(yy / 4.0), np.full(xx.shape, 0.9)], -1).astype(np.float32)
            t_height = b.add_texture(bitmap(bump_img, "bilinear", uscale=3.0, vscale=3.0, scale=0.3))
            t_grid = b.add_texture(gridtexture((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), line_width=0.2, uscale=5.0, vscale=5.0, scale=2e-5))
            t_normal = b.add_texture(bitmap(nrm, "bilinear", uscale=2.0, vscale=2.0))
            M[m_ground] = bumpmap(M[m_ground], t_height)
            M[mats[1]] = normalmap(M[mats[1]], t_normal)
            M[mats[2]] = bumpmap(M[mats[2]], t_grid)
            M[mats[3]] = bumpmap(M[mats[3]], t_height)
            M[mats[5]] = twosided(normalmap(M[mats[5]], t_normal))
    if rough_maps:
        # textured roughness: a checkerboard on both directions of the GGX conductor, an image on alphaU alone of the
        # Beckmann one (anisotropic), an image on a rough dielectric that takes the smooth dielectric's place
        M = b.materials
        t_check = b.add_texture(checkerboard((0.05, 0.05, 0.05), (0.4, 0.4, 0.4), uscale=3.0, vscale=3.0))
        # (a smooth image: the roughness decides where a path goes next, and a value that jumps from texel to texel would
        # turn last-bit differences of a hit position into different paths within a few bounces inside the glass)
        ry, rx = np.mgrid[0:24, 0:24]
        rough = (0.275 + 0.225 * np.sin(rx * (2 * np.pi / 24)) * np.cos(ry * (4 * np.pi / 24))).astype(np.float32)
        t_image = b.add_texture(bitmap(rough, "trilinear"))
        M[mats[1]].update(alpha_u_texture=t_check, alpha_v_texture=t_check)
        M[mats[5]].update(alpha_u_texture=t_image)
        mats[3] = b.add_material(dict(roughdielectric(alpha=0.2, distribution="ggx", int_ior=1.5, ext_ior=1.0), alpha_u_texture=t_image, alpha_v_texture=t_image))
        # the two rough plastics: their transmittance tables stay functions of the roughness (roughplastic.cpp:281-299)
        M[mats[2]].update(alpha_u_texture=t_image, alpha_v_texture=t_image,
                          rtrans=np.asarray(load_rtrans_eta_block("ggx", M[mats[2]]["int_ior"] / M[mats[2]]["ext_ior"]), np.float32))
        M[mats[6]].update(alpha_u_texture=t_check, alpha_v_texture=t_check, variant=1,
                          rtrans=np.asarray(load_rtrans_eta_block("beckmann", M[mats[6]]["int_ior"] / M[mats[6]]["ext_ior"]), np.float32))
    if blend:
        m_ground = b.add_material(twosided(mixture([m_ground, mats[1]], [0.75, 0.25])))
        mats[0] = b.add_material(mixture([mats[0], mats[2], mats[4]], [0.6, 0.5, 0.4]))      # weights above one: rescaled
        mats[5] = b.add_material(mixture([mats[5], mats[6], mats[1], m_light], [0.3, 0.2, 0.0, 0.4]))
        shiny = b.add_material(plastic(int_ior=1.49, diffuse_reflectance=(0.3, 0.5, 0.2)))
        frosted = b.add_material(roughdielectric(alpha=0.2, distribution="ggx", int_ior=1.5, ext_ior=1.0))
        mats[3] = b.add_material(mixture([mats[3], shiny, frosted], [0.4, 0.3, 0.3]))    # Dirac lobes next to a rough dielectric
    for k, m in enumerate(mats):
        p, f, n = blob(subdiv, rng, radius=0.8)
        c = np.array([(k % 4) * 2.2 - 3.3, (k // 4) * 2.4 - 1.2, 1.0], np.float32)
        uv = (p[:, :2] * 0.5 + 0.5).astype(np.float32) if k == 4 else None
        b.add_mesh(p + c, f, m, normals=n if k != 0 else None, texcoords=uv)
    gp, gf = grid(6, 6, (-7, -6), (7, 6), 0.0)
    b.add_mesh(gp, gf, m_ground, texcoords=((gp[:, :2] + np.float32([7, 6])) / np.float32([14, 12])).astype(np.float32) if textured else None)
    b.add_quad([-2, -2, 6], [-2, 2, 6], [2, 2, 6], [2, -2, 6], m_light, radiance=(15, 14, 12) if area_lights else None)
    b.add_quad([4, -1, 3.5], [4, 1, 3.5], [5, 1, 4.5], [5, -1, 4.5], m_light, radiance=(6, 9, 12) if area_lights else None, sampling_weight=0.5)
    if isinstance(sky, str) and sky == "map":
        b.add_envmap_emitter(test_environment_map(), to_world=rotation_about([0.2, 0.3, 1.0], 40.0) @ rotation_about([1, 0, 0], 90.0),
                             scale=0.6, sampling_weight=sky_weight)
    elif sky is not None:
        b.add_constant_emitter(sky, sky_weight)
    if lamps:
        b.add_point_emitter([-3.0, -4.0, 3.0], (30.0, 20.0, 10.0))
        b.add_spot_emitter(look_at([3.0, -5.0, 5.0], [1.0, 0.0, 0.5], [0, 0, 1]), (200.0, 300.0, 400.0), cutoff_angle=25.0, beam_width=12.0,
                           sampling_weight=2.0)
        b.add_directional_emitter(direction=[0.3, 0.4, -1.0], irradiance=(0.8, 0.7, 0.6), sampling_weight=0.5)
    cam = make_camera([0.5, -11, 5.5], [0, 0, 0.8], [0, 0, 1], 40.0, near_clip=0.1, far_clip=100)
    return b.finish(cam, make_film(width, height), dict(max_depth=max_depth, hide_emitters=hide_emitters))


def nested_slivers(n=420, ratio=1.15, first=1e-12) -> SceneData:
    """A deliberately deep hierarchy: n thin triangles that all start at the origin, each `ratio` times larger than the
    one before and slightly higher up. Surface-area splits peel the largest few off at every level, so the tree a
    builder makes of them is a long chain (depth ~ n / 14 with the host builder); used to test the traversal stack at
    and beyond its limit. One diffuse material, no emitters."""
    k = np.arange(n, dtype=np.float64)
    s = first * ratio ** k
    z = 1e-3 * s
    pos = np.zeros((n, 3, 3))
    pos[:, 0, 2] = z
    pos[:, 1, 0], pos[:, 1, 1], pos[:, 1, 2] = s, 0.01 * s, z
    pos[:, 2, 0], pos[:, 2, 1], pos[:, 2, 2] = 0.01 * s, s, z
    b = SceneBuilder()
    m = b.add_material(diffuse())
    b.add_mesh(pos.reshape(-1, 3).astype(np.float32), np.arange(3 * n, dtype=np.uint32).reshape(-1, 3), m)
    return b.finish(make_camera([0.3, 0.3, 5.0], [0.3, 0.3, 0.0], [0, 1, 0], 30.0), make_film(16, 16))


def veach_bidir(width=1024, height=768, max_depth=-1, data_dir=None) -> SceneData:
    """Config 2: the veach_bidir scene (geometry of testScene/veach_bidir/bidir.serialized, materials, emitters and
    camera of bidir.bak) rendered with the `path` integrator. Geometry and scene description are read from the
    package data extracted by tools/make_golden.py (data/veach_bidir/)."""
    from . import loader
    if data_dir is None:
        data_dir = os.path.join(_DATA_DIR, "veach_bidir")
    z = np.load(os.path.join(data_dir, "meshes.npz"))

    def provider(filename, idx):
        return dict(positions=z["p%d" % idx], indices=z["i%d" % idx],
                    normals=z["n%d" % idx] if "n%d" % idx in z.files else None,
                    texcoords=z["t%d" % idx] if "t%d" % idx in z.files else None, face_normals=False)

    sd, _ = loader.load_xml(os.path.join(data_dir, "scene_path.xml"), width=width, height=height,
                            integrator_override=dict(max_depth=max_depth), mesh_provider=provider)
    return sd
