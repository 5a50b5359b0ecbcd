"""Host-side scene ingest for the subset of the Mitsuba 0.5 scene format the `path` hot path needs:
the XML scene description (src/librender/scenehandler.cpp:306-693) restricted to
scene/integrator/sensor/sampler/film/rfilter/bsdf/shape/emitter/ref/transform and the scalar property tags,
and the `.serialized` triangle-mesh container (src/librender/trimesh.cpp:175-294).

The result is a flat `SceneData` for the C ABI. Everything unsupported raises -- nothing is silently
substituted.
"""
from __future__ import annotations

import math
import os
import struct
import xml.etree.ElementTree as ET
import zlib
from typing import Dict, Optional

import numpy as np

from . import scene as S

_FILEFORMAT_HEADER = 0x041C
_HAS_NORMALS, _HAS_TEXCOORDS, _HAS_COLORS, _FACE_NORMALS, _DOUBLE = 0x1, 0x2, 0x8, 0x10, 0x2000


class SceneFormatError(ValueError):
    pass


# ------------------------------------------------------------------------------------------------
# .serialized meshes
# ------------------------------------------------------------------------------------------------
def load_serialized(path_or_bytes, shape_index: int = 0) -> Dict:
    """TriMesh::loadCompressed (src/librender/trimesh.cpp:175-252): header, optional offset dictionary at the
    end of the file (readOffset, :272-294), then one zlib stream per mesh."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    fmt, version = struct.unpack_from("<HH", data, 0)
    if fmt != _FILEFORMAT_HEADER:
        raise SceneFormatError("Encountered an invalid file format!")
    if version not in (3, 4):
        raise SceneFormatError("Encountered an incompatible file version!")
    offset = 0
    if shape_index != 0:
        count = struct.unpack_from("<I", data, len(data) - 4)[0]
        if shape_index < 0 or shape_index > count:
            raise SceneFormatError("Unable to unserialize mesh, shape index is out of range!")
        if version == 4:
            offset = struct.unpack_from("<Q", data, len(data) - 8 * (count - shape_index) - 4)[0]
        else:
            offset = struct.unpack_from("<I", data, len(data) - 4 * (count - shape_index + 1))[0]
    raw = zlib.decompressobj().decompress(data[offset + 4:])
    p = 0
    flags = struct.unpack_from("<I", raw, p)[0]; p += 4
    name = ""
    if version == 4:
        e = raw.index(b"\0", p)
        name = raw[p:e].decode("utf-8", "replace"); p = e + 1
    nv, nt = struct.unpack_from("<QQ", raw, p); p += 16
    ftype = np.dtype("<f8") if flags & _DOUBLE else np.dtype("<f4")

    def take(n, comps):
        nonlocal p
        a = np.frombuffer(raw, ftype, n * comps, p).reshape(n, comps).astype(np.float32)
        p += n * comps * ftype.itemsize
        return a

    out = dict(name=name, positions=take(nv, 3), normals=None, texcoords=None, face_normals=bool(flags & _FACE_NORMALS))
    if flags & _HAS_NORMALS:
        out["normals"] = take(nv, 3)
    if flags & _HAS_TEXCOORDS:
        out["texcoords"] = take(nv, 2)
    if flags & _HAS_COLORS:
        take(nv, 3)
    out["indices"] = np.frombuffer(raw, "<u4", nt * 3, p).reshape(nt, 3).copy()
    return out


def serialized_mesh_count(data: bytes) -> int:
    return struct.unpack_from("<I", data, len(data) - 4)[0]


def rebuild_topology(mesh: Dict, max_angle: float) -> Dict:
    """TriMesh::rebuildTopology (src/librender/trimesh.cpp:468-606): vertex normals are dropped and the vertices re-made so
    that, around every distinct (position, uv), the faces whose normals lie within max_angle of the first unassigned
    face's normal share a vertex (greedy, in the order the faces name the vertex)."""
    f32 = np.float32
    P = np.asarray(mesh["positions"], f32)
    I = np.asarray(mesh["indices"], np.int64).reshape(-1, 3)
    UV = mesh.get("texcoords")
    dp_thresh = f32(np.cos(f32(max_angle) * f32(np.pi / 180.0), dtype=f32))
    a, b = P[I[:, 1]] - P[I[:, 0]], P[I[:, 2]] - P[I[:, 0]]
    n = np.cross(a, b).astype(f32)
    l = np.sqrt((n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1]) + n[:, 2] * n[:, 2], dtype=f32)
    ok = l > f32(2.93873587705571876e-39)
    n = np.where(ok[:, None], n / np.where(ok, l, 1)[:, None], 0).astype(f32)
    entries = []
    for t in range(I.shape[0]):
        for j in range(3):
            v = int(I[t, j])
            key = tuple(P[v]) + (tuple(UV[v]) if UV is not None else (0.0, 0.0))
            entries.append((key, t))
    order = sorted(range(len(entries)), key=lambda e: entries[e][0])      # stable: equal keys stay in insertion order
    new_p, new_uv = [], []
    new_i = np.full(I.shape, -1, np.int64)
    k = 0
    while k < len(order):
        end = k
        while end < len(order) and entries[order[end]][0] == entries[order[k]][0]:
            end += 1
        group = [entries[e] for e in order[k:end]]
        clustered = [False] * len(group)
        for i2, (key, t1) in enumerate(group):
            if clustered[i2]:
                continue
            idx = len(new_p)
            new_p.append(key[:3])
            new_uv.append(key[3:])
            for i3 in range(i2, len(group)):
                if clustered[i3]:
                    continue
                t2 = group[i3][1]
                dot = (n[t1, 0] * n[t2, 0] + n[t1, 1] * n[t2, 1]) + n[t1, 2] * n[t2, 2]
                if np.array_equal(n[t1], n[t2]) or dot > dp_thresh:
                    for c in range(3):
                        if tuple(P[I[t2, c]]) == key[:3]:
                            new_i[t2, c] = idx
                    clustered[i3] = True
        k = end
    assert (new_i >= 0).all()
    out = dict(mesh, positions=np.asarray(new_p, f32).reshape(-1, 3), indices=new_i.astype(np.uint32), normals=None)
    if UV is not None:
        out["texcoords"] = np.asarray(new_uv, f32).reshape(-1, 2)
    return out


def apply_to_world(mesh: Dict, m: np.ndarray) -> Dict:
    """SerializedMesh constructor (src/shapes/serialized.cpp:184-202): positions by the matrix, normals by the
    inverse transpose followed by normalisation, winding swapped when the determinant is negative."""
    m = np.asarray(m, np.float32)
    if np.array_equal(m, np.eye(4, dtype=np.float32)):
        return mesh
    out = dict(mesh)
    p = mesh["positions"]
    out["positions"] = (p @ m[:3, :3].T + m[:3, 3]).astype(np.float32)
    if mesh.get("normals") is not None:
        inv_t = np.linalg.inv(m.astype(np.float64))[:3, :3].T.astype(np.float32)
        n = (mesh["normals"] @ inv_t.T).astype(np.float32)
        ln = np.sqrt((n * n).sum(1, keepdims=True)).astype(np.float32)
        out["normals"] = (n / ln).astype(np.float32)
    if np.linalg.det(m[:3, :3].astype(np.float64)) < 0:
        idx = mesh["indices"].copy()
        idx[:, [0, 1]] = idx[:, [1, 0]]
        out["indices"] = idx
    return out


# ------------------------------------------------------------------------------------------------
# XML
# ------------------------------------------------------------------------------------------------
def _floats(s: str):
    return [float(x) for x in s.replace(",", " ").split()]


def _srgb_to_linear(c):
    return c / 12.92 if c <= 0.04045 else ((c + 0.055) / 1.055) ** 2.4


_spectrum_dir = [None]      # directory against which <spectrum filename=...> is resolved: the scene file's (set by load_xml)


def _spectrum(node) -> tuple:
    """scenehandler.cpp:503-624 in the three-sample build (where Spectrum::getD65() is 1, so that both intents of a
    single value give the same constant)"""
    tag, v = node.tag, node.get("value")
    if tag == "rgb":
        f = _floats(v)
        if len(f) == 1:
            f = f * 3
        return tuple(f[:3])
    if tag == "srgb":
        if v.startswith("#"):
            f = [int(v[i:i + 2], 16) / 255.0 for i in (1, 3, 5)]
        else:
            f = _floats(v)
            if len(f) == 1:
                f = f * 3
        return tuple(_srgb_to_linear(c) for c in f[:3])
    from . import capi      # the conversions below are the library's (host code, no device needed)
    if tag == "blackbody":
        t = (node.get("temperature") or "").strip()
        if t[-1:].upper() == "K":
            t = t[:-1]
        if not t:
            raise SceneFormatError("<blackbody> needs a temperature")
        try:
            return tuple(float(c) for c in capi.blackbody_to_rgb(float(t), float(node.get("scale", 1.0))))
        except capi.MtsamdError as e:
            raise SceneFormatError("<blackbody>: %s" % e)
    if tag == "spectrum":
        has_file, has_intent = node.get("filename") is not None, node.get("intent") is not None
        if (v is not None) == has_file:
            raise SceneFormatError("<spectrum>: please provide one of 'value' or 'filename'")
        if has_file:
            if has_intent:
                raise SceneFormatError("<spectrum>: 'intent' and 'filename' cannot be specified at the same time!")
            path = node.get("filename")
            if not os.path.isabs(path) and _spectrum_dir[0]:
                path = os.path.join(_spectrum_dir[0], path)
            try:
                return tuple(float(c) for c in capi.spd_file_to_rgb(path, scene_value=True))
            except capi.MtsamdError as e:
                raise SceneFormatError(str(e))
        if ":" not in v:
            f = _floats(v)
            if len(f) == 1:
                if has_intent and node.get("intent").lower() not in ("reflectance", "illuminant"):
                    raise SceneFormatError('Invalid intent "%s", must be "reflectance" or "illuminant"' % node.get("intent").lower())
                return (f[0],) * 3
            if has_intent:
                raise SceneFormatError("<spectrum>: 'intent' can only be specified when given a single-valued argument.")
            if len(f) != 3:
                raise SceneFormatError("Invalid spectrum value specified (length does not match the current spectral discretization!)")
            return tuple(f)
        if has_intent:
            raise SceneFormatError("<spectrum>: 'intent' can only be specified when given a single-valued argument.")
        lam, val = [], []
        for token in v.replace(",", " ").split():
            parts = token.split(":")
            if len(parts) != 2 or not parts[0] or not parts[1]:
                raise SceneFormatError("Invalid spectrum->value mapping specified")
            lam.append(float(parts[0])), val.append(float(parts[1]))
        if len(lam) < 2:
            raise SceneFormatError("InterpolatedSpectrum::zeroExtend() -- at least 2 entries are needed!")
        try:
            return tuple(float(c) for c in capi.spd_to_rgb(lam, val, scene_value=True))
        except capi.MtsamdError:
            raise SceneFormatError("InterpolatedSpectrum: spectral power distribution values must be provided in order of "
                                   "increasing wavelength!")
    raise SceneFormatError("unsupported spectrum tag <%s>" % tag)


def _transform(node) -> np.ndarray:
    """<transform> children are applied in order, each pre-multiplied (scenehandler.cpp:574-640)."""
    m = np.eye(4, dtype=np.float64)
    for c in node:
        if c.tag == "matrix":
            f = _floats(c.get("value"))
            t = np.array(f, np.float64).reshape(4, 4)
        elif c.tag == "lookat":
            t = S.look_at(_floats(c.get("origin")), _floats(c.get("target")), _floats(c.get("up", "0,0,1"))).astype(np.float64)
        elif c.tag == "translate":
            t = np.eye(4)
            t[:3, 3] = [float(c.get(k, 0)) for k in "xyz"]
        elif c.tag == "scale":
            t = np.eye(4)
            if c.get("value") is not None:
                s = [float(c.get("value"))] * 3
            else:
                s = [float(c.get(k, 1)) for k in "xyz"]
            t[0, 0], t[1, 1], t[2, 2] = s
        elif c.tag == "rotate":
            ax = np.array([float(c.get(k, 0)) for k in "xyz"], np.float64)
            ax /= np.linalg.norm(ax)
            a = math.radians(float(c.get("angle")))
            x, y, z = ax
            cs, sn = math.cos(a), math.sin(a)
            t = np.eye(4)
            t[:3, :3] = [[cs + x * x * (1 - cs), x * y * (1 - cs) - z * sn, x * z * (1 - cs) + y * sn],
                         [y * x * (1 - cs) + z * sn, cs + y * y * (1 - cs), y * z * (1 - cs) - x * sn],
                         [z * x * (1 - cs) - y * sn, z * y * (1 - cs) + x * sn, cs + z * z * (1 - cs)]]
        else:
            raise SceneFormatError("unsupported transform element <%s>" % c.tag)
        m = t @ m
    return m.astype(np.float32)


def _props(node) -> Dict:
    """Scalar properties of a plugin element."""
    out = {}
    for c in node:
        n = c.get("name")
        if c.tag == "float":
            out[n] = float(c.get("value"))
        elif c.tag == "integer":
            out[n] = int(c.get("value"))
        elif c.tag == "boolean":
            out[n] = c.get("value").lower() == "true"
        elif c.tag == "string":
            out[n] = c.get("value")
        elif c.tag in ("rgb", "srgb", "spectrum", "blackbody"):
            out[n] = _spectrum(c)
        elif c.tag == "transform":
            out[n] = _transform(c)
        elif c.tag in ("point", "vector"):      # scenehandler.cpp:443-459
            out[n] = tuple(float(c.get(k, 0)) for k in "xyz")
    return out


def _ior(props, key, default):
    v = props.get(key, default)
    if isinstance(v, str):
        if v.lower() not in S.IOR:
            raise SceneFormatError("unknown IOR preset '%s'" % v)
        return S.IOR[v.lower()]
    return float(v)


def _conductor_constants(p: Dict, data_dir: Optional[str]):
    """eta and k of the conductor models before the division by extEta (src/bsdfs/roughconductor.cpp:174-190,
    src/bsdfs/conductor.cpp:153-171): explicit RGB values win, "none" stands for (0, 1), anything else names the
    measured spectra data/ior/<material>.{eta,k}.spd (default copper), converted to RGB by mtsamd_spd_file_to_rgb."""
    material = p.get("material", "Cu")
    out = []
    for key, none_value in (("eta", 0.0), ("k", 1.0)):
        if key in p:
            out.append(p[key])
        elif material.lower() == "none":
            out.append((none_value,) * 3)
        else:
            if data_dir is None:
                raise SceneFormatError("conductor material preset '%s' needs the measured spectra: pass the directory that "
                                       "holds ior/%s.%s.spd" % (material, material, key))
            from . import capi
            try:
                out.append(tuple(float(c) for c in capi.spd_file_to_rgb(os.path.join(data_dir, "ior", "%s.%s.spd" % (material, key)))))
            except capi.MtsamdError as e:
                raise SceneFormatError(str(e))
    return out[0], out[1]


def _make_texture(node, ctx) -> int:
    """A <texture> element as an entry of the scene's texture list; returns its 1-based index. `scale` textures
    (src/textures/scale.cpp) fold into the scale factor of a copy of the texture they wrap."""
    t = node.get("type")
    p = _props(node)
    uvt = dict(uoffset=p.get("uoffset", 0.0), voffset=p.get("voffset", 0.0), uscale=p.get("uscale", 1.0), vscale=p.get("vscale", 1.0))
    if t == "scale":
        nested = [c for c in node if c.tag in ("texture", "ref")]
        if len(nested) != 1:
            raise SceneFormatError("scale texture: exactly one nested texture is expected")
        if nested[0].tag == "ref":
            if nested[0].get("id") not in ctx["named_textures"]:
                raise SceneFormatError("reference to unknown texture '%s'" % nested[0].get("id"))
            inner = ctx["named_textures"][nested[0].get("id")]
        else:
            inner = _make_texture(nested[0], ctx)
        f = p.get("scale", 1.0)
        if isinstance(f, tuple):
            if not f[0] == f[1] == f[2]:
                raise SceneFormatError("scale texture: only a scalar (or grey) scale factor is supported")
            f = f[0]
        rec = dict(ctx["textures"][inner - 1])
        rec["scale"] = float(np.float32(rec["scale"]) * np.float32(f))
    elif t == "checkerboard":
        rec = S.checkerboard(p.get("color0", (0.4, 0.4, 0.4)), p.get("color1", (0.2, 0.2, 0.2)), **uvt)
    elif t == "gridtexture":
        rec = S.gridtexture(p.get("color0", (0.2, 0.2, 0.2)), p.get("color1", (0.4, 0.4, 0.4)), line_width=p.get("lineWidth", 0.01), **uvt)
    elif t == "bitmap":     # src/textures/bitmap.cpp:179-235
        if "filename" not in p:
            raise SceneFormatError("bitmap texture: the 'filename' parameter is missing")
        if p.get("channel"):
            raise SceneFormatError("bitmap texture: the 'channel' parameter is not supported")
        from . import capi
        try:
            pixels = capi.image_load(os.path.join(ctx["base"], p["filename"]), gamma=p.get("gamma", 0.0))
        except capi.MtsamdError as e:
            raise SceneFormatError(str(e))
        both = p.get("wrapMode", "repeat")
        try:
            rec = S.bitmap(pixels, p.get("filterType", "ewa"), wrap_mode_u=p.get("wrapModeU", both), wrap_mode_v=p.get("wrapModeV", both),
                           max_anisotropy=p.get("maxAnisotropy", 20.0), **uvt)
        except KeyError as e:
            raise SceneFormatError("Unknown filter type or wrap mode %s" % e)
    else:
        raise SceneFormatError("texture type '%s' is not supported (checkerboard, gridtexture, bitmap, scale)" % t)
    ctx["textures"].append(rec)
    return len(ctx["textures"])


def _attach_textures(node, m: Dict, ctx) -> None:
    """Nested <texture name="..."> elements and <ref name="..." id="..."/> to textures declared at the top level."""
    if ctx is None:
        if any(c.tag == "texture" for c in node):
            raise SceneFormatError("nested <texture> needs a scene to add it to")
        return
    for c in node:
        if c.tag == "texture":
            index = _make_texture(c, ctx)
        elif c.tag == "ref" and c.get("id") in ctx["named_textures"]:
            index = ctx["named_textures"][c.get("id")]
        else:
            continue
        name, ty = c.get("name"), m["type"]
        uses_diffuse = ty in (S.BSDF_DIFFUSE, S.BSDF_ROUGHPLASTIC, S.BSDF_PLASTIC, S.BSDF_WARD)
        if (name == "diffuseReflectance" or (name == "reflectance" and ty == S.BSDF_DIFFUSE)) and uses_diffuse:
            m["diffuse_texture"] = index
        elif name == "specularReflectance" and ty != S.BSDF_DIFFUSE:
            m["specular_texture"] = index
        elif name == "specularTransmittance" and ty in (S.BSDF_DIELECTRIC, S.BSDF_ROUGHDIELECTRIC):
            m["transmittance_texture"] = index
        elif name == "alpha" and ty == S.BSDF_ROUGHPLASTIC:
            # roughplastic.cpp:218,281-299: one isotropic roughness; the transmittance table stays a function of it
            m["alpha_u_texture"] = m["alpha_v_texture"] = index
            eta = m["int_ior"] / m["ext_ior"]
            distr = "ggx" if m["distribution"] == S.DISTR_GGX else "beckmann"
            if ctx.get("data_dir") is not None:
                from . import capi
                dat = open(os.path.join(ctx["data_dir"], "microfacet", distr + ".dat"), "rb").read()
                m["rtrans"] = capi.rtrans_reduce_eta(dat, eta)
            else:
                m["rtrans"] = np.asarray(S.load_rtrans_eta_block(distr, eta), np.float32)
        elif name in ("alpha", "alphaU", "alphaV") and ty in (S.BSDF_ROUGHCONDUCTOR, S.BSDF_ROUGHDIELECTRIC):
            # roughconductor.cpp:207-221, roughdielectric.cpp:224-243: alpha sets both directions
            if name != "alphaV":
                m["alpha_u_texture"] = index
            if name != "alphaU":
                m["alpha_v_texture"] = index
        else:
            raise SceneFormatError("a texture for the parameter '%s' of <bsdf type=\"%s\"> is not supported" % (name, node.get("type")))


def _make_bsdf(node, data_dir: Optional[str], add_material=None, named: Optional[Dict] = None, ctx=None) -> Dict:
    t = node.get("type")
    p = _props(node)
    if t == "twosided":     # src/bsdfs/twosided.cpp:84-106,185-196: one or two nested reflection-only BRDFs
        nested = []
        for c in node:
            if c.tag == "bsdf":
                if c.get("type") == "twosided":
                    raise SceneFormatError("a <bsdf type=\"twosided\"> nested in another one is not supported")
                nested.append(_make_bsdf(c, data_dir, add_material, named, ctx))
            elif c.tag == "ref":
                if named is None or c.get("id") not in named:
                    raise SceneFormatError("reference to unknown bsdf '%s'" % c.get("id"))
                nested.append(dict(named[c.get("id")]))
        if not nested:
            raise SceneFormatError("A nested one-sided material is required!")
        if len(nested) > 2:
            raise SceneFormatError("No more than two nested BRDFs can be added!")
        for m in nested:
            if m["type"] in (S.BSDF_DIELECTRIC, S.BSDF_ROUGHDIELECTRIC):
                raise SceneFormatError("Only materials without a transmission component can be nested!")
        back = -1
        if len(nested) == 2:
            if add_material is None:
                raise SceneFormatError("twosided with two nested BRDFs needs a material list to add the second one to")
            back = add_material(nested[1])
        return S.twosided(nested[0], back)
    if t in ("bumpmap", "normalmap"):   # src/bsdfs/bumpmap.cpp:116-133, src/bsdfs/normalmap.cpp:88-105
        nested = tex = None
        for c in node:
            is_texture_ref = c.tag == "ref" and ctx is not None and c.get("id") in ctx["named_textures"]
            if c.tag == "texture" or is_texture_ref:
                if tex is not None:
                    raise SceneFormatError("Only a single %s texture can be specified!" % ("displacement" if t == "bumpmap" else "normal"))
                tex = c
            elif c.tag in ("bsdf", "ref"):
                if nested is not None:
                    raise SceneFormatError("Only a single nested BSDF can be added!")
                nested = c
        if nested is None:
            raise SceneFormatError("A child BSDF instance is required")
        if tex is None or ctx is None:
            raise SceneFormatError("A displacement texture must be specified" if t == "bumpmap" else "A normal map texture must be specified")
        if tex.tag == "texture" and tex.get("type") == "bitmap" and not any(c.get("name") == "gamma" for c in tex):
            raise SceneFormatError("When using a bitmap texture as a bump map, please explicitly specify the 'gamma' parameter of the bitmap plugin. "
                                   "In most cases the following is the correct choice: <float name=\"gamma\" value=\"1.0\"/>")
        if nested.tag == "ref":
            if named is None or nested.get("id") not in named:
                raise SceneFormatError("reference to unknown bsdf '%s'" % nested.get("id"))
            rec = dict(named[nested.get("id")])
        else:
            rec = _make_bsdf(nested, data_dir, add_material, named, ctx)
        if rec["type"] == S.BSDF_MIXTURE or rec["two_sided"] or rec.get("frame_kind"):
            raise SceneFormatError("%s: the nested BSDF must be a plain model (not a mixturebsdf, twosided, bumpmap or normalmap)" % t)
        index = ctx["named_textures"][tex.get("id")] if tex.tag == "ref" else _make_texture(tex, ctx)
        return S.bumpmap(rec, index) if t == "bumpmap" else S.normalmap(rec, index)
    if t == "mixturebsdf":  # src/bsdfs/mixturebsdf.cpp:66-85,279-287
        try:
            weights = [float(w) for w in p.get("weights", "").replace(",", " ").split()]
        except ValueError:
            raise SceneFormatError("Could not parse the BSDF weights!")
        if any(w < 0 for w in weights):
            raise SceneFormatError("Invalid BSDF weight!")
        if add_material is None:
            raise SceneFormatError("mixturebsdf needs a material list to add the nested BSDFs to")
        children = []

        def nest(rec):
            if rec["two_sided"] or rec["type"] == S.BSDF_MIXTURE:
                raise SceneFormatError("mixturebsdf: twosided and mixturebsdf cannot be nested")
            if rec.get("frame_kind"):
                raise SceneFormatError("mixturebsdf: bumpmap and normalmap cannot be nested")
            children.append(add_material(rec))

        for c in node:
            if c.tag == "bsdf":
                nest(_make_bsdf(c, data_dir, add_material, named, ctx))
            elif c.tag == "ref":
                if named is None or c.get("id") not in named:
                    raise SceneFormatError("reference to unknown bsdf '%s'" % c.get("id"))
                nest(dict(named[c.get("id")]))
        try:
            return S.mixture(children, weights, p.get("ensureEnergyConservation", True))
        except ValueError as e:
            raise SceneFormatError(str(e))
    for c in node:
        if c.tag == "bsdf":
            raise SceneFormatError("nested <%s> inside a <bsdf type=\"%s\"> is not supported" % (c.tag, t))
    ee = p.get("ensureEnergyConservation", True)
    distr = p.get("distribution", "beckmann").lower()
    if "alpha" in p:
        au = av = p["alpha"]
    else:
        au, av = p.get("alphaU", 0.1), p.get("alphaV", 0.1)
    if t == "diffuse":
        m = S.diffuse(p.get("reflectance", p.get("diffuseReflectance", (0.5, 0.5, 0.5))))
    elif t == "dielectric":
        m = S.dielectric(_ior(p, "intIOR", "bk7"), _ior(p, "extIOR", "air"), p.get("specularReflectance", (1, 1, 1)),
                         p.get("specularTransmittance", (1, 1, 1)))
    elif t == "ward":
        a = p.get("alpha", 0.1)
        m = S.ward(alpha=p.get("alphaU", a), alpha_v=p.get("alphaV", a),
                   diffuse_reflectance=p.get("diffuseReflectance", (0.5, 0.5, 0.5)),
                   specular_reflectance=p.get("specularReflectance", (0.2, 0.2, 0.2)), variant=p.get("variant", "balanced").lower())
    elif t == "roughconductor":
        if distr in ("phong", "as"):
            raise SceneFormatError("the Phong/Ashikhmin-Shirley microfacet distribution is not supported")
        eta, k = _conductor_constants(p, data_dir)
        m = S.roughconductor(eta, k, alpha=au, alpha_v=av, distribution=distr,
                             specular_reflectance=p.get("specularReflectance", (1, 1, 1)), ext_eta=_ior(p, "extEta", "air"),
                             sample_visible=p.get("sampleVisible", True))
    elif t == "roughplastic":
        if distr in ("phong", "as"):
            raise SceneFormatError("the Phong microfacet distribution is not supported")
        int_ior, ext_ior = _ior(p, "intIOR", "polypropylene"), _ior(p, "extIOR", "air")
        rt = None
        if data_dir is not None:
            from . import capi
            dat = open(os.path.join(data_dir, "microfacet", distr + ".dat"), "rb").read()
            rt = capi.rtrans_reduce(dat, int_ior / ext_ior, au)
        m = S.roughplastic(alpha=au, distribution=distr, int_ior=int_ior, ext_ior=ext_ior,
                           diffuse_reflectance=p.get("diffuseReflectance", (0.5, 0.5, 0.5)),
                           specular_reflectance=p.get("specularReflectance", (1, 1, 1)), nonlinear=p.get("nonlinear", False),
                           sample_visible=p.get("sampleVisible", True), rtrans=rt)
    elif t == "roughdielectric":
        if distr in ("phong", "as"):
            raise SceneFormatError("the Phong/Ashikhmin-Shirley microfacet distribution is not supported")
        m = S.roughdielectric(alpha=au, alpha_v=av, distribution=distr, int_ior=_ior(p, "intIOR", "bk7"), ext_ior=_ior(p, "extIOR", "air"),
                              specular_reflectance=p.get("specularReflectance", (1, 1, 1)),
                              specular_transmittance=p.get("specularTransmittance", (1, 1, 1)), sample_visible=p.get("sampleVisible", True))
    elif t == "conductor":
        eta, k = _conductor_constants(p, data_dir)
        m = S.conductor(eta, k, specular_reflectance=p.get("specularReflectance", (1, 1, 1)), ext_eta=_ior(p, "extEta", "air"))
    elif t == "plastic":
        m = S.plastic(int_ior=_ior(p, "intIOR", "polypropylene"), ext_ior=_ior(p, "extIOR", "air"),
                      diffuse_reflectance=p.get("diffuseReflectance", (0.5, 0.5, 0.5)),
                      specular_reflectance=p.get("specularReflectance", (1, 1, 1)), nonlinear=p.get("nonlinear", False))
    else:
        raise SceneFormatError("BSDF type '%s' is outside the path hot path of this core" % t)
    m["ensure_energy_conservation"] = int(bool(ee))
    _attach_textures(node, m, ctx)
    return m


def load_xml(path: str, width: Optional[int] = None, height: Optional[int] = None, integrator_override: Optional[Dict] = None,
             data_dir: Optional[str] = None, mesh_provider=None) -> "tuple[S.SceneData, Dict]":
    """Parse a scene file into SceneData. Returns (scene, info) where info carries the sampler's sampleCount
    and the integrator type found in the file. `mesh_provider(filename, shape_index)` may supply mesh dicts
    (positions/normals/texcoords/indices in object space) instead of reading `.serialized` files."""
    root = ET.parse(path).getroot()
    if root.tag != "scene":
        raise SceneFormatError("root element must be <scene>")
    base = os.path.dirname(os.path.abspath(path))
    _spectrum_dir[0] = base
    b = S.SceneBuilder()
    named_bsdfs: Dict[str, int] = {}
    info = dict(integrator_type=None, sample_count=4)
    camera = film = None
    integ = dict(max_depth=-1, rr_depth=5, strict_normals=0, hide_emitters=0)

    named_records: Dict[str, Dict] = {}

    ctx = dict(base=base, textures=b.textures, named_textures={}, data_dir=data_dir)
    for el in root:         # textures first: BSDFs refer to them by id
        if el.tag == "texture":
            idx = _make_texture(el, ctx)
            if el.get("id"):
                ctx["named_textures"][el.get("id")] = idx

    def bsdf_index(node) -> int:
        rec = _make_bsdf(node, data_dir, b.add_material, named_records, ctx)
        idx = b.add_material(rec)
        if node.get("id"):
            named_bsdfs[node.get("id")] = idx
            named_records[node.get("id")] = rec
        return idx

    default_bsdf = None
    for el in root:
        if el.tag == "bsdf":
            bsdf_index(el)
    for el in root:
        if el.tag == "integrator":
            p = _props(el)
            info["integrator_type"] = el.get("type")
            integ.update(max_depth=p.get("maxDepth", -1), rr_depth=p.get("rrDepth", 5),
                         strict_normals=int(p.get("strictNormals", False)), hide_emitters=int(p.get("hideEmitters", False)))
            if el.get("type") == "direct":      # src/integrators/direct/direct.cpp:93-107
                shading = p.get("shadingSamples", 1)
                integ.update(type="direct", emitter_samples=p.get("emitterSamples", shading), bsdf_samples=p.get("bsdfSamples", shading))
            elif el.get("type") == "volpath_simple":    # media are not loaded: the surface-only form of it
                integ.update(type="volpath_simple")
        elif el.tag == "sensor":
            if el.get("type") != "perspective":
                raise SceneFormatError("only the perspective sensor is supported")
            p = _props(el)
            fw, fh, rf, crop = 768, 576, None, None      # Film defaults (src/librender/film.cpp:25-55)
            for c in el:
                if c.tag == "film":
                    fp = _props(c)
                    fw, fh = fp.get("width", 768), fp.get("height", 576)
                    # crop window in pixels, by default the whole film (src/librender/film.cpp:35-48)
                    crop = (fp.get("cropOffsetX", 0), fp.get("cropOffsetY", 0), fp.get("cropWidth", fw), fp.get("cropHeight", fh))
                    if crop[0] < 0 or crop[1] < 0 or crop[2] <= 0 or crop[3] <= 0 or crop[0] + crop[2] > fw or crop[1] + crop[3] > fh:
                        raise SceneFormatError("Invalid crop window specification!")
                    for r in c:
                        if r.tag == "rfilter":
                            rp = _props(r)
                            if r.get("type") == "gaussian":
                                rf = S.gaussian_filter(rp.get("stddev", 0.5))
                            elif r.get("type") == "box":
                                rf = S.box_filter(rp.get("radius", 0.5))
                            elif r.get("type") == "tent":
                                rf = S.tent_filter()
                            elif r.get("type") == "mitchell":
                                rf = S.mitchell_filter(rp.get("B", 1.0 / 3.0), rp.get("C", 1.0 / 3.0))
                            elif r.get("type") == "catmullrom":
                                rf = S.catmullrom_filter()
                            elif r.get("type") == "lanczos":
                                rf = S.lanczos_filter(rp.get("lobes", 3))
                            else:
                                raise SceneFormatError("reconstruction filter '%s' is not supported" % r.get("type"))
                elif c.tag == "sampler":
                    info["sample_count"] = _props(c).get("sampleCount", 4)
            if width is not None:
                fw, fh, crop = width, height, None
            aspect = fw / fh
            # PerspectiveCamera::configure (src/librender/sensor.cpp:239-278): fov along an axis, or a focal length on
            # 35 mm film (diagonal field of view); setYFov / setDiagonalFov (sensor.cpp:296-305)
            if "fov" in p and "focalLength" in p:
                raise SceneFormatError("Please specify either a focal length ('focalLength') or a field of view ('fov')!")
            fov = p.get("fov")
            axis = p.get("fovAxis", "x").lower()
            if fov is None:
                fl = p.get("focalLength", "50mm")
                try:
                    value = float(fl[:-2] if fl.endswith("mm") else fl)
                except ValueError:
                    raise SceneFormatError("Could not parse the focal length (must be of the form <x>mm, where <x> is a positive integer)!")
                fov = math.degrees(2 * math.atan(math.sqrt(36 * 36 + 24 * 24) / (2 * value)))
                axis = "diagonal"
            if axis == "smaller":
                axis = "y" if aspect > 1 else "x"
            elif axis == "larger":
                axis = "x" if aspect > 1 else "y"
            if axis == "y":
                fov = math.degrees(2 * math.atan(math.tan(0.5 * math.radians(fov)) * aspect))
            elif axis == "diagonal":
                diagonal = 2 * math.tan(0.5 * math.radians(fov))
                fov = math.degrees(2 * math.atan(diagonal / math.sqrt(1.0 + 1.0 / (aspect * aspect)) * 0.5))
            elif axis != "x":
                raise SceneFormatError("The 'fovAxis' parameter must be set to one of 'smaller', 'larger', 'diagonal', 'x', or 'y'!")
            if not 0 < fov < 180:
                raise SceneFormatError("The horizontal field of view must be in the interval (0, 180)!")
            camera = dict(to_world=p.get("toWorld", np.eye(4, dtype=np.float32)), x_fov=float(fov),
                          near_clip=float(p.get("nearClip", 1e-2)), far_clip=float(p.get("farClip", 1e4)))
            film = S.make_film(fw, fh, rf, crop)
        elif el.tag == "shape":
            p = _props(el)
            mat = None
            radiance, weight = None, 1.0
            for c in el:
                if c.tag == "ref":
                    if c.get("id") not in named_bsdfs:
                        raise SceneFormatError("reference to unknown id '%s'" % c.get("id"))
                    mat = named_bsdfs[c.get("id")]
                elif c.tag == "bsdf":
                    mat = bsdf_index(c)
                elif c.tag == "emitter":
                    if c.get("type") != "area":
                        raise SceneFormatError("only area emitters are supported")
                    ep = _props(c)
                    radiance, weight = ep.get("radiance", (1, 1, 1)), ep.get("samplingWeight", 1.0)
                elif c.tag in ("medium", "subsurface"):
                    raise SceneFormatError("<%s> is not handled by the path integrator" % c.tag)
            if mat is None:
                if default_bsdf is None:
                    default_bsdf = b.add_material(S.diffuse())      # Shape::configure attaches a default diffuse BSDF
                mat = default_bsdf
            t = el.get("type")
            if t == "serialized":
                idx = p.get("shapeIndex", 0)
                if mesh_provider is not None:
                    mesh = mesh_provider(p["filename"], idx)
                else:
                    mesh = load_serialized(os.path.join(base, p["filename"]), idx)
            else:
                raise SceneFormatError("shape type '%s' is not supported (triangle meshes only)" % t)
            if p.get("faceNormals", False) or mesh.get("face_normals"):
                mesh = dict(mesh, normals=None)
            if p.get("flipNormals", False):
                raise SceneFormatError("flipNormals is not supported")
            mesh = apply_to_world(mesh, p.get("toWorld", np.eye(4, dtype=np.float32)))
            if "maxSmoothAngle" in p:       # serialized.cpp:204-209
                if p.get("faceNormals", False) or mesh.get("face_normals"):
                    raise SceneFormatError("The properties 'maxSmoothAngle' and 'faceNormals' can't be specified at the same time!")
                mesh = rebuild_topology(mesh, p["maxSmoothAngle"])
            normals = mesh.get("normals")
            if normals is None and not (p.get("faceNormals", False) or mesh.get("face_normals")):
                normals = S.vertex_normals(mesh["positions"], mesh["indices"].astype(np.int64))
            b.add_mesh(mesh["positions"], mesh["indices"], mat, normals=normals, texcoords=mesh.get("texcoords"),
                       radiance=radiance, sampling_weight=weight)
        elif el.tag == "emitter":
            t = el.get("type")
            p = _props(el)
            weight = p.get("samplingWeight", 1.0)
            if t == "constant":         # src/emitters/constant.cpp:47-50
                b.add_constant_emitter(p.get("radiance", (1.0, 1.0, 1.0)), weight)
            elif t == "envmap":         # src/emitters/envmap.cpp:105-189
                if "filename" not in p:
                    raise SceneFormatError("envmap: the 'filename' parameter is missing")
                if "intensityScale" in p:
                    raise SceneFormatError("The 'intensityScale' parameter has been deprecated and is now called scale.")
                from . import capi
                try:
                    pixels = capi.image_load(os.path.join(base, p["filename"]), gamma=p.get("gamma", 0.0))
                except capi.MtsamdError as e:
                    raise SceneFormatError(str(e))
                b.add_envmap_emitter(pixels, p.get("toWorld"), p.get("scale", 1.0), weight)
            elif t == "point":          # src/emitters/point.cpp:58-69
                m = np.array(p.get("toWorld", np.eye(4)), np.float32)
                if "position" in p:
                    if "toWorld" in p:
                        raise SceneFormatError("Only one of the parameters 'position' and 'toWorld' can be used!")
                    m[:3, 3] = p["position"]
                b._add_delta_emitter(S.EMITTER_POINT, p.get("intensity", (1.0, 1.0, 1.0)), m, weight)
            elif t == "spot":           # src/emitters/spot.cpp:67-78
                if any(c.tag == "texture" for c in el):
                    raise SceneFormatError("spot emitter: only a constant texture is supported")
                value = np.float32(p.get("intensity", (1.0, 1.0, 1.0))) * np.float32(p.get("texture", (1.0, 1.0, 1.0)))
                b.add_spot_emitter(p.get("toWorld", np.eye(4, dtype=np.float32)), value, p.get("cutoffAngle", 20.0), p.get("beamWidth"), weight)
            elif t == "directional":    # src/emitters/directional.cpp:56-73
                if "direction" in p and "toWorld" in p:
                    raise SceneFormatError("Only one of the parameters 'direction' and 'toWorld' can be used at a time!")
                b.add_directional_emitter(direction=p.get("direction"), irradiance=p.get("irradiance", (1.0, 1.0, 1.0)),
                                          to_world=None if "direction" in p else p.get("toWorld", np.eye(4, dtype=np.float32)),
                                          sampling_weight=weight)
            else:
                raise SceneFormatError("emitter type '%s' is not supported (area, constant, envmap, point, spot, directional)" % t)
    if camera is None:
        raise SceneFormatError("scene has no sensor")
    if integrator_override:
        integ.update(integrator_override)
    return b.finish(camera, film, integ), info
