"""The C++ host (mtsamd-render): scene-file ingest checked against the Python loader and against known geometry on CPU,
and one end-to-end render on the GPU compared with the same scene rendered through the ctypes binding."""
import json
import os
import re
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import ROOT
from mitsuba_amd import loader

CLI = os.path.join(ROOT, "mitsuba_amd", "mtsamd-render")

SCENE = """<?xml version="1.0" encoding="utf-8"?>
<!-- test scene -->
<scene version="0.5.0">
  <integrator type="path"><integer name="maxDepth" value="5"/><integer name="rrDepth" value="3"/></integrator>
  <sensor type="perspective">
    <float name="fov" value="45"/>
    <transform name="toWorld"><lookat origin="0, -6, 2" target="0, 0, 0.5" up="0, 0, 1"/></transform>
    <sampler type="independent"><integer name="sampleCount" value="8"/></sampler>
    <film type="hdrfilm"><integer name="width" value="64"/><integer name="height" value="48"/>
      <rfilter type="gaussian"/></film>
  </sensor>
  <bsdf type="diffuse" id="grey"><rgb name="reflectance" value="0.6, 0.6, 0.6"/></bsdf>
  <bsdf type="roughconductor" id="metal"><string name="distribution" value="ggx"/><float name="alpha" value="0.2"/>
    <rgb name="eta" value="0.2, 0.92, 1.1"/><rgb name="k" value="3.9, 2.45, 2.14"/></bsdf>
  <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="0"/><ref id="grey"/></shape>
  <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="1"/><ref id="metal"/></shape>
  <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="2"/>
    <emitter type="area"><rgb name="radiance" value="30, 28, 25"/></emitter></shape>
  %s
</scene>
"""


def _quad(z, size, flip=False):
    p = np.array([[-size, -size, z], [size, -size, z], [size, size, z], [-size, size, z]], np.float32)
    idx = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    if flip:
        idx = idx[:, ::-1].copy()
    return p, idx


def _write_serialized(path):
    """Three meshes in a v4 container laid out as TriMesh::serialize writes it (src/librender/trimesh.cpp:1138-1200)."""
    meshes = []
    p, idx = _quad(0.0, 4.0)                                    # floor, no normals -> computed
    meshes.append((p, None, idx))
    rng = np.random.default_rng(3)                              # a bumpy blob with its own normals
    from mitsuba_amd import scene as S
    bp, bf, bn = S.blob(2, rng, radius=1.0)
    meshes.append(((bp + np.float32([0, 0, 1.2])).astype(np.float32), bn, bf.astype(np.uint32)))
    p, idx = _quad(4.0, 1.0, flip=True)                         # light facing down
    meshes.append((p, None, idx))
    _serialize(path, meshes)


def _serialize(path, meshes):
    blob, offsets = b"", []
    for k, (p, n, idx) in enumerate(meshes):
        flags = 0x1000 | (0x1 if n is not None else 0)
        body = struct.pack("<I", flags) + b"mesh%d\0" % k + struct.pack("<QQ", len(p), len(idx)) + p.tobytes()
        if n is not None:
            body += n.astype(np.float32).tobytes()
        body += idx.tobytes()
        offsets.append(len(blob))
        blob += struct.pack("<HH", 0x041C, 4) + zlib.compress(body)
    blob += b"".join(struct.pack("<Q", o) for o in offsets) + struct.pack("<I", len(meshes))
    open(path, "wb").write(blob)


@pytest.fixture(scope="module")
def scene_dir(tmp_path_factory):
    if not os.path.exists(CLI):
        pytest.skip("mtsamd-render has not been built")
    d = tmp_path_factory.mktemp("cli")
    _write_serialized(str(d / "meshes.serialized"))
    (d / "scene.xml").write_text(SCENE % "")
    (d / "tri.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3 4/4\n")
    extra = ('<shape type="obj"><string name="filename" value="tri.obj"/><ref id="grey"/></shape>'
             '<shape type="cube"><transform name="toWorld"><scale value="0.5"/><translate x="2" z="0.5"/></transform><ref id="metal"/></shape>')
    (d / "scene_extra.xml").write_text(SCENE % extra)
    return d


def _describe(path, *args):
    out = subprocess.run([CLI, str(path), "--describe", *args], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout)


def test_cli_loader_matches_python_loader(scene_dir):
    d = _describe(scene_dir / "scene.xml")
    sd, info = loader.load_xml(str(scene_dir / "scene.xml"))
    assert d["triangles"] == sd.n_triangles and d["vertices"] == sd.positions.shape[0]
    assert d["shapes"] == len(sd.shapes) and d["emitters"] == len(sd.emitters) == 1
    assert d["material_types"] == [m["type"] for m in sd.materials]
    assert d["sum_indices"] == int(sd.indices.astype(np.uint64).sum())
    assert abs(d["sum_positions"] - float(sd.positions.astype(np.float64).sum())) < 1e-4
    assert abs(d["sum_normals"] - float(sd.normals.astype(np.float64).sum())) < 1e-3     # computed normals: float32 vs float64 accumulation
    assert np.allclose(d["to_world"], sd.camera["to_world"].ravel(), atol=1e-6)
    assert (d["width"], d["height"], d["max_depth"], d["rr_depth"], d["sample_count"]) == (64, 48, 5, 3, info["sample_count"])
    assert _describe(scene_dir / "scene.xml", "--size", "32", "16")["width"] == 32


def test_cli_obj_and_cube(scene_dir):
    base = _describe(scene_dir / "scene.xml")
    d = _describe(scene_dir / "scene_extra.xml")
    # the OBJ quad becomes a fan of two triangles on four vertices; the cube has 24 vertices and 12 triangles
    assert d["shape_triangles"][-2:] == [2, 12]
    assert d["vertices"] - base["vertices"] == 4 + 24
    assert d["has_texcoords"] is True
    # cube scaled by 0.5 and moved to (2, 0, 0.5): its 24 positions sum to 24 * (2, 0, 0.5); the quad's sum to (2, 2, 0)
    assert abs(d["sum_positions"] - base["sum_positions"] - (24 * 2.5 + 4.0)) < 1e-4
    # v is flipped by default (obj.cpp flipTexCoords): quad uv sum 4 stays 4, cube uv sum is 6 faces * 4 = 24
    assert abs(d["sum_texcoords"] - (4.0 + 24.0)) < 1e-5


def test_cli_smooth_bsdfs_and_twosided(scene_dir, tmp_path):
    """conductor / plastic / twosided (one and two nested BRDFs, inline and by reference) through both loaders."""
    body = """
      <bsdf type="plastic" id="pl"><float name="intIOR" value="1.6"/><rgb name="diffuseReflectance" value="0.1, 0.2, 0.3"/>
        <boolean name="nonlinear" value="true"/></bsdf>
      <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="0"/><bsdf type="conductor"><rgb name="eta" value="0.2, 0.9, 1.1"/><rgb name="k" value="3.9, 2.4, 2.1"/></bsdf></shape>
      <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="0"/><ref id="pl"/></shape>
      <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="0"/><bsdf type="twosided"><bsdf type="diffuse"/></bsdf></shape>
      <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="0"/><bsdf type="twosided"><ref id="metal"/><ref id="pl"/></bsdf></shape>"""
    f = tmp_path / "smooth.xml"
    f.write_text(SCENE % body)
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    d = _describe(f)
    sd, _ = loader.load_xml(str(f))
    assert d["material_types"] == [m["type"] for m in sd.materials]
    assert d["two_sided"] == [m["two_sided"] for m in sd.materials]
    assert d["back_materials"] == [m["back_material"] for m in sd.materials]
    from mitsuba_amd import scene as S
    assert S.BSDF_CONDUCTOR in d["material_types"] and S.BSDF_PLASTIC in d["material_types"]
    two = [i for i, t in enumerate(d["two_sided"]) if t]
    assert len(two) == 2 and d["back_materials"][two[0]] == -1
    back = d["back_materials"][two[1]]
    assert d["material_types"][back] == S.BSDF_PLASTIC and d["material_types"][two[1]] == S.BSDF_ROUGHCONDUCTOR
    assert sd.materials[back]["variant"] == 1 and abs(sd.materials[back]["int_ior"] - 1.6) < 1e-6


def test_cli_conductor_material_presets(scene_dir, tmp_path):
    """material="..." names measured spectra under <data dir>/ior/: both loaders convert them with the library's
    mtsamd_spd_file_to_rgb, explicit eta / k override one at a time, "none" is the perfect mirror."""
    import shutil
    from mitsuba_amd import capi
    shape = '<shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="0"/>%s</shape>'
    body = "".join(shape % b for b in (
        '<bsdf type="conductor"/>',
        '<bsdf type="roughconductor"><string name="material" value="Au"/><float name="alpha" value="0.2"/></bsdf>',
        '<bsdf type="conductor"><string name="material" value="a-C"/><rgb name="k" value="0.5, 0.25, 0.125"/></bsdf>',
        '<bsdf type="roughconductor"><string name="material" value="none"/></bsdf>'))
    f = tmp_path / "presets.xml"
    f.write_text(SCENE % body)
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    data = tmp_path / "data"
    shutil.copytree(os.path.join(os.path.dirname(__file__), "golden", "ior_spd"), data / "ior")
    d = _describe(f, "--data-dir", str(data))
    sd, _ = loader.load_xml(str(f), data_dir=str(data))
    mine = [list(m["eta"]) + list(m["k"]) for m in sd.materials]
    assert np.allclose(d["eta_k"], mine, rtol=0, atol=1e-6)     # explicit values: float in one loader, double in the other
    rgb = lambda name: list(capi.spd_file_to_rgb(str(data / "ior" / name)))
    new = mine[-4:]
    assert np.allclose(new[0], rgb("Cu.eta.spd") + rgb("Cu.k.spd"), atol=1e-7)
    assert np.allclose(new[1], rgb("Au.eta.spd") + rgb("Au.k.spd"), atol=1e-7)
    assert np.allclose(new[2], rgb("a-C.eta.spd") + [0.5, 0.25, 0.125], atol=1e-7)
    assert new[3] == [0, 0, 0, 1, 1, 1]
    # without the directory, or with a material it does not hold, both loaders say what is missing
    out = subprocess.run([CLI, str(f), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "ior/Cu.eta.spd" in out.stderr
    with pytest.raises(loader.SceneFormatError, match="ior/Cu.eta.spd"):
        loader.load_xml(str(f))
    f.write_text(SCENE % (shape % '<bsdf type="conductor"><string name="material" value="Xx"/></bsdf>'))
    out = subprocess.run([CLI, str(f), "--describe", "--data-dir", str(data)], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "Xx.eta.spd" in out.stderr
    with pytest.raises(loader.SceneFormatError, match="Xx.eta.spd"):
        loader.load_xml(str(f), data_dir=str(data))


def _write_ply(path, fmt, pos, nrm, faces):
    """Vertices with normals, a colour and a list property to skip over; an extra element before the faces."""
    head = ["ply", "format %s 1.0" % fmt, "comment test", "element vertex %d" % len(pos)]
    head += ["property float x", "property float y", "property float z", "property uchar red",
             "property float nx", "property float ny", "property float nz", "property list uchar int junk"]
    head += ["element edge 2", "property int a", "property short b"]
    head += ["element face %d" % len(faces), "property uchar flag", "property list uchar uint vertex_indices", "end_header"]
    data = ("\n".join(head) + "\n").encode()
    if fmt == "ascii":
        lines = ["%r %r %r 200 %r %r %r 2 7 8" % (*map(float, p), *map(float, n)) for p, n in zip(pos, nrm)]
        lines += ["1 2", "3 4"]
        lines += ["1 %d %s" % (len(f), " ".join(map(str, f))) for f in faces]
        data += ("\n".join(lines) + "\n").encode()
    else:
        e = "<" if fmt == "binary_little_endian" else ">"
        for p, n in zip(pos, nrm):
            data += struct.pack(e + "fffBfffBii", *map(float, p), 200, *map(float, n), 2, 7, 8)
        data += struct.pack(e + "ihih", 1, 2, 3, 4)
        for f in faces:
            data += struct.pack(e + "BB%dI" % len(f), 1, len(f), *f)
    open(path, "wb").write(data)


def test_cli_ply_parameters_include_alias(scene_dir, tmp_path):
    """Loader fidelity: PLY meshes in all three encodings (quads split like ply.cpp:299-313), $parameters with <default> and
    -D, <include>, <alias>, flipNormals."""
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0.25], [0, 1, 0], [2, 0.5, 0.5]], np.float32)
    nrm = np.array([[0, 0, 1], [0, 0, 1], [0, 0.6, 0.8], [0, 0, 1], [1, 0, 0]], np.float32)
    faces = [[0, 1, 2, 3], [1, 4, 2]]
    (tmp_path / "mats.xml").write_text('<scene version="0.5.0"><bsdf type="diffuse" id="inc"><rgb name="reflectance" value="$refl"/></bsdf></scene>')
    descs = {}
    for fmt in ("ascii", "binary_little_endian", "binary_big_endian"):
        _write_ply(str(tmp_path / (fmt + ".ply")), fmt, pos, nrm, faces)
        body = ('<include filename="mats.xml"/><alias id="inc" as="other"/>'
                '<shape type="ply"><string name="filename" value="%s.ply"/><ref id="other"/>'
                '<transform name="toWorld"><translate x="1" y="2" z="3"/></transform></shape>' % fmt)
        f = tmp_path / (fmt + ".xml")
        # a <default> only reaches what follows it in the file (scenehandler.cpp:684-688)
        f.write_text((SCENE % body).replace('name="sampleCount" value="8"', 'name="sampleCount" value="$spp"')
                     .replace('<scene version="0.5.0">', '<scene version="0.5.0"><default name="spp" value="16"/><default name="refl" value="0.25"/>'))
        descs[fmt] = _describe(f)
    base = _describe(scene_dir / "scene.xml")
    for fmt, d in descs.items():
        assert d["sample_count"] == 16, fmt
        assert d["shape_triangles"][-1] == 3 and d["vertices"] - base["vertices"] == 5
        assert abs(d["sum_positions"] - base["sum_positions"] - float(pos.sum() + 5 * 6)) < 1e-4
        assert abs(d["sum_normals"] - base["sum_normals"] - float(nrm.sum())) < 1e-4
        # quad (0,1,2,3) -> (0,1,2) + (3,0,2); indices are offset by the vertices before this mesh
        assert d["sum_indices"] - base["sum_indices"] == (0 + 1 + 2) + (3 + 0 + 2) + (1 + 4 + 2) + 9 * base["vertices"]
        assert d["materials"] == base["materials"] + 1          # the aliased reference adds no record
    assert descs["ascii"] == descs["binary_little_endian"] == descs["binary_big_endian"]
    f = tmp_path / "ascii.xml"
    assert _describe(f, "-D", "spp=32")["sample_count"] == 32 and _describe(f, "-Dspp=64")["sample_count"] == 64
    # flipNormals negates given normals (trimesh.cpp:624-627)
    g = tmp_path / "flip.xml"
    g.write_text(f.read_text().replace('<ref id="other"/>', '<ref id="other"/><boolean name="flipNormals" value="true"/>'))
    assert abs(_describe(g)["sum_normals"] - base["sum_normals"] + float(nrm.sum())) < 1e-4
    # an undefined parameter and a pentagon are errors, with the reference's wording
    bad = tmp_path / "bad.xml"
    bad.write_text(f.read_text().replace("$spp", "$nope"))
    out = subprocess.run([CLI, str(bad), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "undefined parameter" in out.stderr
    _write_ply(str(tmp_path / "ascii.ply"), "ascii", pos, nrm, [[0, 1, 2, 3, 4]])
    out = subprocess.run([CLI, str(f), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "Only triangle and quad-based PLY meshes" in out.stderr


def test_cli_constant_emitter(scene_dir, tmp_path):
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    f = tmp_path / "sky.xml"
    f.write_text(SCENE % '<emitter type="constant"><rgb name="radiance" value="0.5, 0.75, 1.25"/><float name="samplingWeight" value="2"/></emitter>')
    d = _describe(f)
    sd, _ = loader.load_xml(str(f))
    assert d["emitters"] == len(sd.emitters) == 2
    assert sd.emitters[1]["shape"] == -1 and sd.emitters[1]["radiance"] == (0.5, 0.75, 1.25) and sd.emitters[1]["sampling_weight"] == 2.0


def test_cli_max_smooth_angle(scene_dir, tmp_path):
    """maxSmoothAngle -> TriMesh::rebuildTopology (trimesh.cpp:468-606): a box with eight shared vertices gets a vertex
    per face corner (24) and flat normals; a finely tessellated blob keeps its vertex count; both loaders agree."""
    from mitsuba_amd import scene as S
    corners = np.array([[x, y, z] for z in (0, 1) for y in (0, 1) for x in (0, 1)], np.float32)
    quads = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]
    tris = np.array([t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))], np.uint32)
    smooth = (corners - 0.5) / np.linalg.norm(corners - 0.5, axis=1, keepdims=True)
    bp, bf, bn = S.blob(2, np.random.default_rng(3), radius=1.0)
    _serialize(str(tmp_path / "meshes.serialized"), [(corners, smooth.astype(np.float32), tris), (bp.astype(np.float32), bn, bf.astype(np.uint32)),
                                                      (_quad(4.0, 1.0, flip=True)[0], None, _quad(4.0, 1.0, flip=True)[1])])
    text = (scene_dir / "scene.xml").read_text()
    shape0 = '<integer name="shapeIndex" value="0"/>'
    shape1 = '<integer name="shapeIndex" value="1"/>'
    f = tmp_path / "smooth.xml"
    f.write_text(text.replace(shape0, shape0 + '<float name="maxSmoothAngle" value="30"/>')
                 .replace(shape1, shape1 + '<float name="maxSmoothAngle" value="60"/>'))
    d = _describe(f)
    sd, _ = loader.load_xml(str(f))
    assert d["vertices"] == sd.positions.shape[0] == 24 + bp.shape[0] + 4 and d["triangles"] == sd.n_triangles
    assert d["sum_indices"] == int(sd.indices.astype(np.uint64).sum())
    assert abs(d["sum_positions"] - float(sd.positions.astype(np.float64).sum())) < 1e-4
    assert abs(d["sum_normals"] - float(sd.normals.astype(np.float64).sum())) < 1e-3
    # every box vertex now carries the normal of its face
    box_n, box_p, box_i = sd.normals[:24], sd.positions[:24], sd.indices[:12].astype(np.int64)
    fn = np.cross(box_p[box_i[:, 1]] - box_p[box_i[:, 0]], box_p[box_i[:, 2]] - box_p[box_i[:, 0]])
    fn /= np.linalg.norm(fn, axis=1, keepdims=True)
    for k in range(3):
        assert np.allclose(box_n[box_i[:, k]], fn, atol=1e-6)
    # a wide angle leaves the box smooth: eight vertices again
    f.write_text(text.replace(shape0, shape0 + '<float name="maxSmoothAngle" value="100"/>'))
    assert _describe(f)["vertices"] == loader.load_xml(str(f))[0].positions.shape[0] == 8 + bp.shape[0] + 4
    f.write_text(text.replace(shape0, shape0 + '<float name="maxSmoothAngle" value="30"/><boolean name="faceNormals" value="true"/>'))
    out = subprocess.run([CLI, str(f), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "can't be specified at the same time" in out.stderr


def test_cli_focal_length_and_fov_axes(scene_dir, tmp_path):
    """PerspectiveCamera::configure (sensor.cpp:239-305): focal length on 35 mm film, fov along y / the diagonal."""
    import math
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    text = (scene_dir / "scene.xml").read_text()
    fov45 = '<float name="fov" value="45"/>'
    aspect = 64 / 48
    for xml, want in (('<string name="focalLength" value="50mm"/>',
                       math.degrees(2 * math.atan(math.sqrt(36 ** 2 + 24 ** 2) / 100 / math.sqrt(1 + 1 / aspect ** 2)))),
                      ("", math.degrees(2 * math.atan(math.sqrt(36 ** 2 + 24 ** 2) / 100 / math.sqrt(1 + 1 / aspect ** 2)))),
                      ('<float name="fov" value="45"/><string name="fovAxis" value="diagonal"/>',
                       math.degrees(2 * math.atan(math.tan(math.radians(22.5)) / math.sqrt(1 + 1 / aspect ** 2)))),
                      ('<float name="fov" value="45"/><string name="fovAxis" value="smaller"/>',
                       math.degrees(2 * math.atan(math.tan(math.radians(22.5)) * aspect)))):
        f = tmp_path / "cam.xml"
        f.write_text(text.replace(fov45, xml))
        d = _describe(f)
        sd, _ = loader.load_xml(str(f))
        assert d["x_fov"] == pytest.approx(want, rel=1e-6) and sd.camera["x_fov"] == pytest.approx(want, rel=1e-6)
    f.write_text(text.replace(fov45, fov45 + '<string name="focalLength" value="35mm"/>'))
    out = subprocess.run([CLI, str(f), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "either a focal length" in out.stderr


def test_cli_film_crop_window(scene_dir, tmp_path):
    """cropOffsetX/Y, cropWidth/Height (film.cpp:35-48) reach the film record; an invalid window is refused."""
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    text = (scene_dir / "scene.xml").read_text()
    crop = '<integer name="cropOffsetX" value="8"/><integer name="cropOffsetY" value="4"/><integer name="cropWidth" value="40"/><integer name="cropHeight" value="30"/>'
    f = tmp_path / "crop.xml"
    f.write_text(text.replace('<film type="hdrfilm">', '<film type="hdrfilm">' + crop))
    d = _describe(f)
    sd, _ = loader.load_xml(str(f))
    assert d["crop"] == [8, 4, 40, 30] == [sd.film[k] for k in ("crop_x", "crop_y", "crop_w", "crop_h")]
    assert (d["width"], d["height"]) == (64, 48)
    assert _describe(f, "--size", "32", "16")["crop"] == [0, 0, 32, 16]
    assert _describe(scene_dir / "scene.xml")["crop"] == [0, 0, 64, 48]
    f.write_text(text.replace('<film type="hdrfilm">', '<film type="hdrfilm"><integer name="cropOffsetX" value="40"/><integer name="cropWidth" value="40"/>'))
    out = subprocess.run([CLI, str(f), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "Invalid crop window specification" in out.stderr
    with pytest.raises(loader.SceneFormatError, match="Invalid crop window"):
        loader.load_xml(str(f))


def test_cli_direct_integrator(scene_dir, tmp_path):
    """<integrator type="direct"> (direct.cpp:93-107): shadingSamples sets both counts, the finer ones override it."""
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    text = (scene_dir / "scene.xml").read_text()
    head = text[text.index("<integrator"):text.index("</integrator>") + len("</integrator>")]
    for xml, want in (('<integrator type="direct"/>', (1, 1)),
                      ('<integrator type="direct"><integer name="shadingSamples" value="4"/></integrator>', (4, 4)),
                      ('<integrator type="direct"><integer name="shadingSamples" value="4"/><integer name="bsdfSamples" value="0"/>'
                       '<boolean name="hideEmitters" value="true"/></integrator>', (4, 0))):
        f = tmp_path / "direct.xml"
        f.write_text(text.replace(head, xml))
        d = _describe(f)
        sd, info = loader.load_xml(str(f))
        assert d["integrator"] == info["integrator_type"] == "direct" and sd.integrator["type"] == "direct"
        assert (d["emitter_samples"], d["bsdf_samples"]) == want == (sd.integrator["emitter_samples"], sd.integrator["bsdf_samples"])


def test_cli_volpath_simple_integrator(scene_dir, tmp_path):
    """<integrator type="volpath_simple"> on a scene without media: both loaders hand the estimator over with path's
    parameters; an integrator neither of them knows is named."""
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    text = (scene_dir / "scene.xml").read_text()
    head = text[text.index("<integrator"):text.index("</integrator>") + len("</integrator>")]
    f = tmp_path / "simple.xml"
    f.write_text(text.replace(head, '<integrator type="volpath_simple"><integer name="maxDepth" value="7"/><integer name="rrDepth" value="2"/></integrator>'))
    d = _describe(f)
    sd, info = loader.load_xml(str(f))
    assert d["integrator"] == info["integrator_type"] == "volpath_simple" and d["integrator_kind"] == 2 and sd.integrator["type"] == "volpath_simple"
    assert (d["max_depth"], d["rr_depth"]) == (7, 2) == (sd.integrator["max_depth"], sd.integrator["rr_depth"])
    assert _describe(scene_dir / "scene.xml")["integrator_kind"] == 0


def test_cli_mixturebsdf(scene_dir, tmp_path):
    """mixturebsdf.cpp:66-85,279-287: nested BSDFs are appended before the mixture's own record, by both loaders alike."""
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    f = tmp_path / "blend.xml"
    f.write_text(SCENE % """
      <bsdf type="mixturebsdf" id="blend"><string name="weights" value="0.7, 0.2,0.3"/>
        <bsdf type="diffuse"><rgb name="reflectance" value="0.2, 0.4, 0.6"/></bsdf>
        <ref id="metal"/>
        <bsdf type="ward"><float name="alphaU" value="0.2"/><float name="alphaV" value="0.3"/></bsdf></bsdf>
      <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="1"/><ref id="blend"/></shape>
      <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="1"/><transform name="toWorld"><translate x="3"/></transform>
        <bsdf type="twosided"><bsdf type="mixturebsdf"><string name="weights" value="1"/><boolean name="ensureEnergyConservation" value="false"/>
          <ref id="grey"/></bsdf></bsdf></shape>""")
    d = _describe(f)
    sd, _ = loader.load_xml(str(f))
    assert d["material_types"] == [m["type"] for m in sd.materials]
    assert d["two_sided"] == [m["two_sided"] for m in sd.materials]
    mixes = [(i, m) for i, m in enumerate(sd.materials) if m["type"] == 8]
    assert len(mixes) == len(d["mixtures"]) == 2
    for rec, (i, m) in zip(d["mixtures"], mixes):
        assert rec["material"] == i and rec["children"] == [c for c, _ in m["mix"]]
        assert np.allclose(rec["weights"], [w for _, w in m["mix"]])
    assert d["mixtures"][0]["weights"] == pytest.approx([0.7, 0.2, 0.3]) and [sd.materials[c]["type"] for c in d["mixtures"][0]["children"]] == [0, 1, 4]
    for body, needle in (('<bsdf type="mixturebsdf"><string name="weights" value="0.5"/></bsdf>', "BSDF count mismatch"),
                         ('<bsdf type="mixturebsdf"><string name="weights" value="0.5, x"/><ref id="grey"/><ref id="grey"/></bsdf>', "Could not parse the BSDF weights"),
                         ('<shape type="cube"><bsdf type="mixturebsdf"><string name="weights" value="1"/><bsdf type="twosided"><bsdf type="diffuse"/></bsdf></bsdf></shape>', "cannot be nested")):
        bad = tmp_path / "bad_blend.xml"
        bad.write_text(SCENE % body)
        out = subprocess.run([CLI, str(bad), "--describe"], capture_output=True, text=True, timeout=60)
        assert out.returncode == 1 and needle in out.stderr, out.stderr
        with pytest.raises(loader.SceneFormatError, match=needle):
            loader.load_xml(str(bad))


def test_cli_point_spot_directional_emitters(scene_dir, tmp_path):
    """point.cpp:58-69, spot.cpp:67-78, directional.cpp:56-73: both loaders hand over the same records."""
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    f = tmp_path / "lamps.xml"
    f.write_text(SCENE % """
      <emitter type="point"><point name="position" x="1" y="-2" z="3"/><rgb name="intensity" value="10, 20, 30"/></emitter>
      <emitter type="point"><transform name="toWorld"><translate x="-1" y="0.5" z="2"/></transform></emitter>
      <emitter type="spot"><transform name="toWorld"><lookat origin="2, -3, 4" target="0, 0, 0.5" up="0, 0, 1"/></transform>
        <rgb name="intensity" value="100, 100, 50"/><float name="cutoffAngle" value="30"/><float name="samplingWeight" value="3"/></emitter>
      <emitter type="spot"><float name="beamWidth" value="5"/></emitter>
      <emitter type="directional"><vector name="direction" x="0.2" y="0.3" z="-1"/><spectrum name="irradiance" value="2"/></emitter>
      <emitter type="directional"><transform name="toWorld"><rotate x="1" angle="150"/></transform></emitter>""")
    d = _describe(f)
    sd, _ = loader.load_xml(str(f))
    assert d["emitters"] == len(sd.emitters) == 7
    assert [e["type"] for e in d["emitter_records"]] == [0, 1, 1, 2, 2, 3, 3] == [e.get("type", 0) for e in sd.emitters]
    for a, b in zip(d["emitter_records"][1:], sd.emitters[1:]):
        assert np.allclose(a["radiance"], b["radiance"]) and a["sampling_weight"] == b["sampling_weight"]
        assert np.allclose(a["to_world"], np.asarray(b["to_world"]).ravel(), atol=1e-6)
        if a["type"] == 2:
            assert (a["cutoff_angle"], a["beam_width"]) == (b["cutoff_angle"], b["beam_width"])
    r = d["emitter_records"]
    assert r[1]["to_world"][3::4][:3] == [1, -2, 3] and r[1]["radiance"] == [10, 20, 30] and r[2]["radiance"] == [1, 1, 1]
    assert (r[3]["cutoff_angle"], r[3]["beam_width"], r[3]["sampling_weight"]) == (30, 22.5, 3) and (r[4]["cutoff_angle"], r[4]["beam_width"]) == (20, 5)
    want = np.float32([0.2, 0.3, -1]) / np.linalg.norm(np.float32([0.2, 0.3, -1]))
    assert np.allclose(r[5]["to_world"][2::4][:3], want, atol=1e-6) and r[5]["radiance"] == [2, 2, 2]
    for body, needle in (('<emitter type="point"><point name="position" x="1"/><transform name="toWorld"><translate x="1"/></transform></emitter>', "Only one of the parameters"),
                         ('<emitter type="spot"><float name="cutoffAngle" value="10"/><float name="beamWidth" value="20"/></emitter>', "cutoff angle")):
        bad = tmp_path / "bad_lamp.xml"
        bad.write_text(SCENE % body)
        out = subprocess.run([CLI, str(bad), "--describe"], capture_output=True, text=True, timeout=60)
        assert out.returncode == 1 and needle in out.stderr, out.stderr
        with pytest.raises((loader.SceneFormatError, ValueError), match=needle):
            loader.load_xml(str(bad))


def test_cli_rejects_unsupported_content(scene_dir, tmp_path):
    for body, needle in (('<shape type="sphere"/>', "not supported"),
                         ('<shape type="cube"><bsdf type="coating"/></shape>', "outside the path hot path"),
                         ('<shape type="cube"><bsdf type="twosided"><bsdf type="dielectric"/></bsdf></shape>', "transmission component"),
                         ('<shape type="cube"><bsdf type="twosided"/></shape>', "nested one-sided material is required"),
                         ('<emitter type="sunsky"/>', "emitter type 'sunsky' is not supported"),
                         ('<emitter type="envmap"/>', "'filename' parameter is missing"),
                         ('<shape type="cube"><bsdf type="roughconductor"><string name="material" value="Au"/></bsdf></shape>', "needs the measured spectra")):
        f = tmp_path / "bad.xml"
        f.write_text(SCENE % body)
        (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
        out = subprocess.run([CLI, str(f), "--describe"], capture_output=True, text=True, timeout=60)
        assert out.returncode == 1 and needle in out.stderr, out.stderr
    f = tmp_path / "broken.xml"
    f.write_text("<scene><sensor></scene>")
    out = subprocess.run([CLI, str(f), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "XML parse error" in out.stderr


@pytest.mark.gpu
def test_cli_render_matches_binding(scene_dir, tmp_path):
    from mitsuba_amd import capi
    pfm, raw = tmp_path / "out.pfm", tmp_path / "film.raw"
    out = subprocess.run([CLI, str(scene_dir / "scene.xml"), "-o", str(pfm), "--film", str(raw), "--spp", "8", "--seed", "3"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    stats = json.loads(out.stdout)
    film_cli = np.fromfile(str(raw), np.float32).reshape(48, 64, 5)
    sd, _ = loader.load_xml(str(scene_dir / "scene.xml"))
    film_py, st = capi.Scene(sd).render(8, seed=3)
    assert stats["normal_rays"] == st.normal_rays and stats["shadow_rays"] == st.shadow_rays
    assert np.allclose(film_cli, film_py, rtol=1e-4, atol=1e-5)
    # the same job dealt out over two replicas of the scene (--devices; both on this box's one GPU): same film, same rays
    raw2 = tmp_path / "film2.raw"
    out2 = subprocess.run([CLI, str(scene_dir / "scene.xml"), "-o", str(tmp_path / "out2.pfm"), "--film", str(raw2), "--spp", "8", "--seed", "3",
                           "--devices", "0,0"], capture_output=True, text=True, timeout=300)
    assert out2.returncode == 0, out2.stderr
    stats2 = json.loads(out2.stdout)
    assert stats2["devices"] == 2 and (stats2["normal_rays"], stats2["shadow_rays"]) == (stats["normal_rays"], stats["shadow_rays"])
    assert np.allclose(np.fromfile(str(raw2), np.float32).reshape(48, 64, 5), film_cli, rtol=1e-5, atol=1e-6)
    # the PFM holds the developed image bottom row first
    with open(pfm, "rb") as f:
        assert f.readline() == b"PF\n" and f.readline() == b"64 48\n" and f.readline() == b"-1\n"
        img = np.frombuffer(f.read(), np.float32).reshape(48, 64, 3)[::-1]
    w = film_cli[..., 4:5]
    assert np.allclose(img, np.where(w != 0, film_cli[..., :3] / np.where(w != 0, w, 1), 0), rtol=1e-5, atol=1e-6)
    assert img.mean() > 0.01
    # the same render as OpenEXR: header attributes, offset table and the channel planes of every scan line. HDRFilm's
    # defaults are RGB in half precision (hdrfilm.cpp:213-221); the scene file can ask for other channels and float32.
    text = (scene_dir / "scene.xml").read_text()
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    lum = img[..., 0] * np.float32(0.212671) + img[..., 1] * np.float32(0.715160) + img[..., 2] * np.float32(0.072169)
    alpha = np.where(w[..., 0] != 0, film_cli[..., 3] / np.where(w[..., 0] != 0, w[..., 0], 1), 0)
    for props, want, dtype in (("", dict(R=img[..., 0], G=img[..., 1], B=img[..., 2]), np.float16),
                               ('<string name="componentFormat" value="float32"/><string name="pixelFormat" value="rgba"/>',
                                dict(R=img[..., 0], G=img[..., 1], B=img[..., 2], A=alpha), np.float32),
                               ('<string name="pixelFormat" value="luminanceAlpha"/>', dict(Y=lum, A=alpha), np.float16)):
        f = tmp_path / "hdr.xml"
        f.write_text(text.replace('<film type="hdrfilm">', '<film type="hdrfilm">' + props))
        exr = tmp_path / "out.exr"
        out = subprocess.run([CLI, str(f), "-o", str(exr), "--spp", "8", "--seed", "3"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        planes, got = _read_exr(str(exr))
        assert got == dtype and sorted(planes) == sorted(want)
        for c, ref in want.items():
            if dtype == np.float16:     # numpy rounds to nearest even like the writer: the same bits
                assert np.array_equal(planes[c].view(np.uint16), ref.astype(np.float32).astype(np.float16).view(np.uint16)), c
            else:
                assert np.allclose(planes[c], ref, rtol=1e-6, atol=0), c


@pytest.mark.gpu
def test_cli_direct_render_matches_binding(scene_dir, tmp_path):
    """The standalone caller with <integrator type="direct">: same film and ray counts as the binding on the same scene."""
    from mitsuba_amd import capi
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    text = (scene_dir / "scene.xml").read_text()
    head = text[text.index("<integrator"):text.index("</integrator>") + len("</integrator>")]
    f = tmp_path / "direct.xml"
    f.write_text(text.replace(head, '<integrator type="direct"><integer name="emitterSamples" value="3"/><integer name="bsdfSamples" value="2"/></integrator>'))
    raw = tmp_path / "film.raw"
    out = subprocess.run([CLI, str(f), "-o", str(tmp_path / "out.pfm"), "--film", str(raw), "--spp", "4", "--seed", "5"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    stats = json.loads(out.stdout)
    film_cli = np.fromfile(str(raw), np.float32).reshape(48, 64, 5)
    sd, _ = loader.load_xml(str(f))
    film_py, st = capi.Scene(sd).render(4, seed=5)
    assert stats["normal_rays"] == st.normal_rays and stats["shadow_rays"] == st.shadow_rays
    assert st.shadow_rays > 0 and st.normal_rays > 64 * 48 * 4
    assert np.allclose(film_cli, film_py, rtol=1e-4, atol=1e-5) and film_cli[..., :3].mean() > 0.01


def _read_png(path):
    """Minimal reader for 8-bit, non-interlaced PNG files whose rows use filter type 0."""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, head = 8, b"", None
    while pos < len(b):
        n, = struct.unpack_from(">I", b, pos)
        typ, data = b[pos + 4:pos + 8], b[pos + 8:pos + 8 + n]
        assert struct.unpack_from(">I", b, pos + 8 + n)[0] == zlib.crc32(typ + data)
        if typ == b"IHDR":
            head = struct.unpack(">IIBBBBB", data)
        elif typ == b"IDAT":
            idat += data
        pos += 12 + n
    w, h, depth, colour, comp, flt, interlace = head
    assert (depth, comp, flt, interlace) == (8, 0, 0, 0)
    ch = {0: 1, 4: 2, 2: 3, 6: 4}[colour]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch)
    assert np.all(raw[:, 0] == 0)
    return raw[:, 1:].reshape(h, w, ch)


@pytest.mark.gpu
def test_cli_ldrfilm_png(scene_dir, tmp_path):
    """<film type="ldrfilm"> (src/films/ldrfilm.cpp:131-183) and a .png output name: the tone-mapped 8-bit image."""
    from mitsuba_amd import capi
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    text = (scene_dir / "scene.xml").read_text()
    for film_xml, kw in (('<film type="ldrfilm">', dict()),
                         ('<film type="ldrfilm"><string name="pixelFormat" value="rgba"/><string name="tonemapMethod" value="reinhard"/>'
                          '<float name="key" value="0.3"/><float name="burn" value="0.2"/><float name="gamma" value="2.2"/>',
                          dict(pixel_format="rgba", tonemap="reinhard", key=0.3, burn=0.2, gamma=2.2)),
                         ('<film type="ldrfilm"><string name="pixelFormat" value="luminance"/><float name="exposure" value="1.5"/>',
                          dict(pixel_format="luminance", exposure=1.5))):
        f = tmp_path / "ldr.xml"
        f.write_text(text.replace('<film type="hdrfilm">', film_xml))
        png, raw = tmp_path / "out.png", tmp_path / "film.raw"
        out = subprocess.run([CLI, str(f), "-o", str(png), "--film", str(raw), "--spp", "4", "--seed", "2"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        film = np.fromfile(str(raw), np.float32).reshape(48, 64, 5)
        img = _read_png(str(png))
        assert np.array_equal(img, capi.develop_ldr(film, **kw))
        assert img[..., 0].std() > 5
    f.write_text(text.replace('<film type="hdrfilm">', '<film type="ldrfilm"><string name="fileFormat" value="jpeg"/>'))
    out = subprocess.run([CLI, str(f), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "JPEG" in out.stderr


def _read_exr(path):
    """Minimal reader for uncompressed single-part scan-line OpenEXR files with 16- or 32-bit float channels:
    returns ({channel name: (h, w) array}, numpy dtype of the stored components)."""
    b = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", b, 0)
    assert magic == 20000630 and version & ~0x400 == 2        # 0x400: attribute names may be longer than 31 characters
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); typ = b[pos:e].decode(); pos = e + 1
        size, = struct.unpack_from("<i", b, pos); pos += 4
        attrs[name] = (typ, b[pos:pos + size]); pos += size
    pos += 1
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    chans, kinds, p = [], set(), 0
    cl = attrs["channels"][1]
    while cl[p] != 0:
        e = cl.index(b"\0", p); chans.append(cl[p:e].decode()); p = e + 1
        kinds.add(struct.unpack_from("<i", cl, p)[0])
        p += 16
    assert chans == sorted(chans) and len(kinds) == 1
    dtype = {1: np.float16, 2: np.float32}[kinds.pop()]
    bpc = np.dtype(dtype).itemsize
    offsets = struct.unpack_from("<%dQ" % h, b, pos)
    planes = {c: np.zeros((h, w), dtype) for c in chans}
    for y in range(h):
        yy, size = struct.unpack_from("<ii", b, offsets[y])
        assert yy == y and size == w * len(chans) * bpc
        row = np.frombuffer(b, dtype, len(chans) * w, offsets[y] + 8).reshape(len(chans), w)
        for k, c in enumerate(chans):
            planes[c][y] = row[k]
    return planes, dtype


def _write_png(path, rgb):
    """8-bit RGB, unfiltered rows."""
    h, w, _ = rgb.shape

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    rows = b"".join(b"\x00" + rgb[y].astype(np.uint8).tobytes() for y in range(h))
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(rows)) + chunk(b"IEND", b""))


TEXTURED = """
  <texture type="checkerboard" id="check"><rgb name="color0" value="0.8, 0.7, 0.2"/><rgb name="color1" value="0.1"/>
    <float name="uscale" value="3"/><float name="vscale" value="2"/><float name="uoffset" value="0.25"/></texture>
  <texture type="bitmap" id="wood"><string name="filename" value="wood.png"/><string name="wrapModeU" value="mirror"/>
    <float name="maxAnisotropy" value="8"/><float name="uscale" value="2"/></texture>
  <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="0"/>
    <transform name="toWorld"><translate z="0.01"/></transform>
    <bsdf type="diffuse"><ref name="reflectance" id="wood"/></bsdf></shape>
  <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="1"/>
    <transform name="toWorld"><translate x="2.2"/></transform>
    <bsdf type="plastic"><ref name="diffuseReflectance" id="check"/>
      <texture type="scale" name="specularReflectance"><float name="scale" value="0.5"/>
        <texture type="gridtexture"><float name="lineWidth" value="0.1"/><rgb name="color0" value="1.5"/></texture></texture></bsdf></shape>
  <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="1"/>
    <transform name="toWorld"><translate x="-2.2"/></transform>
    <bsdf type="twosided"><bsdf type="conductor"><string name="material" value="none"/>
      <texture type="bitmap" name="specularReflectance"><string name="filename" value="wood.png"/><string name="filterType" value="trilinear"/>
        <float name="gamma" value="1"/></texture></bsdf></bsdf></shape>
  <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="1"/>
    <transform name="toWorld"><translate y="2.2"/></transform>
    <bsdf type="twosided"><bsdf type="bumpmap"><ref id="metal"/>
      <texture type="scale"><float name="scale" value="0.05"/><texture type="bitmap"><string name="filename" value="wood.png"/>
        <float name="gamma" value="1"/><string name="filterType" value="bilinear"/></texture></texture></bsdf></bsdf></shape>
  <shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="1"/>
    <transform name="toWorld"><translate y="-2.2"/></transform>
    <bsdf type="normalmap"><bsdf type="diffuse"/><ref id="wood"/></bsdf></shape>
"""


def _textured_scene(scene_dir, tmp_path):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (24, 40, 3))
    img[::2] //= 2
    _write_png(str(tmp_path / "wood.png"), img)
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    f = tmp_path / "textured.xml"
    f.write_text(SCENE % TEXTURED)
    return f, img


def test_cli_textures(scene_dir, tmp_path):
    """<texture> elements (checkerboard, gridtexture, bitmap, scale; at the top level with an id, or nested and named
    after the BSDF parameter they stand for): both loaders build the same texture list."""
    f, img = _textured_scene(scene_dir, tmp_path)
    d = _describe(f)
    sd, _ = loader.load_xml(str(f))
    assert len(d["textures"]) == len(sd.textures) == 7
    assert d["frames"] == [[m["frame_kind"], m["frame_texture"]] for m in sd.materials]
    assert sorted(f for f in d["frames"] if f[0]) == [[1, 7], [2, 2]]
    for a, b in zip(d["textures"], sd.textures):
        assert a["type"] == b["type"]
        assert np.allclose(a["uv"], [b["u_offset"], b["v_offset"], b["u_scale"], b["v_scale"]])
        assert np.isclose(a["scale"], b["scale"])
        if b["type"] == 3:
            assert a["size"] == [b["pixels"].shape[1], b["pixels"].shape[0], 3]
            assert (a["filter_type"], a["wrap"], a["max_anisotropy"]) == (b["filter_type"], [b["wrap_u"], b["wrap_v"]], b["max_anisotropy"])
            assert np.isclose(a["sum_pixels"], b["pixels"].astype(np.float64).sum(), rtol=1e-6)
        else:
            assert np.allclose(a["color0"], b["color0"]) and np.allclose(a["color1"], b["color1"]) and np.isclose(a["line_width"], b["line_width"])
    assert d["material_textures"] == [[m["diffuse_texture"], m["specular_texture"], m["transmittance_texture"]] for m in sd.materials]
    used = sorted({i for row in d["material_textures"] for i in row if i})
    assert used == [1, 2, 4, 5]          # texture 3 is the grid inside the scale texture: only its scaled copy is used
    assert sd.textures[6]["scale"] == np.float32(0.05) and sd.textures[6]["type"] == 3      # the bump map: texture 6 scaled
    wood, lin = sd.textures[1], sd.textures[4]
    assert (wood["wrap_u"], wood["wrap_v"], wood["filter_type"], wood["max_anisotropy"]) == (2, 1, 3, 8.0)
    assert np.array_equal(lin["pixels"], img.astype(np.float32) * np.float32(1 / 255.0)) and lin["filter_type"] == 2      # gamma 1: linear
    assert sd.textures[3]["scale"] == 0.5 and sd.textures[3]["type"] == 2 and sd.textures[3]["line_width"] == np.float32(0.1)
    # what is not supported says so
    for body, msg in (('<bsdf type="diffuse"><texture type="wood" name="reflectance"/></bsdf>', "texture type 'wood'"),
                      ('<bsdf type="ward"><texture type="checkerboard" name="alphaU"/></bsdf>', "parameter 'alphaU'"),
                      ('<bsdf type="diffuse"><texture type="bitmap" name="reflectance"><string name="filename" value="nope.png"/></texture></bsdf>', "nope.png"),
                      ('<bsdf type="bumpmap"><bsdf type="diffuse"/><texture type="bitmap"><string name="filename" value="wood.png"/></texture></bsdf>', "'gamma' parameter"),
                      ('<bsdf type="bumpmap"><bsdf type="diffuse"/></bsdf>', "displacement texture must be specified"),
                      ('<bsdf type="normalmap"><texture type="checkerboard"/></bsdf>', "child BSDF instance is required")):
        g = tmp_path / "bad.xml"
        g.write_text(SCENE % body)
        out = subprocess.run([CLI, str(g), "--describe"], capture_output=True, text=True, timeout=60)
        assert out.returncode != 0 and msg in out.stderr, out.stderr
        with pytest.raises(loader.SceneFormatError, match=msg):
            loader.load_xml(str(g))


@pytest.mark.gpu
def test_cli_textured_render_matches_binding(scene_dir, tmp_path):
    from mitsuba_amd import capi
    f, _ = _textured_scene(scene_dir, tmp_path)
    raw = tmp_path / "film.raw"
    out = subprocess.run([CLI, str(f), "-o", str(tmp_path / "out.pfm"), "--film", str(raw), "--spp", "8", "--seed", "3"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    stats = json.loads(out.stdout)
    sd, _ = loader.load_xml(str(f))
    film_py, st = capi.Scene(sd).render(8, seed=3)
    assert stats["normal_rays"] == st.normal_rays and stats["shadow_rays"] == st.shadow_rays
    assert np.allclose(np.fromfile(str(raw), np.float32).reshape(48, 64, 5), film_py, rtol=1e-4, atol=1e-5)


ENVMAP = """
  <emitter type="envmap"><string name="filename" value="sky.pfm"/><float name="scale" value="0.5"/><float name="samplingWeight" value="2"/>
    <transform name="toWorld"><rotate x="1" angle="90"/><rotate z="1" angle="30"/></transform></emitter>
"""


def _envmap_scene(scene_dir, tmp_path):
    from mitsuba_amd import scene as S
    img = S.test_environment_map(32, 16)
    with open(tmp_path / "sky.pfm", "wb") as f:
        f.write(b"PF\n32 16\n-1.0\n")
        f.write(img[::-1].astype("<f4").tobytes())
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    f = tmp_path / "env.xml"
    f.write_text(SCENE % ENVMAP)
    return f, img


def test_cli_envmap_emitter(scene_dir, tmp_path):
    f, img = _envmap_scene(scene_dir, tmp_path)
    d = _describe(f)
    sd, _ = loader.load_xml(str(f))
    env = [e for e in sd.emitters if e.get("pixels") is not None]
    assert len(env) == 1 and len(d["envmaps"]) == 1
    a, b = d["envmaps"][0], env[0]
    assert sd.emitters[a["emitter"]] is b
    assert a["size"] == [32, 16] and np.array_equal(b["pixels"], img)
    assert np.isclose(a["sum_pixels"], img.astype(np.float64).sum(), rtol=1e-6)
    assert (a["scale"], a["weight"]) == (0.5, 2.0) == (b["scale"], b["sampling_weight"])
    assert np.allclose(a["to_world"], np.asarray(b["to_world"]).ravel(), atol=1e-6)
    g = tmp_path / "bad.xml"
    g.write_text(SCENE % '<emitter type="envmap"><string name="filename" value="nothing.exr"/></emitter>')
    out = subprocess.run([CLI, str(g), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "nothing.exr" in out.stderr
    with pytest.raises(loader.SceneFormatError, match="nothing.exr"):
        loader.load_xml(str(g))


@pytest.mark.gpu
def test_cli_envmap_render_matches_binding(scene_dir, tmp_path):
    from mitsuba_amd import capi
    f, _ = _envmap_scene(scene_dir, tmp_path)
    raw = tmp_path / "film.raw"
    out = subprocess.run([CLI, str(f), "-o", str(tmp_path / "out.pfm"), "--film", str(raw), "--spp", "8", "--seed", "3"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    stats = json.loads(out.stdout)
    sd, _ = loader.load_xml(str(f))
    film_py, st = capi.Scene(sd).render(8, seed=3)
    assert stats["normal_rays"] == st.normal_rays and stats["shadow_rays"] == st.shadow_rays
    assert np.allclose(np.fromfile(str(raw), np.float32).reshape(48, 64, 5), film_py, rtol=1e-4, atol=1e-5)


def test_cli_obj_groups_and_material_library(scene_dir, tmp_path):
    """WavefrontOBJ (src/shapes/obj.cpp:224-334,423-573,736-757): one mesh per group and per material change, BSDFs from
    the MTL library (diffuse from Kd as sRGB or from map_Kd, dielectric for illum 4, a bump map), overridden by <bsdf
    name="..."> children; shapeIndex picks one mesh, collapse keeps the file whole."""
    rng = np.random.default_rng(4)
    _write_png(str(tmp_path / "tiles.png"), rng.integers(0, 256, (8, 8, 3)))
    (tmp_path / "parts.mtl").write_text(
        "newmtl red\nKd 0.8 0.2 0.1\nillum 1\n"
        "newmtl glass\nKd 1 1 1\nillum 4\n"
        "newmtl tiled\nKd 1 1 1\nmap_Kd tiles.png\nbump tiles.png\nillum 0\n"
        "newmtl shiny\nKd 0.5 0.5 0.5\nKs 0.5 0.5 0.5\nNs 40\nillum 2\n")
    quad = lambda z: "v -1 -1 %g\nv 1 -1 %g\nv 1 1 %g\nv -1 1 %g\n" % (z, z, z, z)
    (tmp_path / "parts.obj").write_text(
        "mtllib parts.mtl\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n" + quad(0) + quad(1) + quad(2) + quad(3) +
        "g first\nusemtl red\nf 1/1 2/2 3/3 4/4\n"
        "usemtl glass\nf 5/1 6/2 7/3\nf 5/1 7/3 8/4\n"
        "g second\nusemtl tiled\nf 9/1 10/2 11/3 12/4\n"
        "usemtl shiny\nf -4/1 -3/2 -2/3 -1/4\n")
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    shape = '<shape type="obj"><string name="filename" value="parts.obj"/>%s</shape>'
    override = '<bsdf type="roughconductor" name="shiny"><string name="material" value="none"/></bsdf>'
    f = tmp_path / "obj.xml"
    f.write_text(SCENE % (shape % override))
    base = _describe(scene_dir / "scene.xml")
    d = _describe(f)
    n0, m0 = base["shapes"], base["materials"]
    assert d["shapes"] == n0 + 4 and d["shape_triangles"][n0:] == [2, 2, 2, 2]
    mats = d["shape_materials"][n0:]
    types = [d["material_types"][m] for m in mats]
    assert types == [0, 3, 0, 1]            # diffuse, dielectric, diffuse (textured, bumped), the override
    srgb = lambda c: c / 12.92 if c <= 0.04045 else ((c + 0.055) / 1.055) ** 2.4
    assert np.allclose(d["diffuse"][mats[0]], [srgb(0.8), srgb(0.2), srgb(0.1)], rtol=1e-5)
    tex = d["material_textures"][mats[2]]
    assert tex[0] > 0 and d["textures"][tex[0] - 1]["type"] == 3 and d["textures"][tex[0] - 1]["size"] == [8, 8, 3]
    assert d["frames"][mats[2]][0] == 1 and d["textures"][d["frames"][mats[2]][1] - 1]["type"] == 3
    assert d["has_texcoords"]
    # without the override the Phong material is refused by name
    f.write_text(SCENE % (shape % ""))
    out = subprocess.run([CLI, str(f), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "'shiny'" in out.stderr and "Phong" in out.stderr
    # an unnamed BSDF covers every mesh; shapeIndex picks one; collapse makes one mesh of the file
    f.write_text(SCENE % (shape % '<ref id="metal"/>'))
    d = _describe(f)
    assert d["shapes"] == n0 + 4 and len(set(d["shape_materials"][n0:])) == 1 and d["materials"] == m0
    f.write_text(SCENE % (shape % '<integer name="shapeIndex" value="1"/>'))
    d = _describe(f)
    assert d["shapes"] == n0 + 1 and d["material_types"][d["shape_materials"][-1]] == 3
    f.write_text(SCENE % (shape % '<boolean name="collapse" value="true"/><ref id="grey"/>'))
    d = _describe(f)
    assert d["shapes"] == n0 + 1 and d["shape_triangles"][-1] == 8
    f.write_text(SCENE % (shape % (override + '<emitter type="area"/>')))
    out = subprocess.run([CLI, str(f), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "multiple objects" in out.stderr
    f.write_text(SCENE % (shape % '<boolean name="loadMaterials" value="false"/>'))
    d = _describe(f)
    assert [d["material_types"][m] for m in d["shape_materials"][n0:]] == [0, 0, 0, 0]


def test_cli_spectral_values(scene_dir, tmp_path):
    """<spectrum value="l:v, ...">, <spectrum filename=...>, <spectrum value="a b c">, single values with an intent and
    <blackbody> (scenehandler.cpp:534-620): both loaders store what the library's conversions return."""
    from mitsuba_amd import capi
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    (tmp_path / "paint.spd").write_text("# measured\n400 0.1\n500 0.3\n600 0.6\n700 0.4\n")
    body = ('<bsdf type="diffuse"><spectrum name="reflectance" value="400:0.2, 500:0.7 600:0.3,700:0.1"/></bsdf>'
            '<bsdf type="diffuse"><spectrum name="reflectance" filename="paint.spd"/></bsdf>'
            '<bsdf type="diffuse"><spectrum name="reflectance" value="0.1 0.2 0.3"/></bsdf>'
            '<bsdf type="diffuse"><spectrum name="reflectance" value="0.4" intent="Reflectance"/></bsdf>'
            '<shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="1"/>'
            '<emitter type="area"><blackbody name="radiance" temperature="3200K" scale="1e-3"/></emitter></shape>'
            '<emitter type="point"><blackbody name="intensity" temperature=" 6504 "/></emitter>'
            '<emitter type="constant"><spectrum name="radiance" value="0.25"/></emitter>')
    f = tmp_path / "spectral.xml"
    f.write_text(SCENE % body)
    base = _describe(scene_dir / "scene.xml")
    d = _describe(f)
    sd, _ = loader.load_xml(str(f))
    m0, e0 = 2, base["emitters"]        # after the two named BSDFs of the fixture; the shapes' default BSDF comes last
    want = [capi.spd_to_rgb([400, 500, 600, 700], [0.2, 0.7, 0.3, 0.1], scene_value=True),
            capi.spd_file_to_rgb(str(tmp_path / "paint.spd"), scene_value=True), np.float32([0.1, 0.2, 0.3]), np.float32([0.4] * 3)]
    for k, w in enumerate(want):
        assert np.array_equal(np.float32(d["diffuse"][m0 + k]), w), k
        assert np.array_equal(np.float32(sd.materials[m0 + k]["diffuse_reflectance"]), w), k
    assert want[0][1] > want[0][0] > 0 and want[0][1] > want[0][2]       # a spectrum that peaks at 500 nm is green
    lights = [capi.blackbody_to_rgb(3200.0, 1e-3), capi.blackbody_to_rgb(6504.0), np.float32([0.25] * 3)]
    got = [np.float32(e["radiance"]) for e in d["emitter_records"][e0:]]
    assert len(got) == 3
    for k, w in enumerate(lights):
        assert np.array_equal(got[k], w), k
        assert np.array_equal(np.float32(sd.emitters[e0 + k]["radiance"]), w), k
    assert lights[0][0] > lights[0][1] > lights[0][2] > 0
    for body, msg in (('<bsdf type="diffuse"><spectrum name="reflectance" value="500:1"/></bsdf>', "at least 2 entries"),
                      ('<bsdf type="diffuse"><spectrum name="reflectance" value="500:1, 400:1"/></bsdf>', "increasing wavelength"),
                      ('<bsdf type="diffuse"><spectrum name="reflectance" value="500:1:2, 600:1"/></bsdf>', "Invalid spectrum->value mapping"),
                      ('<bsdf type="diffuse"><spectrum name="reflectance" value="1 2"/></bsdf>', "length does not match"),
                      ('<bsdf type="diffuse"><spectrum name="reflectance" value="1" filename="paint.spd"/></bsdf>', "one of 'value' or 'filename'"),
                      ('<bsdf type="diffuse"><spectrum name="reflectance" filename="paint.spd" intent="reflectance"/></bsdf>', "cannot be specified at the same time"),
                      ('<bsdf type="diffuse"><spectrum name="reflectance" value="1" intent="warm"/></bsdf>', 'Invalid intent "warm"'),
                      ('<bsdf type="diffuse"><spectrum name="reflectance" value="400:1, 500:1" intent="reflectance"/></bsdf>', "single-valued argument"),
                      ('<bsdf type="diffuse"><spectrum name="reflectance" filename="missing.spd"/></bsdf>', "missing.spd"),
                      ('<bsdf type="diffuse"><blackbody name="reflectance" temperature="K"/></bsdf>', "needs a temperature")):
        g = tmp_path / "bad.xml"
        g.write_text(SCENE % body)
        out = subprocess.run([CLI, str(g), "--describe"], capture_output=True, text=True, timeout=60)
        assert out.returncode != 0 and msg in out.stderr, (msg, out.stderr)
        with pytest.raises(loader.SceneFormatError, match=re.escape(msg)):
            loader.load_xml(str(g))


def _exr_attributes(path):
    b = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", b, 0)
    assert magic == 20000630
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); typ = b[pos:e].decode(); pos = e + 1
        size, = struct.unpack_from("<i", b, pos); pos += 4
        attrs[name] = (typ, b[pos:pos + size]); pos += size
    return version, attrs


def _png_text(path):
    b = open(path, "rb").read()
    pos, out = 8, {}
    while pos < len(b):
        n, = struct.unpack_from(">I", b, pos)
        typ, data = b[pos + 4:pos + 8], b[pos + 8:pos + 8 + n]
        assert struct.unpack_from(">I", b, pos + 8 + n)[0] == zlib.crc32(typ + data)
        if typ == b"tEXt":
            k, v = data.split(b"\0", 1)
            out[k.decode()] = v.decode()
        pos += 12 + n
    return out


ANNOTATED_FILM = ('<film type="%s"><integer name="width" value="64"/><integer name="height" value="48"/><rfilter type="gaussian"/>'
                  '<string name="metadata[\'note\']" value="a $film[\'width\']x$film[ \'height\' ] $film[\'type\'] of $scene[\'sourceFile\'], '
                  '$sampler[\'sampleCount\'] spp ($sampler[\'type\']), fov $sensor[\'fov\'], depth $integrator[\'maxDepth\'], '
                  'blocks of $scene[\'blockSize\'] on $scene[\'coreCount\'] device(s), written to $scene[\'destFile\']"/>'
                  '<string name=" Metadata[\'time taken\'] " value="$scene[\'renderTime\'] / $scene[\'renderTimePrecise\'] / $scene[\'memUsage\']"/>'
                  '<integer name="metadata[\'revision\']" value="42"/><float name="metadata[\'a name that is longer than thirty-one characters\']" value="0.25"/>'
                  '</film>')


def test_cli_annotations(scene_dir, tmp_path):
    """Render-time annotations (src/films/annotations.h:32-155): metadata['key'] properties of the film are collected
    (white space and the case of the prefix do not matter), labels are refused by name, a bad reference is reported."""
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    text = (scene_dir / "scene.xml").read_text()
    film = text[text.index("<film"):text.index("</film>") + len("</film>")]
    f = tmp_path / "annotated.xml"
    f.write_text(text.replace(film, ANNOTATED_FILM % "hdrfilm"))
    d = _describe(f)
    assert d["annotations"] == [["note", 0], ["timetaken", 0], ["revision", 1], ["anamethatislongerthanthirty-onecharacters", 2]]
    loader.load_xml(str(f))         # the Python loader reads past them
    g = tmp_path / "label.xml"
    g.write_text(text.replace(film, film.replace("</film>", '<string name="label[10, 10]" value="hello"/></film>')))
    out = subprocess.run([CLI, str(g), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "label annotations" in out.stderr


@pytest.mark.gpu
def test_cli_annotations_in_the_output(scene_dir, tmp_path):
    """The annotations arrive in the image files: string / int / float attributes of the OpenEXR header (with the
    long-names bit when a key needs it), tEXt chunks of the PNG; references to film / sampler / sensor / integrator
    properties and to the scene's run-time values are replaced; generatedBy is added."""
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    text = (scene_dir / "scene.xml").read_text()
    film = text[text.index("<film"):text.index("</film>") + len("</film>")]
    for kind, name in (("hdrfilm", "out.exr"), ("ldrfilm", "out.png")):
        f = tmp_path / ("annotated_%s.xml" % kind)
        f.write_text(text.replace(film, ANNOTATED_FILM % kind))
        dest = tmp_path / name
        out = subprocess.run([CLI, str(f), "-o", str(dest), "--spp", "2"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        want_note = ("a 64x48 %s of %s, 8 spp (independent), fov 45, depth 5, blocks of 32 on 1 device(s), written to %s" % (kind, f, dest))
        if kind == "hdrfilm":
            version, attrs = _exr_attributes(str(dest))
            assert version & 0x400
            assert attrs["note"] == ("string", want_note.encode())
            assert attrs["revision"] == ("int", struct.pack("<i", 42))
            assert attrs["anamethatislongerthanthirty-onecharacters"] == ("float", struct.pack("<f", 0.25))
            assert attrs["generatedBy"][0] == "string" and attrs["generatedBy"][1].startswith(b"mtsamd-render")
            taken = attrs["timetaken"][1].decode()
            chans, _ = _read_exr(str(dest))         # still a readable image
            assert chans["R"].shape == (48, 64)
        else:
            t = _png_text(str(dest))
            assert t["note"] == want_note and t["revision"] == "42" and t["anamethatislongerthanthirty-onecharacters"] == "0.25"
            assert t["generatedBy"].startswith("mtsamd-render")
            taken = t["timetaken"]
            assert _read_png(str(dest)).shape == (48, 64, 3)
        assert re.fullmatch(r"\d+\.\ds / \d+\.\d{4}s / \d+\.\d [KMG]iB", taken), taken
    bad = tmp_path / "bad.xml"
    bad.write_text(text.replace(film, film.replace("</film>", '<string name="metadata[\'x\']" value="$film[\'nope\']"/></film>')))
    out = subprocess.run([CLI, str(bad), "-o", str(tmp_path / "bad.exr"), "--spp", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and 'Property "nope" has not been specified!' in out.stderr


def test_cli_textured_roughness(scene_dir, tmp_path):
    """A <texture> named alpha / alphaU / alphaV under roughconductor and roughdielectric (roughconductor.cpp:207-221,
    roughdielectric.cpp:224-243): both loaders fill the same slots; other models refuse it."""
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    body = ('<texture type="checkerboard" id="rough"><rgb name="color0" value="0.05"/><rgb name="color1" value="0.4"/></texture>'
            '<bsdf type="roughconductor"><string name="material" value="none"/><ref name="alpha" id="rough"/></bsdf>'
            '<bsdf type="roughconductor"><string name="material" value="none"/><float name="alpha" value="0.2"/>'
            '<texture type="gridtexture" name="alphaV"><rgb name="color0" value="0.1"/><rgb name="color1" value="0.3"/></texture></bsdf>'
            '<bsdf type="roughdielectric"><ref name="alphaU" id="rough"/><float name="alphaV" value="0.15"/></bsdf>')
    f = tmp_path / "rough.xml"
    f.write_text(SCENE % body)
    d = _describe(f)
    sd, _ = loader.load_xml(str(f))
    assert d["alpha_textures"] == [[m["alpha_u_texture"], m["alpha_v_texture"]] for m in sd.materials]
    assert d["alpha_textures"][2:5] == [[1, 1], [0, 2], [1, 0]]
    assert sd.materials[3]["alpha_u"] == np.float32(0.2) and sd.materials[4]["alpha_v"] == np.float32(0.15)
    g = tmp_path / "bad.xml"
    g.write_text(SCENE % '<bsdf type="ward"><texture type="checkerboard" name="alpha"/></bsdf>')        # only the two models above take one
    out = subprocess.run([CLI, str(g), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "parameter 'alpha'" in out.stderr
    with pytest.raises(loader.SceneFormatError, match="parameter 'alpha'"):
        loader.load_xml(str(g))


def test_cli_textured_roughness_on_roughplastic(scene_dir, tmp_path):
    """A <texture> / <ref> named alpha under roughplastic (roughplastic.cpp:218, 281-299): both loaders keep the
    transmittance table a function of the roughness (mtsamd_rtrans_reduce_eta on <data dir>/microfacet/<distribution>.dat)
    and fill both roughness slots; without the texture the table is the one-dimensional slice."""
    import struct
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    n_eta, n_alpha, n_theta = 4, 5, 6
    tab = np.zeros((2, n_eta, n_alpha, n_theta + 1), np.float32)
    tab[...] = 0.3 + 0.5 * (np.arange(n_theta + 1) / n_theta) ** 0.5
    tab *= (1 - 0.1 * np.arange(n_alpha) / n_alpha)[None, None, :, None]
    data = tmp_path / "data"
    (data / "microfacet").mkdir(parents=True)
    (data / "microfacet" / "ggx.dat").write_bytes(b"MTS_TRANSMITTANCE" + struct.pack("<3Q4f", n_eta, n_alpha, n_theta, 1.0, 3.0, 0.0, 2.0) + tab.tobytes())
    body = ('<texture type="checkerboard" id="rough"><rgb name="color0" value="0.05"/><rgb name="color1" value="0.4"/></texture>'
            '<bsdf type="roughplastic"><string name="distribution" value="ggx"/><ref name="alpha" id="rough"/><float name="intIOR" value="1.5"/><float name="extIOR" value="1.0"/></bsdf>'
            '<bsdf type="roughplastic"><string name="distribution" value="ggx"/><float name="alpha" value="0.2"/><float name="intIOR" value="1.5"/><float name="extIOR" value="1.0"/></bsdf>')
    f = tmp_path / "plastic.xml"
    f.write_text(SCENE % body)
    d = _describe(f, "--data-dir", str(data))
    sd, _ = loader.load_xml(str(f), data_dir=str(data))
    assert d["alpha_textures"] == [[m["alpha_u_texture"], m["alpha_v_texture"]] for m in sd.materials]
    assert d["alpha_textures"][2:4] == [[1, 1], [0, 0]]
    block = 4 + n_alpha * n_theta + 2 * n_alpha
    assert [t[0] for t in d["rtrans_tables"][2:4]] == [block, n_theta]
    for t, m in zip(d["rtrans_tables"][2:4], sd.materials[2:4]):
        assert np.asarray(m["rtrans"]).size == t[0] and abs(float(np.asarray(m["rtrans"], np.float64).sum()) - t[1]) < 1e-4 * abs(t[1])
    assert tuple(np.asarray(sd.materials[2]["rtrans"])[:4]) == (n_theta, n_alpha, 0.0, 2.0)
    # alphaU / alphaV are not parameters of this model
    g = tmp_path / "bad.xml"
    g.write_text(SCENE % '<bsdf type="roughplastic"><string name="distribution" value="ggx"/><texture type="checkerboard" name="alphaU"/></bsdf>')
    out = subprocess.run([CLI, str(g), "--describe", "--data-dir", str(data)], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "parameter 'alphaU'" in out.stderr
    with pytest.raises(loader.SceneFormatError, match="parameter 'alphaU'"):
        loader.load_xml(str(g), data_dir=str(data))


def test_cli_rectangle_shape(scene_dir, tmp_path):
    """<shape type="rectangle"> (src/shapes/rectangle.cpp:80-110,150-165): two triangles over [-1, 1]^2 with uv
    (x + 1) / 2, (y + 1) / 2 under toWorld; it may carry an area emitter; other analytic shapes are still refused."""
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    body = ('<shape type="rectangle"><transform name="toWorld"><scale x="2" y="0.5"/><translate x="1" y="2" z="3"/></transform>'
            '<emitter type="area"><rgb name="radiance" value="5"/></emitter></shape>'
            '<shape type="rectangle"><boolean name="flipNormals" value="true"/><ref id="grey"/></shape>')
    f = tmp_path / "rect.xml"
    f.write_text(SCENE % body)
    base, d = _describe(scene_dir / "scene.xml"), _describe(f)
    assert d["shapes"] == base["shapes"] + 2 and d["shape_triangles"][-2:] == [2, 2] and d["vertices"] == base["vertices"] + 8
    assert d["emitters"] == base["emitters"] + 1 and d["emitter_records"][-1]["radiance"] == [5, 5, 5]
    # first rectangle: corners (+-2, +-0.5, 0) + (1, 2, 3) sum to 4 * 6; the second sums to zero; uv sums are 4 each
    assert abs(d["sum_positions"] - base["sum_positions"] - 24.0) < 1e-4
    assert abs(d["sum_texcoords"] - base.get("sum_texcoords", 0.0) - 8.0) < 1e-5
    g = tmp_path / "sphere.xml"
    g.write_text(SCENE % '<shape type="sphere"/>')
    out = subprocess.run([CLI, str(g), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "shape type 'sphere' is not supported" in out.stderr


def test_cli_damaged_mesh_files_are_refused(scene_dir, tmp_path):
    """What a mesh file claims is checked before it is used: a triangle that names a vertex the mesh does not have, vertex
    or triangle counts beyond the stream's length, and a file name that is a directory are errors, not reads out of bounds
    or gigantic allocations."""
    (tmp_path / "meshes.serialized").write_bytes((scene_dir / "meshes.serialized").read_bytes())
    shape = '<shape type="serialized"><string name="filename" value="%s"/><ref id="grey"/></shape>'
    p, idx = _quad(0.0, 1.0)
    bad = idx.copy()
    bad[1, 2] = 7                                               # four vertices
    _serialize(str(tmp_path / "index.serialized"), [(p, None, bad)])
    f = tmp_path / "damaged.xml"
    f.write_text(SCENE % (shape % "index.serialized"))
    out = subprocess.run([CLI, str(f), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "refers to vertex 7 of 4" in out.stderr, out.stderr
    for nv, nt in ((1 << 62, 2), (4, (1 << 64) // 12 + 1), ((1 << 64) - 1, (1 << 64) - 1)):
        body = struct.pack("<I", 0x1000) + b"mesh0\0" + struct.pack("<QQ", nv, nt) + p.tobytes() + idx.tobytes()
        blob = struct.pack("<HH", 0x041C, 4) + zlib.compress(body) + struct.pack("<Q", 0) + struct.pack("<I", 1)
        (tmp_path / "count.serialized").write_bytes(blob)
        f.write_text(SCENE % (shape % "count.serialized"))
        out = subprocess.run([CLI, str(f), "--describe"], capture_output=True, text=True, timeout=60)
        assert out.returncode != 0 and "truncated mesh stream" in out.stderr, out.stderr
    (tmp_path / "a_directory").mkdir()
    f.write_text(SCENE % (shape % "a_directory"))
    out = subprocess.run([CLI, str(f), "--describe"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "cannot read" in out.stderr, out.stderr
