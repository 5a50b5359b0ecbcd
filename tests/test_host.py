"""CPU tests of the host side: the C-ABI library loads and exports what include/mtsamd.h declares, host-only entry
points work without a GPU, scene builders/loaders produce consistent descriptions, and the multi-GPU partition +
film reduction is exact (world_size-2 gloo run with the oracle standing in for the device renderer)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from mitsuba_amd import capi, loader, render, scene as S


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mtsamd.h")).read()
    declared = set(re.findall(r"\b(mtsamd_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    lib = capi.lib()
    for s in declared:
        assert hasattr(lib, s), s


def test_no_device_is_reported_not_faked():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.MtsamdError):
        capi.Scene(S.small_test_scene())


def test_filter_tables_match_reference_construction():
    r, v = capi.filter_gaussian(0.5)
    r2, v2 = S.gaussian_filter(0.5)
    assert r == r2 == 2.0 and np.array_equal(v, v2)
    assert v[31] == 0 and abs(v[:31].sum() * 2 * r / 31 - 1) < 1e-6      # rfilter.cpp:51-54 normalisation
    rb, vb = capi.filter_box(0.5)
    assert np.array_equal(vb, S.box_filter(0.5)[1]) and rb == S.box_filter(0.5)[0] == float(np.float32(0.5) + np.float32(1e-5))
    # tent.cpp, mitchell.cpp, catmullrom.cpp, lanczos.cpp: radius, unit integral of the table, known values
    for kind, args, py in (("tent", (), S.tent_filter()), ("mitchell", (1 / 3, 1 / 3), S.mitchell_filter()),
                           ("mitchell", (0.2, 0.4), S.mitchell_filter(0.2, 0.4)), ("catmullrom", (), S.catmullrom_filter()),
                           ("lanczos", (3,), S.lanczos_filter(3)), ("lanczos", (2,), S.lanczos_filter(2))):
        r, v = capi.filter_table(kind, *args)
        assert r == py[0] == {"tent": 1.0, "mitchell": 2.0, "catmullrom": 2.0, "lanczos": float(args[0]) if args else 0}[kind]
        assert np.allclose(v, py[1], rtol=2e-6, atol=1e-7) and v[31] == 0
        assert abs(v[:31].sum() * 2 * r / 31 - 1) < 1e-5
    r, v = capi.filter_table("tent")
    assert np.allclose(v[:31] / v[0], 1 - np.arange(31) / 31, atol=1e-6)
    r, v = capi.filter_table("catmullrom")
    assert v[15:31].min() < 0 and abs(v[int(round(31 / 2))]) < 0.02 * v[0]       # negative lobe, zero crossing at x = 1
    assert np.array_equal(capi.filter_table("gaussian", 0.5)[1], capi.filter_gaussian(0.5)[1])


def test_rtrans_reduce_product_vs_oracle_and_packaged(oracle_mod):
    ref = "/root/reference/mitsuba-整合版/data/microfacet/beckmann.dat"
    if not os.path.exists(ref):
        pytest.skip("reference data files are not present on this machine")
    dat = open(ref, "rb").read()
    for eta, alpha in ((1.49 / 1.000277, 0.3), (1.5046 / 1.000277, 0.7)):
        a = capi.rtrans_reduce(dat, eta, alpha)
        b = oracle_mod.rtrans_reduce(dat, eta, alpha)
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2]
        c = S.load_rtrans_slice("beckmann", eta, alpha)
        assert np.array_equal(a[0], c[0])
    with pytest.raises(capi.MtsamdError):
        capi.rtrans_reduce(dat[:100], 1.5, 0.1)
    with pytest.raises(capi.MtsamdError):
        capi.rtrans_reduce(dat, 9.0, 0.1)       # checkEta (rtrans.h:399-407)
    # reduced to the index of refraction only (a roughplastic with a textured roughness): product, oracle, packaged
    eta = 1.49 / 1.000277
    blk = capi.rtrans_reduce_eta(dat, eta)
    assert np.array_equal(blk, oracle_mod.rtrans_reduce_eta(dat, eta)) and np.array_equal(blk, S.load_rtrans_eta_block("beckmann", eta))
    with pytest.raises(capi.MtsamdError):
        capi.rtrans_reduce_eta(dat, 9.0)


def test_rtrans_reduce_eta_on_a_synthetic_table(oracle_mod):
    """mtsamd_rtrans_reduce_eta against the oracle's setEta (rtrans.h:292-345) on a file of the reference's layout
    (rtrans.h:100-148) filled with a smooth function: bit-equal blocks; at an index of refraction that is a node of the
    table the slice reproduces the table's own entries; the reduction to one roughness agrees with interpolating
    the block."""
    import struct
    n_eta, n_alpha, n_theta = 5, 6, 7
    eta_min, eta_max, a_min, a_max = 1.0, 3.0, 0.0, 2.0
    rng = np.random.default_rng(1)
    body = []
    tables = np.zeros((2, n_eta, n_alpha, n_theta + 1), np.float32)
    for half in range(2):
        for i in range(n_eta):
            for j in range(n_alpha):
                row = 0.2 + 0.6 * (np.arange(n_theta + 1) / n_theta) ** 0.5 * (1 - 0.1 * j / n_alpha) * (1 - 0.05 * i / n_eta) * (0.9 if half else 1.0)
                tables[half, i, j] = row + rng.uniform(-0.01, 0.01, n_theta + 1)
    dat = b"MTS_TRANSMITTANCE" + struct.pack("<3Q4f", n_eta, n_alpha, n_theta, eta_min, eta_max, a_min, a_max) + tables.tobytes()
    for eta in (1.5, 1.0 / 1.5, eta_min + (eta_max - eta_min) * 0.5 ** 4):
        blk = capi.rtrans_reduce_eta(dat, eta)
        assert np.array_equal(blk, oracle_mod.rtrans_reduce_eta(dat, eta))
        assert blk.size == 4 + n_alpha * n_theta + 2 * n_alpha and tuple(blk[:4]) == (n_theta, n_alpha, a_min, a_max)
    # warped eta = 0.5 is the middle node of the five: the exterior slice is that node's rows
    assert np.allclose(blk[4:4 + n_alpha * n_theta].reshape(n_alpha, n_theta), tables[0, 2, :, :n_theta], atol=1e-6)
    assert np.allclose(blk[4 + n_alpha * n_theta:4 + n_alpha * n_theta + n_alpha], tables[0, 2, :, n_theta], atol=1e-6)
    # the one-dimensional reduction at a roughness that is a node of the table: a row of the block
    eta = 1.5
    blk = capi.rtrans_reduce_eta(dat, eta)
    alpha = a_min + (a_max - a_min) * (2 / (n_alpha - 1)) ** 4
    T, d_ext, d_int = capi.rtrans_reduce(dat, eta, alpha)
    T2 = blk[4:4 + n_alpha * n_theta].reshape(n_alpha, n_theta)
    assert np.allclose(T, T2[2], atol=2e-6) and abs(d_ext - blk[4 + n_alpha * n_theta + 2]) < 2e-6 and abs(d_int - blk[4 + n_alpha * n_theta + n_alpha + 2]) < 2e-6


def test_scene_builders_are_well_formed():
    for sd in (S.small_test_scene(), S.cornell_box(32, 32), S.veach_bidir(64, 48)):
        assert sd.indices.max() < sd.positions.shape[0]
        covered = np.zeros(sd.n_triangles, int)
        for s in sd.shapes:
            covered[s["first_triangle"]:s["first_triangle"] + s["n_triangles"]] += 1
        assert (covered == 1).all()
        for i, e in enumerate(sd.emitters):
            assert sd.shapes[e["shape"]]["emitter"] == i
    v = S.veach_bidir(64, 48)
    assert v.n_triangles == 3392 and len(v.shapes) == 11 and len(v.emitters) == 2
    assert [m["type"] for m in v.materials[:2]] == [S.BSDF_DIELECTRIC, S.BSDF_WARD]


def test_blob_normals_point_outwards():
    p, f, n = S.blob(3, np.random.default_rng(0))
    assert (np.einsum("ij,ij->i", p, n) > 0).all()


def test_loader_rejects_what_it_does_not_support(tmp_path):
    xml = tmp_path / "s.xml"
    xml.write_text('<scene version="0.5.0"><sensor type="perspective"><float name="fov" value="40"/></sensor>'
                   '<shape type="sphere"/></scene>')
    with pytest.raises(loader.SceneFormatError):
        loader.load_xml(str(xml))
    xml.write_text('<scene version="0.5.0"><bsdf type="plastic" id="a"/></scene>')
    with pytest.raises(loader.SceneFormatError):
        loader.load_xml(str(xml))


def test_serialized_roundtrip(tmp_path):
    """A v4 container written as TriMesh::serialize lays it out (src/librender/trimesh.cpp:1138-1200) loads back."""
    import struct, zlib
    rng = np.random.default_rng(1)
    meshes = []
    blob, offsets = b"", []
    for k in range(3):
        nv, nt = 10 + k, 5 + k
        p, n = rng.normal(size=(nv, 3)).astype(np.float32), rng.normal(size=(nv, 3)).astype(np.float32)
        idx = rng.integers(0, nv, (nt, 3)).astype(np.uint32)
        meshes.append((p, n, idx))
        body = struct.pack("<I", 0x1 | 0x1000) + b"m%d\0" % k + struct.pack("<QQ", nv, nt) + p.tobytes() + n.tobytes() + idx.tobytes()
        offsets.append(len(blob))
        blob += struct.pack("<HH", 0x041C, 4) + zlib.compress(body)
    blob += b"".join(struct.pack("<Q", o) for o in offsets) + struct.pack("<I", 3)
    assert loader.serialized_mesh_count(blob) == 3
    for k, (p, n, idx) in enumerate(meshes):
        m = loader.load_serialized(blob, k)
        assert m["name"] == "m%d" % k
        assert np.array_equal(m["positions"], p) and np.array_equal(m["normals"], n) and np.array_equal(m["indices"], idx)
    flipped = loader.apply_to_world(loader.load_serialized(blob, 0), np.diag([1, 1, -1, 1]).astype(np.float32))
    assert np.array_equal(flipped["indices"][:, 0], meshes[0][2][:, 1])      # negative determinant swaps the winding


def test_tile_schedules_partition_the_image():
    for w, h, tile, ws in ((100, 60, 16, 3), (1920, 1080, 128, 8), (33, 7, 64, 2)):
        cover = np.zeros((h, w), int)
        for r in range(ws):
            for x0, y0, x1, y1 in render.tile_schedule(w, h, tile, ws, r):
                cover[y0:y1, x0:x1] += 1
        assert (cover == 1).all()
    for w, h, ws in ((1920, 1080, 8), (64, 5, 4)):
        cover = np.zeros((h, w), int)
        for r in range(ws):
            for x0, y0, x1, y1 in render.band_schedule(w, h, ws, r, 2):
                cover[y0:y1, x0:x1] += 1
        assert (cover == 1).all()
    with pytest.raises(ValueError):
        render.tile_schedule(10, 10, 0, 1, 0)
    # the interleaved full-width tiles of the GPU path: the rectangles and the (first row, rows, period) triple agree
    for w, h, rows, ws in ((3840, 2160, 16, 8), (64, 48, 5, 3), (64, 5, 4, 4), (10, 100, 1, 7)):
        cover = np.zeros((h, w), int)
        for r in range(ws):
            rects = render.band_rects(w, h, rows, ws, r)
            b = render.interleaved_bands(h, rows, ws, r)
            assert (b is None) == (not rects)
            for x0, y0, x1, y1 in rects:
                cover[y0:y1, x0:x1] += 1
                assert (y0 - b[0]) % b[2] == 0 and y1 - y0 <= b[1] and (x0, x1) == (0, w)
        assert (cover == 1).all()


_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from mitsuba_amd import render, scene as S
from oracle import pyoracle as po
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
sd = S.small_test_scene()
o = po.OracleScene(sd)
w, h = sd.film["crop_w"], sd.film["crop_h"]
film = np.zeros((h, w, 5), np.float32)
px = render.render_partitioned(lambda r: o.render(3, film=film, seed=5, rect=r, n_threads=2), render.tile_schedule(w, h, 16, world, rank))
t = torch.from_numpy(film)
render.reduce_film(t, world)
# the interleaved full-width tiles the multi-GPU benchmark deals out
film2 = np.zeros((h, w, 5), np.float32)
px2 = render.render_partitioned(lambda r: o.render(3, film=film2, seed=5, rect=r, n_threads=2), render.band_rects(w, h, 5, world, rank))
t2 = torch.from_numpy(film2)
render.reduce_film(t2, world)
npx = torch.tensor([px, px2]); dist.all_reduce(npx)
if rank == 0:
    full, _ = o.render(3, seed=5, n_threads=2)
    assert npx.tolist() == [w * h, w * h]
    assert np.allclose(t.numpy(), full, rtol=1e-5, atol=1e-6), np.abs(t.numpy() - full).max()
    assert np.allclose(t2.numpy(), full, rtol=1e-5, atol=1e-6), np.abs(t2.numpy() - full).max()
    print("DIST_OK")
dist.destroy_process_group()
'''


def test_two_rank_partition_and_film_reduce_gloo(tmp_path, oracle_mod):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29611", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert "DIST_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_bvh_depth_never_exceeds_the_traversal_stack():
    """The host builder on geometry whose surface-area tree is a long chain (mitsuba_amd.scene.nested_slivers): up to
    the depth limit the chain is kept; beyond it the tree is rebuilt with median splits below the top levels, so the
    depth the traversal kernels must hold is never exceeded and no triangle is lost."""
    limit = capi.bvh_max_depth()
    assert limit >= 29
    depths = {}
    for n in (100, 420, 450, 470):
        sd = S.nested_slivers(n)
        info = capi.bvh_host_info(sd.positions, sd.indices)
        assert info.n_triangles == n and 1 <= info.depth <= limit, (n, info.depth)
        depths[n] = info.depth
    assert depths[420] >= limit - 2          # the chain reaches the limit ...
    assert depths[450] < depths[420]         # ... and the longer one was rebuilt shallower
    # degenerate inputs: nothing, one triangle, coincident triangles, coordinates whose box areas overflow
    assert capi.bvh_host_info(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32)).n_triangles == 0
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    assert capi.bvh_host_info(tri, [[0, 1, 2]]).depth == 1
    same = capi.bvh_host_info(tri, [[0, 1, 2]] * 40)
    assert same.n_triangles == 40 and same.depth <= limit
    huge = S.nested_slivers(300, ratio=1.25, first=1.0)        # sizes up to 1e29: areas overflow to infinity
    assert np.isfinite(huge.positions).all()
    info = capi.bvh_host_info(huge.positions, huge.indices)
    assert info.n_triangles == 300 and info.depth <= limit
