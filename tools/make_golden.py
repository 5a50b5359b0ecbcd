"""Generates the committed fixtures under tests/golden/ and mitsuba_amd/data/ from DATA files of the reference
checkout (no reference source is copied): run in the build container where /root/reference exists.

  tests/golden/kd_clip.json            the five known answers of test01_sutherlandHodgman
                                       (src/tests/test_kd.cpp:34-84), transcribed as data
  tests/golden/rtrans_nodes.json       raw grid-node values of data/microfacet/{beckmann,ggx}.dat (header
                                       ranges + ~100 table entries each) -- used to pin the microfacet / Fresnel code
  mitsuba_amd/data/veach_bidir/meshes.npz      the 11 meshes of testScene/veach_bidir/bidir.serialized (object space)
  mitsuba_amd/data/veach_bidir/scene_path.xml  testScene/veach_bidir/bidir.bak with the integrator switched to `path`
  tests/golden/bidir_ref_lowres.npz    testScene/veach_bidir/bidir.exr (768x576) box-downsampled to 96x72
  tests/golden/bidir_ref_cells.npz     the same image as means over 24-pixel cells, the bounding box of the banner the film
                                       stamped into it, and the lines of the log embedded in the file that say how its final
                                       version was rendered
  mitsuba_amd/csrc/cie1931.inc         the CIE 1931 2-degree colour matching functions at 1 nm (360-830 nm), one
                                       {lambda, x, y, z} row per wavelength: the numbers of the tables at
                                       src/libcore/spectrum.cpp:743-1141
  tests/golden/ior_spd/                three of the measured spectra under data/ior/ (Cu, Au, a-C), unchanged
  mitsuba_amd/data/rtrans_slices.json  reduced rough-transmittance tables for the bundled scenes (mtsamd_rtrans_reduce)
  mitsuba_amd/data/rtrans_eta_blocks.npz  the same tables reduced to an index of refraction only (mtsamd_rtrans_reduce_eta)
"""
import json
import os
import re
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
R = os.path.join(REF, "mitsuba-整合版")
T = os.path.join(REF, "testScene")
GOLD = os.path.join(ROOT, "tests", "golden")


def kd_clip():
    tri = [[0, 0, 0], [1, 0, 0], [1, 1, 0]]
    cases = [
        dict(box=[[0, .5, -1], [1, 1, 1]], min=[.5, .5, 0], max=[1, 1, 0], valid=True),
        dict(box=[[2, 2, 2], [3, 3, 3]], valid=False),
        dict(box=[[-1, -1, -1], [1, 1, 1]], min=[0, 0, 0], max=[1, 1, 0], valid=True),
        dict(box=[[-100, -100, 0], [100, 100, 0]], min=[0, 0, 0], max=[1, 1, 0], valid=True),
        dict(box=[[0, 1, 0], [1, 2, 0]], min=[1, 1, 0], max=[1, 1, 0], valid=True),
    ]
    json.dump(dict(source="src/tests/test_kd.cpp:34-84", triangle=tri, cases=cases), open(os.path.join(GOLD, "kd_clip.json"), "w"), indent=1)


def read_rtrans(path):
    d = open(path, "rb").read()
    assert d[:17] == b"MTS_TRANSMITTANCE"
    ne, na, nt = struct.unpack_from("<QQQ", d, 17)
    rng = struct.unpack_from("<ffff", d, 41)
    arr = np.frombuffer(d, "<f4", 2 * ne * na * (nt + 1), 57).reshape(2 * ne, na, nt + 1)
    return ne, na, nt, rng, arr


def rtrans_nodes():
    out = {}
    rs = np.random.RandomState(11)
    for name in ("beckmann", "ggx"):
        ne, na, nt, rng, arr = read_rtrans(os.path.join(R, "data", "microfacet", name + ".dat"))
        pts = []
        for _ in range(120):
            block = int(rs.randint(0, 2))        # 0: eta >= 1 block, 1: eta < 1 block
            k, i, j = int(rs.randint(2, ne)), int(rs.randint(3, 30)), int(rs.randint(20, nt))
            pts.append(dict(block=block, eta_index=k, alpha_index=i, theta_index=j,
                            T=float(arr[block * ne + k, i, j]), diffuse=float(arr[block * ne + k, i, nt])))
        out[name] = dict(eta_samples=ne, alpha_samples=na, theta_samples=nt, eta_min=rng[0], eta_max=rng[1],
                         alpha_min=rng[2], alpha_max=rng[3], points=pts)
    json.dump(dict(source="data/microfacet/*.dat (layout: src/bsdfs/rtrans.h:100-148)", tables=out),
              open(os.path.join(GOLD, "rtrans_nodes.json"), "w"))


def rtrans_slices():
    from mitsuba_amd import capi, scene as S
    out = {}
    for distr in ("beckmann", "ggx"):
        dat = open(os.path.join(R, "data", "microfacet", distr + ".dat"), "rb").read()
        for int_ior, ext_ior in ((1.49, 1.000277), (1.5046, 1.000277), (1.5, 1.0)):
            for alpha in (0.05, 0.1, 0.15, 0.2, 0.3, 0.5, 0.7):
                eta = int_ior / ext_ior
                Tt, de, di = capi.rtrans_reduce(dat, eta, alpha)
                out[S.rtrans_key(distr, eta, alpha)] = dict(T=[float(x) for x in Tt], diffuse_ext=de, diffuse_int=di)
    json.dump(out, open(os.path.join(ROOT, "mitsuba_amd", "data", "rtrans_slices.json"), "w"))
    # the same files reduced to an index of refraction only (roughplastic with a textured roughness)
    blocks = {}
    for distr in ("beckmann", "ggx"):
        dat = open(os.path.join(R, "data", "microfacet", distr + ".dat"), "rb").read()
        for int_ior, ext_ior in ((1.49, 1.000277), (1.5, 1.0)):
            eta = int_ior / ext_ior
            blocks["%s_%.6f" % (distr, eta)] = capi.rtrans_reduce_eta(dat, eta)
    np.savez_compressed(os.path.join(ROOT, "mitsuba_amd", "data", "rtrans_eta_blocks.npz"), **blocks)


def veach():
    from mitsuba_amd import loader
    data = open(os.path.join(T, "veach_bidir", "bidir.serialized"), "rb").read()
    n = loader.serialized_mesh_count(data)
    arrs = {}
    for i in range(n):
        m = loader.load_serialized(data, i)
        arrs["p%d" % i] = m["positions"]
        arrs["i%d" % i] = m["indices"].astype(np.uint32)
        if m["normals"] is not None:
            arrs["n%d" % i] = m["normals"]
        if m["texcoords"] is not None:
            arrs["t%d" % i] = m["texcoords"]
    vdir = os.path.join(ROOT, "mitsuba_amd", "data", "veach_bidir")
    os.makedirs(vdir, exist_ok=True)
    np.savez_compressed(os.path.join(vdir, "meshes.npz"), count=n, **arrs)
    xml = open(os.path.join(T, "veach_bidir", "bidir.bak")).read()
    xml = xml.replace('<integrator type="bdpt">', '<integrator type="path">')
    open(os.path.join(vdir, "scene_path.xml"), "w").write(xml)


def read_exr_zip_half(path):
    """Minimal OpenEXR reader: scanline, ZIP (16-line blocks), HALF channels."""
    d = open(path, "rb").read()
    assert struct.unpack_from("<I", d, 0)[0] == 20000630
    p = 8
    attrs = {}
    while True:
        e = d.index(b"\0", p)
        name = d[p:e].decode()
        if not name:
            p = e + 1
            break
        p = e + 1
        e = d.index(b"\0", p); typ = d[p:e].decode(); p = e + 1
        sz = struct.unpack_from("<i", d, p)[0]; p += 4
        attrs[name] = (typ, d[p:p + sz]); p += sz
    assert attrs["compression"][1][0] == 3, "ZIP compression expected"
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    chans = []
    c = attrs["channels"][1]
    q = 0
    while c[q] != 0:
        e = c.index(b"\0", q)
        nm = c[q:e].decode(); q = e + 1
        ptype = struct.unpack_from("<i", c, q)[0]; q += 16
        assert ptype == 1, "HALF pixels expected"
        chans.append(nm)
    nblocks = (h + 15) // 16
    offsets = struct.unpack_from("<%dQ" % nblocks, d, p)
    img = np.zeros((h, w, len(chans)), np.float32)
    for off in offsets:
        y, size = struct.unpack_from("<ii", d, off)
        raw = zlib.decompress(d[off + 8:off + 8 + size])
        a = np.frombuffer(raw, np.uint8).astype(np.int32)
        # predictor
        a = a.copy()
        a[1:] = a[1:] - 128
        a = (np.cumsum(a) & 0xff).astype(np.uint8)
        # de-interleave
        half = (len(a) + 1) // 2
        out = np.empty(len(a), np.uint8)
        out[0::2] = a[:half]
        out[1::2] = a[half:]
        lines = min(16, h - (y - y0))
        blk = out.view("<f2").reshape(lines, len(chans), w)
        img[y - y0:y - y0 + lines] = np.transpose(blk, (0, 2, 1)).astype(np.float32)
    return img, chans, attrs


def bidir_ref():
    img, chans, attrs = read_exr_zip_half(os.path.join(T, "veach_bidir", "bidir.exr"))
    order = [chans.index(c) for c in "RGB"]
    rgb = img[..., order]
    h, w, _ = rgb.shape
    f = 8
    low = rgb.reshape(h // f, f, w // f, f, 3).mean((1, 3))
    log = attrs.get("log", ("", b""))[1].decode("utf-8", "replace")
    np.savez_compressed(os.path.join(GOLD, "bidir_ref_lowres.npz"), image=low.astype(np.float32), full_size=[w, h], factor=f)
    open("/tmp/bidir_log.txt", "w").write(log)
    print("bidir.exr", rgb.shape, "mean", rgb.mean((0, 1)))


def bidir_ref_cells():
    """Cell means of bidir.exr plus what the file says about itself. The log attribute is the GUI session's whole
    log up to the moment the file was (over)written; the lines kept are the `Starting render job` / `Writing image`
    pairs and the integrator configuration, i.e. which integrator produced the pixels that are in the file."""
    img, chans, attrs = read_exr_zip_half(os.path.join(T, "veach_bidir", "bidir.exr"))
    rgb = img[..., [chans.index(c) for c in "RGB"]]
    h, w, _ = rgb.shape
    c = 24
    cells = rgb.reshape(h // c, c, w // c, c, 3).mean((1, 3)).astype(np.float32)
    # the banner is drawn with one saturated value (hdrfilm.cpp:496-506 stamps it 5 pixels from the corner)
    top = rgb.max()
    ys, xs = np.nonzero((rgb == top).all(-1))
    banner = [int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1]
    log = attrs["log"][1].decode("utf-8", "replace").splitlines()
    keep = [ln.strip() for ln in log if any(k in ln for k in ("Starting render job", "Writing image to", "did not complete", "Render time",
                                                              "photon map full", "Maximum path depth", "Mitsuba version", "Loading plugin"))]
    np.savez_compressed(os.path.join(GOLD, "bidir_ref_cells.npz"), cells=cells, cell=c, full_size=[w, h], banner_box=banner,
                        banner_value=np.float32(top), log_lines=np.array(keep))
    print("cells", cells.shape, "banner box", banner, "value", top)
    for ln in keep[-12:]:
        print("   ", ln[:150])


def cie_tables():
    text = open(os.path.join(R, "src", "libcore", "spectrum.cpp")).read()

    def table(name):
        body = re.search(r"const Float %s\[CIE_samples\] = \{(.*?)\};" % name, text, re.S).group(1)
        return [t for t in re.split(r"[\s,]+", body) if t]

    lam, x, y, z = (table(n) for n in ("CIE_wavelengths", "CIE_X_entries", "CIE_Y_entries", "CIE_Z_entries"))
    assert len(lam) == len(x) == len(y) == len(z) == 471
    with open(os.path.join(ROOT, "mitsuba_amd", "csrc", "cie1931.inc"), "w") as f:
        f.write("// CIE 1931 2-degree standard observer, 1 nm steps: {wavelength [nm], x-bar, y-bar, z-bar}.\n"
                "// Written by tools/make_golden.py; included by spectrum.cpp.\n")
        for row in zip(lam, x, y, z):
            f.write("{%sf, %sf, %sf, %sf},\n" % tuple(t + ("." if "." not in t and "e" not in t.lower() else "") for t in row))


def ior_spd():
    import shutil
    out = os.path.join(GOLD, "ior_spd")
    os.makedirs(out, exist_ok=True)
    for name in ("Cu", "Au", "a-C"):
        for part in ("eta", "k"):
            shutil.copy(os.path.join(R, "data", "ior", "%s.%s.spd" % (name, part)), out)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    kd_clip()
    rtrans_nodes()
    rtrans_slices()
    veach()
    bidir_ref()
    bidir_ref_cells()
    cie_tables()
    ior_spd()
    print("done")
