"""Bitmap textures on the CPU: the oracle's restatement of the reference's MIP map (include/mitsuba/render/mipmap.h),
resampler (include/mitsuba/core/rfilter.h:107-468) and uv partials (src/librender/intersection.cpp:5-76) against closed
forms, and the library's image reader (mtsamd_image_load, no device needed) against values computed here.
The reference holds no expected texture values (its texture tests are visual), so these lookups are parity unpinned:
the restatement is checked by the properties below, the GPU against the restatement (tests/test_gpu_parity.py)."""
import os
import struct
import zlib

import numpy as np
import pytest

from mitsuba_amd import scene as S


def _scene_with(textures):
    sd = S.cornell_box(16, 16)
    sd.textures = list(textures)
    return sd


def _half(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def test_half_storage_nearest_lookup_and_statistics(oracle_mod):
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 3, (7, 11, 3)).astype(np.float32)
    img[0, 0] = [1e-7, 65504.0, 1e5]          # a half denormal, the largest half, overflow to infinity
    o = oracle_mod.OracleScene(_scene_with([S.bitmap(img, "nearest")]))
    ys, xs = np.mgrid[0:7, 0:11]
    uv = np.stack([(xs.ravel() + 0.5) / 11, (ys.ravel() + 0.5) / 7], -1)
    got = o.texture_eval(1, uv).reshape(7, 11, 3)
    with np.errstate(over="ignore"):
        assert np.array_equal(got, _half(img))      # evalBox: the texel itself, stored as half (round to nearest even)
    # repeat wrap: one period further on is the same texel
    assert np.array_equal(o.texture_eval(1, uv + [3, -2]).reshape(7, 11, 3), got)


def test_bilinear_lookup_wrap_modes_and_gradient(oracle_mod):
    rng = np.random.default_rng(1)
    img = rng.uniform(0, 1, (5, 6)).astype(np.float32)
    h = _half(img)
    texs = [S.bitmap(img, "bilinear", wrap_mode=m) for m in ("repeat", "clamp", "mirror", "zero", "one")]
    o = oracle_mod.OracleScene(_scene_with(texs))
    # at texel centres bilinear interpolation returns the texel
    ys, xs = np.mgrid[0:5, 0:6]
    centres = np.stack([(xs.ravel() + 0.5) / 6, (ys.ravel() + 0.5) / 5], -1)
    for ti in range(1, 6):
        assert np.allclose(o.texture_eval(ti, centres)[:, 0].reshape(5, 6), h, atol=1e-6)
    # half a texel left of the first column: the average of the first column and whatever the wrap mode puts beside it
    uv = np.float32([[0.0, 0.5 / 5]])
    first, last = h[0, 0], h[0, -1]
    expect = {1: (first + last) / 2, 2: first, 3: first, 4: first / 2, 5: (first + 1) / 2}
    for ti, e in expect.items():
        assert np.isclose(o.texture_eval(ti, uv)[0, 0], e, atol=1e-6), ti
    # gradient of a ramp along u (evalGradientBilinear, mipmap.h:601-626): slope per unit of u
    ramp = np.tile(np.linspace(0, 1, 9, dtype=np.float32) * np.float32(0.5), (4, 1))     # exact in half
    o2 = oracle_mod.OracleScene(_scene_with([S.bitmap(ramp, "bilinear", wrap_mode="clamp", uscale=2.0, scale=3.0)]))
    g = o2.texture_eval(1, np.float32([[0.2, 0.4]]), gradient=True)[0]
    # texel step 0.5 / 8 per 1 / 9 of u, times uscale (Texture2D::evalGradient) and the scale texture's factor
    assert np.allclose(g[0], 0.5 / 8 * 9 * 2.0 * 3.0, rtol=1e-4) and np.allclose(g[1], 0, atol=1e-5)


def test_pyramid_levels_of_a_constant_and_of_a_ramp(oracle_mod):
    """Levels halve rounding up until 1 x 1 (mipmap.h:244-270); resampling keeps a constant constant (normalised
    weights) and a ramp a ramp away from the borders (the Lanczos filter is symmetric)."""
    const = np.full((13, 21, 3), 0.375, np.float32)
    w = 64
    ramp = np.tile(np.linspace(0.1, 0.9, w, dtype=np.float32), (w, 1))
    o = oracle_mod.OracleScene(_scene_with([S.bitmap(const, "trilinear"), S.bitmap(ramp, "trilinear", wrap_mode="clamp"),
                                            S.bitmap(const, "ewa")]))
    rng = np.random.default_rng(2)
    uv = rng.uniform(0, 1, (200, 2)).astype(np.float32)
    for footprint in (1e-4, 0.01, 0.1, 0.5, 3.0):         # from below one texel to beyond the whole image
        duv = np.tile(np.float32([footprint, 0, 0, footprint]), (200, 1))
        assert np.allclose(o.texture_eval(1, uv, duv), 0.375, atol=2e-4)
        assert np.allclose(o.texture_eval(3, uv, duv), 0.375, atol=2e-4)
    # a footprint of 4 texels selects level 2 (log2 of the major radius, mipmap.h:654-668); in the interior the ramp's value
    inner = uv[(uv[:, 0] > 0.3) & (uv[:, 0] < 0.7)]
    duv = np.tile(np.float32([4.0 / w, 0, 0, 4.0 / w]), (inner.shape[0], 1))
    got = o.texture_eval(2, inner, duv)[:, 0]
    expect = 0.1 + 0.8 * (inner[:, 0] * w - 0.5) / (w - 1)
    assert np.allclose(got, expect, atol=3e-3)


def test_ewa_lookup_properties(oracle_mod):
    rng = np.random.default_rng(3)
    w = 128
    ramp = np.tile(np.linspace(0.0, 1.0, w, dtype=np.float32)[None, :], (w, 1))
    noise = rng.uniform(0, 1, (w, w, 3)).astype(np.float32)
    o = oracle_mod.OracleScene(_scene_with([S.bitmap(ramp, "ewa", wrap_mode="clamp"), S.bitmap(noise, "ewa"),
                                            S.bitmap(noise, "ewa", max_anisotropy=2.0), S.bitmap(noise, "bilinear")]))
    uv = rng.uniform(0.3, 0.7, (300, 2)).astype(np.float32)
    # an elliptical average of a ramp is the ramp's value at the centre (the weights are symmetric), whatever the
    # orientation and the eccentricity of the ellipse
    for d in ([0.03, 0.0, 0.0, 0.03], [0.05, 0.02, -0.01, 0.02], [0.08, 0.0, 0.0, 0.004]):
        got = o.texture_eval(1, uv, np.tile(np.float32(d), (300, 1)))[:, 0]
        expect = (uv[:, 0] * w - 0.5) / (w - 1)
        assert np.allclose(got, expect, atol=6e-3), d
    # footprints below one texel fall back to the bilinear lookup of level 0 (mipmap.h:709-710)
    tiny = np.tile(np.float32([1e-4, 0, 0, 1e-4]), (300, 1))
    assert np.allclose(o.texture_eval(2, uv, tiny), o.texture_eval(4, uv), atol=1e-6)
    # a wider footprint averages more: the variance over positions drops
    v = [o.texture_eval(2, uv, np.tile(np.float32([f, 0, 0, f]), (300, 1))).var() for f in (0.01, 0.04, 0.16)]
    assert v[0] > v[1] > v[2]
    # a tighter anisotropy bound widens the minor axis of a skinny ellipse: more blur
    skinny = np.tile(np.float32([0.2, 0, 0, 0.004]), (300, 1))
    assert o.texture_eval(3, uv, skinny).var() < o.texture_eval(2, uv, skinny).var()


def test_camera_uv_partials_on_a_facing_plane(oracle_mod):
    """Intersection::computePartials for a plane facing a perspective camera: one pixel step moves the hit point by
    2 d tan(fov / 2) / width, scaled by 1 / sqrt(sampleCount) (integrator.cpp:150,177), and uv follows through dpdu /
    dpdv. Seen through a trilinear lookup of a one-texel checker: a footprint of 4 texels reads level 2 (the checker
    averaged out), the same view at 64 samples per pixel has a footprint of half a texel and reads the checker itself.
    The albedo a sample saw is its radiance divided by that of the same scene with a constant albedo (same sample
    stream, same path: the diffuse BSDF's sampling does not depend on its colour)."""
    n = 64
    yy, xx = np.mgrid[0:n, 0:n]
    checker = ((xx + yy) % 2).astype(np.float32)

    def scene(texture):
        b = S.SceneBuilder()
        m = b.add_material(S.diffuse((0.5, 0.5, 0.5)))
        if texture is not None:
            b.materials[m]["diffuse_texture"] = b.add_texture(texture)
        half = 2.0      # uv runs over [0, 2]: two world units per unit of uv, texels of 1 / 32
        b.add_mesh(np.float32([[-half, -half, 0], [half, -half, 0], [half, half, 0], [-half, half, 0]]),
                   np.uint32([[0, 1, 2], [0, 2, 3]]), m, texcoords=np.float32([[0, 0], [2, 0], [2, 2], [0, 2]]))
        b.add_constant_emitter((1.0, 1.0, 1.0))
        dist = 10.0     # 16 pixels across 2 world units at the plane: 4 texels per pixel
        cam = S.make_camera([0, 0, dist], [0, 0, 0], [0, 1, 0], float(2 * np.degrees(np.arctan(1.0 / dist))), near_clip=0.1, far_clip=100)
        return b.finish(cam, S.make_film(16, 16), dict(max_depth=2))

    flat = oracle_mod.OracleScene(scene(None))
    tex = oracle_mod.OracleScene(scene(S.bitmap(checker, "trilinear")))
    albedo = {}
    for spp_total in (1, 64):
        Lt, pt, _ = tex.li(1, spp_total=spp_total, seed=4)
        Lf, pf, _ = flat.li(1, spp_total=spp_total, seed=4)
        assert np.array_equal(pt, pf)
        ok = Lf[..., 0] > 1e-3
        assert ok.mean() > 0.9
        albedo[spp_total] = 0.5 * Lt[..., 0][ok] / Lf[..., 0][ok]
    assert np.allclose(albedo[1], 0.5, atol=0.02)           # level 2 of a one-texel checker: grey
    assert albedo[64].std() > 0.12 and albedo[64].min() < 0.2 and albedo[64].max() > 0.8    # level 0: the checker, interpolated


# ---- image files -------------------------------------------------------------------------------------------------------
def _png(path, arr, colour, depth, extra=b"", palette=None):
    h, w = arr.shape[:2]

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    rows = []
    for y in range(h):
        row = arr[y]
        if depth == 16:
            data = row.astype(">u2").tobytes()
        elif depth == 8:
            data = row.astype(np.uint8).tobytes()
        else:
            bits = "".join(format(int(v), "0%db" % depth) for v in row.ravel())
            bits += "0" * (-len(bits) % 8)
            data = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
        # alternate the row filters; all but "none" need the previous bytes, so only none / sub are written here
        rows.append(b"\x00" + data)
    body = chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, 0)) + extra
    if palette is not None:
        body += chunk(b"PLTE", bytes(palette))
    body += chunk(b"IDAT", zlib.compress(b"".join(rows))) + chunk(b"IEND", b"")
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + body)
    return chunk


def _srgb_to_linear(v):
    v = np.asarray(v, np.float32)
    lo = v * np.float32(1.0 / 12.92)
    hi = np.power(((v + np.float32(0.055)) * np.float32(1.0 / 1.055)).astype(np.float32), np.float32(2.4)).astype(np.float32)
    return np.where(v <= np.float32(0.04045), lo, hi)


def test_image_load_png(tmp_path):
    from mitsuba_amd import capi
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (5, 7, 3))
    p = str(tmp_path / "rgb8.png")
    _png(p, rgb, 2, 8)
    got = capi.image_load(p)
    assert got.shape == (5, 7, 3)
    assert np.allclose(got, _srgb_to_linear(rgb.astype(np.float32) * np.float32(1 / 255.0)), rtol=2e-6, atol=1e-7)
    # the texture's gamma property overrides the file's (here: linear)
    assert np.array_equal(capi.image_load(p, gamma=1.0), rgb.astype(np.float32) * np.float32(1 / 255.0))
    assert np.allclose(capi.image_load(p, gamma=2.2), np.power(rgb.astype(np.float32) * np.float32(1 / 255.0), np.float32(2.2)), rtol=2e-6)
    # 16-bit grey, grey + alpha (alpha dropped), RGBA, 4-bit grey, palette
    g16 = rng.integers(0, 65536, (4, 3))
    _png(p, g16, 0, 16)
    assert np.allclose(capi.image_load(p), _srgb_to_linear(g16.astype(np.float32) * np.float32(1 / 65535.0)), rtol=2e-6, atol=1e-8)
    ga = rng.integers(0, 256, (3, 4, 2))
    _png(p, ga, 4, 8)
    assert np.allclose(capi.image_load(p), _srgb_to_linear(ga[..., 0].astype(np.float32) * np.float32(1 / 255.0)), rtol=2e-6, atol=1e-7)
    rgba = rng.integers(0, 256, (3, 4, 4))
    _png(p, rgba, 6, 8)
    assert np.allclose(capi.image_load(p), _srgb_to_linear(rgba[..., :3].astype(np.float32) * np.float32(1 / 255.0)), rtol=2e-6, atol=1e-7)
    g4 = rng.integers(0, 16, (3, 5))
    _png(p, g4, 0, 4)
    assert np.allclose(capi.image_load(p), _srgb_to_linear((g4 * 17).astype(np.float32) * np.float32(1 / 255.0)), rtol=2e-6, atol=1e-7)
    pal = rng.integers(0, 256, (6, 3))
    idx = rng.integers(0, 6, (4, 4))
    _png(p, idx, 3, 8, palette=pal.ravel().tolist())
    assert np.allclose(capi.image_load(p), _srgb_to_linear(pal[idx].astype(np.float32) * np.float32(1 / 255.0)), rtol=2e-6, atol=1e-7)
    # a gAMA chunk without an sRGB chunk sets the decoding exponent (bitmap.cpp:2534-2541)
    chunk = _png(p, rgb, 2, 8)
    _png(p, rgb, 2, 8, extra=chunk(b"gAMA", struct.pack(">I", 50000)))
    assert np.allclose(capi.image_load(p), np.power(rgb.astype(np.float32) * np.float32(1 / 255.0), np.float32(2.0)), rtol=2e-6)
    # filtered rows (sub / up / average / Paeth) as zlib-compressed by a real encoder: the CLI's own PNG writer is
    # covered in tests/test_cli.py; here a hand-filtered image
    h, w = 6, 5
    img = rng.integers(0, 256, (h, w, 3)).astype(np.int64)
    raw = bytearray()
    prev = np.zeros(w * 3, np.int64)
    for y in range(h):
        cur = img[y].ravel()
        ft = (y % 4) + 1
        left = np.concatenate([np.zeros(3, np.int64), cur[:-3]])
        upleft = np.concatenate([np.zeros(3, np.int64), prev[:-3]])
        if ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) // 2
        else:
            pa, pb, pc = abs(prev - upleft), abs(left - upleft), abs(left + prev - 2 * upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        raw += bytes([ft]) + bytes(((cur - pred) % 256).astype(np.uint8))
        prev = cur

    def chunk2(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(p, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk2(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                        chunk2(b"IDAT", zlib.compress(bytes(raw))) + chunk2(b"IEND", b""))
    assert np.array_equal(capi.image_load(p, gamma=1.0), img.astype(np.float32) * np.float32(1 / 255.0))
    with pytest.raises(capi.MtsamdError):
        capi.image_load(str(tmp_path / "missing.png"))
    open(p, "wb").write(b"\xff\xd8\xff\xe0 not really a jpeg")
    with pytest.raises(capi.MtsamdError, match="JPEG"):
        capi.image_load(p)


def _piz_block(planes_rows, word_sizes, width):
    """One PIZ block the way an OpenEXR writer stores it (the published format: ImfPizCompressor, ImfWav, ImfHuf; test
    data only, no run-length codes): planes_rows[c] = the block's rows of channel c as uint16 words."""
    import heapq
    data = [np.concatenate(rows).astype(np.int64) for rows in planes_rows]
    everything = np.concatenate(data)
    present = np.zeros(65536, bool)
    present[everything] = True
    bitmap = np.packbits(present, bitorder="little")
    bitmap[0] &= 0xfe
    nz = np.nonzero(bitmap)[0]
    lo, hi = (int(nz[0]), int(nz[-1])) if len(nz) else (8191, 0)
    present[0] = True
    rank = np.cumsum(present) - 1
    max_value = int(rank[-1])
    w14 = max_value < (1 << 14)

    def enc(a, b):
        if w14:
            a = a - 65536 if a >= 32768 else a
            b = b - 65536 if b >= 32768 else b
            return ((a + b) >> 1) & 0xffff, (a - b) & 0xffff
        ao = (a + 0x8000) & 0xffff
        m, d = (ao + b) >> 1, ao - b
        if d < 0:
            m = (m + 0x8000) & 0xffff
        return m, d & 0xffff

    coded = []
    for c, words in enumerate(data):
        a = [int(v) for v in rank[words]]
        sz, ny = word_sizes[c], len(planes_rows[c])
        for j in range(sz):
            nx, ox, oy = width, sz, width * sz
            pw, p2 = 1, 2
            while p2 <= min(nx, ny):
                py = j
                while py <= j + oy * (ny - p2):
                    px = py
                    while px <= py + ox * (nx - p2):
                        p01, p10 = px + ox * pw, px + oy * pw
                        p11 = p10 + ox * pw
                        i00, i01 = enc(a[px], a[p01])
                        i10, i11 = enc(a[p10], a[p11])
                        a[px], a[p10] = enc(i00, i10)
                        a[p01], a[p11] = enc(i01, i11)
                        px += ox * p2
                    if nx & pw:
                        a[px], a[px + oy * pw] = enc(a[px], a[px + oy * pw])
                    py += oy * p2
                if ny & pw:
                    px = py
                    while px <= py + ox * (nx - p2):
                        a[px], a[px + ox * pw] = enc(a[px], a[px + ox * pw])
                        px += ox * p2
                pw, p2 = p2, p2 << 1
        coded += a
    freq = {}
    for v in coded:
        freq[v] = freq.get(v, 0) + 1
    im, iM = min(freq), max(freq) + 1
    freq[iM] = 1                                    # the run-length symbol (never used below)
    heap = [(f, s, (s,)) for s, f in freq.items()]
    heapq.heapify(heap)
    length = dict.fromkeys(freq, 0)
    while len(heap) > 1:
        f1, s1, m1 = heapq.heappop(heap)
        f2, s2, m2 = heapq.heappop(heap)
        for sym in m1 + m2:
            length[sym] += 1
        heapq.heappush(heap, (f1 + f2, min(s1, s2), m1 + m2))
    assert max(length.values()) <= 58
    count = [0] * 59
    for l in length.values():
        count[l] += 1
    first, c = [0] * 59, 0
    for l in range(58, 0, -1):                      # canonical codes, the longest first
        first[l], c = c, (c + count[l]) >> 1
    code, nxt = {}, list(first)
    for sym in sorted(length):
        code[sym] = nxt[length[sym]]
        nxt[length[sym]] += 1
    bits = []
    i = im
    while i <= iM:                                  # six bits per code length, runs of zeros packed
        l = length.get(i, 0)
        if l:
            bits.append((l, 6)); i += 1
            continue
        run = 1
        while i + run <= iM and length.get(i + run, 0) == 0 and run < 255 + 6:
            run += 1
        if run >= 6:
            bits.append((63, 6)); bits.append((run - 6, 8))
        elif run >= 2:
            bits.append((59 + run - 2, 6))
        else:
            bits.append((0, 6))
        i += run

    def pack(pairs):
        acc, n = 0, 0
        for v, k in pairs:
            acc, n = (acc << k) | v, n + k
        pad = -n % 8
        return ((acc << pad).to_bytes((n + pad) // 8, "big") if n else b""), n
    table, _ = pack(bits)
    payload, n_bits = pack([(code[v], length[v]) for v in coded])
    huf = struct.pack("<IIIII", im, iM, len(table), n_bits, 0) + table + payload
    head = struct.pack("<HH", lo, hi) + (bytes(bitmap[lo:hi + 1]) if lo <= hi else b"")
    return head + struct.pack("<I", len(huf)) + huf, w14


def test_image_load_pfm_and_openexr(tmp_path):
    from mitsuba_amd import capi
    rng = np.random.default_rng(6)
    img = rng.uniform(0, 4, (5, 3, 3)).astype(np.float32)
    p = str(tmp_path / "a.pfm")
    with open(p, "wb") as f:        # rows bottom to top, little endian, scale 2
        f.write(b"PF\n3 5\n-2.0\n")
        f.write(img[::-1].astype("<f4").tobytes())
    assert np.array_equal(capi.image_load(p), img * np.float32(2.0))
    lum = rng.uniform(0, 1, (4, 6)).astype(np.float32)
    with open(p, "wb") as f:        # big endian, luminance
        f.write(b"Pf\n6 4\n1.0\n")
        f.write(lum[::-1].astype(">f4").tobytes())
    assert np.array_equal(capi.image_load(p), lum)

    wavelets = []

    def exr(path, planes, names, types, compression):
        h, w = planes[0].shape
        order = np.argsort(names)
        hdr = b"\x76\x2f\x31\x01" + struct.pack("<I", 2)

        def attr(name, typ, data):
            return name + b"\0" + typ + b"\0" + struct.pack("<I", len(data)) + data
        chl = b"".join(names[i].encode() + b"\0" + struct.pack("<IBBBBII", types[i], 0, 0, 0, 0, 1, 1) for i in order) + b"\0"
        hdr += attr(b"channels", b"chlist", chl) + attr(b"compression", b"compression", bytes([compression]))
        box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
        hdr += attr(b"dataWindow", b"box2i", box) + attr(b"displayWindow", b"box2i", box)
        hdr += attr(b"lineOrder", b"lineOrder", b"\0") + attr(b"pixelAspectRatio", b"float", struct.pack("<f", 1))
        hdr += attr(b"screenWindowCenter", b"v2f", struct.pack("<ff", 0, 0)) + attr(b"screenWindowWidth", b"float", struct.pack("<f", 1)) + b"\0"
        lines = {0: 1, 1: 1, 2: 1, 3: 16, 4: 32}[compression]
        blocks = []
        for y0 in range(0, h, lines):
            raw = b""
            for y in range(y0, min(h, y0 + lines)):
                for i in order:
                    raw += planes[i][y].astype("<f2" if types[i] == 1 else "<f4").tobytes()
            data = raw
            if compression == 4:
                rows = [[np.frombuffer(planes[i][y].astype("<f2" if types[i] == 1 else "<f4").tobytes(), "<u2")
                         for y in range(y0, min(h, y0 + lines))] for i in order]
                comp, w14 = _piz_block(rows, [1 if types[i] == 1 else 2 for i in order], w)
                data = comp if len(comp) < len(raw) else raw        # what a writer does
                wavelets.append(w14 if data is comp else None)
            elif compression:
                a = np.frombuffer(raw, np.uint8)
                t = np.concatenate([a[0::2], a[1::2]]).astype(np.int64)
                t[1:] = (t[1:] - t[:-1] + 128 + 256) % 256
                comp = zlib.compress(bytes(t.astype(np.uint8)))
                data = comp if len(comp) < len(raw) else raw
            blocks.append(struct.pack("<iI", y0, len(data)) + data)
        table_at = len(hdr)
        offs, pos = [], table_at + 8 * len(blocks)
        for bl in blocks:
            offs.append(pos)
            pos += len(bl)
        open(path, "wb").write(hdr + b"".join(struct.pack("<Q", o) for o in offs) + b"".join(blocks))

    p = str(tmp_path / "a.exr")
    big = rng.uniform(0, 2, (37, 9, 3)).astype(np.float32)
    exr(p, [big[..., 0], big[..., 1], big[..., 2]], ["R", "G", "B"], [2, 2, 2], 0)
    assert np.array_equal(capi.image_load(p), big)
    exr(p, [big[..., 0], big[..., 1], big[..., 2]], ["R", "G", "B"], [1, 1, 1], 3)     # ZIP, half
    assert np.array_equal(capi.image_load(p), big.astype(np.float16).astype(np.float32))
    exr(p, [big[..., 0], big[..., 1], big[..., 2], big[..., 0]], ["R", "G", "B", "A"], [1, 2, 1, 2], 2)   # ZIPS, mixed types, alpha ignored
    expect = big.copy()
    expect[..., 0] = _half(big[..., 0]); expect[..., 2] = _half(big[..., 2])
    assert np.array_equal(capi.image_load(p), expect)
    exr(p, [big[..., 1]], ["Y"], [2], 3)
    assert np.array_equal(capi.image_load(p), big[..., 1])
    # PIZ: half and float channels, odd sizes, a second block stored as it is; few distinct values (the 14-bit wavelet)
    yy, xx = np.mgrid[0:37, 0:45]
    smooth = (np.round(8 * np.sin(xx / 7.0) * np.cos(yy / 5.0)) / 8 + 2).astype(np.float32)
    steps = np.stack([smooth, smooth.T[:37, :37].mean() + 0 * smooth, np.round(smooth * 3) / 4], -1).astype(np.float32)
    exr(p, [steps[..., 0], steps[..., 1], steps[..., 2], steps[..., 0]], ["R", "G", "B", "A"], [1, 2, 1, 2], 4)
    assert np.array_equal(capi.image_load(p), steps)
    # ... and more than 2^14 distinct 16-bit words in a block (the 16-bit wavelet): full-precision floats of a smooth image
    yy, xx = np.mgrid[0:33, 0:400]
    fine = np.stack([np.sin(xx / 40.0 + c) * np.cos(yy / 30.0) + 1.5 for c in range(3)], -1)
    fine = (fine + rng.uniform(0, 1e-3, fine.shape)).astype(np.float32)
    exr(p, [fine[..., 0], fine[..., 1], fine[..., 2]], ["R", "G", "B"], [2, 2, 2], 4)
    assert np.array_equal(capi.image_load(p), fine)
    assert wavelets == [True, None, False, None]      # the short last blocks are smaller as they are
    # a header that claims a gigantic image is refused before anything is allocated
    exr(p, [big[..., 1]], ["Y"], [2], 0)
    data = bytearray(open(p, "rb").read())
    at = data.index(b"dataWindow\0box2i\0") + len(b"dataWindow\0box2i\0") + 4
    data[at + 8:at + 12] = struct.pack("<i", 1 << 30)
    open(p, "wb").write(bytes(data))
    with pytest.raises(capi.MtsamdError, match="out of range"):
        capi.image_load(p)
    # run-length compression is not read: refused with a message that says so (PIZ, which the reference's own files
    # use, is: tests/test_envmap.py reads one)
    exr(p, [big[..., 1]], ["Y"], [2], 1)
    with pytest.raises(capi.MtsamdError, match="compression"):
        capi.image_load(p)


def test_bump_and_normal_map_frames(oracle_mod):
    """BumpMap::getFrame (bumpmap.cpp:135-159) and NormalMap::getFrame (normalmap.cpp:107-124) through a diffuse BSDF,
    whose value is albedo / pi times the cosine to the perturbed normal. The batch queries have no intersection: they
    use the canonical frame (dpdu = x, dpdv = y, n = z), so a height a * u tilts the normal to (-a, 0, 1) / sqrt(1 + a^2)."""
    w = 33
    ramp = np.tile(np.linspace(0.0, 1.0, w, dtype=np.float32)[None, :], (4, 1))
    sd = S.cornell_box(16, 16)
    slope = 0.5 * (w / (w - 1.0))          # one unit of u spans w texels, the ramp rises by 1 / (w - 1) per texel; times the scale
    sd.textures = [S.bitmap(ramp, "bilinear", wrap_mode="clamp", scale=0.5),
                   S.bitmap(np.full((4, 4, 3), 0.5, np.float32) * [1, 1, 2], "nearest"),          # (0.5, 0.5, 1): the unperturbed normal
                   S.bitmap(np.full((4, 4, 3), 1.0, np.float32) * [0.75, 0.5, 1.0], "bilinear"),  # n = (0.5, 0, 1) normalised (exact in half)
                   S.checkerboard((0.3, 0.3, 0.3), (0.3, 0.3, 0.3))]                             # a flat displacement
    base = S.diffuse((0.6, 0.4, 0.2))
    sd.materials = list(sd.materials) + [S.bumpmap(base, 1), S.normalmap(base, 2), S.normalmap(base, 3), S.bumpmap(base, 4), base,
                                         S.twosided(S.bumpmap(base, 1))]
    k = len(sd.materials) - 6
    o = oracle_mod.OracleScene(sd)
    rng = np.random.default_rng(8)
    n = 4000
    wo = rng.normal(size=(n, 3)).astype(np.float32)
    wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    wi = np.tile(np.float32([0.1, 0.2, 0.97]), (n, 1))
    wi /= np.linalg.norm(wi[0])
    o.bsdf_query_uv(0.4, 0.5)
    albedo = np.float32([0.6, 0.4, 0.2])

    def lambert(normal):
        normal = np.float32(normal) / np.linalg.norm(normal)
        c = wo @ normal
        ok = (c > 0) & (wi[0] @ normal > 0) & (wo[:, 2] * c > 0)      # same side in both frames (bumpmap.cpp:169-170)
        return np.where(ok[:, None], albedo[None, :] / np.pi * c[:, None], 0)

    assert np.allclose(o.bsdf_eval(k, wi, wo), lambert([-slope, 0, 1]), atol=2e-5)
    assert np.allclose(o.bsdf_eval(k + 1, wi, wo), o.bsdf_eval(k + 4, wi, wo), atol=1e-6)       # (0.5, 0.5, 1) changes nothing
    assert np.allclose(o.bsdf_eval(k + 2, wi, wo), lambert([0.5, 0.0, 1.0]), atol=2e-5)
    assert np.allclose(o.bsdf_eval(k + 3, wi, wo), o.bsdf_eval(k + 4, wi, wo), atol=1e-6)       # zero gradient changes nothing
    # pdf and sample agree with eval in the perturbed frame: weight = eval / pdf = albedo, directions on the tilted hemisphere
    pdf = o.bsdf_pdf(k, wi, wo)
    normal = np.float32([-slope, 0, 1]) / np.linalg.norm([-slope, 0, 1])
    assert np.allclose(pdf, np.where(lambert(normal)[:, 0] > 0, (wo @ normal) / np.pi, 0), atol=2e-5)
    swo, weight, spdf, eta, typ = o.bsdf_sample(k, wi, rng.random((n, 2)).astype(np.float32))
    live = weight[:, 0] > 0
    assert live.mean() > 0.7 and np.allclose(weight[live], albedo, atol=1e-5)
    assert ((swo[live] @ normal) > 0).all() and (swo[live][:, 2] > 0).all()
    assert np.allclose(np.linalg.norm(swo[live], axis=1), 1, atol=1e-5)
    # seen from behind through twosided: the mirrored query of the bumped BRDF (twosided.cpp:108-183)
    back_wi, back_wo = wi * [1, 1, -1], wo * [1, 1, -1]
    assert np.allclose(o.bsdf_eval(k + 5, back_wi.astype(np.float32), back_wo.astype(np.float32)), o.bsdf_eval(k, wi, wo), atol=1e-6)
    # adapters cannot sit inside a mixture, and need a texture
    bad = S.cornell_box(16, 16)
    bad.textures = sd.textures
    bad.materials = list(bad.materials) + [S.bumpmap(base, 1)]
    bad.materials.append(S.mixture([len(bad.materials) - 1], [1.0]))
    with pytest.raises(RuntimeError, match="cannot be nested"):
        oracle_mod.OracleScene(bad)
    bad.materials = list(bad.materials)[:-2] + [S.bumpmap(base, 9)]
    with pytest.raises(RuntimeError, match="out of range"):
        oracle_mod.OracleScene(bad)


def test_textured_roughness(oracle_mod):
    """m_alphaU->eval(bRec.its).average() (roughconductor.cpp:264-268, roughdielectric.cpp:303-308): a roughness texture
    gives, at every uv, what the record with that constant roughness gives; `alpha` sets both directions, alphaU alone
    makes the record anisotropic; the texture's own colours count (no energy-conservation factor: values above one stay);
    records that cannot take one say so."""
    b = S.SceneBuilder()
    t_check = b.add_texture(S.checkerboard((0.0625, 0.0625, 0.0625), (0.5, 0.5, 0.5), uscale=2.0, vscale=2.0))
    t_flat = b.add_texture(S.bitmap(np.full((4, 4), 0.25, np.float32), "bilinear"))
    t_big = b.add_texture(S.bitmap(np.full((4, 4, 3), (1.5, 1.5, 1.5), np.float32), "nearest"))
    cond = dict(eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), distribution="ggx")
    rec = {}
    rec["check"] = b.add_material(dict(S.roughconductor(alpha=0.3, **cond), alpha_u_texture=t_check, alpha_v_texture=t_check))
    rec["lo"] = b.add_material(S.roughconductor(alpha=0.0625, **cond))
    rec["hi"] = b.add_material(S.roughconductor(alpha=0.5, **cond))
    rec["flat_u"] = b.add_material(dict(S.roughconductor(alpha=0.1, **cond), alpha_u_texture=t_flat))
    aniso = S.roughconductor(alpha=0.1, **cond)
    aniso.update(alpha_u=0.25, alpha_v=0.1)
    rec["aniso"] = b.add_material(aniso)
    rec["glass_tex"] = b.add_material(dict(S.roughdielectric(alpha=0.1, distribution="beckmann", int_ior=1.5, ext_ior=1.0), alpha_u_texture=t_flat, alpha_v_texture=t_flat))
    rec["glass"] = b.add_material(S.roughdielectric(alpha=0.25, distribution="beckmann", int_ior=1.5, ext_ior=1.0))
    rec["big_tex"] = b.add_material(dict(S.roughconductor(alpha=0.1, **cond), alpha_u_texture=t_big, alpha_v_texture=t_big))
    rec["big"] = b.add_material(S.roughconductor(alpha=1.5, **cond))
    b.add_quad([-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0], rec["check"], radiance=(1, 1, 1))
    sd = b.finish(S.make_camera([0, -3, 2], [0, 0, 0], [0, 0, 1], 40), S.make_film(8, 8))
    o = oracle_mod.OracleScene(sd)
    rng = np.random.default_rng(3)
    n = 4000
    d = rng.normal(size=(2, n, 3))
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    wi, wo = d[0].astype(np.float32), d[1].astype(np.float32)
    smp, extra = rng.random((n, 2)).astype(np.float32), rng.random(n).astype(np.float32)

    def same(a, c, uv):
        o.bsdf_query_uv(*uv)
        assert np.allclose(o.bsdf_eval(rec[a], wi, wo), o.bsdf_eval(rec[c], wi, wo), rtol=2e-5, atol=1e-7), (a, c, uv)
        assert np.allclose(o.bsdf_pdf(rec[a], wi, wo), o.bsdf_pdf(rec[c], wi, wo), rtol=2e-5, atol=1e-7), (a, c, uv)
        sa, sc = o.bsdf_sample(rec[a], wi, smp, extra), o.bsdf_sample(rec[c], wi, smp, extra)
        for x, y in zip(sa, sc):
            assert np.allclose(x, y, rtol=2e-4, atol=1e-5), (a, c, uv)
    same("check", "lo", (0.1, 0.1))         # first colour of the checkerboard ...
    same("check", "hi", (0.1, 0.35))        # ... and the second
    same("check", "lo", (0.35, 0.35))
    same("flat_u", "aniso", (0.4, 0.6))
    same("glass_tex", "glass", (0.7, 0.2))
    same("big_tex", "big", (0.5, 0.5))
    o.bsdf_query_uv(0.0, 0.0)
    assert np.abs(o.bsdf_eval(rec["lo"], wi, wo) - o.bsdf_eval(rec["hi"], wi, wo)).max() > 0.1
    # a record that cannot take one says so; a rough plastic needs its transmittance table as a function of the roughness
    for bad, message in ((S.diffuse(), "roughconductor, roughdielectric and roughplastic only"),
                         (S.roughplastic(alpha=0.1, distribution="ggx", int_ior=1.49), "two-dimensional transmittance table")):
        b2 = S.SceneBuilder()
        t = b2.add_texture(S.checkerboard())
        m = b2.add_material(dict(bad, alpha_u_texture=t, alpha_v_texture=t))
        b2.add_quad([-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0], m, radiance=(1, 1, 1))
        with pytest.raises(RuntimeError, match=message):
            oracle_mod.OracleScene(b2.finish(S.make_camera([0, -3, 2], [0, 0, 0], [0, 0, 1], 40), S.make_film(8, 8)))


def test_textured_roughness_on_roughplastic(oracle_mod):
    """RoughPlastic with a non-constant alpha (roughplastic.cpp:281-299, 336-341, 361-363): the microfacet lobe uses the
    roughness found at the intersection, and the transmittances T12, T21 and the interior diffuse transmittance are
    looked up at that roughness in the table reduced to the index of refraction only (rtrans.h:195-205, 254-259).
    Where the texture is flat the record must give what the record with that constant roughness gives -- exactly in the
    microfacet lobe, and in the diffuse lobe up to the difference between interpolating the two-dimensional table and
    interpolating its slice (setAlpha samples the same interpolant at the table's own angles, rtrans.h:353-388)."""
    b = S.SceneBuilder()
    t_flat = b.add_texture(S.bitmap(np.full((4, 4), 0.2, np.float32), "bilinear"))
    t_check = b.add_texture(S.checkerboard((0.1, 0.1, 0.1), (0.5, 0.5, 0.5), uscale=2.0, vscale=2.0))
    rec = {}
    for distr in ("ggx", "beckmann"):
        kw = dict(distribution=distr, int_ior=1.49, diffuse_reflectance=(0.5, 0.3, 0.2))
        rec[distr, "flat"] = b.add_material(S.roughplastic(alpha=0.7, alpha_texture=t_flat, **kw))
        rec[distr, 0.2] = b.add_material(S.roughplastic(alpha=0.2, **kw))
        rec[distr, "check"] = b.add_material(S.roughplastic(alpha=0.7, alpha_texture=t_check, nonlinear=True, **kw))
        rec[distr, 0.1] = b.add_material(S.roughplastic(alpha=0.1, nonlinear=True, **kw))
        rec[distr, 0.5] = b.add_material(S.roughplastic(alpha=0.5, nonlinear=True, **kw))
    b.add_quad([-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0], rec["ggx", "flat"], radiance=(1, 1, 1))
    sd = b.finish(S.make_camera([0, -3, 2], [0, 0, 0], [0, 0, 1], 40), S.make_film(8, 8))
    o = oracle_mod.OracleScene(sd)
    rng = np.random.default_rng(5)
    n = 4000
    d = rng.normal(size=(2, n, 3))
    d[:, :, 2] = np.abs(d[:, :, 2])
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    wi, wo = d[0].astype(np.float32), d[1].astype(np.float32)
    for distr in ("ggx", "beckmann"):
        for a, c, uv in (("flat", 0.2, (0.4, 0.6)), ("check", 0.1, (0.1, 0.1)), ("check", 0.5, (0.1, 0.35))):
            o.bsdf_query_uv(*uv)
            ea, ec = o.bsdf_eval(rec[distr, a], wi, wo), o.bsdf_eval(rec[distr, c], wi, wo)
            assert np.allclose(ea, ec, rtol=3e-3, atol=1e-6), (distr, a, c, np.abs(ea - ec).max())
            assert np.allclose(o.bsdf_pdf(rec[distr, a], wi, wo), o.bsdf_pdf(rec[distr, c], wi, wo), rtol=3e-3, atol=1e-6), (distr, a, c)
        # the two colours of the checkerboard are two different materials
        o.bsdf_query_uv(0.1, 0.1)
        lo = o.bsdf_eval(rec[distr, "check"], wi, wo)
        o.bsdf_query_uv(0.1, 0.35)
        assert np.abs(lo - o.bsdf_eval(rec[distr, "check"], wi, wo)).max() > 0.05
    o.bsdf_query_uv(0.0, 0.0)
    # the roughness texture must stay inside the table's range (RoughTransmittance::checkAlpha, rtrans.h:400-407)
    # and a rough plastic takes one isotropic roughness
    b3 = S.SceneBuilder()
    t1, t2 = b3.add_texture(S.checkerboard()), b3.add_texture(S.checkerboard((0.2, 0.2, 0.2), (0.3, 0.3, 0.3)))
    m = b3.add_material(dict(S.roughplastic(alpha_texture=t1, distribution="ggx", int_ior=1.49), alpha_v_texture=t2))
    b3.add_quad([-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0], m, radiance=(1, 1, 1))
    with pytest.raises(RuntimeError, match="anisotropic"):
        oracle_mod.OracleScene(b3.finish(S.make_camera([0, -3, 2], [0, 0, 0], [0, 0, 1], 40), S.make_film(8, 8)))


def test_rtrans_eta_block_contains_the_reduced_slices(oracle_mod):
    """The packaged two-dimensional tables against the packaged one-dimensional ones (both made from the reference's
    data/microfacet/*.dat): setAlpha evaluates the two-dimensional interpolant at the table's own angles
    (rtrans.h:371-377), so looking the block up there must reproduce the slice; the diffuse transmittances likewise."""
    for distr in ("ggx", "beckmann"):
        eta = 1.49 / 1.000277
        blk = np.asarray(S.load_rtrans_eta_block(distr, eta), np.float32)
        n_theta, n_alpha, a_min, a_max = int(blk[0]), int(blk[1]), blk[2], blk[3]
        assert blk.size == 4 + n_theta * n_alpha + 2 * n_alpha and a_max > a_min
        T2 = blk[4:4 + n_theta * n_alpha].reshape(n_alpha, n_theta)
        assert ((T2 >= -1e-3) & (T2 <= 1 + 1e-3)).all()
        for alpha in (0.1, 0.3):
            T1, d_ext, d_int = S.load_rtrans_slice(distr, eta, alpha)
            b = S.SceneBuilder()
            t = b.add_texture(S.bitmap(np.full((2, 2), alpha, np.float32), "nearest"))
            mt = b.add_material(S.roughplastic(alpha_texture=t, distribution=distr, int_ior=1.49, diffuse_reflectance=(0.5, 0.5, 0.5), specular_reflectance=(0, 0, 0)))
            mc = b.add_material(S.roughplastic(alpha=alpha, distribution=distr, int_ior=1.49, diffuse_reflectance=(0.5, 0.5, 0.5), specular_reflectance=(0, 0, 0)))
            b.add_quad([-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0], mt, radiance=(1, 1, 1))
            o = oracle_mod.OracleScene(b.finish(S.make_camera([0, -3, 2], [0, 0, 0], [0, 0, 1], 40), S.make_film(8, 8)))
            # directions whose cosines are the table's angles: cos = (j / (n - 1))^4
            cos = (np.arange(1, n_theta, dtype=np.float64) / (n_theta - 1)) ** 4
            cos = cos[cos > 1e-3]
            w = np.stack([np.sqrt(1 - cos * cos), np.zeros_like(cos), cos], 1).astype(np.float32)
            et, ec = o.bsdf_eval(mt, w, w[::-1].copy()), o.bsdf_eval(mc, w, w[::-1].copy())
            assert np.allclose(et, ec, rtol=2e-4, atol=1e-7), (distr, alpha, np.abs(et / np.maximum(ec, 1e-9) - 1).max())
