"""Ad-hoc GPU bring-up check: traversal parity, per-path radiance parity and a short timing run.
Run on the GPU box: python tools/gpu_check.py [--big]"""
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mitsuba_amd import capi, scene as S
from oracle import pyoracle as po


def random_rays(sd, n, seed=0):
    rng = np.random.default_rng(seed)
    lo, hi = sd.positions.min(0), sd.positions.max(0)
    c, r = (lo + hi) / 2, np.linalg.norm(hi - lo) / 2
    a = rng.normal(size=(n, 3)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.normal(size=(n, 3)); b /= np.linalg.norm(b, axis=1, keepdims=True)
    o = c + a * r
    d = (c + b * r) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, 1e-4, d, np.inf
    return rays


def check_scene(name, sd, nrays=200000, spp=4):
    t = time.time(); g = capi.Scene(sd); tb = time.time() - t
    info = g.bvh_info()
    print(f"[{name}] tris={sd.n_triangles} scene_create={tb:.2f}s bvh nodes={info.n_nodes} depth={info.depth} build={info.build_seconds:.2f}s", flush=True)
    t = time.time(); o = po.OracleScene(sd); print(f"  oracle build {time.time()-t:.2f}s", flush=True)
    rays = np.concatenate([random_rays(sd, nrays // 2), g.camera_rays(np.random.default_rng(1).uniform(0, 1, (nrays // 2, 2)) * [sd.film['crop_w'], sd.film['crop_h']])])
    hg = g.trace(rays); ho = o.trace(rays)
    mism = (hg['shape'] != ho['shape']) | (hg['prim'] != ho['prim'])
    tie = mism & (hg['t'] == ho['t'])
    bad = mism & ~tie
    both = (hg['prim'] != 0xFFFFFFFF) & ~mism
    print(f"  trace: hits={np.mean(ho['prim']!=0xFFFFFFFF):.3f} mismatches={bad.sum()} ties={tie.sum()} "
          f"t bitexact={np.mean(hg['t'][both]==ho['t'][both]):.6f} uv bitexact={np.mean((hg['u'][both]==ho['u'][both])&(hg['v'][both]==ho['v'][both])):.6f}", flush=True)
    if bad.sum():
        idx = np.nonzero(bad)[0][:5]
        hb = o.trace(rays[idx], brute=True)
        for k, i in enumerate(idx):
            print("   ray", i, "gpu", hg[i], "oracle", (ho['t'][i], ho['shape'][i], ho['prim'][i]), "brute", (hb['t'][k], hb['shape'][k], hb['prim'][k]))
    og = g.occluded(rays); oo = o.occluded(rays)
    print(f"  occluded mismatches={np.sum(og != oo)} of {len(og)}", flush=True)
    ig = g.intersect(rays[:20000]); io = o.trace(rays[:20000])
    ok = (ig['prim'] == io['prim']) & (ig['prim'] != 0xFFFFFFFF)
    for f in ('p', 'geo_n', 'sh_n', 'sh_s', 'sh_t', 'wi', 'uv'):
        d = np.abs(ig[f][ok] - io[f][ok]).max() if ok.any() else 0
        print(f"    isect {f}: max abs diff {d:.3e}")
    # per-sample radiance
    w, h = sd.film['crop_w'], sd.film['crop_h']
    rect = (0, 0, min(w, 64), min(h, 48))
    Lg, pg, sg = g.li(spp, rect=rect, seed=3)
    Lo, pos, so = o.li(spp, rect=rect, seed=3)
    print(f"  li: pos max diff {np.abs(pg-pos).max():.2e}; rays gpu {sg.normal_rays}/{sg.shadow_rays} oracle {so.normal_rays}/{so.shadow_rays}")
    rel = np.abs(Lg[..., :3] - Lo[..., :3]).max(-1) / (np.abs(Lo[..., :3]).max(-1) + 1e-3)
    print(f"  li: frac samples within 1e-3: {np.mean(rel < 1e-3):.5f}; mean L gpu {Lg[...,:3].mean():.5f} oracle {Lo[...,:3].mean():.5f}; alpha equal {np.mean(Lg[...,3]==Lo[...,3]):.5f}", flush=True)
    # film
    fg, st = g.render(spp, seed=3)
    fo, sto = o.render(spp, seed=3)
    dg, do = g.develop(fg), o.develop(fo)
    print(f"  film: weight max rel diff {np.abs(fg[...,4]-fo[...,4]).max()/fo[...,4].max():.2e}; img mean gpu {dg[...,:3].mean():.5f} oracle {do[...,:3].mean():.5f}; "
          f"relMSE {np.mean((dg[...,:3]-do[...,:3])**2/(do[...,:3]**2+1e-2)):.3e}")
    print(f"  gpu render: {st.seconds:.3f}s rays {(st.normal_rays+st.shadow_rays)/st.seconds/1e6:.1f} Mrays/s iters {st.bounce_iterations} "
          f"extend {st.extend_ms:.1f}ms shade {st.shade_ms:.1f}ms shadow {st.shadow_ms:.1f}ms film {st.film_ms:.1f}ms raygen {st.raygen_ms:.1f}ms", flush=True)
    return g, o


if __name__ == "__main__":
    capi.init(0)
    s = capi.rng_stream(5, 123456789, 16)
    print("rng parity:", np.array_equal(s, po.rng_stream(5, 123456789, 16)))
    check_scene("small", S.small_test_scene())
    check_scene("cornell", S.cornell_box(128, 128))
    if "--big" in sys.argv:
        sd = S.config3(1920, 1080)
        t = time.time(); g = capi.Scene(sd); print("config3 create", time.time() - t, "tris", sd.n_triangles, flush=True)
        info = g.bvh_info(); print("nodes", info.n_nodes, "depth", info.depth, "build", info.build_seconds, flush=True)
        import torch
        film = torch.zeros(1080, 1920, 5, device="cuda")
        for spp in (1, 4, 8):
            st = g.render_device(spp, film.data_ptr(), seed=1)
            print(f"config3 spp={spp}: {st.seconds:.3f}s {(st.normal_rays+st.shadow_rays)/st.seconds/1e6:.1f} Mrays/s iters {st.bounce_iterations} "
                  f"extend {st.extend_ms:.1f} shade {st.shade_ms:.1f} shadow {st.shadow_ms:.1f} film {st.film_ms:.1f} raygen {st.raygen_ms:.1f} ms", flush=True)
        img = g.develop(film.cpu().numpy())
        print("mean", img[..., :3].mean(), "nan", np.isnan(img).sum())
        np.save("gpurun_out/config3_img.npy", img[::4, ::4, :3].astype(np.float16))
