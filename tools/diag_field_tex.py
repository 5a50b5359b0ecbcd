"""Diagnostic: per-sample radiance of the small textured field, GPU against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mitsuba_amd import capi, scene as S
from oracle import pyoracle
capi.init(0)
for flt in (None, "bilinear"):
    for md in (2, 3, -1):
        sd = S.microfacet_field(n_objects=6, subdiv=2, ground_res=8, width=96, height=54, textured=flt, max_depth=md)
        for m in sd.materials:
            m["frame_kind"] = 0; m["frame_texture"] = 0
        g, o = capi.Scene(sd), pyoracle.OracleScene(sd)
        Lg, pg, sg = g.li(4, rect=(0, 0, 64, 48), seed=3)
        Lo, po, so = o.li(4, rect=(0, 0, 64, 48), seed=3)
        rel = np.abs(Lg[..., :3] - Lo[..., :3]).max(-1) / (np.abs(Lo[..., :3]).max(-1) + 1e-3)
        print(flt, md, "rays", sg.normal_rays, so.normal_rays, sg.shadow_rays, so.shadow_rays,
              "frac<1e-3", np.mean(rel < 1e-3), "n>1e-3", int((rel >= 1e-3).sum()), "n>1e-5", int((rel >= 1e-5).sum()), "of", rel.size)
        bad = np.argwhere(rel >= 1e-3)[:6]
        for b in bad:
            print("   ", tuple(b), Lg[tuple(b)], Lo[tuple(b)])
