"""Where the triangle-phase threshold of the traversal loop (MTSAMD_TRI_SHARE, read when a scene is created) should
change with the size of the scene: the field scene at several triangle counts, a few 64-spp passes each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mitsuba_amd import capi, scene as S

capi.init(0)
for n_objects, subdiv, ground in ((8, 2, 8), (16, 3, 16), (32, 3, 32), (48, 4, 48), (64, 4, 64)):
    sd = S.microfacet_field(n_objects=n_objects, subdiv=subdiv, ground_res=ground, width=1280, height=720)
    line = "%8d triangles:" % sd.n_triangles
    for ts in (12, 16, 24, 32):
        os.environ["MTSAMD_TRI_SHARE"] = str(ts)
        g = capi.Scene(sd)
        g.render(64, seed=1)
        best = 0.0
        for rep in range(3):
            film, st = g.render(128, seed=2 + rep)
            best = max(best, (st.normal_rays + st.shadow_rays) / st.seconds / 1e6)
        line += "  share %d: %7.1f" % (ts, best)
        del g
    print(line, flush=True)
