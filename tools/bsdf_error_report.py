#!/usr/bin/env python3
"""Prints, per material of the BSDF parity list (tests/test_gpu_parity.py), how far the GPU's eval / pdf / sample are from
the oracle's: fraction of inputs beyond 1e-4 relative and the 99.9th percentile of the relative difference."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mitsuba_amd import capi
from oracle import pyoracle as po
import test_gpu_parity as T

capi.init(0)
mats, g, o = T.material_pair._get_wrapped_function()(capi, po)
rng = np.random.default_rng(22)
n = 200000


def rel(a, b):
    return np.abs(a - b) / (np.maximum(np.abs(a), np.abs(b)) + 1e-6)


for mi, m in enumerate(mats):
    wi, wo = T._wi(n, rng), T._wi(n, rng)
    wo[::7, 2] *= -1
    e = rel(g.bsdf_eval(mi, wi, wo), o.bsdf_eval(mi, wi, wo)).max(-1)
    p = rel(g.bsdf_pdf(mi, wi, wo), o.bsdf_pdf(mi, wi, wo))
    if m["type"] in (3, 7, 8) or m["two_sided"]:
        wi[::2, 2] *= -1
    s = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    u = rng.uniform(0, 1, n).astype(np.float32)
    a, b = g.bsdf_sample(mi, wi, s, u), o.bsdf_sample(mi, wi, s, u)
    ok = a[4] == b[4]
    rw, rwt, rp = rel(a[0][ok], b[0][ok]).max(-1), rel(a[1][ok], b[1][ok]).max(-1), rel(a[2][ok], b[2][ok])
    f = lambda x: "%.1e/%.1e" % (np.mean(x > 1e-4), np.percentile(x, 99.9))
    print("mat %2d type %d distr %d: eval %s pdf %s | type flips %.1e  wo %s weight %s pdf %s" % (mi, m["type"], m["distribution"], f(e), f(p), 1 - ok.mean(), f(rw), f(rwt), f(rp)))
