#!/usr/bin/env python3
"""Ray-batch micro-benchmark of the traversal kernels alone: 2^24 random chords through the bounding sphere of the
scene (the ray distribution of the reference's kd-tree benchmark, src/tests/test_kd.cpp:112-117), already resident in
device memory, traced as closest-hit and as any-hit queries through the C ABI's device entry points.

  python tools/raybatch_bench.py [--workload config3|config5|cornell|veach] [--log2-rays 24] [--repeat 5]

Prints one JSON line: rays, hit fraction, Grays/s of both queries (best and median launch). These rays are incoherent from the first node on, unlike camera rays."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mitsuba_amd import capi  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config3")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--log2-rays", type=int, default=24)
    ap.add_argument("--repeat", type=int, default=5)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    capi.init(0)
    sd, workload = bench.build_scene(args)
    g = capi.Scene(sd)
    n = 1 << args.log2_rays
    gen = torch.Generator(device="cuda").manual_seed(1)
    lo = torch.tensor(sd.positions.min(0), device="cuda")
    hi = torch.tensor(sd.positions.max(0), device="cuda")
    c, r = (lo + hi) / 2, torch.linalg.norm(hi - lo) / 2
    a = torch.nn.functional.normalize(torch.randn(n, 3, device="cuda", generator=gen), dim=1)
    b = torch.nn.functional.normalize(torch.randn(n, 3, device="cuda", generator=gen), dim=1)
    o = c + a * r
    d = torch.nn.functional.normalize((c + b * r) - o, dim=1)
    rays = torch.empty(n, 8, device="cuda", dtype=torch.float32)
    rays[:, :3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, 1e-4, d, float("inf")
    hits = torch.empty(n, 5, device="cuda", dtype=torch.float32)        # mtsamd_hit: t, u, v, shape, prim
    occ = torch.empty(n, device="cuda", dtype=torch.uint8)
    torch.cuda.synchronize()

    closest, anyhit = [], []
    for _ in range(args.repeat + 1):
        closest.append(g.trace_device(rays.data_ptr(), n, hits.data_ptr()))
        anyhit.append(g.occluded_device(rays.data_ptr(), n, occ.data_ptr()))
    closest, anyhit = closest[1:], anyhit[1:]                            # first launch warms the caches
    prim = hits.view(torch.int32)[:, 4]
    hit_fraction = float((prim != -1).float().mean())
    assert abs(float(occ.float().mean()) - hit_fraction) < 1e-3          # both queries agree on what is hit

    out = {"workload": workload, "triangles": sd.n_triangles, "rays": n, "hit_fraction": hit_fraction,
           "closest_hit_grays_per_s": {"best": n / min(closest) / 1e9, "median": n / float(np.median(closest)) / 1e9},
           "any_hit_grays_per_s": {"best": n / min(anyhit) / 1e9, "median": n / float(np.median(anyhit)) / 1e9},
           "closest_hit_ms": [round(t * 1e3, 3) for t in closest], "any_hit_ms": [round(t * 1e3, 3) for t in anyhit]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
