#!/usr/bin/env python3
"""Throughput of the `direct` integrator (src/integrators/direct/direct.cpp) on the headline scene: one pass of
`--spp` camera samples per pixel with `--emitter-samples` / `--bsdf-samples` each, film resident on the device.

  python tools/direct_bench.py [--spp 64] [--emitter-samples 1] [--bsdf-samples 1] [--repeat 3]

Prints one JSON line: rays per second (camera + BSDF + shadow rays the reference would trace) and the kernel times."""
import argparse
import json
import os
import sys

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
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--emitter-samples", type=int, default=1)
    ap.add_argument("--bsdf-samples", type=int, default=1)
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    capi.init(0)
    sd, workload = bench.build_scene(args)
    sd.integrator.update(type="direct", emitter_samples=args.emitter_samples, bsdf_samples=args.bsdf_samples)
    g = capi.Scene(sd)
    film = torch.zeros(args.height, args.width, 5, device="cuda")
    runs = []
    for r in range(args.repeat + 1):
        st = g.render_device(args.spp, film.data_ptr(), seed=1 + r)
        runs.append(st)
    best = min(runs[1:], key=lambda s: s.seconds)
    rays = best.normal_rays + best.shadow_rays
    print(json.dumps({"workload": workload.replace("path integrator", "direct integrator"), "spp": args.spp,
                      "emitter_samples": args.emitter_samples, "bsdf_samples": args.bsdf_samples,
                      "mrays_per_s": rays / best.seconds / 1e6, "mpix_spp_per_s": args.width * args.height * args.spp / best.seconds / 1e6,
                      "seconds": best.seconds, "normal_rays": best.normal_rays, "shadow_rays": best.shadow_rays,
                      "kernel_ms": {"raygen": best.raygen_ms, "extend": best.extend_ms, "shade": best.shade_ms,
                                    "shadow": best.shadow_ms, "film": best.film_ms}}))


if __name__ == "__main__":
    main()
