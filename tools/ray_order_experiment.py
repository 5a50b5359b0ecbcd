#!/usr/bin/env python3
"""What the ORDER of the rays in the queue is worth to the closest-hit kernel, bounce by bounce.

Rays start at the camera of the benchmark scene (pixel order) and are scattered diffusely at every hit, the survivors
kept in queue order like the wavefront loop keeps them. At every bounce the same set of rays is traced three times
through mtsamd_trace_device: in queue order, sorted by (direction octant, Morton code of the origin in a 1024^3 grid),
and shuffled. Prints one JSON line per bounce with the three rates (the traversal launch's duration, no copies).

  python tools/ray_order_experiment.py [grid side, default 4096 -> 16.8 M camera rays] [bounces, default 5]
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba_amd import capi, scene as S  # noqa: E402


def spread_bits(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x30000ff
    v = (v | (v << 8)) & 0x300f00f
    v = (v | (v << 4)) & 0x30c30c3
    v = (v | (v << 2)) & 0x9249249
    return v


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    bounces = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    capi.init(0)
    dev = torch.device("cuda:0")
    sd = S.config3(1920, 1080)
    g = capi.Scene(sd)
    gen = torch.Generator(device=dev).manual_seed(1)
    pos = torch.from_numpy(sd.positions).to(dev)
    idx = torch.from_numpy(sd.indices.astype(np.int64)).to(dev)
    first = torch.tensor([s["first_triangle"] for s in sd.shapes], device=dev, dtype=torch.int64)
    lo, hi = pos.min(0).values, pos.max(0).values

    m = torch.from_numpy(np.asarray(sd.camera["to_world"], np.float32)).to(dev)
    t = float(np.tan(np.radians(sd.camera["x_fov"]) / 2))
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, side, device=dev), torch.linspace(-1, 1, side, device=dev), indexing="ij")
    d = torch.stack([xs * t, ys * t * 1080 / 1920, torch.ones_like(xs)], -1).reshape(-1, 3)
    d = torch.nn.functional.normalize(d @ m[:3, :3].T, dim=-1)
    o = m[:3, 3].expand_as(d)

    def pack(o, d):
        n = o.shape[0]
        rays = torch.empty((n, 8), device=dev, dtype=torch.float32)
        rays[:, 0:3] = o; rays[:, 3] = 1e-4          # mint = Epsilon: the adaptive offset of the reference
        rays[:, 4:7] = d; rays[:, 7] = float("inf")
        return rays

    def trace(rays):
        hits = torch.empty((rays.shape[0], 5), device=dev, dtype=torch.float32)
        torch.cuda.synchronize()
        sec = min(g.trace_device(rays.data_ptr(), rays.shape[0], hits.data_ptr()) for _ in range(3))
        return hits, sec

    for b in range(bounces + 1):
        rays = pack(o, d)
        n = rays.shape[0]
        hits, sec = trace(rays)
        cell = ((o - lo) / (hi - lo) * 1023.0).clamp(0, 1023).to(torch.int64)
        key = spread_bits(cell[:, 0]) | (spread_bits(cell[:, 1]) << 1) | (spread_bits(cell[:, 2]) << 2)
        key = key | ((d[:, 0] < 0).to(torch.int64) << 30) | ((d[:, 1] < 0).to(torch.int64) << 31) | ((d[:, 2] < 0).to(torch.int64) << 32)
        order = torch.argsort(key)
        _, sec_sorted = trace(rays[order].contiguous())
        order12 = torch.argsort((key >> 21) & 0xfff, stable=True)        # octant + 8^3 cells only: what two radix passes give
        _, sec_coarse = trace(rays[order12].contiguous())
        _, sec_shuffled = trace(rays[torch.randperm(n, device=dev, generator=gen)].contiguous())
        tt = hits[:, 0]
        hit = torch.isfinite(tt) & (hits[:, 3].view(torch.int32) != -1)
        print(json.dumps({"bounce": b, "rays": n, "hit_fraction": round(float(hit.float().mean()), 3),
                          "queue_order_grays": round(n / sec / 1e9, 2), "sorted_grays": round(n / sec_sorted / 1e9, 2),
                          "coarse_sorted_grays": round(n / sec_coarse / 1e9, 2), "shuffled_grays": round(n / sec_shuffled / 1e9, 2)}), flush=True)
        # diffuse scattering at the hits, survivors in queue order
        sel = hit.nonzero().squeeze(1)
        shape = hits[sel, 3].view(torch.int32).to(torch.int64)
        prim = hits[sel, 4].view(torch.int32).to(torch.int64)
        tri = idx[first[shape] + prim]
        p0, p1, p2 = pos[tri[:, 0]], pos[tri[:, 1]], pos[tri[:, 2]]
        nrm = torch.nn.functional.normalize(torch.cross(p1 - p0, p2 - p0, dim=-1), dim=-1)
        din = d[sel]
        nrm = torch.where((nrm * din).sum(-1, keepdim=True) > 0, -nrm, nrm)
        p = o[sel] + tt[sel, None] * din
        u = torch.rand((sel.shape[0], 2), device=dev, generator=gen)
        r, phi = torch.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
        a = torch.where(nrm[:, 0:1].abs() > 0.9, torch.tensor([0.0, 1.0, 0.0], device=dev), torch.tensor([1.0, 0.0, 0.0], device=dev))
        s = torch.nn.functional.normalize(torch.cross(nrm, a.expand_as(nrm), dim=-1), dim=-1)
        tb = torch.cross(nrm, s, dim=-1)
        d = s * (r * torch.cos(phi))[:, None] + tb * (r * torch.sin(phi))[:, None] + nrm * torch.sqrt((1 - u[:, 0]).clamp_min(0))[:, None]
        o = p
        if o.shape[0] == 0:
            break


if __name__ == "__main__":
    main()
