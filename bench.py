#!/usr/bin/env python3
"""Benchmark of the `path` hot path on MI355X: Mrays/s (camera + bounce + shadow rays, counted as the reference
counts "Normal rays traced" + "Shadow rays traced", src/librender/skdtree.cpp:46-47,123,210).

  python bench.py --gpus N --steps K --warmup W

One GPU: the synthetic ~1M-triangle microfacet scene (roughconductor GGX + roughplastic GGX + dielectric over a
diffuse ground, four quad area lights) at 1920x1080, `path` integrator with default parameters; a step is one
wavefront pass of `--spp-step` (256) samples per pixel over the whole frame.

N > 1 GPUs: the same scene at 3840x2160, dealt out as an image-tile partition: full-width tiles of `--band-rows` (16)
rows, rank r owning tiles r, r + N, r + 2N, ... (the scene is replicated). Every rank traces wavefront passes of the
SAME size whatever N is -- 530 841 600 paths, the 1920x1080 x 256 samples of the one-GPU run: 64 N samples per pixel
of its 1/N share of the 4K frame -- because the traversal kernels' rate depends on the pass size; a step is
max(256, 64 N) samples per pixel of the whole frame (one or two such passes per rank). Per-GPU work per pass is fixed
as N grows (weak scaling); with the default 4 steps the job covers the 1 024 samples per pixel the configuration
names. After the K timed steps the per-rank films are summed with one reduction over RCCL inside the timed region --
Film::put(block) of the reference. Every rank reports the time it was busy, so that the balance of the partition is
visible.

All inputs (scene, BVH, film) are resident in HBM before the timed region.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from mitsuba_amd import capi, gates, render, scene as S

NODE_BYTES, TRI_BYTES = 80, 48
PASS_PATHS = 1920 * 1080 * 256      # paths a rank traces in one wavefront pass, on any number of GPUs
# Ceilings of the memory system for dependent fetches of small records, chip-wide, GB/s (MI355X measurements of
# indexed-row gathers: rows served by the XCD's L2 16.8-18.8 TB/s; rows of a table resident in the Infinity Cache
# 8.6 TB/s at 38 MB and 7.4-7.9 TB/s at 151 MB; rows from HBM 6.0-6.3 TB/s)
GATHER_L2_GBS = 17800.0
GATHER_IC_SMALL_GBS, GATHER_IC_LARGE_GBS, GATHER_HBM_GBS = 8600.0, 7650.0, 6150.0
INFINITY_CACHE_BYTES = 256 << 20
# Dependent fetches of uniformly random 80-B records, 5 x 16 B per lane, G records/s chip-wide (tools/microbench/gather.hip,
# MI355X): table of 2 MB (lives in every L2), 8 MB, 58 MB (Infinity Cache)
RECORD_GATHER_GRS = {"l2": 134.0, "8MB": 79.7, "58MB": 41.1}


def build_scene(args):
    if args.workload == "config3":
        return S.config3(args.width, args.height), "synthetic 1M-triangle microfacet scene (roughconductor GGX + roughplastic GGX + dielectric), %dx%d, path integrator" % (args.width, args.height)
    if args.workload == "config3simple":    # the same scene under volpath_simple's estimator (no media)
        sd = S.config3(args.width, args.height)
        sd.integrator["type"] = "volpath_simple"
        return sd, "synthetic 1M-triangle microfacet scene, %dx%d, volpath_simple integrator" % (args.width, args.height)
    if args.workload.startswith("config3tex"):       # config3tex, config3tex:trilinear, ...
        flt = args.workload.partition(":")[2] or "ewa"
        return S.config3_textured(args.width, args.height, flt), ("config 3 with 1024x1024 bitmap textures (%s) on ground, plastic and conductor colours and a "
                                                                  "bump-mapped ground, %dx%d, path integrator" % (flt, args.width, args.height))
    if args.workload == "config5":
        return S.config5(args.width, args.height), "synthetic 10M-triangle many-emitter scene, max depth 16, %dx%d" % (args.width, args.height)
    if args.workload == "cornell":
        return S.cornell_box(args.width, args.height), "Cornell box, diffuse, %dx%d" % (args.width, args.height)
    if args.workload == "veach":
        return S.veach_bidir(args.width, args.height), "veach_bidir, path integrator, %dx%d" % (args.width, args.height)
    raise SystemExit("unknown workload")


def host_threads():
    """Threads the CPU baseline runs on: the cores this process may use (its affinity mask), but no more than the
    share of the machine the job was given (OMP_NUM_THREADS, which the GPU box sets to its per-GPU CPU share)."""
    visible = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = visible
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = min(affinity, share) if share > 0 else affinity
    return max(1, threads), visible, affinity, share


def cpu_baseline(sd, seconds_target=10.0, seconds_cap=30.0):
    """The CPU restatement of the reference path integrator (oracle/, built with the reference's optimisation flags),
    timed on this host on a bounded sample of the same workload: the WHOLE frame at a few samples per pixel, repeated
    until at least `seconds_target` of rendering have been timed."""
    from oracle import pyoracle as po
    t0 = time.time()
    sc = po.OracleScene(sd, fast=True)
    build_s = time.time() - t0
    threads, visible, affinity, share = host_threads()
    total_rays, total_s, spp_done, spp = 0, 0.0, 0, 1
    while total_s < seconds_target:
        _, st = sc.render(spp, seed=1, n_threads=threads, spp_total=4096, sample_offset=spp_done)
        total_rays += st.normal_rays + st.shadow_rays
        total_s += st.seconds
        spp_done += spp
        # aim at the target with what has been measured so far, without passing the cap
        per_spp = total_s / spp_done
        spp = int(max(1, min((seconds_target - total_s) / per_spp + 1, (seconds_cap - total_s) / per_spp)))
        if total_s + per_spp > seconds_cap:
            break
    return dict(value=total_rays / total_s / 1e6, unit="Mrays/s", cores=threads, kind="port",
                host=dict(cpu_count=visible, affinity=affinity, omp_num_threads=share),
                sample="whole %dx%d frame x %d spp of the same scene (%d rays, %.1f s of rendering on %d threads), kd-tree build %.1f s excluded"
                       % (sd.film["crop_w"], sd.film["crop_h"], spp_done, total_rays, total_s, threads, build_s))


def extra_workloads(min_steps=2, min_seconds=1.5, spp=256):
    """Short runs of the configurations the headline is not quoted on: Cornell box 512x512 (config 1's scene on the GPU), veach_bidir
    1024x768 under `path` (config 2), the 10M-triangle many-emitter scene, depth 16, at 1080p (config 5 on one GPU). Two
    untimed passes, then passes of `spp` samples per pixel until at least `min_steps` of them and `min_seconds` have been
    timed; whole-job rates like the headline."""
    res = {}
    for name, make in (("cornell_512x512", lambda: S.cornell_box(512, 512)), ("veach_bidir_1024x768", lambda: S.veach_bidir(1024, 768)),
                       ("config5_10M_triangles_1920x1080", lambda: S.config5(1920, 1080))):
        sd = make()
        g = capi.Scene(sd)
        W, H = sd.film["crop_w"], sd.film["crop_h"]
        film = torch.zeros(H, W, 5, device="cuda", dtype=torch.float32)
        kw = dict(spp_total=spp * 4096, seed=1, spp_per_pass=spp)
        for i in range(2):
            g.render_device(spp, film.data_ptr(), sample_offset=i * spp, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rays = paths = steps = 0
        while steps < min_steps or time.perf_counter() - t0 < min_seconds:
            st = g.render_device(spp, film.data_ptr(), sample_offset=(steps + 2) * spp, **kw)
            rays += st.normal_rays + st.shadow_rays
            paths += st.paths
            steps += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = {"mrays_s": rays / dt / 1e6, "mpix_spp_per_s": paths / dt / 1e6, "ms_per_step": dt / steps * 1e3, "spp": spp * steps,
                     "triangles": sd.n_triangles}
        del g, film
        torch.cuda.empty_cache()
    return res


def kernel_source_hash():
    """Identifies the build a counter profile was taken on: hash of the device sources."""
    h = hashlib.sha1()
    for path in sorted(glob.glob(os.path.join(ROOT, "mitsuba_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "mitsuba_amd", "csrc", "*.hip"))):
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_counters(args):
    """Hardware counters of the closest-hit launches (tools/profile_round.sh runs the separate rocprofv3 --pmc passes
    and writes profiles/*_extend_pmc.json). Only a profile taken on THIS build (same hash of the device sources) of this
    workload and pass size is used; otherwise None."""
    if (args.width, args.height) != (1920, 1080):
        return None, None
    want = kernel_source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_extend_pmc.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        if d.get("source_hash") == want and d.get("spp_step") == args.spp_step and d.get("workload", "config3") == args.workload:
            return d, "profiles/" + os.path.basename(path)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="config3")
    ap.add_argument("--width", type=int, default=0, help="0: 1920x1080 on one GPU, 3840x2160 on several")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp-step", type=int, default=0, help="samples per pixel per step (0: 256 on one GPU, max(256, 64 N) on N)")
    ap.add_argument("--no-extra-workloads", action="store_true", help="skip the short runs of the other configurations (one GPU only)")
    ap.add_argument("--band-rows", type=int, default=16, help="rows of a tile of the multi-GPU partition")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal switches for a box with fewer GPUs than ranks: MTSAMD_DIST_BACKEND=gloo moves the collectives to the
    # CPU, MTSAMD_SHARE_DEVICE=1 puts every rank on device 0. The driver's runs use neither.
    backend = os.environ.get("MTSAMD_DIST_BACKEND", "nccl")
    device_index = 0 if os.environ.get("MTSAMD_SHARE_DEVICE") else local_rank
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(device_index)
    capi.init(device_index)
    coll_device = "cuda" if backend == "nccl" else "cpu"
    if args.width == 0 or args.height == 0:
        args.width, args.height = (1920, 1080) if world == 1 else (3840, 2160)
    # Pass size: what one rank traces at once. On N GPUs a rank owns 1/N of the rows, so 64 N samples per pixel of the 4K
    # frame are the 530 841 600 paths of the one-GPU pass (1920 x 1080 x 256).
    if args.spp_step <= 0:
        args.spp_step = 256 if world == 1 else max(256, 64 * world)
    spp_pass = args.spp_step if world == 1 else max(1, min(args.spp_step, (PASS_PATHS * world) // (args.width * args.height)))

    sd, workload = build_scene(args)
    g = capi.Scene(sd)
    info = g.bvh_info()
    W, H = sd.film["crop_w"], sd.film["crop_h"]
    film = torch.zeros(H, W, 5, device="cuda", dtype=torch.float32)
    n_steps_total = args.steps + args.warmup + 2
    spp_total = args.spp_step * n_steps_total
    if world > 1:
        band = render.interleaved_bands(H, args.band_rows, world, rank)
        rect, bands = (0, band[0], W, H), (band[1], band[2])
        partition = ("image tiles: full-width tiles of %d rows dealt round-robin over %d GPUs (%d tiles each); every GPU traces passes of %d samples "
                     "per pixel of its share = %d paths, the pass size of the one-GPU run; a step is %d samples per pixel of the whole frame; scene "
                     "replicated, one film reduction over RCCL inside the timed region"
                     % (args.band_rows, world, len(render.band_rects(W, H, args.band_rows, world, rank)), spp_pass,
                        spp_pass * sum(r[3] - r[1] for r in render.band_rects(W, H, args.band_rows, world, rank)) * W, args.spp_step))
    else:
        rect, bands = (0, 0, W, H), None
        partition = "one GPU: the whole frame in one wavefront pass per step"

    def step(i):
        st = g.render_device(args.spp_step, film.data_ptr(), spp_total=spp_total, sample_offset=i * args.spp_step, seed=1, rect=rect,
                             bands=bands, spp_per_pass=spp_pass)
        return dict(normal=st.normal_rays, shadow=st.shadow_rays, ext_ms=st.extend_ms, shd_ms=st.shadow_ms, shade_ms=st.shade_ms,
                    film_ms=st.film_ms, tail_ms=st.tail_ms, iters=st.bounce_iterations, paths=st.paths, busy_s=st.seconds)

    for i in range(args.warmup):
        step(i)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    acc = None
    for i in range(args.steps):
        s = step(args.warmup + i)
        acc = s if acc is None else {k: acc[k] + s[k] for k in acc}
    t_render = time.perf_counter() - t0
    if world > 1 and backend != "nccl":
        host_film = film.cpu()
        render.reduce_film(host_film, world)
        film.copy_(host_film)
    else:
        render.reduce_film(film, world)
    barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed, float(acc["normal"] + acc["shadow"]), float(acc["paths"])], device=coll_device, dtype=torch.float64)
    busy = torch.zeros(world, device=coll_device, dtype=torch.float64)
    busy[rank] = t_render
    if world > 1:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dist.all_reduce(busy, op=dist.ReduceOp.SUM)
        elapsed, rays, paths = tmax[0].item(), tsum[1].item(), tsum[2].item()
    else:
        rays, paths = t[1].item(), t[2].item()

    out = None
    if rank == 0:
        busy_list = [round(x, 4) for x in busy.tolist()]
        out = {
            "metric": "Mrays/s (camera+bounce+shadow)", "value": rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if world > 1 else "n/a", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "mpix_spp_per_s": paths / elapsed / 1e6,
            "config": {"workload": workload, "triangles": sd.n_triangles, "spp_per_step": args.spp_step, "spp_timed": args.spp_step * args.steps,
                       "paths_per_step": int(paths / args.steps), "partition": partition, "gates": gates.as_dict(),
                       "bvh": {"nodes": info.n_nodes, "depth": info.depth, "node_bytes": NODE_BYTES, "tri_bytes": TRI_BYTES,
                               "build_s": round(info.build_seconds, 3), "binary_tree_s": round(info.binary_seconds, 3),
                               "builder": "device" if info.builder == 1 else "host"}},
            "rank_busy_s": busy_list,
            "partition_balance": (sum(busy_list) / len(busy_list)) / max(busy_list) if max(busy_list) > 0 else None,
            "kernel_ms_rank0": {"extend": acc["ext_ms"], "shade": acc["shade_ms"], "shadow": acc["shd_ms"], "film": acc["film_ms"], "tail": acc["tail_ms"],
                                "bounce_iterations": acc["iters"]},
        }

    # ---- roofline of the dominant kernel (closest-hit traversal): algorithmic bytes from counted node visits and
    #      triangle tests, against the rate at which the cache hierarchy can serve dependent fetches of such records ----
    if rank == 0 and not args.no_roofline:
        g.set_counting(True)
        scratch = torch.zeros_like(film)
        kw = dict(spp_total=spp_total, sample_offset=args.warmup * args.spp_step, seed=1, rect=rect, bands=bands, spp_per_pass=spp_pass)
        stc = g.render_device(args.spp_step, scratch.data_ptr(), **kw)
        g.set_counting(False)
        st = g.render_device(args.spp_step, scratch.data_ptr(), **kw)
        nodes_per_ray = stc.node_visits / max(1, stc.normal_rays)
        tris_per_ray = stc.tri_tests / max(1, stc.normal_rays)
        bytes_per_ray = 32 + 20 + nodes_per_ray * NODE_BYTES + tris_per_ray * TRI_BYTES
        launches = max(1, st.extend_launches)
        # the launch that finishes the pass traces its few rays inside another kernel: not part of this kernel's figure
        wave_rays = st.normal_rays - st.tail_normal_rays
        alg = bytes_per_ray * wave_rays / launches            # algorithmic bytes per launch
        avg_ms = st.extend_ms / launches
        achieved = alg / (avg_ms * 1e-3) / 1e9
        structure = info.node_bytes + info.triangle_bytes
        if structure <= (64 << 20):
            fabric_rate, served = GATHER_IC_SMALL_GBS, "Infinity Cache (structure %.0f MB)" % (structure / 1e6)
        elif structure <= INFINITY_CACHE_BYTES * 3 // 4:
            fabric_rate, served = GATHER_IC_LARGE_GBS, "Infinity Cache (structure %.0f MB)" % (structure / 1e6)
        else:
            fabric_rate, served = GATHER_HBM_GBS, "HBM (structure %.0f MB exceeds the Infinity Cache)" % (structure / 1e6)
        pmc, pmc_src = measured_counters(args)
        traffic = pmc["fabric_bytes_per_launch"] if pmc else None
        if traffic is not None:
            beyond_l2 = min(traffic, alg)
            t_min = (alg - beyond_l2) / (GATHER_L2_GBS * 1e9) + beyond_l2 / (fabric_rate * 1e9)
            basis = "measured: %.1f GB of the %.1f GB algorithmic bytes per launch leave the L2s (%s)" % (beyond_l2 / 1e9, alg / 1e9, pmc_src)
        else:
            # no counter profile of this build: the most favourable split (everything served by the L2s) gives the
            # highest ceiling, so the fraction printed is a lower bound
            t_min = alg / (GATHER_L2_GBS * 1e9)
            basis = "no counter profile of this build: every byte priced at the L2 gather rate (the fraction is a lower bound)"
        peak = alg / t_min / 1e9
        # the same launches as a rate of records (80-B nodes + 48-B triangle records), next to what a kernel that does nothing
        # but fetch uniformly random 80-B records reaches on this chip (tools/microbench/gather.hip)
        records_per_s = (nodes_per_ray + tris_per_ray) * wave_rays / launches / (avg_ms * 1e-3)
        micro = RECORD_GATHER_GRS["l2"] if structure <= (4 << 20) else RECORD_GATHER_GRS["8MB"] if structure <= (16 << 20) else RECORD_GATHER_GRS["58MB"]
        out["roofline"] = {"kernel": "k_trace<TRACE_CLOSEST> (extend)",
                           "bound": "latency of dependent 16-byte fetches through L1 / L2 with the vector ALUs nearly full (not HBM, not MFMA): +13 % vector instructions cost 5 %, a fifth fewer node loads or a sixth wavefront per SIMD change nothing, one wavefront fewer costs 5 %",
                           "record_rate": {"achieved_Grecords_s": records_per_s / 1e9,
                                           "uniform_random_80B_records_Grecords_s": RECORD_GATHER_GRS,
                                           "frac_of_uniform_random_at_this_structure_size": records_per_s / 1e9 / micro,
                                           "frac_of_l2_resident_rate": records_per_s / 1e9 / RECORD_GATHER_GRS["l2"],
                                           "note": "the traversal beats the uniform-random rate at its structure size because the upper levels of the tree stay in the L1s and L2s; "
                                                   "against the rate for a table that fits the L2s it stands just below 1"},
                           "ceiling": "memory hierarchy serving dependent 80-B node / 48-B triangle fetches: bytes that stay in the L2s at %.0f GB/s, bytes beyond them at %.0f GB/s from %s"
                                      % (GATHER_L2_GBS, fabric_rate, served),
                           "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                           "traffic": traffic, "traffic_source": pmc_src, "ceiling_basis": basis,
                           "frac_of_hbm_peak": achieved / 8000.0,
                           "per_ray": {"node_visits": nodes_per_ray, "tri_tests": tris_per_ray, "algorithmic_bytes": bytes_per_ray,
                                       "shadow_node_visits": stc.shadow_node_visits / max(1, stc.shadow_rays),
                                       "shadow_tri_tests": stc.shadow_tri_tests / max(1, stc.shadow_rays)},
                           "launches": int(launches), "avg_launch_ms": avg_ms, "rays_per_launch": wave_rays / launches,
                           "algorithmic_bytes_per_launch": alg,
                           "counters": {k: pmc[k] for k in ("l1_hit", "l2_hit", "valu_lane_utilisation", "wait_fraction", "valu_issue_bound_ms") if pmc and k in pmc} or None,
                           "note": ("the structure (BVH %.1f MB + triangle records %.1f MB) lives in the L2s and the Infinity Cache, so HBM is not the ceiling of this kernel: "
                                    "frac_of_hbm_peak is printed for reference only" if structure <= INFINITY_CACHE_BYTES * 3 // 4 else
                                    "the structure (BVH %.1f MB + triangle records %.1f MB) exceeds the Infinity Cache: what leaves the L2s is served by HBM at its "
                                    "measured rate for dependent record fetches") % (info.node_bytes / 1e6, info.triangle_bytes / 1e6)}
        del scratch

    # ---- the other single-GPU configurations, two timed passes each (one GPU, default workload only) ----
    if rank == 0 and world == 1 and args.workload == "config3" and not args.no_extra_workloads:
        del g
        torch.cuda.empty_cache()
        out["extra"] = {"workloads": extra_workloads()}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sd)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
