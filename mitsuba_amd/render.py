"""Image-tile partitioning of a render over the GPUs of one node.

The reference hands 32x32 blocks to worker threads (or network workers) and sums the finished, bordered blocks
into the film under a mutex (src/librender/imageproc.cpp:28-78, src/librender/renderproc.cpp:142-149). Here each
process owns one GPU and a fixed, interleaved subset of larger tiles; the scene is replicated; every rank
accumulates its tiles (filter border included) into a private full-frame (R,G,B,alpha,weight) film and a single
sum-reduction over RCCL (or gloo on CPU) at the end plays the role of Film::put(block). No collective runs during
tracing.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

Rect = Tuple[int, int, int, int]


def tile_schedule(width: int, height: int, tile: int, world_size: int, rank: int) -> List[Rect]:
    """Tiles owned by `rank`: tiles are numbered row-major and dealt round-robin with a per-row shift so that
    expensive image regions spread over all ranks. Every pixel belongs to exactly one tile of exactly one rank."""
    if tile <= 0 or world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad tile schedule arguments")
    nx, ny = (width + tile - 1) // tile, (height + tile - 1) // tile
    out = []
    for ty in range(ny):
        for tx in range(nx):
            if (tx + ty * (nx + 1)) % world_size == rank:
                x0, y0 = tx * tile, ty * tile
                out.append((x0, y0, min(x0 + tile, width), min(y0 + tile, height)))
    return out


def band_schedule(width: int, height: int, world_size: int, rank: int, bands_per_rank: int = 4) -> List[Rect]:
    """Alternative partition into full-width horizontal bands, interleaved over the ranks. Bands keep the wavefront
    passes large (width x band height x spp paths), which matters more for throughput than fine-grained balance."""
    n = world_size * bands_per_rank
    edges = [round(i * height / n) for i in range(n + 1)]
    return [(0, edges[i], width, edges[i + 1]) for i in range(n) if i % world_size == rank and edges[i + 1] > edges[i]]


def interleaved_bands(height: int, band_rows: int, world_size: int, rank: int):
    """The tile partition the GPU path uses: full-width tiles of `band_rows` rows dealt round-robin, rank r owning tiles
    r, r + n, r + 2n, ... -- the counterpart of the reference's blocks (src/librender/imageproc.cpp:43-78), wide
    instead of square so that one GPU's whole share is ONE wavefront pass (mtsamd_render_opts.band_rows /
    band_period). Returns (first row, band_rows, period) for the render call, or None when the rank owns no row."""
    if band_rows <= 0 or world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad band partition arguments")
    y0 = rank * band_rows
    if y0 >= height:
        return None
    return y0, band_rows, band_rows * world_size


def band_rects(width: int, height: int, band_rows: int, world_size: int, rank: int) -> List[Rect]:
    """The same partition as explicit rectangles (what a renderer without band support is given, e.g. the CPU oracle)."""
    out = []
    y = rank * band_rows
    while y < height:
        out.append((0, y, width, min(y + band_rows, height)))
        y += band_rows * world_size
    return out


def render_partitioned(render_rect: Callable[[Rect], None], rects: List[Rect]) -> int:
    """Calls render_rect for every rectangle; returns the number of pixels covered."""
    px = 0
    for r in rects:
        render_rect(r)
        px += (r[2] - r[0]) * (r[3] - r[1])
    return px


def reduce_film(film, world_size: int, dst: int = 0):
    """Sum the per-rank films onto rank `dst` (torch.distributed must be initialised when world_size > 1).
    `film` is a torch tensor (device tensor with the nccl backend = RCCL, CPU tensor with gloo)."""
    if world_size > 1:
        import torch.distributed as dist
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film
