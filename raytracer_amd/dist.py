"""Multi-GPU sharding of the render path: one process per GPU, scene replicated.

The reference renders every pixel of every frame on one device; there is nothing to mirror
here.  Work is split two ways, both without any exchange during rendering because every
pixel-sample is independent (its RNG is keyed by pixel and frame index only):

* samples per pixel: rank r renders the frames f with (f - first_frame) % world == r;
* image tiles: 64x64-pixel macro tiles are dealt round-robin in Morton order
  (``tile_owner``), the rule the kernels apply when ``rt_render_sum`` is given nshards > 1.

Each rank adds linear radiance into its own fp32 RGBA sum image; the images are combined once
per run with a sum-reduce to rank 0 (``torch.distributed.reduce``: RCCL over xGMI on GPUs, gloo
on CPU), after which rank 0 divides by the sample count and sRGB-encodes.
"""

from __future__ import annotations

import numpy as np


def frames_of_rank(first_frame: int, nframes: int, rank: int, world: int) -> list[int]:
    """Frame indices rank `rank` renders under samples-per-pixel sharding."""
    return [first_frame + i for i in range(nframes) if i % world == rank]


def _part1by1(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.uint32) & 0xFFFF
    v = (v | (v << 8)) & 0x00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v


def tile_owner(x, y, nshards: int) -> np.ndarray:
    """Shard that owns pixel (x, y): Morton index of its 64x64 macro tile modulo nshards."""
    x = np.asarray(x, dtype=np.uint32) >> 6
    y = np.asarray(y, dtype=np.uint32) >> 6
    return ((_part1by1(x) | (_part1by1(y) << 1)) % np.uint32(nshards)).astype(np.uint32)


def owner_mask(width: int, height: int, shard: int, nshards: int) -> np.ndarray:
    """(H, W) bool mask of the pixels shard `shard` renders (only the area the dispatch covers)."""
    ww, hh = (width // 8) * 8, (height // 4) * 4
    yy, xx = np.mgrid[0:height, 0:width]
    return (tile_owner(xx, yy, nshards) == shard) & (xx < ww) & (yy < hh)


def reduce_sum_to_root(sum_tensor, dst: int = 0):
    """One collective per run: element-wise sum of every rank's image into rank `dst`."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(sum_tensor, dst=dst, op=dist.ReduceOp.SUM)
    return sum_tensor


def finalize_mean_srgb(sum_rgba: np.ndarray, spp: int) -> np.ndarray:
    """Host mirror of rt_finalize_sum for tests: srgb(sum / spp) with plain libm pow.

    The device version uses the reference's own pow polynomial; this one is only used to compare
    sharded against unsharded sums, where both sides go through the same function."""
    lin = np.asarray(sum_rgba, dtype=np.float32)[..., :3] / np.float32(spp)
    out = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(np.maximum(lin, 0), 1 / 2.4) - 0.055)
    return np.concatenate([out, np.ones(out.shape[:-1] + (1,), out.dtype)], axis=-1).astype(np.float32)
