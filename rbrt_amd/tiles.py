"""Pixel-tile sharding index math (host side, numpy) — the same layout the kernels use.

An image is cut into 8x8 tiles; tile NUMBER t belongs to rank t % world, where number t is the tile in tile row
ty = t // tiles_x at tile column (t % tiles_x + SKEW * ty) % tiles_x (include/rbrt_hip.h "How tiles are dealt to
ranks": every tile row is rotated a little more than the one above, so that a rank's tiles are not fixed columns); a rank's
packed buffer holds its tiles in ascending number, 64 pixel slots per tile (slot p = (y%8)*8 +
x%8; slots outside a ragged image edge are padding). bench.py gathers the packed buffers with one
RCCL gather and rank 0 de-interleaves them (rbrt_hip_unpack_tiles on the GPU; `unpack` here is the
numpy restatement used by the CPU tests).
"""
from __future__ import annotations

import numpy as np

TILE = 8
SKEW = 3  # RBRT_TILE_SKEW


def tile_number(ty, tx, tiles_x: int):
    """Number of the tile in tile row ty at tile column tx (ints or numpy arrays): rbrt_hip_tile_number."""
    return ty * tiles_x + (tx + tiles_x - (SKEW * ty) % tiles_x) % tiles_x


def n_tiles(width: int, height: int) -> int:
    return ((width + TILE - 1) // TILE) * ((height + TILE - 1) // TILE)


def local_tiles(width: int, height: int, rank: int, world: int) -> int:
    n = n_tiles(width, height)
    return (n - rank + world - 1) // world if rank < n else 0


def packed_pixels(width: int, height: int, rank: int, world: int) -> int:
    return local_tiles(width, height, rank, world) * TILE * TILE


def _index_maps(width: int, height: int, world: int):
    """For every image pixel: (owner rank, slot index inside the owner's packed buffer)."""
    tiles_x = (width + TILE - 1) // TILE
    ys, xs = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    ty, tx = ys // TILE, xs // TILE
    tile = tile_number(ty, tx, tiles_x)
    return tile % world, (tile // world) * TILE * TILE + (ys % TILE) * TILE + xs % TILE


def pack(image: np.ndarray, rank: int, world: int) -> np.ndarray:
    """image [H,W,C] -> this rank's packed buffer [packed_pixels, C] (padding slots are zero)."""
    h, w, c = image.shape
    out = np.zeros((packed_pixels(w, h, rank, world), c), image.dtype)
    owner, slot = _index_maps(w, h, world)
    mine = owner == rank
    out[slot[mine]] = image[mine]
    return out


def unpack(packed_per_rank, width: int, height: int) -> np.ndarray:
    """list of per-rank packed buffers (rank order) -> image [H,W,C]."""
    world = len(packed_per_rank)
    c = packed_per_rank[0].shape[1]
    out = np.zeros((height, width, c), packed_per_rank[0].dtype)
    owner, slot = _index_maps(width, height, world)
    for r in range(world):
        mine = owner == r
        out[mine] = packed_per_rank[r][slot[mine]]
    return out
