"""Tile partition of a frame over the GPUs of one node -- the host-side arithmetic.

The frame is cut into 8x8 pixel tiles numbered row-major; rank r of N renders tiles r, r+N, r+2N, ... and
writes them PACKED (tile after tile, 64 pixels x 3 floats, pixel k of a tile = row k//8, column k%8) --
that is what crt_render_tiles_device produces.  One all_gather of the equal-sized packed buffers collects
the frame; crt_unpack_tiles_device scatters it into row-major order on rank 0.  The numpy functions here
are the reference for those two kernels and are what the CPU-side (gloo) test of the multi-GPU path uses.
"""
import numpy as np

TILE = 8


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def tiles_of_rank(width, height, rank, world):
    tx, ty = tile_grid(width, height)
    return np.arange(rank, tx * ty, world, dtype=np.int64)


def tiles_per_rank(width, height, world):
    """Packed buffer length per rank, in tiles: the same for every rank (padded) so one all_gather fits."""
    tx, ty = tile_grid(width, height)
    return (tx * ty + world - 1) // world


def pack_tiles(frame, rank, world):
    """What rank `rank` contributes: float32 [tiles_per_rank, 64, 3], zero where a tile is outside the frame."""
    h, w = frame.shape[:2]
    tx, _ = tile_grid(w, h)
    ids = tiles_of_rank(w, h, rank, world)
    out = np.zeros((tiles_per_rank(w, h, world), 64, 3), dtype=np.float32)
    for j, t in enumerate(ids):
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        block = frame[y0:y0 + TILE, x0:x0 + TILE]
        full = np.zeros((TILE, TILE, 3), dtype=np.float32)
        full[:block.shape[0], :block.shape[1]] = block
        out[j] = full.reshape(64, 3)
    return out


def unpack_tiles(gathered, width, height, world):
    """Inverse of pack_tiles over all ranks: gathered float32 [world, tiles_per_rank, 64, 3] -> frame [H, W, 3]."""
    tx, ty = tile_grid(width, height)
    frame = np.zeros((ty * TILE, tx * TILE, 3), dtype=np.float32)
    for t in range(tx * ty):
        part, local = t % world, t // world
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        frame[y0:y0 + TILE, x0:x0 + TILE] = gathered[part, local].reshape(TILE, TILE, 3)
    return frame[:height, :width].copy()
