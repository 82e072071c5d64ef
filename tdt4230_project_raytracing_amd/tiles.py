"""Host-side arithmetic of the multi-GPU tile partition (numpy; mirrors csrc/tdt_rt.hip's
cover_of / tiles_of / assemble_kernel so the N > 1 path can be sized and unit-tested without a GPU).

The covered image of ComputeShader::dispatch_compute(w, h, d) (compute_shader.rs:28-38) is cut
into the reference's own 32x32 work-groups, numbered row-major t = gy * tiles_x + gx; rank r of
`world` owns the groups with t % world == r and stores them packed as a tile buffer
[k][32][32][4] with k = 0,1,.. <-> t = r + k * world.
"""
import numpy as np

GROUP = 32


def cover(image_width, image_height, dispatch_w, dispatch_h):
    """Pixels a dispatch covers: groups = max(dim // 32, 1) by floor division; stores outside the image are dropped."""
    gx = max(dispatch_w // GROUP, 1)
    gy = max(dispatch_h // GROUP, 1)
    return max(min(gx * GROUP, image_width), 0), max(min(gy * GROUP, image_height), 0)


def tile_grid(cover_w, cover_h):
    tx = -(-cover_w // GROUP)
    ty = -(-cover_h // GROUP)
    return tx, ty, tx * ty


def owned_tiles(total, rank, world):
    return (total - rank + world - 1) // world if total > rank else 0


def tiles_per_rank(total, world):
    """Tile-buffer capacity every rank allocates (what rank 0 owns: the most)."""
    return -(-total // world)


def owned_pixels(cover_w, cover_h, rank, world):
    tx, ty, total = tile_grid(cover_w, cover_h)
    n = 0
    for k in range(owned_tiles(total, rank, world)):
        t = rank + k * world
        gx, gy = t % tx, t // tx
        n += min(GROUP, cover_w - gx * GROUP) * min(GROUP, cover_h - gy * GROUP)
    return n


def pack_tiles(image, cover_w, cover_h, rank, world, capacity=None):
    """Full image [H][W][4] -> this rank's tile buffer [capacity][32][32][4] (zero padded)."""
    tx, ty, total = tile_grid(cover_w, cover_h)
    n = owned_tiles(total, rank, world)
    cap = capacity if capacity is not None else n
    buf = np.zeros((cap, GROUP, GROUP, 4), np.float32)
    for k in range(n):
        t = rank + k * world
        gx, gy = t % tx, t // tx
        w = min(GROUP, cover_w - gx * GROUP)
        h = min(GROUP, cover_h - gy * GROUP)
        buf[k, :h, :w] = image[gy * GROUP:gy * GROUP + h, gx * GROUP:gx * GROUP + w]
    return buf


def assemble(gathered, image_width, image_height, cover_w, cover_h, world, out=None):
    """Gathered tile buffers [world][capacity][32][32][4] -> full image (pixels outside the cover untouched)."""
    tx, ty, total = tile_grid(cover_w, cover_h)
    img = out if out is not None else np.zeros((image_height, image_width, 4), np.float32)
    for t in range(total):
        r, k = t % world, t // world
        gx, gy = t % tx, t // tx
        w = min(GROUP, cover_w - gx * GROUP)
        h = min(GROUP, cover_h - gy * GROUP)
        img[gy * GROUP:gy * GROUP + h, gx * GROUP:gx * GROUP + w] = gathered[r, k, :h, :w]
    return img
