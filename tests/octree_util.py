"""Test helpers: turn an indirect-cell octree (the cells payload raytracer.comp reads) back into its voxel list, and write
voxel lists as the ASCII PLY the reference's loader reads."""
import numpy as np

from tdt4230_project_raytracing_amd import host

EMPTY, PARENT, LEAF = 0, 1, 2


def expand_cells(cells_u32, depth):
    """(n, 4) int32 voxels {x, y, z, material + 1} of the tree whose root cell is cell 0 and whose finest level is `depth`
    (a LEAF above the finest level stands for the whole block under it).  Vectorised level by level."""
    cells = np.asarray(cells_u32, np.uint32).reshape(-1, 8, 2)
    cell = np.zeros(1, np.int64)
    base = np.zeros((1, 3), np.int64)
    out = []
    child = np.array([[c >> 2, (c >> 1) & 1, c & 1] for c in range(8)], np.int64)
    for level in range(1, depth + 1):
        nodes = cells[cell]                                   # (m, 8, 2)
        size = 1 << (depth - level)
        pos = (base[:, None, :] * 2 + child[None, :, :])      # (m, 8, 3) in units of `size`
        typ, val = nodes[..., 1], nodes[..., 0]
        leaf = typ == LEAF
        if leaf.any():
            p = pos[leaf] * size
            m = val[leaf].astype(np.int64) + 1
            if size == 1:
                out.append(np.concatenate([p, m[:, None]], axis=1))
            else:
                g = np.stack(np.meshgrid(*([np.arange(size)] * 3), indexing="ij"), axis=-1).reshape(-1, 3)
                q = (p[:, None, :] + g[None, :, :]).reshape(-1, 3)
                out.append(np.concatenate([q, np.repeat(m, g.shape[0])[:, None]], axis=1))
        par = typ == PARENT
        cell = val[par].astype(np.int64)
        base = pos[par]
        if cell.size == 0:
            break
    if not out:
        return np.zeros((0, 4), np.int32)
    return np.concatenate(out).astype(np.int32)


def ply_bytes(xyz, rgb, eol="\r\n"):
    """ASCII PLY as MagicaVoxel writes it (the format ply_point_loader.rs:102-319 reads): integer positions, uchar colours."""
    xyz = np.asarray(xyz, np.int64).reshape(-1, 3)
    rgb = np.asarray(rgb, np.int64).reshape(-1, 3)
    head = ["ply", "format ascii 1.0", "comment : MagicaVoxel @ Ephtracy", f"element vertex {len(xyz)}", "property float x",
            "property float y", "property float z", "property uchar red", "property uchar green", "property uchar blue", "end_header"]
    body = [f"{p[0]} {p[1]} {p[2]} {c[0]} {c[1]} {c[2]}" for p, c in zip(xyz, rgb)]
    return (eol.join(head + body) + eol).encode()


def points_of(ply):
    """What tdt_octree_build_from_points takes, from a parsed host.Ply: voxels {x,y,z,key}, min_point, sorted palette."""
    vox = np.concatenate([ply.positions.astype(np.int32), ply.albedo_keys.view(np.int32)[:, None]], axis=1)
    keys = np.array(sorted(ply.albedos), np.uint32)
    rgb = np.array([ply.albedos[int(k)] for k in keys], np.uint8).reshape(-1, 3)
    return vox, ply.min_point, keys, rgb


def points_of_scene(blobs, depth, min_point=(0, 0, 0), z_up=True):
    """Invert tdt_scene_from_ply for a scene it built: a voxel list / palette that builds exactly these payloads again
    (positions in file axes; the palette is read off the albedo table: rgb = albedo * 255, key = the loader's Cantor pairing)."""
    vox = expand_cells(blobs[0], depth)
    n = 1 << depth
    ext_x = int(vox[:, 0].max() - vox[:, 0].min() + 1)
    ox = (n - ext_x) // 2
    assert vox[:, 0].min() == ox and vox[:, 1].min() == 0 and vox[:, 2].min() == 0
    alb = np.asarray(blobs[2], np.float32).reshape(-1, 3)
    rgb = np.rint(alb.astype(np.float64) * 255.0).astype(np.int64)
    assert (rgb.astype(np.float32) / np.float32(255.0) == alb).all()
    keys = np.array([host.cantor_pair(*[float(c) for c in col]) for col in rgb], np.uint32)
    assert (np.diff(keys.astype(np.int64)) > 0).all()          # materials are numbered in ascending key order
    f = np.zeros((len(vox), 4), np.int32)
    f[:, 0] = vox[:, 0] - ox + min_point[0]
    ya, za = (2, 1) if z_up else (1, 2)                         # octree y <- file axis ya, octree z <- file axis za
    f[:, ya] = vox[:, 1] + min_point[ya]
    f[:, za] = vox[:, 2] + min_point[za]
    f[:, 3] = keys[vox[:, 3] - 1].view(np.int32)
    return f, list(min_point), keys, rgb.astype(np.uint8)


# ---- voxel edits (octree_update.comp) --------------------------------------------------------------------------------
def edit_setup(scene, counter0, delta_floats):
    from tdt4230_project_raytracing_amd import rt
    cam = host.camera_reference_pose(64, 64, 1, 2)
    r = rt.Renderer(scene, cam)
    upd = rt.ComputeShader(r.ctx, rt.PROGRAM_OCTREE_UPDATE)
    counter = rt.VertexBufferObject(r.ctx, np.array([counter0], np.uint32))
    r.ctx.bind_buffer_base(rt.ATOMIC_COUNTER_BUFFER, 0, counter)
    dv = rt.VertexBufferObject(r.ctx, np.ascontiguousarray(delta_floats, np.float32))
    r.ctx.bind_buffer_base(rt.SHADER_STORAGE_BUFFER, 5, dv)
    return r, upd, counter


def written_node(cells, p, depth):
    """Index of the node of the tree AS IT IS that an edit at p (strictly inside a finest-level cell, so treeLookupLeaf's
    float index arithmetic is plain binary digits) writes first: its first EMPTY node, or the last node of a full walk."""
    g = 1 << depth
    q = [int(c * g) for c in p]
    value, index = 0, 0
    for level in range(1, depth):                        # max_depth - 1 levels (octree_update.comp:63)
        sh = depth - level
        index = (((2 * value + ((q[0] >> sh) & 1)) << 1) + ((q[1] >> sh) & 1) << 1) + ((q[2] >> sh) & 1)
        if cells[2 * index + 1] == 0:
            return index
        if cells[2 * index + 1] == 2 and level < depth - 1:
            return None                                  # a LEAF above the last level: the shader follows its material index as
        value = int(cells[2 * index])                    # if it were a cell index, into the top of the tree — a real collision
    return index


def distinct_deltas(rng, n, depth, cells=None):
    """Up to n edits, each in its own cell of the level the walk ends on, and — when `cells` is given — no two of them
    writing the same node of the tree as it is (two edits under one EMPTY node both want to turn it into a PARENT)."""
    g = 1 << (depth - 1)
    idx = rng.permutation(g ** 3)[: (n if cells is None else 60 * n)]
    p = np.stack([idx // (g * g), (idx // g) % g, idx % g], 1).astype(np.float32)
    pos = (p + rng.uniform(0.3, 0.7, size=p.shape).astype(np.float32)) / np.float32(g)
    if cells is not None:
        seen, keep = set(), []
        for i, q in enumerate(pos):
            w = written_node(cells, q, depth)
            if w is not None and w not in seen:
                seen.add(w); keep.append(i)
                if len(keep) == n:
                    break
        pos = pos[keep]
    d = np.zeros((len(pos), 8), np.float32)
    d[:, :3] = pos
    d[:, 3] = rng.integers(0, 3, size=len(pos))
    d[:, 4] = rng.integers(0, 12, size=len(pos))
    return d
