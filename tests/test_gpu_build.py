"""SURVEY §8f-1 on the GPU: tdt_octree_build_cells / tdt_octree_build_from_points (Morton sort + prefix sums + breadth-first
emission in HIP) must produce, byte for byte, what the host builder of libtdthost.so produces — for the reference's own
3x3x3 model (regenerated), for its 156 942-voxel monument (the payloads stored in tests/golden/monu1_ply_*.npz), for the
synthetic BASELINE scenes and for random ragged voxel lists with duplicates — and the trace must render through it."""
import json
import os

import numpy as np
import pytest

from octree_util import expand_cells, ply_bytes, points_of, points_of_scene
from test_ply_ingest import cube_edges_ply
from tdt4230_project_raytracing_amd import host, rt

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    c = rt.Context(0)
    yield c
    c.close()


def read_all(vbos, scene):
    out = {}
    for slot in (0, 1, 2, 3, 4, 6, 7):
        nbytes = np.ascontiguousarray(scene.blobs[slot]).nbytes
        vbos[slot].nbytes = nbytes
        out[slot] = vbos[slot].read(np.uint32)
    return out


def assert_same_payloads(ctx, ply, max_iter=100, z_up=True):
    scene = ply.to_scene(max_iter=max_iter, z_up=z_up)                  # the host builder
    vox, mp, keys, rgb = points_of(ply)
    vbos, depth, cc = rt.octree_build_from_points(ctx, vox, mp, keys, rgb, z_up=z_up, max_iter=max_iter)
    assert depth == scene.max_depth and cc == int(scene.blobs[7][2])
    got = read_all(vbos, scene)
    for slot in (0, 1, 2, 3, 4, 6, 7):
        want = np.ascontiguousarray(scene.blobs[slot]).view(np.uint32)
        assert got[slot].size == want.size and (got[slot] == want).all(), f"slot {slot} differs from the host builder"
    return scene, vbos


def test_reference_3x3x3_model(ctx):
    data, _ = cube_edges_ply("\r\n")
    assert_same_payloads(ctx, host.Ply(data))


@pytest.mark.parametrize("seed", range(6))
def test_random_ragged_models(ctx, seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(1, 6000))
    span = int(rng.choice([1, 3, 9, 40, 130, 300]))
    lo = rng.integers(-50, 50, size=3)
    xyz = lo + rng.integers(0, span + 1, size=(n, 3))                  # duplicates are likely for small spans
    if seed % 2:                                                       # solid blocks: uniform subtrees must merge into LEAFs
        g = np.stack(np.meshgrid(*([np.arange(8)] * 3), indexing="ij"), -1).reshape(-1, 3)
        xyz = np.concatenate([xyz, lo + g + 8 * rng.integers(0, 3, size=3)])
    pal = rng.integers(0, 256, size=(int(rng.integers(1, 60)), 3))
    rgb = pal[rng.integers(0, len(pal), size=len(xyz))]
    if seed % 2:
        rgb[-512:] = pal[0]
    assert_same_payloads(ctx, host.Ply(ply_bytes(xyz, rgb)), max_iter=50, z_up=bool(seed & 2))


@pytest.mark.parametrize("cfg", [1, 2, 3])
def test_synthetic_scene_cells(ctx, cfg):
    scene = host.Scene.config(cfg)
    vox = expand_cells(scene.blobs[0], scene.max_depth)
    rng = np.random.default_rng(cfg)
    vox = vox[rng.permutation(len(vox))]                               # any order, the tree is the same
    vbo, n_cells = rt.octree_build_cells(ctx, vox, scene.max_depth)
    want = np.ascontiguousarray(scene.blobs[0]).view(np.uint32)
    assert n_cells * 16 == want.size
    assert (vbo.read(np.uint32) == want).all()


def test_duplicates_last_one_wins_and_out_of_grid_is_dropped(ctx):
    vox = np.array([[1, 1, 1, 3], [0, 0, 0, 1], [1, 1, 1, 7], [9, 0, 0, 2], [0, -1, 0, 2], [1, 1, 1, 5]], np.int32)
    vbo, n_cells = rt.octree_build_cells(ctx, vox, 2)
    cells = vbo.read(np.uint32).reshape(-1, 8, 2)
    assert n_cells == 2 and cells[0, 0].tolist() == [1, 1]             # root octant 0 is a PARENT of cell 1
    assert cells[1, 0].tolist() == [0, 2] and cells[1, 7].tolist() == [4, 2]   # (0,0,0) -> material 0; (1,1,1) -> the LAST: 5 - 1
    assert (cells[0, 1:] == 0).all() and (cells[1, 1:7] == 0).all()
    empty, n0 = rt.octree_build_cells(ctx, np.zeros((0, 4), np.int32), 3)
    assert n0 == 1 and (empty.read(np.uint32) == 0).all()


def test_errors(ctx):
    data, _ = cube_edges_ply("\r\n")
    vox, mp, keys, rgb = points_of(host.Ply(data))
    with pytest.raises(rt.TdtError) as e:
        rt.octree_build_from_points(ctx, vox, mp, keys[:-1], rgb[:-1])
    assert "not in the palette" in str(e.value)
    with pytest.raises(rt.TdtError) as e:
        rt.octree_build_from_points(ctx, vox, mp, keys[::-1].copy(), rgb[::-1].copy())
    assert "ascending" in str(e.value)
    far = vox.copy(); far[0, 0] = 600
    with pytest.raises(rt.TdtError) as e:
        rt.octree_build_from_points(ctx, far, mp, keys, rgb)
    assert "512" in str(e.value)
    many = np.array([[i, 0, 0, i] for i in range(300)], np.int32)
    with pytest.raises(rt.TdtError) as e:
        rt.octree_build_from_points(ctx, many, [0, 0, 0], np.arange(300, dtype=np.uint32), np.zeros((300, 3), np.uint8))
    assert "254" in str(e.value)


def test_reference_monument_builds_and_renders(ctx, oracle):
    """The reference's monu1_point.ply: its built payloads are stored in the render golden; the voxel list is read back off
    them (the 3.2 MB asset itself stays in the reference checkout), rebuilt on the GPU, and the reference's render of it must
    come out of the trace through the GPU-built buffers."""
    z = np.load(os.path.join(GOLDEN, "monu1_ply_320x240_spp2_b6.npz"))
    meta = json.loads(str(z["meta"]))
    blobs = {s: z[f"blob_{s}"] for s in (0, 1, 2, 3, 4, 6, 7)}
    depth = int(blobs[7].view(np.int32)[0])
    vox, mp, keys, rgb = points_of_scene(blobs, depth, min_point=(-29, -52, 0))
    assert len(vox) == 156942 and depth == 7
    vbos, d, cc = rt.octree_build_from_points(ctx, vox, mp, keys, rgb, z_up=True, max_iter=int(blobs[7].view(np.int32)[1]))
    assert d == depth and cc == int(blobs[7].view(np.int32)[2])
    for s in (0, 1, 2, 3, 4, 6, 7):
        want = np.ascontiguousarray(blobs[s]).view(np.uint32)
        vbos[s].nbytes = want.nbytes
        assert (vbos[s].read(np.uint32) == want).all(), f"slot {s}"
    # render through the buffers the GPU built (bound as main.rs:352-448 / octree.rs:67-100 bind theirs)
    cam = host.CameraUniforms()
    for k, v in meta["camera"].items():
        if isinstance(v, list):
            getattr(cam, k)[:] = v
        else:
            setattr(cam, k, v)
    for s in (0, 1, 2, 3, 4, 6, 7):
        ctx.bind_buffer_base(rt.SHADER_STORAGE_BUFFER, s, vbos[s])
    shader = rt.ComputeShader(ctx)
    rt.initial_uniforms(cam, shader.program)
    tex = rt.Texture.new_2d(ctx, cam.image_width, cam.image_height)
    shader.dispatch_compute(cam.image_width + 1, cam.image_height + 1, 1)
    img = tex.read()
    assert (img.view(np.uint32) == z["image"].view(np.uint32)).all()
