"""The octree -> voxel-list helper the GPU builder tests rest on, checked on the CPU against the host builder's own counts."""
import numpy as np
import pytest

from octree_util import expand_cells, ply_bytes, points_of, points_of_scene
from tdt4230_project_raytracing_amd import host


@pytest.mark.parametrize("cfg", [1, 2])
def test_expand_counts_and_range(cfg):
    scene = host.Scene.config(cfg)
    vox = expand_cells(scene.blobs[0], scene.max_depth)
    assert len(vox) == scene.counts["voxels"]
    n = 1 << scene.max_depth
    assert vox[:, :3].min() >= 0 and vox[:, :3].max() < n and vox[:, 3].min() >= 1
    assert len(np.unique(vox[:, :3], axis=0)) == len(vox)


def test_ply_round_trip_through_the_host_builder():
    rng = np.random.default_rng(3)
    xyz = rng.integers(-7, 9, size=(300, 3))
    pal = rng.integers(0, 256, size=(5, 3))
    rgb = pal[rng.integers(0, 5, size=300)]
    ply = host.Ply(ply_bytes(xyz, rgb))
    scene = ply.to_scene(max_iter=64)
    vox, mp, keys, prgb = points_of(ply)
    assert len(vox) == 300 and mp == [int(v) for v in xyz.min(axis=0)]
    back, mp2, keys2, rgb2 = points_of_scene(scene.blobs, scene.max_depth, min_point=mp)
    # duplicates collapsed (the last one wins), otherwise the same set of voxels with the same colour keys
    last = {}
    for p, k in zip(xyz.tolist(), vox[:, 3].tolist()):
        last[tuple(p)] = k
    assert {tuple(r[:3]): r[3] for r in back.tolist()} == last
    assert set(keys2.tolist()) <= set(keys.tolist())
