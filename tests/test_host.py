"""Host-side restatements: camera uniforms (camera.rs:135-196), octree payloads (octree.rs:40-100),
the demo scene literal (main.rs:235-463) and the synthetic generators."""
import numpy as np
import pytest

from tdt4230_project_raytracing_amd import host, tiles


def test_reference_pose_uniforms():
    cam = host.camera_reference_pose(1280, 720, 4, 6)       # main.rs:26,165-168 + camera.ron:3-4
    assert (cam.image_width, cam.image_height) == (1280, 720)
    aspect = np.float32(1280) / np.float32(720)
    assert list(cam.horizontal) == [float(aspect * np.float32(2.0)), 0.0, 0.0]
    assert list(cam.vertical) == [0.0, 2.0, 0.0]
    assert list(cam.origin) == [0.0, float(np.float32(-0.1)), float(np.float32(-0.3))]
    llc = [np.float32(0) - np.float32(cam.horizontal[0]) * np.float32(.5), np.float32(-0.1) - np.float32(1.0), np.float32(-0.3) - np.float32(1.0)]
    assert list(cam.lower_left_corner) == [float(v) for v in llc]
    assert (cam.samples_per_pixel, cam.max_bounce) == (4, 6)


def test_camera_builder_defaults():
    cam = host.camera_build(90.0, 1600)                      # camera.rs:136-157 defaults
    assert cam.image_height == 900                           # (1600 / (16/9)) as i32
    assert (cam.samples_per_pixel, cam.max_bounce) == (10, 3)
    assert list(cam.origin) == [0.0, 0.0, 0.0]
    assert cam.vertical[1] == pytest.approx(2.0, abs=1e-6)


@pytest.mark.parametrize("w,h", [(256, 256), (1280, 720), (1920, 1080), (3840, 2160), (7680, 4320), (320, 180)])
def test_image_height_round_trip(w, h):
    assert host.camera_reference_pose(w, h, 1, 1).image_height == h


def test_demo_scene_literal():
    s = host.Scene.demo()
    assert s.blobs[0].size == 100144                         # 304 u32 + (100000 - 160) zeros, main.rs:339-341
    assert s.blobs[0][:16].tolist() == [1, 1, 1, 0, 1, 0, 10, 1, 1, 0, 1, 0, 1, 0, 1, 1]     # cell 0, main.rs:240-244
    assert s.blobs[0][9 * 16:10 * 16].tolist() == [0, 2, 2, 2, 0, 2, 1, 2, 0, 2, 3, 2, 0, 2, 0, 2]   # cell 9
    assert s.blobs[0][18 * 16:19 * 16].tolist() == [7, 2, 2, 2, 6, 2, 1, 2, 0, 2, 3, 2, 0, 2, 5, 2]  # cell 18
    assert not s.blobs[0][19 * 16:].any()
    assert s.blobs[1].size == 39 and s.blobs[2].size == 21 and s.blobs[3].size == 4 and s.blobs[4].size == 1
    assert s.blobs[6].tolist() == [-0.5, -0.5, -1.0, 0.0, 1.0, 1.0, float(np.float32(1.0) / np.float32(100000))]
    assert s.blobs[7].tolist() == [10, 100, 100000]
    assert s.counts["cells"] == 19 and s.counts["leaves"] == 16


def test_generators_are_deterministic_and_well_formed():
    a = host.Scene.generate(host.SCENE_TERRAIN, 6, 1 << 16, 100, 0x5EED0003)
    b = host.Scene.config(2)
    for slot in host.SLOTS:
        assert np.array_equal(a.blobs[slot], b.blobs[slot])
    cells = b.blobs[0].reshape(-1, 8, 2)
    n = b.counts["cells"]
    assert cells.shape[0] == n
    types = cells[..., 1]
    assert set(np.unique(types)) <= {0, 1, 2}
    parents = cells[..., 0][types == 1]
    assert parents.min() >= 1 and parents.max() == n - 1 and np.unique(parents).size == parents.size   # a tree: every cell referenced once
    assert (np.diff(parents.reshape(-1)) > 0).all()          # breadth-first: children indices ascend in emission order
    leaves = cells[..., 0][types == 2]
    assert leaves.max() < b.counts["materials"]
    assert b.cell_count == 1 << 16 and b.max_depth == 6 and b.max_iter == 100
    assert b.blobs[6][6] == np.float32(1.0) / np.float32(1 << 16)


def test_generator_rejects_bad_parameters():
    with pytest.raises(RuntimeError):
        host.Scene.generate(host.SCENE_TERRAIN, 6, 1000, 100, 1)       # cell_count must be a power of two
    with pytest.raises(RuntimeError):
        host.Scene.generate(host.SCENE_TERRAIN, 8, 64, 100, 1)         # does not fit cell_count


@pytest.mark.parametrize("W,H,dw,dh,expect", [
    (1280, 720, 1281, 721, (1280, 704)),    # main.rs:579 at the shipped window: rows 704..719 never written
    (1920, 1080, 1921, 1081, (1920, 1056)),
    (3840, 2160, 3841, 2161, (3840, 2144)),
    (256, 256, 257, 257, (256, 256)),
    (20, 20, 21, 21, (20, 20)),             # fewer than one group: max(.., 1)
    (100, 50, 65, 33, (64, 32)),
])
def test_dispatch_cover_arithmetic(W, H, dw, dh, expect):
    assert tiles.cover(W, H, dw, dh) == expect


def test_tile_partition_round_trip():
    rng = np.random.default_rng(5)
    W, H = 200, 120
    cw, ch = tiles.cover(W, H, W + 1, H + 1)          # (192, 96)
    img = np.zeros((H, W, 4), np.float32)
    img[:ch, :cw] = rng.random((ch, cw, 4), dtype=np.float32)
    for world in (1, 2, 3, 8):
        total = tiles.tile_grid(cw, ch)[2]
        cap = tiles.tiles_per_rank(total, world)
        g = np.stack([tiles.pack_tiles(img, cw, ch, r, world, cap) for r in range(world)])
        assert sum(tiles.owned_pixels(cw, ch, r, world) for r in range(world)) == cw * ch
        assert np.array_equal(tiles.assemble(g, W, H, cw, ch, world), img)
