"""Live comparison against the reference shader on llvmpipe — only where /root/reference and the
harness exist (the build container); skipped on the GPU box.  Complements the committed fixtures
with freshly generated scenes."""
import numpy as np
import pytest

from tdt4230_project_raytracing_amd import host


@pytest.mark.parametrize("kind,depth,seed", [(0, 3, 11), (0, 4, 12), (1, 5, 13), (2, 6, 14)])
def test_oracle_vs_reference_shader_fresh_scene(oracle, glref, kind, depth, seed):
    scene = host.Scene.generate(kind, depth, 1 << 14, 100, seed)
    cam = host.camera_reference_pose(96, 64, 2, 5)
    ref = glref.render(scene, cam)
    got = oracle.render(scene, cam, threads=4)
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.parametrize("kind,depth,seed,cc", [(1, 5, 21, 100000), (2, 6, 22, 99999), (0, 7, 23, 100000), (1, 6, 24, 3000)])
def test_oracle_vs_reference_shader_under_other_cell_counts(oracle, glref, kind, depth, seed, cc):
    """cell_count as the reference's own host passes it (100000: not a power of two) and neighbours of it: the index arithmetic of
    treeLookup then reads the previous cell at the bottom of many cells — live against the shader on fresh scenes, rays along cell
    boundaries included."""
    scene = host.scene_with_cell_count(host.Scene.generate(kind, depth, 1 << 20, 100, seed), cc, 1000)
    for origin in (None, (0.0, 0.0, -0.5)):
        cam = host.camera_reference_pose(96, 64, 2, 5) if origin is None else host.camera_build(
            90.0, 96, aspect_ratio=1.5, viewport_height=2.0, origin=origin, samples_per_pixel=2, max_bounce=5)
        ref = glref.render(scene, cam)
        got = oracle.render(scene, cam, threads=4)
        assert (got.view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.parametrize("cfg,origin,yaw,pitch,fov", [(0, (0.2, -0.1, 0.25), -17.0, 8.0, 60.0), (2, (0.62, 0.3, 0.7), 43.0, -20.0, 60.0),
                                                       (1, (-0.7, 0.9, 0.6), -57.0, -40.0, 60.0)])
def test_oracle_vs_reference_shader_turned_camera(oracle, glref, cfg, origin, yaw, pitch, fov):
    """Turned cameras (camera.rs:68-82: every component of horizontal / vertical / lower_left_corner non-zero but horizontal.y, which the
    controller keeps at 0), inside the octree and outside looking in: the uniforms of the host's controller mirror, whatever their
    last bit, through the shader and the oracle.  (The controller turns by 2 * angle * turn_rate radians, turn_rate 0.025.)"""
    scene = host.Scene.config(cfg)
    c = host.Camera(fov, 96, aspect_ratio=1.5, viewport_height=2.0, origin=origin, samples_per_pixel=3, max_bounce=5)
    c.turn_yaw(float(np.radians(yaw)) / 0.05)
    c.turn_pitch(float(np.radians(pitch)) / 0.05)
    cam = c.uniforms()
    assert sum(abs(x) > 1e-3 for v in (cam.horizontal, cam.vertical, cam.lower_left_corner) for x in v) >= 8
    ref = glref.render(scene, cam)
    got = oracle.render(scene, cam, threads=4)
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()


def test_reference_work_group_size_and_layout(glref):
    """compute_shader.rs:18 queries {32,32,1}; the `layout(shared)` blocks are packed like std430,
    which is what the tightly packed host payloads assume (SURVEY.md §7 hard parts)."""
    import ctypes
    glref.program()
    gs = (ctypes.c_int * 3)()
    glref.L.glref_group_size(gs)
    assert list(gs) == [32, 32, 1]
    expect = {b"indirect_cells[0].value": (0, 8), b"indirect_cells[0].type": (4, 8),
              b"octree_floats[0].scale": (16, None), b"octree_floats[0].inv_scale": (20, None),
              b"octree_floats[0].inv_cell_count": (24, None), b"octree_ints[0].max_iter": (4, None),
              b"octree_ints[0].cell_count": (8, None), b"materials[0].albedo_index": (8, 12), b"albedos[0].z": (8, 12)}
    for name, (off, stride) in expect.items():
        out = (ctypes.c_int * 2)()
        assert glref.L.glref_buffer_variable(name, out) == 0, name
        assert out[0] == off, (name, out[0])
        if stride is not None:
            assert out[1] == stride, (name, out[1])


def test_presentation_vs_reference_quad_pass_fresh_frames(glref):
    """quad.vert / quad.frag on llvmpipe against tdt_present_rgba8, on frames the fixtures do not hold."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py
    rng = np.random.default_rng(99)
    for shape in ((33, 70), (64, 64), (5, 257)):
        img = rng.uniform(-0.2, 1.2, shape + (4,)).astype(np.float32)
        img[rng.random(shape) < 0.05] = np.float32("nan")
        got = host.present_rgba8(img, top_down=False)
        assert (got == oracle_py.glref_present(img)).all(), shape
