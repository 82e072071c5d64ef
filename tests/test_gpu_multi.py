"""ONE context over n devices (tdt_ctx_create_multi) — uploads replicate, a raytracer dispatch is sharded
over the devices' work-groups, gathered to the first device and de-interleaved there.  On the one-GPU test box the device
list repeats device 0 (shares on one GPU, peer-copy transport); the RCCL transport is exercised with a one-device
communicator (n = 1), which still loads librccl, creates the communicator and runs ncclGather."""
import numpy as np
import pytest

import oracle_py
from tdt4230_project_raytracing_amd import host, rt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frame():
    scene = host.Scene.config(2)
    cam = host.camera_reference_pose(480, 270, 16, 8)       # spp >= 16: every share is a two-phase frame
    r = rt.Renderer(scene, cam)
    img = r.render()
    r.close()
    return scene, cam, img


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], [0] * 8])
def test_multi_context_renders_the_single_device_frame(frame, devices):
    scene, cam, img = frame
    r = rt.Renderer(scene, cam, devices=devices)
    try:
        assert r.ctx.device_count() == len(devices)
        first = r.render()
        assert (first.view(np.uint32) == img.view(np.uint32)).all()
        assert r.ctx.multi_transport() == ("rccl" if len(devices) == 1 else "copy")
        again = r.render()                                  # replay: per-share cost order
        assert (again.view(np.uint32) == img.view(np.uint32)).all()
        trace_ms, gather_ms, assemble_ms = r.ctx.multi_timing()
        assert len(trace_ms) == len(devices) and all(t > 0 for t in trace_ms) and gather_ms >= 0 and assemble_ms > 0
        assert r.shader.covered_pixels(cam.image_width + 1, cam.image_height + 1) == int((img[..., 3] == 1).sum())
        # presentation reads the assembled frame too
        single = rt.Renderer(scene, cam)
        single.dispatch()
        assert (r.texture.read_rgba8() == single.texture.read_rgba8()).all()
        single.close()
    finally:
        r.close()


def test_multi_context_camera_move_and_edit_reach_every_device(frame, oracle):
    scene, cam, _ = frame
    scene = host.Scene.config(1)
    scene.blobs[0] = np.concatenate([scene.blobs[0], np.zeros(16 * 32, np.uint32)])
    cam = host.camera_reference_pose(160, 96, 4, 6)
    r = rt.Renderer(scene, cam, devices=[0, 0, 0])
    try:
        assert (r.render().view(np.uint32) == oracle.render(scene, cam, threads=4).view(np.uint32)).all()
        moved = cam.copy()
        moved.origin[0] += 0.05
        moved.lower_left_corner[0] += 0.05
        r.shader.program.set_vector3_f32("camera.origin", moved.origin)                      # camera.rs:78-81
        r.shader.program.set_vector3_f32("camera.lower_left_corner", moved.lower_left_corner)
        assert (r.render().view(np.uint32) == oracle.render(scene, moved, threads=4).view(np.uint32)).all()
        # a voxel edit (octree.rs:170-183) must change the scene replica of every device
        upd = rt.ComputeShader(r.ctx, rt.PROGRAM_OCTREE_UPDATE)
        counter = rt.VertexBufferObject(r.ctx, np.array([scene.counts["cells"]], np.uint32))
        r.ctx.bind_buffer_base(rt.ATOMIC_COUNTER_BUFFER, 0, counter)
        dv = rt.VertexBufferObject(r.ctx, np.zeros(1000, np.float32))
        r.ctx.bind_buffer_base(rt.SHADER_STORAGE_BUFFER, 5, dv)
        delta = np.zeros(500, np.float32)
        delta[:5] = [0.52, 0.45, 0.55, 2.0, 1.0]
        rt.update_vbo(r.ctx, dv, delta, 5, upd)
        d8 = np.zeros((1, 8), np.float32); d8[0, :5] = delta[:5]
        cells, cnt = oracle_py.oracle_octree_update(oracle, scene, d8, scene.counts["cells"], (0, 1, 0))
        assert np.array_equal(r.vbos[0].read(np.uint32), cells) and int(counter.read(np.uint32)[0]) == cnt
        scene.blobs[0] = cells
        assert (r.render().view(np.uint32) == oracle.render(scene, moved, threads=4).view(np.uint32)).all()
    finally:
        r.close()


def test_multi_context_error_behaviour(frame):
    scene, cam, _ = frame
    r = rt.Renderer(scene, cam, devices=[0, 0])
    try:
        with pytest.raises(rt.TdtError) as e:
            r.shader.program.set_i32("camera.nope", 1)
        assert e.value.code == rt.ERR_VARIABLE_NOT_FOUND
        with pytest.raises(rt.TdtError) as e:
            r.shader.set_partition(0, 2)
        assert e.value.code == rt.ERR_INVALID_OPERATION
        with pytest.raises(rt.TdtError) as e:
            r.shader.dispatch_accumulate(64, 64, 1, 0, 1)
        assert e.value.code == rt.ERR_INVALID_OPERATION
        counts = r.shader.dispatch_counted(cam.image_width + 1, cam.image_height + 1)
        one = rt.Renderer(scene, cam)
        assert counts == one.shader.dispatch_counted(cam.image_width + 1, cam.image_height + 1)
        one.close()
    finally:
        r.close()
    with pytest.raises(rt.TdtError):
        rt.Context(devices=[])
    with pytest.raises(rt.TdtError):
        rt.Context(devices=[0, 99])
