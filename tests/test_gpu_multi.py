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
        with pytest.raises(rt.TdtError) as e:                # a resolve with nothing accumulated
            r.shader.dispatch_resolve(cam.image_width + 1, cam.image_height + 1, 1, 16)
        assert e.value.code == rt.ERR_INVALID_OPERATION
        with pytest.raises(rt.TdtError) as e:                # a pass that continues sums that do not exist
            r.shader.dispatch_accumulate(cam.image_width + 1, cam.image_height + 1, 1, 4, 4)
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


@pytest.mark.parametrize("devices,fail_at", [([0, 0, 0], 2), ([0, 0, 0], 0), ([0] * 5, 3)])
def test_half_launched_frame_is_drained_and_the_next_frame_is_whole(frame, devices, fail_at):
    """tdt_multi.hip: a dispatch that fails at device i after devices < i were launched drains them, resets the frame state
    (no stale ev_gathered wait, no timing of a frame that never assembled) and the next dispatch renders the whole frame."""
    scene, cam, img = frame
    r = rt.Renderer(scene, cam, devices=devices)
    try:
        assert (r.render().view(np.uint32) == img.view(np.uint32)).all()
        r.ctx.multi_fail(fail_at)
        with pytest.raises(rt.TdtError) as e:
            r.dispatch()
        assert e.value.code == rt.ERR_INVALID_OPERATION and "injected" in str(e.value)
        with pytest.raises(rt.TdtError):                     # no frame to time
            r.ctx.multi_timing()
        for _ in range(2):
            assert (r.render().view(np.uint32) == img.view(np.uint32)).all()
        assert len(r.ctx.multi_timing()[0]) == len(devices)
        assert r.ctx.multi_rccl_ranks() == 0                 # shares of one GPU: peer copies, no communicator
    finally:
        r.close()


def test_rccl_ranks_of_a_one_device_communicator(frame):
    scene, cam, img = frame
    r = rt.Renderer(scene, cam, devices=[0])
    try:
        assert (r.render().view(np.uint32) == img.view(np.uint32)).all()
        assert r.ctx.multi_transport() == "rccl" and r.ctx.multi_rccl_ranks() == 1
    finally:
        r.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0]])
@pytest.mark.parametrize("use_carry", [True, False])
def test_progressive_passes_on_a_multi_device_context(frame, devices, use_carry):
    """BASELINE configs[4]'s shape on a node: passes accumulate per device (running sums in the tile buffers, hit-record carry
    beside them), one resolve + gather + assemble at the end: the bits of the one-pass frame when the records are carried."""
    import torch
    scene, cam, img = frame
    dw, dh = cam.image_width + 1, cam.image_height + 1
    spp = cam.samples_per_pixel
    r = rt.Renderer(scene, cam, devices=devices)
    single = rt.Renderer(scene, cam)
    acc = torch.zeros((cam.image_height, cam.image_width, 4), dtype=torch.float32, device="cuda:0")
    carry = torch.zeros((cam.image_height, cam.image_width, 16), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    try:
        for rep in range(2):                                 # (the second round starts over on the buffers of the first)
            for b, n in ((0, 5), (5, 3), (8, spp - 8)):
                r.shader.dispatch_accumulate(dw, dh, 1, b, n, 1 if use_carry else None)
            r.shader.dispatch_resolve(dw, dh, 1, spp)
            got = r.texture.read()
            if use_carry:
                assert (got.view(np.uint32) == img.view(np.uint32)).all()
            else:                                            # without carry: the single-device path without carry is the definition
                t = rt.Texture.wrap_device(single.ctx, acc.data_ptr(), cam.image_width, cam.image_height)
                for b, n in ((0, 5), (5, 3), (8, spp - 8)):
                    single.shader.dispatch_accumulate(dw, dh, 1, b, n, None)
                single.shader.dispatch_resolve(dw, dh, 1, spp)
                assert (got.view(np.uint32) == t.read().view(np.uint32)).all()
        assert (r.render().view(np.uint32) == img.view(np.uint32)).all()      # and an ordinary frame afterwards
    finally:
        r.close(); single.close()
