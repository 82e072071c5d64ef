"""Cameras OUTSIDE the octree: the miss pre-pass (csrc/tdt_rt.hip miss_prepass_kernel) finishes the pixels whose primary rays
all miss the root cube and takes them out of the hand-out order of the frame's launches.  Same bits as the oracle, as the
trace without the pre-pass, across schedules (one pass, two-phase, replay), partitions and a multi-device context."""
import numpy as np
import pytest

from tdt4230_project_raytracing_amd import host, rt

pytestmark = pytest.mark.gpu


def _eq(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)).all()


def _cam(W, H, spp, origin, yaw=0.0, pitch=0.0, fov=90.0, bounce=6):
    c = host.Camera(fov, W, aspect_ratio=np.float32(W) / np.float32(H), origin=origin, viewport_height=2.0, samples_per_pixel=spp, max_bounce=bounce)
    if yaw:
        c.turn_yaw(yaw)
    if pitch:
        c.turn_pitch(pitch)
    return c.uniforms()


POSES = [
    ("in front, looking in", (0.0, 0.2, 0.9), 0.0, -8.0, 90.0),
    ("corner, narrow lens", (1.3, 1.0, 0.7), -48.0, -28.0, 40.0),
    ("grazing a face", (-0.5005, 0.1, 0.4), 1.5, 0.0, 90.0),
    ("looking away: nothing in view", (0.0, 0.0, 1.5), 180.0, 0.0, 60.0),
    ("on the boundary plane", (0.5, 0.0, -0.5), 90.0, 0.0, 90.0),
]


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("pose", POSES, ids=[p[0] for p in POSES])
@pytest.mark.parametrize("spp", [3, 16])
def test_outside_cameras_equal_the_oracle(oracle, cfg, pose, spp, monkeypatch):
    _, origin, yaw, pitch, fov = pose
    scene = host.Scene.config(cfg)
    cam = _cam(136, 100, spp, origin, yaw, pitch, fov)           # (136 x 100: the dispatch covers 128 x 96 of it)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        first, again = r.render(), r.render()                       # history-free (two-phase at 16 spp), then the replay
    finally:
        r.close()
    assert _eq(first, ref) and _eq(again, ref)
    monkeypatch.setenv("TDT_NO_PREPASS", "1")
    r = rt.Renderer(scene, cam)
    try:
        assert _eq(r.render(), ref)
    finally:
        r.close()


def test_prepass_finishes_the_missing_pixels(oracle):
    """Looking away from the octree every pixel is finished by the pre-pass: the frame equals the oracle's sky."""
    scene = host.Scene.config(2)
    cam = _cam(256, 160, 16, (0.0, 0.0, 1.5), yaw=180.0)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        got = r.render()
        counts = r.shader.dispatch_counted(cam.image_width + 1, cam.image_height + 1, 1)
    finally:
        r.close()
    assert _eq(got, ref) and counts["lambertian"] + counts["metal"] + counts["dielectric"] == 0
    assert (got[:160, :, 3] == 1).all()


@pytest.mark.parametrize("world", [2, 3])
def test_prepass_with_partitions_and_tile_buffers(oracle, world):
    import torch
    from tdt4230_project_raytracing_amd import tiles
    scene = host.Scene.config(2)
    cam = _cam(200, 120, 16, (0.9, 0.6, 0.8), yaw=-45.0, pitch=-20.0, fov=60.0)
    W, H = cam.image_width, cam.image_height
    dw, dh = W + 1, H + 1
    cw, ch = tiles.cover(W, H, dw, dh)
    total = tiles.tile_grid(cw, ch)[2]
    cap = tiles.tiles_per_rank(total, world)
    ref = oracle.render(scene, cam, threads=8)
    gathered = torch.zeros((world, cap, 32, 32, 4), dtype=torch.float32, device="cuda:0")
    full = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    last = None
    for rank in range(world):
        r = rt.Renderer(scene, cam, rank=rank, world=world, image_ptr=gathered[rank].data_ptr(), tile_buffer_tiles=cap)
        for _ in range(2):
            r.dispatch()
        r.ctx.finish()
        if last is not None:
            last.close()
        last = r
    tex = rt.Texture.wrap_device(last.ctx, full.data_ptr(), W, H, bind=False)
    last.shader.assemble_tiles(gathered.data_ptr(), world, cap, tex, dw, dh)
    last.ctx.finish()
    got = full.cpu().numpy()
    last.close()
    assert _eq(got[:ch, :cw], ref[:ch, :cw])


def test_prepass_on_a_multi_device_context_and_scaled_octree(oracle):
    scene = host.Scene.config(2)
    of = scene.blobs[6].copy()
    s = np.float32(2.0)
    of[0:3] = np.array([-0.5, -0.5, -1.0], np.float32) * s + np.array([0.25, -0.125, 0.5], np.float32)
    of[4] = s
    of[5] = np.float32(1.0) / s
    scene.blobs[6] = of
    cam = _cam(160, 96, 16, (3.0, 1.0, 2.5), yaw=-50.0, pitch=-15.0, fov=50.0)
    ref = oracle.render(scene, cam, threads=8)
    assert (ref[..., :3] != ref[0, 0, :3]).any()
    for devices in (None, [0, 0, 0]):
        r = rt.Renderer(scene, cam, devices=devices)
        try:
            assert _eq(r.render(), ref) and _eq(r.render(), ref)
        finally:
            r.close()


@pytest.mark.parametrize("cfg,spp,origin", [(2, 16, None), (0, 4, None), (2, 16, (0.9, 0.6, 0.8)), (3, 3, (1.4, 0.2, 0.3))])
def test_still_camera_frames_reuse_the_order(oracle, cfg, spp, origin, monkeypatch):
    """From the third identical frame on the sort is skipped and the previous hand-out order is reused (launch(): order_exact) — with
    and without a filtered order (camera outside), after a two-phase first frame or a one-pass one; then a moved camera, then still again."""
    scene = host.Scene.config(cfg)
    cam = host.camera_reference_pose(168, 100, spp, 6) if origin is None else _cam(168, 100, spp, origin, yaw=-130.0 if cfg == 2 else -100.0, pitch=-20.0, fov=70.0)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        for _ in range(5):
            assert _eq(r.render(), ref)
        moved = cam.copy()
        moved.origin[1] += 0.01
        moved.lower_left_corner[1] += 0.01
        rt.initial_uniforms(moved, r.shader.program)
        ref2 = oracle.render(scene, moved, threads=8)
        for _ in range(4):
            assert _eq(r.render(), ref2)
    finally:
        r.close()


@pytest.mark.parametrize("env", [{"TDT_NO_COST_ORDER": "1"}, {"TDT_NO_TWO_PHASE": "1"}, {"TDT_NO_ORDER_REUSE": "1"}, {"TDT_NO_SPECIALISE": "1"}])
def test_prepass_under_the_schedule_switches(oracle, env, monkeypatch):
    """The filtered hand-out order in image order (no cost feedback at all), without two-phase frames, with a sort every frame, and
    in front of the general kernel: same bits."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    scene = host.Scene.config(2)
    cam = _cam(200, 136, 16, (0.9, 0.6, 0.8), yaw=-130.0, pitch=-20.0, fov=70.0)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        for _ in range(4):
            assert _eq(r.render(), ref)
    finally:
        r.close()
