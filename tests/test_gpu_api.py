"""The C ABI on a real GPU: partition + tile buffers + assemble, progressive accumulate/resolve,
error behaviour of the GL-wrapper stand-ins, and the committed reference renders."""
import ctypes
import glob
import json
import os

import numpy as np
import pytest

from tdt4230_project_raytracing_amd import host, rt, tiles

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _scene(z, spec):
    if spec[0] == "config":
        return host.Scene.config(spec[1])
    if spec[0] == "config_cc":    # the reference host's conventions on a BASELINE scene: (config, cell_count, zero nodes appended)
        return host.scene_with_cell_count(host.Scene.config(spec[1]), spec[2], spec[3])
    if spec[0] == "ply":          # built from the reference's model file when the fixture was made: payloads stored
        return host.Scene({int(k[5:]): z[k] for k in z.files if k.startswith("blob_")}, name=spec[1])
    return host.Scene.generate(*spec[1:])


def _camera(meta):
    """main.rs:165-168's camera for the fixture's size, optionally moved to the fixture's origin; a turned camera (round 4): the
    uniforms the fixture stores, as the reference shader was sent them."""
    if meta.get("camera_explicit"):
        u = host.CameraUniforms()
        for k, v in meta["camera"].items():
            if isinstance(v, list):
                getattr(u, k)[:] = v
            else:
                setattr(u, k, v)
        return u
    if meta.get("origin") is None:
        return host.camera_reference_pose(meta["W"], meta["H"], meta["spp"], meta["max_bounce"])
    aspect = float(np.float32(meta["W"]) / np.float32(meta["H"]))
    return host.camera_build(90.0, meta["W"], aspect_ratio=aspect, viewport_height=2.0, origin=meta["origin"],
                             samples_per_pixel=meta["spp"], max_bounce=meta["max_bounce"])
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if "math_table" not in p and not os.path.basename(p).startswith(("edit_", "present_")))


def _eq(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.parametrize("name", CASES)
def test_gpu_bit_exact_vs_reference_render(name):
    """HIP path vs the image the reference shader itself wrote (llvmpipe), same scene and uniforms."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    spec = meta["scene"]
    scene = _scene(z, spec)
    cam = _camera(meta)
    r = rt.Renderer(scene, cam)
    try:
        img = r.render()
    finally:
        r.close()
    if meta["crop"]:
        x0, y0, w, h = meta["crop"]
        img = img[y0:y0 + h, x0:x0 + w]
    golden = z["image"]
    eq = (img.view(np.uint32) == golden.view(np.uint32)).all(axis=2)
    assert eq.all(), f"{int((~eq).sum())} pixels differ (max |d| {np.nanmax(np.abs(img - golden)):.3g}; tolerance of the north star: 1e-4)"


@pytest.mark.parametrize("name", [n for n in CASES if "_turned_" in n or "_rolled_" in n])
@pytest.mark.parametrize("switch", ["TDT_NO_PREPASS", "TDT_NO_TWO_PHASE", "TDT_NO_SPECIALISE"])
def test_gpu_turned_cameras_vs_reference_render_under_switches(name, switch, monkeypatch):
    """The turned-camera fixtures (every component of horizontal / vertical / lower_left_corner non-zero, except horizontal.y, which the
    reference's controller keeps at 0 — camera.rs:70 — and only the rolled fixture sets) against the reference render with the miss pre-pass off (every pixel through the trace kernel), in one pass, and through the general kernel; and the
    replay of the frame under the defaults."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    scene, cam, golden = _scene(z, meta["scene"]), _camera(meta), z["image"]
    comps = [c for k in ("horizontal", "vertical", "lower_left_corner") for c in meta["camera"][k]]
    assert sum(abs(c) > 1e-3 for c in comps) >= (9 if "_rolled_" in name else 8), "not a turned camera"
    r = rt.Renderer(scene, cam)
    try:
        r.render()
        assert _eq(r.render(), golden), "replay under the defaults"
    finally:
        r.close()
    monkeypatch.setenv(switch, "1")
    r = rt.Renderer(scene, cam)
    try:
        assert _eq(r.render(), golden), switch
    finally:
        r.close()


def test_pass_statistics_are_refused_by_the_product_library():
    """tdt_debug_stats belongs to -DTDT_STATS builds (tools/loss_budget.py); the product library must say so, not return zeros."""
    if os.environ.get("TDT_LIB"):
        pytest.skip("an alternative build is loaded")
    ctx = rt.Context()
    try:
        with pytest.raises(rt.TdtError) as e:
            ctx.stats()
        assert e.value.code == rt.ERR_INVALID_OPERATION and "TDT_STATS" in str(e.value)
    finally:
        ctx.close()


def test_resolve_on_its_own_resolves_every_pixel():
    """tdt_dispatch_resolve is main()'s sqrt / clamp / store for EVERY covered pixel (include/tdt_rt.h), whatever alpha the caller's
    running sums carry — only the library's own resolve of a frame whose miss pre-pass ran skips the pixels that pass finished."""
    import torch
    scene = host.Scene.config(1)
    cam = host.camera_reference_pose(64, 64, 4, 2)
    buf = torch.empty((64, 64, 4), dtype=torch.float32, device="cuda:0")
    buf[..., 0], buf[..., 1], buf[..., 2], buf[..., 3] = 1.0, 0.25, 9.0, 1.0      # "sums" with a non-zero alpha
    torch.cuda.synchronize()
    r = rt.Renderer(scene, cam, image_ptr=buf.data_ptr())
    try:
        r.shader.dispatch_resolve(65, 65, 1, 4)
        r.ctx.finish()
    finally:
        r.close()
    got = buf.cpu().numpy()
    assert np.allclose(got[..., 0], 0.5) and np.allclose(got[..., 1], 0.25) and (got[..., 2] == 1.0).all() and (got[..., 3] == 1.0).all()


def test_first_moved_frame_after_a_still_camera_keeps_a_prior(oracle):
    """Frames of a still camera reuse their hand-out order (no sort, no cost stores) from the third on; the first frame after that with
    other inputs must still be the right pixels — and is ordered from the still frames' sums (bench: reference_default.still_then_move_ms)."""
    scene = host.Scene.config(2)
    c = host.Camera(90.0, 160, aspect_ratio=160 / 96, origin=(0.0, -0.1, -0.3), viewport_height=2.0, samples_per_pixel=4, max_bounce=5)
    r = rt.Renderer(scene, c.uniforms())
    try:
        for _ in range(5):
            still = r.render()
        c.translate("Front", 0.5); c.turn_yaw(3.0)
        rt.initial_uniforms(c.uniforms(), r.shader.program)
        moved = r.render()
        again = r.render()
    finally:
        r.close()
    assert _eq(still, oracle.render(scene, host.Camera(90.0, 160, aspect_ratio=160 / 96, origin=(0.0, -0.1, -0.3), viewport_height=2.0, samples_per_pixel=4, max_bounce=5).uniforms(), threads=8))
    ref = oracle.render(scene, c.uniforms(), threads=8)
    assert _eq(moved, ref) and _eq(again, ref)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_partition_tile_buffers_and_assemble(oracle, world):
    """Every rank's tile buffer, gathered and de-interleaved by tdt_assemble_tiles, equals the one-rank image."""
    import torch
    scene = host.Scene.config(2)
    W, H = 200, 120
    cam = host.camera_reference_pose(W, H, 2, 6)
    dw, dh = W + 1, H + 1
    cw, ch = tiles.cover(W, H, dw, dh)
    total = tiles.tile_grid(cw, ch)[2]
    cap = tiles.tiles_per_rank(total, world)
    ref = oracle.render(scene, cam, threads=8)
    gathered = torch.zeros((world, cap, 32, 32, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()          # torch fills on its own stream; the contexts below launch on theirs
    pixels = 0
    for rank in range(world):
        r = rt.Renderer(scene, cam, rank=rank, world=world, image_ptr=gathered[rank].data_ptr(), tile_buffer_tiles=cap)
        try:
            assert r.shader.owned_tiles(dw, dh) == (tiles.owned_tiles(total, rank, world), tiles.tile_grid(cw, ch)[0], total)
            assert r.shader.covered_pixels(dw, dh) == tiles.owned_pixels(cw, ch, rank, world)
            pixels += r.shader.covered_pixels(dw, dh)
            r.dispatch()
            r.ctx.finish()
        finally:
            r.close()
    assert pixels == cw * ch
    # numpy de-interleave (host arithmetic) and the HIP assemble kernel agree with the oracle
    assert _eq(tiles.assemble(gathered.cpu().numpy(), W, H, cw, ch, world), ref)
    full = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    r0 = rt.Renderer(scene, cam, rank=0, world=world, image_ptr=gathered[0].data_ptr(), tile_buffer_tiles=cap)
    try:
        tex = rt.Texture.wrap_device(r0.ctx, full.data_ptr(), W, H, bind=False)
        r0.shader.assemble_tiles(gathered.data_ptr(), world, cap, tex, dw, dh)
        r0.ctx.finish()
    finally:
        r0.close()
    torch.cuda.synchronize()
    assert _eq(full.cpu().numpy(), ref)


def test_partition_into_full_size_image(oracle):
    """With the full-size image bound, a rank writes only its own work-groups."""
    scene = host.Scene.demo()
    cam = host.camera_reference_pose(128, 96, 1, 4)
    ref = oracle.render(scene, cam, threads=4)
    acc = np.zeros_like(ref)
    for rank in range(3):
        r = rt.Renderer(scene, cam, rank=rank, world=3)
        try:
            img = r.render()
        finally:
            r.close()
        mask = img[..., 3] == 1
        assert not (acc[mask].any())                    # disjoint
        acc[mask] = img[mask]
    assert _eq(acc, ref)


def test_progressive_accumulate_equals_one_pass(oracle):
    """BASELINE configs[4] style: k passes carrying fp32 running sums (+ the shader's loop-carried
    temporaries) in sample order, one resolve — bit-identical to a single dispatch."""
    import torch
    scene = host.Scene.config(2)
    W, H, spp = 160, 96, 12
    cam = host.camera_reference_pose(W, H, spp, 8)
    one = oracle.render(scene, cam, threads=8)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    carry = torch.zeros((H, W, 16), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    r = rt.Renderer(scene, cam, image_ptr=accum.data_ptr())
    try:
        for begin, count in ((0, 5), (5, 1), (6, 6)):
            r.shader.dispatch_accumulate(W + 1, H + 1, 1, begin, count, carry.data_ptr())
        r.shader.dispatch_resolve(W + 1, H + 1, 1, spp)
        img = r.texture.read()
    finally:
        r.close()
    assert _eq(img, one)


def test_generic_kernel_equals_exact_comparison_kernel(oracle, monkeypatch):
    """TDT_FORCE_GENERIC=1 runs the literal float treeLookup on a power-of-two scene: same bits."""
    scene = host.Scene.config(2)
    cam = host.camera_reference_pose(160, 96, 2, 8)
    ref = oracle.render(scene, cam, threads=8)
    monkeypatch.setenv("TDT_FORCE_GENERIC", "1")
    r = rt.Renderer(scene, cam)
    try:
        assert _eq(r.render(), ref)
    finally:
        r.close()


@pytest.mark.parametrize("cfg,W,H", [(2, 160, 96), (3, 128, 72)])
def test_unspecialised_kernel_equals_specialised(oracle, monkeypatch, cfg, W, H):
    """TDT_NO_SPECIALISE=1 keeps the run-time-depth / non-resident lookup: same bits as the
    scene-specialised variants the dispatcher normally picks."""
    scene = host.Scene.config(cfg)
    cam = host.camera_reference_pose(W, H, 2, 8)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        assert _eq(r.render(), ref)
    finally:
        r.close()
    monkeypatch.setenv("TDT_NO_SPECIALISE", "1")
    r = rt.Renderer(scene, cam)
    try:
        assert _eq(r.render(), ref)
    finally:
        r.close()


@pytest.mark.parametrize("cfg", [2, 3])
def test_cost_feedback_order_changes_only_the_schedule(oracle, monkeypatch, cfg):
    """From the second dispatch of a context on, pixels are handed out most expensive first (work counts recorded
    by the previous dispatch) — for the LDS-resident 64^3 tree and for the 256^3 one that lives in L2/HBM alike.
    Every frame must still be the same bits."""
    scene = host.Scene.config(cfg)
    cam = host.camera_reference_pose(320, 192, 3, 8)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        for _ in range(3):                      # image order, then twice in cost order
            assert _eq(r.render(), ref)
    finally:
        r.close()
    monkeypatch.setenv("TDT_NO_COST_ORDER", "1")
    r = rt.Renderer(scene, cam)
    try:
        for _ in range(2):
            assert _eq(r.render(), ref)
    finally:
        r.close()


@pytest.mark.parametrize("env", [("TDT_ORDER_SMOOTH", "1"), ("TDT_MAX_SHARE", "0"), ("TDT_MAX_SHARE", "100"), ("TDT_NO_COST_ACCUM", "1")])
def test_every_scheduling_variant_is_the_same_bits(oracle, monkeypatch, env):
    """Tile-sum order, exact / batched queue draws, costs from the last dispatch only: schedules, not arithmetic."""
    monkeypatch.setenv(*env)
    scene = host.Scene.config(3)
    cam = host.camera_reference_pose(256, 160, 3, 8)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        for _ in range(3):
            assert _eq(r.render(), ref)
    finally:
        r.close()


@pytest.mark.parametrize("cfg", [2, 3])
def test_two_phase_frames_are_the_same_bits(oracle, monkeypatch, cfg):
    """A frame without a usable cost history (first frame, moved camera) with spp >= 16 is traced as spp/16 probe samples,
    then the rest in the cost order of its own probe, then a resolve — sums and hit-record carry through HBM in between.
    It must equal the one-pass frame (TDT_NO_TWO_PHASE=1) and the oracle bit for bit."""
    scene = host.Scene.config(cfg)
    cam = host.camera_reference_pose(192, 128, 40, 6)                # probe = 2 samples
    ref = oracle.render(scene, cam, threads=16)
    r = rt.Renderer(scene, cam)
    try:
        assert _eq(r.render(), ref)                          # two-phase
        assert _eq(r.render(), ref)                          # replay: one pass, exact order
        cam2 = cam.copy(); cam2.origin[0] = 0.05; cam2.lower_left_corner[0] += 0.05
        rt.initial_uniforms(cam2, r.shader.program)
        assert _eq(r.render(), oracle.render(scene, cam2, threads=16))       # moved: two-phase again, stale tile order for the probe
    finally:
        r.close()
    monkeypatch.setenv("TDT_NO_TWO_PHASE", "1")
    r = rt.Renderer(scene, cam)
    try:
        assert _eq(r.render(), ref)
    finally:
        r.close()


def test_cost_feedback_after_the_camera_moved(oracle):
    """When the inputs of a dispatch differ from what the recorded costs were measured on (camera moved, scene edited),
    8x8 tiles are ordered by their summed cost instead of single pixels; a repeat of the same view then goes back to
    the per-pixel order.  Same bits every time."""
    scene = host.Scene.config(2)
    cam_a = host.camera_reference_pose(320, 192, 3, 8)
    moved = host.Camera(90.0, 320, aspect_ratio=np.float32(320) / np.float32(192), origin=(0.0, -0.1, -0.3), viewport_height=2.0,
                        samples_per_pixel=3, max_bounce=8)
    moved.turn_yaw(2.0)
    moved.translate("Front", 0.5)
    cam_b = moved.uniforms()
    ref_a, ref_b = oracle.render(scene, cam_a, threads=8), oracle.render(scene, cam_b, threads=8)
    assert not _eq(ref_a, ref_b)
    r = rt.Renderer(scene, cam_a)
    try:
        assert _eq(r.render(), ref_a)                       # image order
        rt.initial_uniforms(cam_b, r.shader.program)
        assert _eq(r.render(), ref_b)                       # tiles ordered by the costs of view A
        assert _eq(r.render(), ref_b)                       # pixels ordered by the costs of view B
        rt.initial_uniforms(cam_a, r.shader.program)
        assert _eq(r.render(), ref_a)
    finally:
        r.close()


def test_zero_samples_and_zero_bounces(oracle):
    scene = host.Scene.demo()
    for spp, bounce in ((0, 4), (2, 0)):
        cam = host.camera_reference_pose(64, 64, spp, bounce)
        r = rt.Renderer(scene, cam)
        try:
            got = r.render()
        finally:
            r.close()
        assert _eq(got, oracle.render(scene, cam)), (spp, bounce)


def test_error_behaviour():
    with rt.Context(0) as ctx:
        cs = rt.ComputeShader(ctx)
        assert cs.group_size == [32, 32, 1]                                   # compute_shader.rs:18
        with pytest.raises(rt.TdtError) as e:
            cs.program.set_i32("camera.no_such_uniform", 1)                   # program.rs:144-165
        assert e.value.code == rt.ERR_VARIABLE_NOT_FOUND and "camera.no_such_uniform" in str(e.value)
        with pytest.raises(rt.TdtError) as e:
            cs.program.set_f32("camera.image_width", 1.0)                     # wrong type -> TypedVariableNotFound
        assert e.value.code == rt.ERR_VARIABLE_NOT_FOUND
        with pytest.raises(rt.TdtError) as e:
            cs.dispatch_compute(33, 33, 1)                                    # nothing bound
        assert e.value.code == rt.ERR_INCOMPLETE
        with pytest.raises(rt.TdtError) as e:
            rt.ComputeShader(ctx, 7)                                           # no such program
        assert e.value.code == rt.ERR_INVALID_ENUM
        upd = rt.ComputeShader(ctx, rt.PROGRAM_OCTREE_UPDATE)                 # main.rs:226-230
        assert upd.group_size == [1, 1, 1]
        with pytest.raises(rt.TdtError) as e:
            upd.dispatch_compute(0, 1, 0)                                     # nothing bound
        assert e.value.code == rt.ERR_INCOMPLETE
        vbo = rt.VertexBufferObject(ctx, np.zeros(4, np.float32))
        with pytest.raises(rt.TdtError) as e:
            ctx.bind_buffer_base(rt.SHADER_STORAGE_BUFFER, 9, vbo)
        assert e.value.code == rt.ERR_INVALID_VALUE
        with pytest.raises(rt.TdtError) as e:
            ctx.bind_buffer_base(0x1234, 0, vbo)
        assert e.value.code == rt.ERR_INVALID_ENUM
        ctx.bind_buffer_base(rt.ATOMIC_COUNTER_BUFFER, 0, vbo)                # octree.rs:115: accepted, unused


def test_out_of_range_cells_read_as_empty(oracle):
    """Robust buffer access: the demo scene's shader divides by cell_count = 100000 but the cell
    buffer is shorter; truncating it further must behave exactly like the oracle's zero reads."""
    scene = host.Scene.demo()
    scene.blobs[0] = scene.blobs[0][:19 * 16 - 6].copy()      # cut into the last cell, odd dword count
    cam = host.camera_reference_pose(96, 96, 1, 6)
    r = rt.Renderer(scene, cam)
    try:
        got = r.render()
    finally:
        r.close()
    assert _eq(got, oracle.render(scene, cam, threads=4))


@pytest.mark.parametrize("slot,keep_bytes", [(1, 12 * 2 + 4), (1, 12 * 1 + 8), (1, 12), (2, 12 * 2 + 4), (2, 8), (3, 4), (3, 0), (4, 4), (4, 0)])
def test_out_of_range_material_tables_read_as_zero(oracle, slot, keep_bytes):
    """Robust buffer access on the material tables (dword by dword, as llvmpipe checks them): a materials / albedos /
    metal / dielectric buffer cut inside a record, or down to nothing, must give what the oracle's zero reads give."""
    scene = host.Scene.generate(host.SCENE_TERRAIN, 5, 1 << 14, 100, 0x5EED0011)
    m = scene.blobs[1].reshape(-1, 3).copy()             # every record non-zero in every dword: a record cut in two must not read as all zeros
    m[:, 0] = 1 + (np.arange(len(m)) % 2); m[:, 1] = 1 + (np.arange(len(m)) % 3); m[:, 2] = 1 + (np.arange(len(m)) % 5)
    scene.blobs[1] = m.reshape(-1)
    keep = max(keep_bytes, 4) if keep_bytes else 4       # (a zero-size buffer cannot be created: one dword of zeros reads the same)
    blob = np.frombuffer(scene.blobs[slot].tobytes()[:keep_bytes].ljust(keep, b"\0"), dtype=np.uint32).copy()
    scene.blobs[slot] = blob
    cam = host.camera_reference_pose(96, 64, 2, 6)
    r = rt.Renderer(scene, cam)
    try:
        got = r.render()
    finally:
        r.close()
    assert _eq(got, oracle.render(scene, cam, threads=4))


def test_short_rounding_forms_exhaustively():
    """The kernels' 3/5-instruction rcp / sqrt / rsq equal the IEEE expressions on ALL 2^32 inputs
    (and the harness does detect an inexact form: the raw hardware reciprocal seed fails)."""
    with rt.Context(0) as ctx:
        assert ctx.selftest(0) == 0      # 1/x
        assert ctx.selftest(1) == 0      # sqrt(x)
        assert ctx.selftest(2) == 0      # 1/sqrt(x), two roundings
        assert ctx.selftest(3) > 1000000
        assert ctx.selftest(4) == 0      # v_fract_f32 == x - floor(x) for every x >= 0
        assert ctx.selftest(5) == 0      # 4-level jump table: x decision == binary digit outside the bands, all c in [0,1), all v below the bounds
        assert ctx.selftest(6) > 0       # ... and not without the bands (harness check)
        assert ctx.selftest(7) == 0      # the 5-level table of trees that are not LDS-resident (cell indices up to 8191 at level 5)
        assert ctx.selftest(8) > 0
        assert ctx.selftest(9) == 0      # CubeHit's normal: +-1 / signed zeros where the guard holds == the literal normalise-orient-normalise
        assert ctx.selftest(10) > 0      # ... and not without the guard
        assert ctx.selftest(11) == 0     # (m-1)/(m+1) and (1-x)/(1+x) through reciprocal + residual step == IEEE division
        assert ctx.selftest(12) > 0
        assert ctx.selftest(13) == 0     # fl(v + f) - v is a function of floor(log2 v) and f (the bricks of depth-8 trees)
        assert ctx.selftest(14) > 0
        assert ctx.selftest(15) == 0     # ... and their level-5 table's bands, one per cell index
        assert ctx.selftest(16) > 0
