"""Parity of the HIP path (through the C ABI) against the oracle — bit-exact fp32.

The stated tolerance of the north star is 1e-4 per channel; because the integrand is chaotic
that is only reachable by matching bit patterns, so these tests assert EQUAL BITS and report
max |diff| as context.
"""
import numpy as np
import pytest

from tdt4230_project_raytracing_amd import host, rt

pytestmark = pytest.mark.gpu


def _bits_equal(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)).all(axis=2)


def _render_gpu(scene, cam, dispatch=None, **kw):
    r = rt.Renderer(scene, cam, device=0, **kw)
    try:
        if dispatch:
            return r.render(*dispatch)
        return r.render()
    finally:
        r.close()


CASES = [
    # (scene config, W, H, spp, bounce)
    (0, 256, 256, 1, 6),
    (0, 256, 256, 4, 6),
    (0, 320, 180, 4, 6),      # 16:9 like the reference's 1280x720 window; floor-div leaves rows unwritten
    (1, 256, 256, 1, 1),      # BASELINE config 1
    (2, 480, 270, 4, 8),      # BASELINE config 2 scene at quarter resolution
    (2, 256, 144, 16, 8),
]


@pytest.mark.parametrize("cfg,W,H,spp,bounce", CASES)
def test_bit_exact_vs_oracle(oracle, cfg, W, H, spp, bounce):
    scene = host.Scene.config(cfg)
    cam = host.camera_reference_pose(W, H, spp, bounce)
    assert (cam.image_width, cam.image_height) == (W, H)
    got = _render_gpu(scene, cam)
    ref = oracle.render(scene, cam, threads=8)
    eq = _bits_equal(got, ref)
    bad = int((~eq).sum())
    maxd = float(np.nanmax(np.abs(got - ref)))
    assert bad == 0, f"{bad} of {W * H} pixels differ from the oracle (max |diff| {maxd:.3g})"


def test_dispatch_floor_division_coverage(oracle):
    """compute_shader.rs:30-32: groups = max(dim / 32, 1); 320x180 -> 10 x 5 groups, rows 160..179 never written."""
    scene = host.Scene.demo()
    cam = host.camera_reference_pose(320, 180, 1, 2)
    got = _render_gpu(scene, cam)
    assert (got[160:, :, :] == 0).all()
    assert (got[:160, :, 3] == 1).all()


def test_event_counts_equal_oracle(oracle):
    """The instrumented dispatch counts exactly the events the oracle counts: they define the
    algorithmic bytes that bench.py's roofline uses."""
    scene = host.Scene.config(2)
    cam = host.camera_reference_pose(256, 144, 4, 8)
    r = rt.Renderer(scene, cam, device=0)
    try:
        got = r.shader.dispatch_counted(cam.image_width + 1, cam.image_height + 1, 1)
        img = r.texture.read()
    finally:
        r.close()
    ref, st = oracle.render(scene, cam, threads=8, want_stats=True)
    assert _bits_equal(img, ref).all()
    for k in ("pixels", "octree_hit_calls", "iterations", "node_loads", "lambertian", "metal", "dielectric", "unknown_material"):
        assert got[k] == st[k], (k, got[k], st[k])


def test_relocated_top_cells_disable_the_jump_table(oracle):
    """The top-3-level jump table (build_top_grid) is exact only while the top PARENT nodes point at cell indices < 128.
    A tree whose first-level cell was re-allocated at the end of the buffer (what octree_update.comp's allocator does to
    edited trees) must fall back to the level-by-level descent and still match the oracle bit for bit."""
    scene = host.Scene.config(3)                                  # 256^3: not LDS-resident -> the table is in use
    cells = scene.blobs[0].reshape(-1, 8, 2).copy()
    n = cells.shape[0]
    j = int(np.nonzero(cells[0, :, 1] == 1)[0][0])                # a PARENT node of the root cell
    moved = cells[int(cells[0, j, 0])].copy()
    cells = np.concatenate([cells, moved[None]], axis=0)
    cells[0, j, 0] = n                                            # ... now lives in cell n (>= 128)
    scene.blobs[0] = np.ascontiguousarray(cells.reshape(-1))
    cam = host.camera_reference_pose(256, 160, 2, 6)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        for _ in range(2):
            got = r.render()
            assert (got.view(np.uint32) == ref.view(np.uint32)).all()
    finally:
        r.close()


@pytest.mark.parametrize("cfg,case", [(3, "wide_materials"), (5, "wide_materials"), (3, "scattered_cells"), (5, "scattered_cells"), (3, "zero_child"),
                                      (2, "wide_materials")])   # (the whole-depth table of small trees: no table at all, the 4-level form runs)
def test_positions_whose_bricks_cannot_be_built_walk(oracle, cfg, case):
    """The bricks of depth-8 / 9 trees (build_bricks_kernel) mark a level-5 position whose sub-tree they cannot represent — a
    material index that does not fit the entry, level-7 / level-8 cells that do not share floor(log2(index)), a PARENT that
    points at cell 0 — and waves that meet one walk the levels: same bits as the oracle, next to positions that do have bricks."""
    scene = host.Scene.config(cfg)
    cells = scene.blobs[0].reshape(-1, 8, 2).copy()
    rng = np.random.default_rng(11 + cfg)
    leaves = np.argwhere(cells[:, :, 1] == 2)
    parents = np.argwhere(cells[:, :, 1] == 1)
    if case == "wide_materials":                                   # every third LEAF: an index past what a 16-bit entry holds
        pick = leaves[rng.random(len(leaves)) < 0.33]              # (beyond the material table: robust access reads zeros)
        cells[pick[:, 0], pick[:, 1], 0] += 70000 if cfg == 5 else 3000
        cells[pick[::7, 0], pick[::7, 1], 0] += 1 << 26
    elif case == "scattered_cells":                                # deep cells re-allocated far away (what edits do): other binades
        deep = parents[parents[:, 0] > 5000]
        pick = deep[rng.choice(len(deep), size=min(400, len(deep)), replace=False)]
        n = cells.shape[0]
        pad = (1 << 21) + 5 - n if cfg == 3 else 64                # config 3: past 2^21 (another exponent AND >= 2^20 for level-6 cells)
        moved = cells[cells[pick[:, 0], pick[:, 1], 0].astype(np.int64)].copy()
        cells = np.concatenate([cells, np.zeros((max(pad, 0), 8, 2), np.uint32), moved], axis=0)
        cells[pick[:, 0], pick[:, 1], 0] = n + max(pad, 0) + np.arange(len(pick), dtype=np.uint32)
    else:                                                          # a PARENT whose value is 0: the walk re-enters the root cell
        pick = parents[parents[:, 0] > 600][::97]
        cells[pick[:, 0], pick[:, 1], 0] = 0
    scene.blobs[0] = np.ascontiguousarray(cells.reshape(-1))
    cam = host.camera_reference_pose(192, 112, 4, 6)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        for _ in range(2):
            assert _bits_equal(r.render(), ref).all()
    finally:
        r.close()


@pytest.mark.parametrize("cfg,scale", [(2, 2.0), (2, 0.75), (3, 2.0), (1, 3.0)])
def test_octree_scale_other_than_one(oracle, cfg, scale):
    """Octree::new takes any scale (main.rs:455-463 passes 1.0, and the kernels have builds that skip the multiplications by an
    exact 1.0f): a scene moved and scaled in world space, the camera moved and scaled with it — the builds that DO multiply,
    against the oracle (resident whole-depth table, 5-level table and small-tree forms)."""
    scene = host.Scene.config(cfg)
    of = scene.blobs[6].copy()
    s = np.float32(scale)
    of[0:3] = np.array([-0.5, -0.5, -1.0], np.float32) * s + np.array([0.25, -0.125, 0.5], np.float32)
    of[4] = s
    of[5] = np.float32(1.0) / s                                        # octree.rs:47: inv_scale = 1.0 / scale
    scene.blobs[6] = of
    cam = host.camera_build(90.0, 200, aspect_ratio=200 / 120, viewport_height=2.0,
                            origin=(float(0.0 * s + 0.25), float(-0.1 * s - 0.125), float(-0.3 * s + 0.5)), samples_per_pixel=16, max_bounce=6)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        first, again = r.render(), r.render()
    finally:
        r.close()
    assert _bits_equal(first, ref).all() and _bits_equal(again, ref).all()
    assert (ref[..., :3] != ref[0, 0, :3]).any()                       # the scene is in view


@pytest.mark.parametrize("cfg,max_iter", [(5, None), (3, None), (5, 40), (3, 1000)])
def test_rays_that_stop_advancing_leave_the_loop_with_the_same_bits(oracle, monkeypatch, cfg, max_iter):
    """OctreeHit's loop (raytracer.comp:410-447) is a function of (t_stride, inv_pow_depth) and the ray.  At the finest levels of the
    256^3 / 512^3 trees treeLookup's float index arithmetic (rc:376-378) can hand back a cell the sample point is not in, whose slab
    interval along the ray is empty at t_stride: the step leaves both unchanged, and so does every later one until max_iter — more than
    half of all iterations of these frames (the oracle's own count below: 20-70 iterations per OctreeHit call where a 64^3 tree takes
    nine).  The brick builds end such a lane's loop at the first repeat; the walk build (TDT_NO_BRICKS) runs every iteration.  Both
    are the oracle's bits, whatever max_iter is (a stuck ray ends as `false` either way; a ray that merely runs out of iterations is not
    touched), in a two-phase frame and in its replay."""
    scene = host.Scene.config(cfg)
    if max_iter is not None:
        oi = scene.blobs[7].copy(); oi[1] = max_iter; scene.blobs[7] = oi
    cam = host.camera_reference_pose(192, 112, 16, 8)
    ref, st = oracle.render(scene, cam, threads=8, want_stats=True)
    if max_iter is None:
        assert st["iterations"] > 15 * st["octree_hit_calls"], st       # the premise: this frame is full of rays that stop advancing (64^3: 9 per call)
    for walk in (False, True):
        if walk:
            monkeypatch.setenv("TDT_NO_BRICKS", "1")
        r = rt.Renderer(scene, cam)
        try:
            for _ in range(2):
                assert _bits_equal(r.render(), ref).all()
            assert r.ctx.last_variant()["brick"] == (0 if walk else 1)
        finally:
            r.close()


def test_rays_that_stop_advancing_are_not_iterated(monkeypatch):
    """The guard for the shortcut above, by its effect (device-side HIP-event timing of the launches): on the 512^3 scene, where six of
    seven iterations of the reference are repeats, the brick build must beat the walk build — which runs every iteration — by far more
    than the bricks themselves ever bought (round 3: 1.3-1.5 x).  Measured 7.4 x (1.9 against 14.2 ms); the assertion asks for 3 x."""
    scene = host.Scene.config(5)
    cam = host.camera_reference_pose(480, 270, 16, 8)
    ms = {}
    for walk in (False, True):
        if walk:
            monkeypatch.setenv("TDT_NO_BRICKS", "1")
        r = rt.Renderer(scene, cam)
        try:
            r.dispatch(); r.ctx.finish()
            r.ctx.phase_timing(True)
            best = None
            for _ in range(3):
                r.ctx.forget_costs(); r.dispatch(); r.ctx.finish()
                t = sum(r.ctx.phase_timing(True))
                best = t if best is None or t < best else best
            ms[walk] = best
        finally:
            r.close()
    assert ms[True] > 3.0 * ms[False], ms
