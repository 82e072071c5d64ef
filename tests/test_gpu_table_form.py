"""A cell_count that is not a power of two — the reference's own scene uses 100000 (main.rs:459) — takes treeLookup's x index
through per-cell thresholds (FORM_TABLE builds: csrc/trace_device.hpp x_thresholds) instead of the float formula, and a
pre-allocated cells buffer (main.rs:339-341: a tail of zero nodes) counts as LDS-resident when its live part fits the table.
The claim is checked exhaustively against the literal formula, the kernels against the oracle and against the literal kernel."""
import os
import sys

import numpy as np
import pytest

from tdt4230_project_raytracing_amd import host, rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def _eq(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.parametrize("cc,n_cells", [(100000, 5120), (99999, 2048), (12345, 2048), (65537, 1024), (3000, 4096), (7, 512), (1 << 16, 1024),
                                        (1000003, 1024)])
def test_thresholds_equal_the_literal_formula_for_every_coordinate(cc, n_cells):
    """ix(v, f) = 2v + (f >= F1(v)) + (f >= F2(v)) for EVERY f in [0, 1) and every cell index below n_cells."""
    with rt.Context(0) as ctx:
        ic = float(np.float32(1.0) / np.float32(cc))
        bad, shape_ok = ctx.selftest_index(cc, ic, n_cells)
        assert shape_ok and bad == 0
        for shift in (1, -1):                                      # the harness: thresholds one ulp off must be caught
            bad, _ = ctx.selftest_index(cc, ic, n_cells, shift=shift)
            assert bad >= n_cells


def test_unusual_index_uniforms_are_refused_not_guessed():
    """cell_count / inv_cell_count pairs for which the index is not 2v + two steps (a host that wrote an unrelated
    inv_cell_count): shape_ok = 0, and such a scene runs the literal kernel (next test)."""
    with rt.Context(0) as ctx:
        for cc, ic in ((100000, 2e-5), (100000, 0.0), (100000, -1e-5), (1, 3.0)):
            _, shape_ok = ctx.selftest_index(cc, ic, 64)
            assert not shape_ok, (cc, ic)


SCENES = [("config", 1), ("config", 2), ("gen", 1, 4, 1 << 14, 100, 7), ("gen", 0, 5, 1 << 16, 100, 9), ("gen", 2, 7, 1 << 16, 256, 11),
          ("gen", 1, 8, 1 << 20, 256, 13), ("config", 3), ("config", 5), ("gen", 0, 7, 1 << 20, 100, 17), ("gen", 2, 9, 1 << 20, 512, 19), ("demo",)]


def _make(spec):
    if spec[0] == "config":
        return host.Scene.config(spec[1])
    if spec[0] == "demo":
        return host.Scene.demo()
    return host.Scene.generate(*spec[1:])


@pytest.mark.parametrize("spec", SCENES)
@pytest.mark.parametrize("cc", [100000, 12345, 1000003])
def test_table_form_equals_oracle_and_literal_kernel(oracle, spec, cc, monkeypatch):
    import fuzz_parity
    scene = fuzz_parity.with_cell_count(_make(spec), cc)
    if spec[0] != "demo":
        scene = fuzz_parity.with_zero_tail(scene, 100144 // 2)     # a pre-allocated buffer, as the reference's host makes it
    cam = host.camera_reference_pose(200, 120, 4, 6)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        first, again = r.render(), r.render()
    finally:
        r.close()
    assert _eq(first, ref) and _eq(again, ref)
    monkeypatch.setenv("TDT_NO_TABLE_FORM", "1")                   # the literal kernel on the same inputs
    r = rt.Renderer(scene, cam)
    try:
        assert _eq(r.render(), ref)
    finally:
        r.close()


def test_wrong_inv_cell_count_runs_the_literal_kernel(oracle):
    """inv_cell_count unrelated to cell_count (the x index is then not 2v + ...: every lookup lands in other cells): whatever the
    reference computes from such uniforms, the kernel computes too."""
    scene = host.Scene.config(2)
    blobs = {k: v.copy() for k, v in scene.blobs.items()}
    blobs[6][6] = np.float32(3.0e-5)
    blobs[7][2] = 100000
    scene = host.Scene(blobs, scene.counts, "wrong_inv")
    cam = host.camera_reference_pose(128, 96, 2, 4)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        assert _eq(r.render(), ref)
    finally:
        r.close()


@pytest.mark.parametrize("cfg", [1, 2])
def test_zero_tail_counts_as_resident(oracle, cfg):
    """A power-of-two cell_count with a pre-allocated buffer: the live part decides residency (whole-depth / 4-level table builds)."""
    import fuzz_parity
    scene = fuzz_parity.with_zero_tail(host.Scene.config(cfg), 70000)
    cam = host.camera_reference_pose(160, 96, 16, 6)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        assert _eq(r.render(), ref) and _eq(r.render(), ref)
    finally:
        r.close()


def test_parent_pointing_past_the_thresholds_runs_the_literal_kernel(oracle):
    """A PARENT whose value is beyond the cells that have thresholds (a cell index into the zero tail): not eligible, same bits."""
    import fuzz_parity
    scene = fuzz_parity.with_cell_count(fuzz_parity.with_zero_tail(host.Scene.config(2), 60000), 100000)
    cells = scene.blobs[0].reshape(-1, 8, 2).copy()
    leaves = np.argwhere(cells[:, :, 1] == 2)
    pick = leaves[::53]
    cells[pick[:, 0], pick[:, 1], 1] = 1
    cells[pick[:, 0], pick[:, 1], 0] = 6000 + np.arange(len(pick), dtype=np.uint32) % 500
    scene.blobs[0] = np.ascontiguousarray(cells.reshape(-1))
    cam = host.camera_reference_pose(160, 96, 2, 6)
    ref = oracle.render(scene, cam, threads=8)
    r = rt.Renderer(scene, cam)
    try:
        assert _eq(r.render(), ref)
    finally:
        r.close()
