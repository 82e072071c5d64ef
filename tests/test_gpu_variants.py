"""Which build of the trace kernel a scene runs.  Every build writes the same pixels, so a scene that silently fell back to the general
kernel (an eligibility test gone wrong) would pass every parity test and only show up as a slower bench: this pins the dispatch."""
import numpy as np
import pytest

from tdt4230_project_raytracing_amd import host, rt

pytestmark = pytest.mark.gpu

LITERAL, POW2, TABLE = 0, 1, 2


def _variant(scene, cam, frames=1, **env):
    r = rt.Renderer(scene, cam)
    try:
        for _ in range(frames):
            r.dispatch()
        r.ctx.finish()
        return r.ctx.last_variant()
    finally:
        r.close()


def test_baseline_configs_run_their_specialised_builds():
    cam = host.camera_reference_pose(128, 96, 2, 4)
    v = _variant(host.Scene.config(1), cam)
    assert (v["form"], v["depth"], v["resident"], v["full"], v["brick"], v["unit"]) == (POW2, 3, 1, 0, 0, 1)
    v = _variant(host.Scene.config(2), cam)                         # 64^3, inside the LDS table: the whole-depth table
    assert (v["form"], v["depth"], v["resident"], v["full"], v["brick"], v["unit"]) == (POW2, 6, 1, 1, 0, 1)
    v = _variant(host.Scene.config(3), cam)                         # 256^3: bricks
    assert (v["form"], v["depth"], v["resident"], v["full"], v["brick"], v["unit"]) == (POW2, 8, 0, 0, 1, 1)
    v = _variant(host.Scene.config(5), cam)                         # 512^3 sparse: bricks, four levels
    assert (v["form"], v["depth"], v["resident"], v["full"], v["brick"], v["unit"]) == (POW2, 9, 0, 0, 1, 1)


def test_the_references_own_scene_runs_the_threshold_build():
    """demo scene: cell_count 100000, max_depth 10, 19 live cells in a buffer of 6259 (main.rs:235-463)"""
    v = _variant(host.Scene.demo(), host.camera_reference_pose(128, 96, 4, 6))
    assert (v["form"], v["depth"], v["resident"], v["full"], v["brick"], v["unit"]) == (TABLE, 10, 1, 0, 0, 1)


def test_other_cell_counts_and_zero_tails():
    cam = host.camera_reference_pose(128, 96, 2, 4)
    v = _variant(host.scene_with_cell_count(host.Scene.config(2), 100000, 30000), cam)      # resident by its live part
    assert (v["form"], v["depth"], v["resident"]) == (TABLE, 6, 1)
    v = _variant(host.scene_with_cell_count(host.Scene.config(3), 100000, 0), cam)          # outside the LDS table
    assert (v["form"], v["depth"], v["resident"], v["brick"]) == (TABLE, 8, 0, 0)
    v = _variant(host.scene_with_cell_count(host.Scene.config(2), 1 << 16, 50000), cam)     # power of two, zero tail: still the whole-depth table
    assert (v["form"], v["full"]) == (POW2, 1)
    blobs = {k: a.copy() for k, a in host.Scene.config(2).blobs.items()}
    blobs[6][6] = np.float32(3e-5)                                                          # inv_cell_count unrelated to the count
    blobs[7][2] = 100000
    v = _variant(host.Scene(blobs), cam)
    assert (v["form"], v["depth"]) == (LITERAL, 0)                                          # refused: the literal kernel


def test_probe_launch_and_scaled_octrees_run_the_multiplying_build():
    scene = host.Scene.config(2)
    cam = host.camera_reference_pose(128, 96, 16, 4)                # 16 spp: a two-phase first frame; its LAST launch is the main one
    assert _variant(scene, cam)["unit"] == 1
    of = scene.blobs[6].copy()
    of[4] = np.float32(2.0); of[5] = np.float32(0.5)
    scene.blobs[6] = of
    v = _variant(scene, cam)
    assert (v["form"], v["full"], v["unit"]) == (POW2, 1, 0)


def test_switches_select_the_general_builds(monkeypatch):
    cam = host.camera_reference_pose(128, 96, 2, 4)
    monkeypatch.setenv("TDT_NO_SPECIALISE", "1")
    assert _variant(host.Scene.config(2), cam)["depth"] == 0
    monkeypatch.delenv("TDT_NO_SPECIALISE")
    monkeypatch.setenv("TDT_NO_TABLE_FORM", "1")
    v = _variant(host.Scene.demo(), cam)
    assert (v["form"], v["depth"]) == (LITERAL, 0)
    monkeypatch.delenv("TDT_NO_TABLE_FORM")
    monkeypatch.setenv("TDT_NO_BRICKS", "1")
    v = _variant(host.Scene.config(3), cam)
    assert (v["form"], v["depth"], v["brick"]) == (POW2, 8, 0)
