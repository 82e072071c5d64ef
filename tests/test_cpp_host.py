"""The C++ host mirror (include/renderer.hpp) end to end: csrc/demo_main.cpp is the reference's main.rs made
headless — context, programs, CameraBuilder, the scene literal's buffers, Octree::init_global_buffers,
(optionally) one update_vbo click, dispatch_compute(w+1, h+1, 1) — and must write the frame the reference wrote."""
import json
import os
import subprocess

import numpy as np
import pytest

from tdt4230_project_raytracing_amd import build

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_pfm4(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF4"
        w, h = map(int, f.readline().split())
        assert float(f.readline()) < 0          # little endian
        return np.frombuffer(f.read(), "<f4").reshape(h, w, 4)


def test_demo_binary_is_built_and_fails_loudly_without_gpu():
    exe = build.build_demo()
    assert os.access(exe, os.X_OK)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, "--size", "64x64"], capture_output=True, text=True)
    assert r.returncode == 1 and "InitializeErr" in r.stderr        # no CPU fallback


@pytest.mark.gpu
def test_headless_main_writes_the_reference_frame(tmp_path):
    exe = build.build_demo()
    out = str(tmp_path / "frame.pfm")
    r = subprocess.run([exe, "--size", "160x96", "--spp", "4", "--bounce", "6", "--out", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    golden = np.load(os.path.join(GOLDEN, "demo_160x96_spp4_b6.npz"))["image"]
    assert (read_pfm4(out).view(np.uint32) == golden.view(np.uint32)).all()
    assert "counter 19" in r.stdout                                    # active_cell_count untouched (octree.rs:105-110)


@pytest.mark.gpu
def test_headless_main_with_one_click_edit(tmp_path):
    exe = build.build_demo()
    out = str(tmp_path / "frame.pfm")
    z = np.load(os.path.join(GOLDEN, "edit_demo_click.npz"))
    meta = json.loads(str(z["meta"]))
    d = z["delta"][0]
    r = subprocess.run([exe, "--size", "128x96", "--spp", "2", "--bounce", "6", "--edit", ",".join(repr(float(v)) for v in d[:5]),
                        "--out", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"counter {meta['counter_after']}" in r.stdout
    assert (read_pfm4(out).view(np.uint32) == z["image"].view(np.uint32)).all()


@pytest.mark.gpu
def test_headless_main_presents_a_png_and_moves_the_camera(tmp_path):
    """--png = the quad pass's frame as a file; --settings / --move drive the camera controller
    (§8f-3) exactly as the Python mirror does, so both hosts must render the same moved frame."""
    import zlib, struct
    from tdt4230_project_raytracing_amd import host, rt
    exe = build.build_demo()
    ron = tmp_path / "camera.ron"
    ron.write_text("CameraSettings(samples_per_pixel: 2, max_bounce: 5, turn_rate: 0.05, normal_speed: 0.03, sprint_speed: 0.15)")
    out, png = str(tmp_path / "frame.pfm"), str(tmp_path / "frame.png")
    r = subprocess.run([exe, "--size", "160x96", "--settings", str(ron), "--move", "eeSwwNdrj", "--out", out, "--png", png],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "spp 2 bounce 5" in r.stdout
    # the same session through the Python mirror
    cam = host.Camera(90.0, 160, aspect_ratio=np.float32(160) / np.float32(96), origin=(0.0, -0.1, -0.3), viewport_height=2.0,
                      samples_per_pixel=4, max_bounce=6, turn_rate=0.05, normal_speed=0.03, sprint_speed=0.15)
    cam.apply_settings(host.CameraSettings.from_ron(ron.read_text()))
    cam.set_speed_to_normal()
    dt = 1.0 / 60.0
    for k in "eeSwwNdrj":
        {"e": lambda: cam.turn_yaw(1.0), "r": lambda: cam.turn_pitch(-1.0), "S": cam.set_speed_to_sprint, "N": cam.set_speed_to_normal,
         "w": lambda: cam.translate("Front", dt), "d": lambda: cam.translate("Rigth", dt), "j": lambda: cam.translate("Down", dt)}[k]()
    rr = rt.Renderer(host.Scene.demo(), cam.uniforms())
    try:
        want = rr.render()
        want8 = rr.texture.read_rgba8(top_down=True)
    finally:
        rr.close()
    assert (read_pfm4(out).view(np.uint32) == want.view(np.uint32)).all()
    moved = np.load(os.path.join(GOLDEN, "demo_160x96_spp4_b6.npz"))["image"]
    assert (want[..., :3] != moved[..., :3]).any()                     # the camera did move
    data = open(png, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    idat = data[data.index(b"IDAT") + 4:data.index(b"IEND") - 8]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(96, 1 + 160 * 3)
    assert (raw[:, 1:].reshape(96, 160, 3) == want8[..., :3]).all()
