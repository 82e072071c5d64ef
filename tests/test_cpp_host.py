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
