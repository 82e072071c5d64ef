"""The oracle (CPU restatement) against fixtures rendered by the REFERENCE shader itself on Mesa
llvmpipe (tests/golden/*.npz, made by oracle/make_goldens.py).  Bit-exact: this is what pins the
oracle, and through it every GPU parity test."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from tdt4230_project_raytracing_amd import host

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


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    spec = meta["scene"]
    scene = _scene(z, spec)
    cam = _camera(meta)
    return z["image"], meta, scene, cam


def region(meta):
    if meta["crop"]:
        x0, y0, w, h = meta["crop"]
    else:
        x0, y0, w, h = 0, 0, meta["W"], meta["H"]
    return x0, y0, w, h


def test_fixtures_present():
    assert len(CASES) >= 8


@pytest.mark.parametrize("name", CASES)
def test_scene_and_camera_reproduce(name):
    """The generator / camera builder give byte-identical inputs to the ones the reference rendered."""
    _, meta, scene, cam = load_case(name)
    for slot, digest in meta["scene_sha256"].items():
        assert hashlib.sha256(np.ascontiguousarray(scene.blobs[int(slot)]).tobytes()).hexdigest() == digest, f"payload {slot}"
    for k, v in meta["camera"].items():
        got = getattr(cam, k)
        got = list(got) if hasattr(got, "__len__") else got
        assert got == v, k


@pytest.mark.parametrize("name", CASES)
def test_oracle_bit_exact_vs_reference_render(oracle, name):
    golden, meta, scene, cam = load_case(name)
    x0, y0, w, h = region(meta)
    img = oracle.render(scene, cam, rows=(y0, y0 + h), threads=8)
    got = img[y0:y0 + h, x0:x0 + w]
    eq = (got.view(np.uint32) == golden.view(np.uint32)).all(axis=2)
    assert eq.all(), f"{int((~eq).sum())} of {w * h} pixels differ from the reference render (max |d| {np.nanmax(np.abs(got - golden)):.3g})"


def test_math_models_vs_llvmpipe_table(oracle):
    """sin / cos / pow(x,5): the reference target's polynomial forms, bit for bit."""
    z = np.load(os.path.join(GOLDEN, "math_table.npz"))
    x, tab = z["x"], z["table"]
    assert (oracle.sin(x).view(np.uint32) == tab[:, 0].view(np.uint32)).all()
    assert (oracle.cos(x).view(np.uint32) == tab[:, 1].view(np.uint32)).all()
    pos = x > 0
    assert (oracle.pow(x[pos], 5.0).view(np.uint32) == tab[pos, 2].view(np.uint32)).all()
    # the ops the oracle takes from IEEE arithmetic really are IEEE on the reference target
    with np.errstate(all="ignore"):
        assert ((np.float32(1) / np.sqrt(x[pos])).view(np.uint32) == tab[pos, 3].view(np.uint32)).all()
        nz = x != 0
        assert ((np.float32(1) / x[nz]).view(np.uint32) == tab[nz, 4].view(np.uint32)).all()
        assert (np.sqrt(x[pos]).view(np.uint32) == tab[pos, 5].view(np.uint32)).all()
        assert ((x - np.floor(x)).view(np.uint32) == tab[:, 6].view(np.uint32)).all()
