"""SURVEY §8f-4 on the GPU: tdt_image_read_rgba8 (present_kernel) gives the bytes of the reference's quad pass — the
fixtures made by running quad.vert / quad.frag on llvmpipe (tests/golden/present_*.npz) — and the demo frame rendered on
the GPU presents to the fixture's bytes end to end."""
import os

import numpy as np
import pytest
import torch

from tdt4230_project_raytracing_amd import host, rt

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return z["image"].view(np.float32), z["rgba8"]


@pytest.mark.parametrize("name", ["present_values", "present_demo"])
def test_gpu_conversion_equals_reference_quad_pass(name):
    image, want = load(name)
    H, W = image.shape[:2]
    with rt.Context(0) as ctx:
        dev = torch.from_numpy(np.ascontiguousarray(image)).cuda()
        torch.cuda.synchronize()
        tex = rt.Texture.wrap_device(ctx, dev.data_ptr(), W, H, bind=False)
        assert (tex.read_rgba8(top_down=False) == want).all()
        assert (tex.read_rgba8(top_down=True) == want[::-1]).all()
        assert (host.present_rgba8(tex.read(), top_down=False) == want).all()        # host and device conversions agree


def test_rendered_demo_frame_presents_to_the_reference_bytes(tmp_path):
    _, want = load("present_demo")                              # render + quad pass, both by the reference on llvmpipe
    scene, cam = host.Scene.demo(), host.camera_reference_pose(160, 100, 4, 6)
    r = rt.Renderer(scene, cam)
    try:
        r.dispatch()
        frame = r.texture.read_rgba8(top_down=True)
    finally:
        r.close()
    assert (frame == want[::-1]).all()
    path = tmp_path / "demo.png"
    host.png_write(str(path), frame)
    assert path.stat().st_size > 1000
