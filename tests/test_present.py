"""SURVEY §8f-4, presentation: the RGBA8 frame of the reference's quad pass (assets/shaders/quad.vert, quad.frag:10;
main.rs:113-153, 582-600).  tests/golden/present_*.npz were produced by running those two shader files on llvmpipe into an
RGBA8 colour buffer (oracle/make_goldens.py present()); the host function must give the same bytes, and the PNG writer
must store exactly those bytes."""
import json
import os
import struct
import zlib

import numpy as np
import pytest

from tdt4230_project_raytracing_amd import host

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return z["image"].view(np.float32), z["rgba8"]


def decode_png(data):
    """Minimal PNG reader (8-bit RGB / RGBA, filter type 0 only — what tdt_png_encode writes)."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(typ + body) & 0xFFFFFFFF, typ
        chunks.append((typ, body))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype, comp, flt, inter = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, comp, flt, inter) == (8, 0, 0, 0) and ctype in (2, 6)
    ch = 3 if ctype == 2 else 4
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + w * ch)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, ch)


@pytest.mark.parametrize("name", ["present_values", "present_demo"])
def test_present_equals_reference_quad_pass(name):
    image, want = load(name)
    got = host.present_rgba8(image, top_down=False)
    bad = np.argwhere(got != want)
    assert bad.size == 0, f"{name}: {len(bad)} bytes differ, first {bad[:3].tolist()}"
    assert (host.present_rgba8(image, top_down=True) == want[::-1]).all()      # file order: top scan-line first


def test_conversion_rule_on_every_float_class():
    # the rule the fixtures pin: clamp to [0,1] with NaN -> 0, x255 in fp32, round half to even
    image, want = load("present_values")
    x = np.clip(np.nan_to_num(image, nan=0.0, posinf=1.0, neginf=0.0), 0, 1).astype(np.float32) * np.float32(255)
    assert (np.rint(x).astype(np.uint8) == want).all()
    assert len(np.unique(want)) == 256                                         # every code occurs in the fixture


@pytest.mark.parametrize("with_alpha", [False, True])
def test_png_round_trip(with_alpha, tmp_path):
    image, want = load("present_demo")
    frame = host.present_rgba8(image, top_down=True)
    back = decode_png(host.png_encode(frame, with_alpha))
    assert (back == (frame if with_alpha else frame[..., :3])).all()
    path = tmp_path / "frame.png"
    host.png_write(str(path), frame, with_alpha)
    assert path.read_bytes() == host.png_encode(frame, with_alpha)
    assert len(path.read_bytes()) < frame.size                                 # deflate did something


def test_png_errors():
    with pytest.raises(RuntimeError):
        host.png_write("/nonexistent-dir/x.png", np.zeros((2, 2, 4), np.uint8))
