"""the camera controller and its settings file (src/renderer/camera.rs:8-102, main.rs:170-176, 492-546).

PARITY UNPINNED for the controller arithmetic: it is the cgmath crate's (Cargo.lock: cgmath 0.18.0), which is neither
vendored nor buildable here, and the reference holds no fixture of camera values.  These tests pin what can be pinned:
the identity-orientation case against CameraBuilder::build (which the render goldens cover), exact cases of the
formulas, the algebra against an independent float64 model, and the settings grammar against the reference's camera.ron
values."""
import math

import numpy as np
import pytest

from tdt4230_project_raytracing_amd import host

POSE = dict(vertical_fov=90.0, image_width=640, aspect_ratio=640 / 360, origin=(0.0, -0.1, -0.3), viewport_height=2.0,
            samples_per_pixel=4, max_bounce=6)
# assets/settings/camera.ron (the values main.rs:170-176 feeds the builder)
RON = "CameraSettings(\n    samples_per_pixel: 4,\n    max_bounce: 6,\n    turn_rate: 0.05,\n    normal_speed: 0.03,\n    sprint_speed: 0.15,\n)"


def vec(a):
    return np.array(list(a), np.float64)


def test_identity_orientation_equals_the_builder():
    cam = host.Camera(**POSE)
    u, want = cam.uniforms(), host.camera_build(**POSE)
    for name, _ in host.CameraUniforms._fields_:
        a, b = getattr(u, name), getattr(want, name)
        assert (list(a) == list(b)) if hasattr(a, "__len__") else (a == b), name
    assert list(cam.c.pitch) == [1, 0, 0, 0] and list(cam.c.yaw) == [1, 0, 0, 0]                 # camera.rs:178-179
    s = cam.c.settings
    assert (s.turn_rate, s.normal_speed, s.sprint_speed, cam.c.movement_speed) == (np.float32(0.025), 1.0, 2.0, 1.0)   # :167-169, 190


def test_builder_controller_options():
    cam = host.Camera(**POSE, turn_rate=0.05, normal_speed=0.03)
    s = cam.c.settings
    assert s.turn_rate == np.float32(0.05) and s.normal_speed == np.float32(0.03)
    assert s.sprint_speed == np.float32(0.03) * np.float32(2.0)                                  # unwrap_or(normal_speed * 2.0)
    assert host.Camera(**POSE, sprint_speed=0.15).c.settings.sprint_speed == np.float32(0.15)


def test_translate_is_exact_at_identity_orientation():
    cam = host.Camera(**POSE, normal_speed=0.03, sprint_speed=0.15)
    dt = 1 / 60
    cam.translate("Front", dt)                              # origin += rotate(identity, by * dt as f32 * speed) = that vector
    step = np.float32(-1.0) * np.float32(dt) * np.float32(0.03)
    assert list(cam.c.origin) == [0.0, np.float32(-0.1), np.float32(-0.3) + step]
    cam.set_speed_to_sprint()
    cam.translate("Rigth", dt)
    assert cam.c.origin[0] == np.float32(1.0) * np.float32(dt) * np.float32(0.15)
    cam.set_speed_to_normal()
    assert cam.c.movement_speed == np.float32(0.03)
    u = cam.uniforms()                                      # lower_left_corner follows the origin (camera.rs:76)
    llc = (np.float32(u.origin[2]) - np.float32(u.horizontal[2]) * np.float32(0.5) - np.float32(u.vertical[2]) * np.float32(0.5)) - np.float32(1.0)
    assert u.lower_left_corner[2] == llc


def test_look_at_world_point():
    cam = host.Camera(**POSE)
    assert cam.look_at_world_point(0.25) == [0.0, np.float32(-0.1), np.float32(-0.25) + np.float32(-0.3)]      # main.rs:555
    cam.turn_yaw(math.pi / 2 / 0.025 / 2)                   # half-angle = angle * turn_rate: a quarter turn about +y
    p = vec(cam.look_at_world_point(1.0)) - vec(cam.c.origin)
    assert np.allclose(p, [-1.0, 0.0, 0.0], atol=2e-6)      # -unit_z rotated by +90 degrees about y


# ---- an independent float64 model of the same formulas -------------------------------------------------------------
def qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def qrot(q, v):
    qv = q[1:]
    return v + 2.0 * np.cross(qv, np.cross(qv, v) + q[0] * v)


class Model:
    def __init__(self, cam):
        self.origin, self.pitch, self.yaw = vec(cam.c.origin), vec(cam.c.pitch), vec(cam.c.yaw)
        self.vw, self.vh, self.rate, self.speed = cam.c.viewport_width, cam.c.viewport_height, cam.c.settings.turn_rate, cam.c.movement_speed

    def q(self):
        q = qmul(self.yaw, self.pitch)
        return q / np.linalg.norm(q)

    def basis(self):
        f = qrot(self.q(), np.array([0.0, 0.0, 1.0])); f /= np.linalg.norm(f)
        r = np.cross([0.0, 1.0, 0.0], f); r /= np.linalg.norm(r)
        u = np.cross(f, r); u /= np.linalg.norm(u)
        return f, r, u


def test_random_walk_matches_float64_model():
    rng = np.random.default_rng(7)
    cam = host.Camera(**POSE, turn_rate=0.05, normal_speed=0.03, sprint_speed=0.15)
    m = Model(cam)
    for _ in range(200):
        k = rng.integers(0, 3)
        a = float(rng.uniform(-1.5, 1.5))
        if k == 0:
            cam.turn_yaw(a)
            h = np.float32(a) * np.float32(m.rate)
            m.yaw = qmul(m.yaw, [math.cos(h), 0.0, math.sin(h), 0.0])
        elif k == 1 and abs(m.basis()[0][1]) < 0.9:         # keep away from the pole where `right` degenerates (as the reference would)
            cam.turn_pitch(a * 0.2)
            h = np.float32(a * 0.2) * np.float32(m.rate)
            m.pitch = qmul(m.pitch, [math.cos(h), math.sin(h), 0.0, 0.0])
        else:
            by = list(host.DIRECTION.values())[rng.integers(0, 6)]
            cam.translate(by, 1 / 60)
            m.origin = m.origin + qrot(m.q(), vec(by) * np.float32(1 / 60) * m.speed)
    f, r, u = m.basis()
    got = cam.uniforms()
    assert np.allclose(vec(got.origin), m.origin, atol=2e-5)
    assert np.allclose(vec(got.horizontal), r * m.vw, atol=5e-5) and np.allclose(vec(got.vertical), u * m.vh, atol=5e-5)
    assert np.allclose(vec(got.lower_left_corner), m.origin - r * m.vw / 2 - u * m.vh / 2 - f, atol=1e-4)
    # the frame stays orthogonal and keeps the viewport's size
    h, v = vec(got.horizontal), vec(got.vertical)
    assert abs(h @ v) < 1e-5 and abs(np.linalg.norm(h) - m.vw) < 1e-5 and abs(np.linalg.norm(v) - m.vh) < 1e-5
    assert abs(np.linalg.norm(vec(cam.c.yaw)) - 1) < 1e-4 and abs(np.linalg.norm(vec(cam.c.pitch)) - 1) < 1e-4


def test_settings_file():
    s = host.CameraSettings.from_ron(RON)
    assert (s.samples_per_pixel, s.max_bounce) == (4, 6)
    assert (s.turn_rate, s.normal_speed, s.sprint_speed) == (np.float32(0.05), np.float32(0.03), np.float32(0.15))
    # RON freedoms serde's derive accepts: no struct name, any order, comments, no trailing comma, integers for floats
    t = host.CameraSettings.from_ron("( sprint_speed: 1, /* nested /* comment */ */ normal_speed: 5e-1, // x\n turn_rate: .25, max_bounce: 3, samples_per_pixel: 1_0 )")
    assert (t.samples_per_pixel, t.max_bounce, t.turn_rate, t.normal_speed, t.sprint_speed) == (10, 3, 0.25, 0.5, 1.0)
    for bad, msg in (("CameraSettings(samples_per_pixel: 4, max_bounce: 6, turn_rate: 0.05, normal_speed: 0.03)", "missing field `sprint_speed`"),
                     (RON.replace("4,", "4.5,"), "must be an integer"), (RON.replace("CameraSettings", "Settings"), "expected `CameraSettings(`"),
                     (RON + " x", "trailing"), (RON.replace("max_bounce: 6,", "max_bounce: 6, max_bounce: 7,"), "duplicate"), ("", "expected `(`")):
        with pytest.raises(RuntimeError) as e:
            host.CameraSettings.from_ron(bad)
        assert msg in str(e.value), (bad, str(e.value))


def test_apply_settings_changes_what_the_next_dispatch_uses():
    cam = host.Camera(**POSE)
    cam.apply_settings(host.CameraSettings.from_ron(RON))                                       # camera.rs:96-101
    u = cam.uniforms()
    assert (u.samples_per_pixel, u.max_bounce) == (4, 6)
    assert cam.c.movement_speed == 1.0                       # as in the reference: unchanged until set_speed_to_* is called
    cam.set_speed_to_sprint()
    assert cam.c.movement_speed == np.float32(0.15)
