"""ctypes binding of libtdthost.so (include/tdt_host.h): camera uniforms and scene payloads.

Host-side restatement of the reference's Rust host code — CameraBuilder::build
(src/renderer/camera.rs:135-196), Octree::init_global_buffers (src/renderer/octree.rs:40-100),
the demo scene literal (src/main.rs:235-463) — plus the synthetic-scene generators.  Pure
host code; no HIP, no oracle.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtdthost.so")

SLOTS = (0, 1, 2, 3, 4, 6, 7)
SLOT_DTYPES = {0: np.uint32, 1: np.uint32, 2: np.float32, 3: np.float32, 4: np.float32, 6: np.float32, 7: np.int32}
SLOT_NAMES = {0: "cells", 1: "materials", 2: "albedos", 3: "metal", 4: "dielectric", 6: "octree_floats", 7: "octree_ints"}


class CameraBuilderC(ctypes.Structure):
    _fields_ = [
        ("vertical_fov", ctypes.c_float), ("image_width", ctypes.c_int32),
        ("has_aspect_ratio", ctypes.c_int32), ("aspect_ratio", ctypes.c_float),
        ("has_viewport_height", ctypes.c_int32), ("viewport_height", ctypes.c_float),
        ("has_origin", ctypes.c_int32), ("origin", ctypes.c_float * 3),
        ("has_samples_per_pixel", ctypes.c_int32), ("samples_per_pixel", ctypes.c_int32),
        ("has_max_bounce", ctypes.c_int32), ("max_bounce", ctypes.c_int32),
        ("has_turn_rate", ctypes.c_int32), ("turn_rate", ctypes.c_float),
        ("has_normal_speed", ctypes.c_int32), ("normal_speed", ctypes.c_float),
        ("has_sprint_speed", ctypes.c_int32), ("sprint_speed", ctypes.c_float),
    ]


class CameraUniforms(ctypes.Structure):
    """`uniform Camera camera` (raytracer.comp:133-146) as initial_uniforms sends it (camera.rs:241-253)."""
    _fields_ = [
        ("image_width", ctypes.c_int32), ("image_height", ctypes.c_int32),
        ("horizontal", ctypes.c_float * 3), ("vertical", ctypes.c_float * 3),
        ("lower_left_corner", ctypes.c_float * 3), ("origin", ctypes.c_float * 3),
        ("samples_per_pixel", ctypes.c_int32), ("max_bounce", ctypes.c_int32),
    ]

    def copy(self):
        c = CameraUniforms()
        ctypes.memmove(ctypes.byref(c), ctypes.byref(self), ctypes.sizeof(self))
        return c


class CameraSettings(ctypes.Structure):
    """CameraSettings (camera.rs:9-16) = assets/settings/camera.ron."""
    _fields_ = [("samples_per_pixel", ctypes.c_int32), ("max_bounce", ctypes.c_int32), ("turn_rate", ctypes.c_float),
                ("normal_speed", ctypes.c_float), ("sprint_speed", ctypes.c_float)]

    @classmethod
    def from_ron(cls, text):
        """ron::de::from_bytes::<CameraSettings> (main.rs:171, 493)."""
        data = text.encode() if isinstance(text, str) else bytes(text)
        out = cls()
        _check(lib().tdt_camera_settings_from_ron(data, len(data), ctypes.byref(out)))
        return out


class CameraC(ctypes.Structure):
    _fields_ = [("horizontal", ctypes.c_float * 3), ("vertical", ctypes.c_float * 3),
                ("viewport_width", ctypes.c_float), ("viewport_height", ctypes.c_float),
                ("lower_left_corner", ctypes.c_float * 3), ("origin", ctypes.c_float * 3),
                ("pitch", ctypes.c_float * 4), ("yaw", ctypes.c_float * 4),
                ("image_width", ctypes.c_int32), ("image_height", ctypes.c_int32),
                ("settings", CameraSettings), ("movement_speed", ctypes.c_float)]


# Direction::into_vector3 (utility/mod.rs:15-26)
DIRECTION = {"Front": (0.0, 0.0, -1.0), "Back": (0.0, 0.0, 1.0), "Rigth": (1.0, 0.0, 0.0), "Left": (-1.0, 0.0, 0.0),
             "Up": (0.0, 1.0, 0.0), "Down": (0.0, -1.0, 0.0)}


class Camera:
    """The reference's Camera (camera.rs:20-102) without the GL objects: the controller state and the uniforms it sends
    (cgmath restated, parity unpinned — see include/tdt_host.h)."""

    def __init__(self, vertical_fov, image_width, aspect_ratio=None, viewport_height=None, origin=None, samples_per_pixel=None,
                 max_bounce=None, turn_rate=None, normal_speed=None, sprint_speed=None):
        b = _builder(vertical_fov, image_width, aspect_ratio, viewport_height, origin, samples_per_pixel, max_bounce)
        for name, v in (("turn_rate", turn_rate), ("normal_speed", normal_speed), ("sprint_speed", sprint_speed)):
            if v is not None:
                setattr(b, "has_" + name, 1)
                setattr(b, name, v)
        self.c = CameraC()
        _check(lib().tdt_camera_init(ctypes.byref(b), ctypes.byref(self.c)))

    def translate(self, by, deltatime):
        v = (ctypes.c_float * 3)(*(DIRECTION[by] if isinstance(by, str) else by))
        _check(lib().tdt_camera_translate(ctypes.byref(self.c), v, ctypes.c_double(deltatime)))

    def turn_pitch(self, angle):
        _check(lib().tdt_camera_turn_pitch(ctypes.byref(self.c), ctypes.c_float(angle)))

    def turn_yaw(self, angle):
        _check(lib().tdt_camera_turn_yaw(ctypes.byref(self.c), ctypes.c_float(angle)))

    def set_speed_to_normal(self):
        lib().tdt_camera_set_speed_to_normal(ctypes.byref(self.c))

    def set_speed_to_sprint(self):
        lib().tdt_camera_set_speed_to_sprint(ctypes.byref(self.c))

    def look_at_world_point(self, distance):
        out = (ctypes.c_float * 3)()
        _check(lib().tdt_camera_look_at_world_point(ctypes.byref(self.c), ctypes.c_float(distance), out))
        return list(out)

    def apply_settings(self, settings):
        _check(lib().tdt_camera_apply_settings(ctypes.byref(self.c), ctypes.byref(settings)))

    def uniforms(self):
        u = CameraUniforms()
        _check(lib().tdt_camera_get_uniforms(ctypes.byref(self.c), ctypes.byref(u)))
        return u


# ---------------------------------------------------------------- presentation ---
def present_rgba8(image, top_down=True):
    """What the reference's quad pass leaves in an RGBA8 back buffer (quad.frag:10, main.rs:582-600) for an (H, W, 4) float32
    render texture with row 0 at the bottom; host-side counterpart of rt.Texture.read_rgba8."""
    image = np.ascontiguousarray(image, np.float32)
    h, w = image.shape[:2]
    out = np.empty((h, w, 4), np.uint8)
    _check(lib().tdt_present_rgba8(image.ctypes.data, w, h, 1 if top_down else 0, out.ctypes.data))
    return out


def png_encode(rgba8, with_alpha=False):
    """8-bit PNG bytes of a top-down (H, W, 4) uint8 frame."""
    rgba8 = np.ascontiguousarray(rgba8, np.uint8)
    h, w = rgba8.shape[:2]
    buf, n = ctypes.c_void_p(), ctypes.c_size_t(0)
    _check(lib().tdt_png_encode(rgba8.ctypes.data, w, h, 1 if with_alpha else 0, ctypes.byref(buf), ctypes.byref(n)))
    try:
        return ctypes.string_at(buf, n.value)
    finally:
        lib().tdt_host_free(buf)


def png_write(path, rgba8, with_alpha=False):
    rgba8 = np.ascontiguousarray(rgba8, np.uint8)
    h, w = rgba8.shape[:2]
    _check(lib().tdt_png_write(os.fsencode(path), rgba8.ctypes.data, w, h, 1 if with_alpha else 0))


class SceneParams(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("max_depth", ctypes.c_int32), ("cell_count", ctypes.c_int32),
                ("max_iter", ctypes.c_int32), ("seed", ctypes.c_uint64)]


SCENE_HASH_GRID, SCENE_TERRAIN, SCENE_SHELLS = 0, 1, 2

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = ctypes.CDLL(_LIB_PATH)
        L.tdt_host_last_error.restype = ctypes.c_char_p
        L.tdt_camera_build.argtypes = [ctypes.POINTER(CameraBuilderC), ctypes.POINTER(CameraUniforms)]
        L.tdt_camera_reference_pose.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(CameraUniforms)]
        L.tdt_scene_demo.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        L.tdt_scene_generate.argtypes = [ctypes.POINTER(SceneParams), ctypes.POINTER(ctypes.c_void_p)]
        L.tdt_scene_config.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.tdt_scene_from_blobs.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                           ctypes.POINTER(ctypes.c_void_p)]
        L.tdt_scene_destroy.argtypes = [ctypes.c_void_p]
        L.tdt_scene_destroy.restype = None
        L.tdt_scene_blob.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]
        L.tdt_scene_blob.restype = ctypes.c_void_p
        L.tdt_scene_counts.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
        L.tdt_camera_init.argtypes = [ctypes.POINTER(CameraBuilderC), ctypes.POINTER(CameraC)]
        L.tdt_camera_translate.argtypes = [ctypes.POINTER(CameraC), ctypes.POINTER(ctypes.c_float), ctypes.c_double]
        L.tdt_camera_turn_pitch.argtypes = [ctypes.POINTER(CameraC), ctypes.c_float]
        L.tdt_camera_turn_yaw.argtypes = [ctypes.POINTER(CameraC), ctypes.c_float]
        L.tdt_camera_set_speed_to_normal.argtypes = [ctypes.POINTER(CameraC)]
        L.tdt_camera_set_speed_to_normal.restype = None
        L.tdt_camera_set_speed_to_sprint.argtypes = [ctypes.POINTER(CameraC)]
        L.tdt_camera_set_speed_to_sprint.restype = None
        L.tdt_camera_look_at_world_point.argtypes = [ctypes.POINTER(CameraC), ctypes.c_float, ctypes.POINTER(ctypes.c_float)]
        L.tdt_camera_apply_settings.argtypes = [ctypes.POINTER(CameraC), ctypes.POINTER(CameraSettings)]
        L.tdt_camera_get_uniforms.argtypes = [ctypes.POINTER(CameraC), ctypes.POINTER(CameraUniforms)]
        L.tdt_camera_settings_from_ron.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(CameraSettings)]
        L.tdt_present_rgba8.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.tdt_png_encode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.POINTER(ctypes.c_size_t)]
        L.tdt_png_write.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.tdt_host_free.argtypes = [ctypes.c_void_p]
        L.tdt_host_free.restype = None
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("libtdthost: " + lib().tdt_host_last_error().decode())


class Scene:
    """The seven SSBO payloads of one scene, as numpy arrays (copies) keyed by binding slot."""

    def __init__(self, blobs, counts=None, name=""):
        self.blobs = {int(k): np.ascontiguousarray(v) for k, v in blobs.items()}
        self.counts = counts
        self.name = name

    @classmethod
    def _from_handle(cls, h, name):
        L = lib()
        blobs = {}
        for slot in SLOTS:
            n = ctypes.c_size_t(0)
            p = L.tdt_scene_blob(h, slot, ctypes.byref(n))
            dt = np.dtype(SLOT_DTYPES[slot])
            if n.value:
                buf = (ctypes.c_char * n.value).from_address(p)
                blobs[slot] = np.frombuffer(buf, dtype=dt).copy()
            else:
                blobs[slot] = np.zeros(0, dt)
        cnt = (ctypes.c_int64 * 6)()
        _check(L.tdt_scene_counts(h, cnt))
        L.tdt_scene_destroy(h)
        keys = ("cells", "parents", "leaves", "empties", "materials", "voxels")
        return cls(blobs, dict(zip(keys, list(cnt))), name)

    @classmethod
    def demo(cls):
        h = ctypes.c_void_p()
        _check(lib().tdt_scene_demo(ctypes.byref(h)))
        return cls._from_handle(h, "demo")

    @classmethod
    def config(cls, n):
        h = ctypes.c_void_p()
        _check(lib().tdt_scene_config(int(n), ctypes.byref(h)))
        return cls._from_handle(h, f"config{n}")

    @classmethod
    def generate(cls, kind, max_depth, cell_count, max_iter, seed):
        p = SceneParams(kind, max_depth, cell_count, max_iter, seed)
        h = ctypes.c_void_p()
        _check(lib().tdt_scene_generate(ctypes.byref(p), ctypes.byref(h)))
        return cls._from_handle(h, f"gen{kind}_d{max_depth}_s{seed:x}")

    @property
    def max_depth(self):
        return int(self.blobs[7][0])

    @property
    def max_iter(self):
        return int(self.blobs[7][1])

    @property
    def cell_count(self):
        return int(self.blobs[7][2])

    def nbytes(self):
        return sum(a.nbytes for a in self.blobs.values())


def scene_with_cell_count(scene, cell_count, zero_tail_nodes=0):
    """The same tree as a host with other conventions uploads it: Octree::init_global_buffers writes floats[6] = 1.0 / cell_count as
    f32 and ints[2] = cell_count (octree.rs:49, 79) — the reference's main.rs passes 100000 — and a pre-allocated cells buffer ends in
    zero nodes (main.rs:339-341)."""
    blobs = {k: v.copy() for k, v in scene.blobs.items()}
    blobs[6][6] = np.float32(1.0) / np.float32(cell_count)
    blobs[7][2] = cell_count
    if zero_tail_nodes:
        blobs[0] = np.concatenate([blobs[0], np.zeros(2 * int(zero_tail_nodes), np.uint32)])
    return Scene(blobs, scene.counts, f"{scene.name}_cc{cell_count}" + (f"_tail{zero_tail_nodes}" if zero_tail_nodes else ""))


def _builder(vertical_fov, image_width, aspect_ratio=None, viewport_height=None, origin=None, samples_per_pixel=None,
             max_bounce=None):
    b = CameraBuilderC()
    b.vertical_fov = vertical_fov
    b.image_width = image_width
    if aspect_ratio is not None:
        b.has_aspect_ratio, b.aspect_ratio = 1, aspect_ratio
    if viewport_height is not None:
        b.has_viewport_height, b.viewport_height = 1, viewport_height
    if origin is not None:
        b.has_origin = 1
        b.origin[:] = list(origin)
    if samples_per_pixel is not None:
        b.has_samples_per_pixel, b.samples_per_pixel = 1, samples_per_pixel
    if max_bounce is not None:
        b.has_max_bounce, b.max_bounce = 1, max_bounce
    return b


def camera_build(vertical_fov, image_width, aspect_ratio=None, viewport_height=None, origin=None,
                 samples_per_pixel=None, max_bounce=None):
    """CameraBuilder::new(fov, width).with_*().build() -> the uniforms it uploads (camera.rs:119-253)."""
    b = _builder(vertical_fov, image_width, aspect_ratio, viewport_height, origin, samples_per_pixel, max_bounce)
    u = CameraUniforms()
    _check(lib().tdt_camera_build(ctypes.byref(b), ctypes.byref(u)))
    return u


def camera_reference_pose(width, height, spp, max_bounce):
    """The camera main.rs:165-168 builds, for a width x height window."""
    u = CameraUniforms()
    _check(lib().tdt_camera_reference_pose(width, height, spp, max_bounce, ctypes.byref(u)))
    return u


# ---------------------------------------------------------------- scene ingest ---
def cantor_pair(r, g, b):
    """ply_point_loader.rs:228-241 (f64 arithmetic, saturating `as u32`)."""
    fd = 0.5 * (r + g) * (r + g + 1.0) + g
    h = 0.5 * (fd + b) * (fd + b + 1.0) + b
    return 0xFFFFFFFF if h >= 4294967295.0 else int(h)


class Ply:
    """PlyFileContent of ply_point_loader::from_resources (ply_point_loader.rs:84-93) for a byte buffer."""

    def __init__(self, data, strict_crlf=True):
        L = lib()
        L.tdt_ply_parse.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.tdt_ply_destroy.argtypes = [ctypes.c_void_p]
        L.tdt_ply_destroy.restype = None
        L.tdt_ply_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                   ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)]
        L.tdt_ply_voxels.argtypes = [ctypes.c_void_p]
        L.tdt_ply_voxels.restype = ctypes.POINTER(ctypes.c_int32)
        L.tdt_ply_albedos.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        L.tdt_ply_albedos.restype = ctypes.c_int64
        L.tdt_scene_from_ply.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        h = ctypes.c_void_p()
        _check(L.tdt_ply_parse(bytes(data), len(data), 1 if strict_crlf else 0, ctypes.byref(h)))
        self._h = h
        hv, nv, na = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        mp = (ctypes.c_int32 * 3)()
        _check(L.tdt_ply_info(h, ctypes.byref(hv), ctypes.byref(nv), mp, ctypes.byref(na)))
        self.header_vertex, self.min_point = hv.value, list(mp)
        n = nv.value
        v = np.ctypeslib.as_array(L.tdt_ply_voxels(h), shape=(n, 4)).copy() if n else np.zeros((0, 4), np.int32)
        self.positions = v[:, :3]
        self.albedo_keys = v[:, 3].view(np.uint32) if n else np.zeros(0, np.uint32)
        keys = np.zeros(na.value, np.uint32)
        rgb = np.zeros((na.value, 3), np.uint8)
        L.tdt_ply_albedos(h, keys.ctypes.data, rgb.ctypes.data, na.value)
        self.albedos = {int(k): tuple(int(c) for c in col) for k, col in zip(keys, rgb)}

    def to_scene(self, max_iter=256, z_up=True):
        h = ctypes.c_void_p()
        _check(lib().tdt_scene_from_ply(self._h, max_iter, 1 if z_up else 0, ctypes.byref(h)))
        return Scene._from_handle(h, "ply")

    def __del__(self):
        try:
            if self._h:
                lib().tdt_ply_destroy(self._h)
        except Exception:
            pass
