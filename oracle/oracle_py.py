"""TEST INFRASTRUCTURE ONLY — ctypes wrappers of the two checkers.

  Oracle : liboracle.so, the CPU restatement (pathtrace_oracle.c)
  GLRef  : _ref/libglref.so, the reference's own shader on Mesa llvmpipe (this container only:
           needs /root/reference at run time)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.environ.get("REF_DIR", "/root/reference")
REF_SHADER = os.path.join(REF_DIR, "assets", "shaders", "raytracer.comp")

STAT_FIELDS = ["pixels", "samples", "octree_hit_calls", "iterations", "node_loads", "lambertian", "metal",
               "dielectric", "unknown_material"]
_SLOT_NAMES = {0: "cells", 1: "materials", 2: "albedos", 3: "metal", 4: "dielectric", 6: "octree_floats", 7: "octree_ints"}


class _Cam(ctypes.Structure):
    _fields_ = [("image_width", ctypes.c_int32), ("image_height", ctypes.c_int32),
                ("horizontal", ctypes.c_float * 3), ("vertical", ctypes.c_float * 3),
                ("lower_left_corner", ctypes.c_float * 3), ("origin", ctypes.c_float * 3),
                ("samples_per_pixel", ctypes.c_int32), ("max_bounce", ctypes.c_int32)]


class _Scene(ctypes.Structure):
    _fields_ = [f for nm in ["cells", "materials", "albedos", "metal", "dielectric", "octree_floats", "octree_ints"]
                for f in [(nm, ctypes.c_void_p), (nm + "_bytes", ctypes.c_size_t)]]


class _Stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in STAT_FIELDS]


def _cam_struct(cam):
    c = _Cam()
    for name, _ in _Cam._fields_:
        v = getattr(cam, name)
        if name in ("horizontal", "vertical", "lower_left_corner", "origin"):
            getattr(c, name)[:] = [float(x) for x in v]
        else:
            setattr(c, name, int(v))
    return c


def algorithmic_bytes(stats):
    """bytes the path must read (node loads + material/albedo/attribute reads +
    the 40 B of octree uniforms) and write (16 B per pixel), from oracle event counts."""
    read = (8 * stats["node_loads"] + 24 * stats["lambertian"] + 28 * stats["metal"] + 16 * stats["dielectric"] + 40)
    return {"read": int(read), "write": int(16 * stats["pixels"])}


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = ctypes.CDLL(path)
        L.oracle_render.argtypes = [ctypes.POINTER(_Scene), ctypes.POINTER(_Cam)] + [ctypes.c_int] * 4 + \
            [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(_Stats)]
        L.oracle_accumulate_carry.argtypes = [ctypes.POINTER(_Scene), ctypes.POINTER(_Cam)] + [ctypes.c_int] * 6 + \
            [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(_Stats)]
        L.oracle_resolve.argtypes = [ctypes.POINTER(_Cam)] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
        for f in ("oracle_sin", "oracle_cos"):
            getattr(L, f).argtypes = [ctypes.c_float]
            getattr(L, f).restype = ctypes.c_float
        L.oracle_pow.argtypes = [ctypes.c_float, ctypes.c_float]
        L.oracle_pow.restype = ctypes.c_float
        self.L = L

    @staticmethod
    def _scene_struct(scene):
        s = _Scene()
        keep = []
        for slot, nm in _SLOT_NAMES.items():
            a = np.ascontiguousarray(scene.blobs[slot])
            keep.append(a)
            setattr(s, nm, a.ctypes.data if a.size else None)
            setattr(s, nm + "_bytes", a.nbytes)
        return s, keep

    def render(self, scene, cam, dispatch=None, rows=(0, 2**31 - 1), threads=1, want_stats=False, image=None):
        """Image of ComputeShader::dispatch_compute(*dispatch) (default (W+1, H+1, 1) as main.rs:579)."""
        W, H = cam.image_width, cam.image_height
        dw, dh = dispatch if dispatch else (W + 1, H + 1)
        s, keep = self._scene_struct(scene)
        c = _cam_struct(cam)
        img = image if image is not None else np.zeros((H, W, 4), np.float32)
        st = _Stats()
        self.L.oracle_render(ctypes.byref(s), ctypes.byref(c), dw, dh, rows[0], rows[1], img.ctypes.data, threads,
                             ctypes.byref(st) if want_stats else None)
        del keep
        if want_stats:
            return img, {n: int(getattr(st, n)) for n in STAT_FIELDS}
        return img

    def accumulate(self, scene, cam, accum, carry, spp_begin, spp_count, dispatch=None, rows=(0, 2**31 - 1), threads=1):
        W, H = cam.image_width, cam.image_height
        dw, dh = dispatch if dispatch else (W + 1, H + 1)
        s, keep = self._scene_struct(scene)
        c = _cam_struct(cam)
        self.L.oracle_accumulate_carry(ctypes.byref(s), ctypes.byref(c), dw, dh, rows[0], rows[1], spp_begin, spp_count,
                                       accum.ctypes.data, carry.ctypes.data if carry is not None else None, threads, None)
        del keep

    def resolve(self, cam, accum, total_spp, dispatch=None, rows=(0, 2**31 - 1)):
        W, H = cam.image_width, cam.image_height
        dw, dh = dispatch if dispatch else (W + 1, H + 1)
        img = np.zeros((H, W, 4), np.float32)
        c = _cam_struct(cam)
        self.L.oracle_resolve(ctypes.byref(c), dw, dh, rows[0], rows[1], total_spp, accum.ctypes.data, img.ctypes.data)
        return img

    def sin(self, x):
        return np.array([self.L.oracle_sin(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32)

    def cos(self, x):
        return np.array([self.L.oracle_cos(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32)

    def pow(self, x, y):
        return np.array([self.L.oracle_pow(float(v), float(y)) for v in np.asarray(x, np.float32).ravel()], np.float32)


def glref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libglref.so")) and os.path.exists(REF_SHADER)


class GLRef:
    """The reference shader itself on llvmpipe.  One GL context per process."""
    _inst = None

    def __init__(self):
        path = os.path.join(_HERE, "_ref", "libglref.so")
        L = ctypes.CDLL(path)
        L.glref_last_error.restype = ctypes.c_char_p
        L.glref_renderer.restype = ctypes.c_char_p
        L.glref_version.restype = ctypes.c_char_p
        L.glref_program.argtypes = [ctypes.c_char_p]
        L.glref_ssbo.argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_size_t]
        L.glref_ssbo_read.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
        L.glref_image_read.argtypes = [ctypes.c_void_p]
        L.glref_set_i32.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.glref_set_f32.argtypes = [ctypes.c_char_p, ctypes.c_float]
        L.glref_set_vec3f.argtypes = [ctypes.c_char_p, ctypes.c_float, ctypes.c_float, ctypes.c_float]
        L.glref_dispatch_compute.argtypes = [ctypes.c_int] * 3
        L.glref_dispatch_compute.restype = ctypes.c_double
        L.glref_buffer_variable.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        L.glref_image_write.argtypes = [ctypes.c_void_p]
        L.glref_present.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        if L.glref_init() != 0:
            raise RuntimeError(L.glref_last_error().decode())
        self.L = L
        self._prog = None

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def renderer(self):
        return self.L.glref_renderer().decode() + " / " + self.L.glref_version().decode()

    def program(self, path=REF_SHADER):
        if self._prog != path:
            rc = self.L.glref_program(path.encode())
            if rc != 0:
                raise RuntimeError(f"shader {path}: {self.L.glref_last_error().decode()}")
            self._prog = path

    def upload_scene(self, scene):
        self.L.glref_free_buffers()
        for slot in (0, 1, 2, 3, 4, 6, 7):
            a = np.ascontiguousarray(scene.blobs[slot])
            rc = self.L.glref_ssbo(slot, a.ctypes.data, a.nbytes)
            if rc != 0:
                raise RuntimeError(f"glref_ssbo slot {slot}: GL error {rc:#x}")

    def set_camera(self, cam):
        L = self.L
        def chk(rc):
            if rc != 0:
                raise RuntimeError(L.glref_last_error().decode())
        chk(L.glref_set_i32(b"camera.image_width", cam.image_width))
        chk(L.glref_set_i32(b"camera.image_height", cam.image_height))
        for n in ("horizontal", "vertical", "lower_left_corner", "origin"):
            v = getattr(cam, n)
            chk(L.glref_set_vec3f(("camera." + n).encode(), v[0], v[1], v[2]))
        chk(L.glref_set_i32(b"camera.samples_per_pixel", cam.samples_per_pixel))
        chk(L.glref_set_i32(b"camera.max_bounce", cam.max_bounce))

    def render(self, scene, cam, dispatch=None, want_time=False, upload=True):
        """Exactly the reference's frame: upload (main.rs:234-470), uniforms (camera.rs:241-253),
        dispatch_compute(W+1, H+1, 1) (main.rs:579), then read the texture back."""
        self.program()
        if upload:
            self.upload_scene(scene)
        W, H = cam.image_width, cam.image_height
        self.L.glref_image(W, H)
        self.set_camera(cam)
        dw, dh = dispatch if dispatch else (W + 1, H + 1)
        t = self.L.glref_dispatch_compute(dw, dh, 1)
        if t < 0:
            raise RuntimeError("glref dispatch failed")
        img = np.zeros((H, W, 4), np.float32)
        self.L.glref_image_read(img.ctypes.data)
        return (img, t) if want_time else img


def glref_present(image, viewport=None):
    """The reference's presentation pass (quad.vert + quad.frag, main.rs:113-153, 582-600) on llvmpipe for a W x H
    RGBA32F render texture (row 0 = bottom): the RGBA8 frame a window of `viewport` = (w, h) pixels would hold,
    as glReadPixels returns it (row 0 = bottom)."""
    g = GLRef.get()
    image = np.ascontiguousarray(image, np.float32)
    H, W = image.shape[:2]
    vw, vh = viewport if viewport else (W, H)
    g.L.glref_image(W, H)
    assert g.L.glref_image_write(image.ctypes.data) == 0
    out = np.zeros((vh, vw, 4), np.uint8)
    rc = g.L.glref_present(os.path.join(REF_DIR, "assets", "shaders", "quad.vert").encode(),
                           os.path.join(REF_DIR, "assets", "shaders", "quad.frag").encode(), vw, vh, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"glref_present: {g.L.glref_last_error().decode()}")
    return out


def glref_math_table(xs):
    """Run oracle/glref/probe_math.comp (our own test shader) on llvmpipe: returns [n][8] =
    sin, cos, pow(x,5), inversesqrt, 1/x, sqrt, fract, fract(sin(x)*43758.5453) of every input."""
    g = GLRef.get()
    xs = np.ascontiguousarray(xs, np.float32)
    n = (xs.size + 63) // 64 * 64
    xin = np.zeros(n, np.float32)
    xin[:xs.size] = xs
    out = np.zeros(n * 8, np.float32)
    g.program(os.path.join(_HERE, "glref", "probe_math.comp"))
    g.L.glref_free_buffers()
    assert g.L.glref_ssbo(0, xin.ctypes.data, xin.nbytes) == 0
    assert g.L.glref_ssbo(1, out.ctypes.data, out.nbytes) == 0
    t = g.L.glref_dispatch_compute(n, 1, 1)       # floor(n / 64) groups
    assert t >= 0
    assert g.L.glref_ssbo_read(1, out.ctypes.data, out.nbytes) == 0
    g._prog = None
    g.L.glref_free_buffers()
    return out.reshape(n, 8)[:xs.size]


# ---- next row §8f-2: assets/shaders/octree_update.comp ---------------------------------------------
UPDATE_SHADER = os.path.join(REF_DIR, "assets", "shaders", "octree_update.comp")


def oracle_octree_update(oracle, scene, delta, counter, dispatch):
    """Oracle restatement of the edit kernel: returns (cells_after, counter_after)."""
    L = oracle.L
    L.oracle_octree_update.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.c_int, ctypes.c_int]
    cells = np.ascontiguousarray(scene.blobs[0]).copy()
    d = np.ascontiguousarray(delta, np.float32)
    of, oi = np.ascontiguousarray(scene.blobs[6]), np.ascontiguousarray(scene.blobs[7])
    c = ctypes.c_uint32(counter)
    L.oracle_octree_update(cells.ctypes.data, cells.nbytes, d.ctypes.data, d.nbytes, of.ctypes.data, of.nbytes,
                           oi.ctypes.data, oi.nbytes, ctypes.byref(c), *dispatch)
    return cells, int(c.value)


def glref_octree_update(scene, delta, counter, dispatch):
    """The reference's own octree_update.comp on llvmpipe: returns (cells_after, counter_after)."""
    g = GLRef.get()
    L = g.L
    L.glref_atomic_counter.argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_size_t]
    g.program(UPDATE_SHADER)
    g._prog = None
    L.glref_free_buffers()
    idx = {}
    for slot in (0, 6, 7):
        a = np.ascontiguousarray(scene.blobs[slot])
        idx[slot] = L.glref_buffer_count()
        assert L.glref_ssbo(slot, a.ctypes.data, a.nbytes) == 0
    d = np.ascontiguousarray(delta, np.float32)
    assert L.glref_ssbo(5, d.ctypes.data, d.nbytes) == 0
    c = np.array([counter], np.uint32)
    ic = L.glref_buffer_count()
    assert L.glref_atomic_counter(0, c.ctypes.data, 4) == 0
    assert L.glref_dispatch_compute(*dispatch) >= 0
    cells = np.zeros_like(np.ascontiguousarray(scene.blobs[0]))
    assert L.glref_ssbo_read(idx[0], cells.ctypes.data, cells.nbytes) == 0
    cnt = np.zeros(1, np.uint32)
    assert L.glref_ssbo_read(ic, cnt.ctypes.data, 4) == 0
    L.glref_free_buffers()
    return cells, int(cnt[0])
