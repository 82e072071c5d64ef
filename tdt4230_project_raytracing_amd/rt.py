"""ctypes binding of libtdtrt.so (include/tdt_rt.h) and a Python mirror of the reference's
`src/renderer` wrapper API over it (same names and argument meaning, so tests read like the
reference's host code in main.rs:156-470, 579):

    ComputeShader(ctx).dispatch_compute(w, h, d)      compute_shader.rs:15-38
    Program.set_i32 / set_f32 / set_vector3_f32 / set_vector3_i32     program.rs:35-83
    VertexBufferObject(ctx, array) + bind_buffer_base(SSBO, slot, vbo)  vbo.rs:32-55, main.rs:352
    Texture.new_2d(ctx, w, h)                         texture.rs:47-75

There is no fallback: if the library or a HIP device is missing, construction raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TDT_LIB: alternative build of the same library (A/B experiments); never a different implementation
LIB_PATH = os.environ.get("TDT_LIB") or os.path.join(_HERE, "libtdtrt.so")

OK, ERR_NO_DEVICE, ERR_HIP, ERR_VARIABLE_NOT_FOUND, ERR_INCOMPLETE = 0, 1, 2, 3, 4
ERR_INVALID_ENUM, ERR_INVALID_VALUE, ERR_INVALID_OPERATION = 0x0500, 0x0501, 0x0502
PROGRAM_RAYTRACER, PROGRAM_OCTREE_UPDATE = 0, 1
SHADER_STORAGE_BUFFER, ATOMIC_COUNTER_BUFFER = 0x90D2, 0x92C0

# every symbol include/tdt_rt.h declares: (name, restype, argtypes)
_P, _I, _U, _F, _S = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_float, ctypes.c_size_t
_PP = ctypes.POINTER(ctypes.c_void_p)
SYMBOLS = [
    ("tdt_ctx_create", _I, [_I, _P, _PP]),
    ("tdt_ctx_create_multi", _I, [_I, ctypes.POINTER(ctypes.c_int), _PP]),
    ("tdt_ctx_device_count", _I, [_P]),
    ("tdt_ctx_destroy", None, [_P]),
    ("tdt_finish", _I, [_P]),
    ("tdt_last_error", ctypes.c_char_p, [_P]),
    ("tdt_strerror", ctypes.c_char_p, [_I]),
    ("tdt_compute_create", _I, [_P, _I, _PP]),
    ("tdt_compute_destroy", None, [_P]),
    ("tdt_compute_group_size", _I, [_P, ctypes.POINTER(ctypes.c_int)]),
    ("tdt_set_i32", _I, [_P, ctypes.c_char_p, ctypes.c_int32]),
    ("tdt_set_f32", _I, [_P, ctypes.c_char_p, _F]),
    ("tdt_set_vec3f", _I, [_P, ctypes.c_char_p, _F, _F, _F]),
    ("tdt_set_vec3i", _I, [_P, ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    ("tdt_buffer_create", _I, [_P, _P, _S, _PP]),
    ("tdt_buffer_destroy", None, [_P]),
    ("tdt_bind_buffer_base", _I, [_P, _I, _U, _P]),
    ("tdt_buffer_sub_data", _I, [_P, _S, _S, _P]),
    ("tdt_buffer_read", _I, [_P, _S, _S, _P]),
    ("tdt_image_create_rgba32f", _I, [_P, _I, _I, _PP]),
    ("tdt_image_wrap_device", _I, [_P, _P, _I, _I, _PP]),
    ("tdt_image_destroy", None, [_P]),
    ("tdt_bind_image", _I, [_P, _U, _P]),
    ("tdt_image_width", _I, [_P]),
    ("tdt_image_height", _I, [_P]),
    ("tdt_image_device_ptr", _P, [_P]),
    ("tdt_image_read", _I, [_P, _P]),
    ("tdt_image_read_rgba8", _I, [_P, _I, _P]),
    ("tdt_dispatch_compute", _I, [_P, _I, _I, _I]),
    ("tdt_set_partition", _I, [_P, _I, _I]),
    ("tdt_dispatch_accumulate", _I, [_P, _I, _I, _I, _I, _I, _P]),
    ("tdt_dispatch_resolve", _I, [_P, _I, _I, _I, _I]),
    ("tdt_covered_pixels", ctypes.c_int64, [_P, _I, _I, _I]),
    ("tdt_owned_tiles", _I, [_P, _I, _I, _I, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    ("tdt_assemble_tiles", _I, [_P, _P, _I, _I, _P, _I, _I, _I]),
    ("tdt_dispatch_counted", _I, [_P, _I, _I, _I, ctypes.POINTER(ctypes.c_uint64)]),
    ("tdt_dispatch_counted_range", _I, [_P, _I, _I, _I, _I, _I, _P, ctypes.POINTER(ctypes.c_uint64)]),
    ("tdt_forget_costs", _I, [_P]),
    ("tdt_debug_phase_timing", _I, [_P, _I, ctypes.POINTER(ctypes.c_float)]),
    ("tdt_debug_multi_timing", _I, [_P, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]),
    ("tdt_debug_multi_transport", ctypes.c_char_p, [_P]),
    ("tdt_debug_multi_rccl_ranks", _I, [_P]),
    ("tdt_debug_multi_fail", _I, [_P, _I]),
    ("tdt_octree_build_cells", _I, [_P, _P, _S, _I, _PP, ctypes.POINTER(ctypes.c_uint32)]),
    ("tdt_octree_build_from_points", _I, [_P, _P, _S, ctypes.POINTER(ctypes.c_int32), _P, _P, _S, _I, _I, _PP,
                                          ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    ("tdt_debug_edit_mode", _I, [_P, _I]),
    ("tdt_debug_last_edit_path", _I, [_P]),
    ("tdt_debug_last_variant", _I, [_P, ctypes.POINTER(ctypes.c_int)]),
    ("tdt_debug_counters", _I, [_P, ctypes.POINTER(ctypes.c_uint64)]),
    ("tdt_debug_stats", _I, [_P, ctypes.POINTER(ctypes.c_uint64), _I, _I]),
    ("tdt_debug_wave_ends", _I, [_P, ctypes.POINTER(ctypes.c_uint64), _I]),
    ("tdt_debug_pixel_log", _I, [_P, ctypes.c_void_p, ctypes.c_size_t]),
    ("tdt_selftest", _I, [_P, _I, ctypes.POINTER(ctypes.c_uint64)]),
    ("tdt_selftest_index", _I, [_P, ctypes.c_int32, _F, ctypes.c_uint32, _I, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]),
]

_lib = None


def lib():
    """Load libtdtrt.so (raises if it has not been built: there is no other implementation)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                               "the trace has no CPU fallback")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64; if ours (from /opt/rocm) were
        # loaded first, torch would later find "no HIP GPUs".  Importing torch first makes both share one.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            try:
                f = getattr(L, name)
            except AttributeError:
                if os.environ.get("TDT_LIB"):                # an older build loaded for an A/B run: newer entry points are absent
                    continue
                raise
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class TdtError(RuntimeError):
    """InitializeErr (renderer/mod.rs:28-33) analogue: carries the integer code of the C ABI."""

    def __init__(self, code, message):
        super().__init__(f"[{code:#x}] {message}")
        self.code = code


class Context:
    """The GL context of main.rs:58-61,108-112: one HIP device + stream."""

    def __init__(self, device=0, stream=None, devices=None):
        """devices = [ids]: ONE context over several GPUs (tdt_ctx_create_multi): uploads replicate, a raytracer dispatch is
        sharded over the devices and gathered + assembled on the first; otherwise a single-device context."""
        h = ctypes.c_void_p()
        if devices is not None:
            ids = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
            rc = lib().tdt_ctx_create_multi(len(devices), ids, ctypes.byref(h))
            device = devices[0] if len(devices) else 0
        else:
            rc = lib().tdt_ctx_create(int(device), ctypes.c_void_p(stream) if stream else None, ctypes.byref(h))
        if rc != OK:
            raise TdtError(rc, lib().tdt_last_error(None).decode())
        self.h = h
        self.device = device

    def device_count(self):
        return int(lib().tdt_ctx_device_count(self.h))

    def forget_costs(self):
        """Drop the per-pixel cost history: the next dispatch_compute is scheduled like a context's first frame."""
        self.check(lib().tdt_forget_costs(self.h))

    def phase_timing(self, enable=True):
        """(probe_ms, main_ms, resolve_ms) of the last dispatch_compute (blocks); switches the recording on / off."""
        ms = (ctypes.c_float * 3)()
        self.check(lib().tdt_debug_phase_timing(self.h, 1 if enable else 0, ms))
        return tuple(float(v) for v in ms)

    def multi_timing(self):
        """Multi-device context: ([trace ms per device], gather ms, assemble ms) of the last raytracer dispatch (blocks)."""
        n = self.device_count()
        tr, g, a = (ctypes.c_float * n)(), ctypes.c_float(), ctypes.c_float()
        self.check(lib().tdt_debug_multi_timing(self.h, tr, ctypes.byref(g), ctypes.byref(a)))
        return [float(v) for v in tr], float(g.value), float(a.value)

    def multi_transport(self):
        return lib().tdt_debug_multi_transport(self.h).decode()

    def multi_rccl_ranks(self):
        """Ranks of the RCCL communicator a multi-device context created (0: none / copy transport)."""
        return int(lib().tdt_debug_multi_rccl_ranks(self.h))

    def multi_fail(self, member):
        """Test hook: the next raytracer dispatch fails at device index `member` (after the devices before it were launched)."""
        self.check(lib().tdt_debug_multi_fail(self.h, int(member)))

    def edit_mode(self, mode):
        """0: edits run in parallel when that is provably the ordered result; 1: always the ordered one-lane walk."""
        self.check(lib().tdt_debug_edit_mode(self.h, mode))

    def last_edit_path(self):
        """1 = the last edit dispatch took the ordered walk, 2 = the parallel form, 0 = none yet."""
        return int(lib().tdt_debug_last_edit_path(self.h))

    def check(self, rc):
        if rc != OK:
            raise TdtError(rc, lib().tdt_last_error(self.h).decode() or lib().tdt_strerror(rc).decode())

    def finish(self):
        self.check(lib().tdt_finish(self.h))

    def bind_buffer_base(self, target, slot, vbo):
        """gl::BindBufferBase(target, slot, vbo.id()) — main.rs:352,383,408,430,448; octree.rs:67,98,115,144."""
        self.check(lib().tdt_bind_buffer_base(self.h, target, slot, vbo.h if vbo is not None else None))

    def selftest(self, which):
        """Mismatches of the short rcp (0) / sqrt (1) / rsq (2) forms vs IEEE over all 2^32 inputs."""
        n = ctypes.c_uint64(0)
        self.check(lib().tdt_selftest(self.h, which, ctypes.byref(n)))
        return n.value

    def last_variant(self):
        """The build of the trace kernel the last trace launch ran: dict(form, depth, resident, full, brick, unit); depth 0 = general kernel."""
        v = (ctypes.c_int * 6)()
        self.check(lib().tdt_debug_last_variant(self.h, v))
        return dict(zip(("form", "depth", "resident", "full", "brick", "unit"), [int(x) for x in v]))

    STAT_NAMES = ("trav_pass", "trav_lanes", "inside_lanes", "alive_lanes", "event_pass", "event_lanes", "hit_pass", "hit_lanes", "lamb_pass", "lamb_lanes",
                  "metal_pass", "metal_lanes", "diel_pass", "diel_lanes", "end_pass", "end_lanes", "pixel_end_pass", "pixel_end_lanes", "fetch_pass", "fetch_lanes",
                  "primary_pass", "primary_lanes", "newray_pass", "newray_lanes", "gate_wait_lanes", "drained_trav_pass", "drained_event_pass", "loop_pass",
                  "wave_ticks", "waves", "t_first", "t_last")

    def stats(self, reset=True):
        """Pass / lane statistics of the product trace kernels since the last reset (a -DTDT_STATS build of the library only)."""
        n = 32 + 1 + 8192
        v = (ctypes.c_uint64 * n)()
        self.check(lib().tdt_debug_stats(self.h, v, n, 1 if reset else 0))
        d = dict(zip(self.STAT_NAMES, [int(x) for x in v[:32]]))
        d["t_queue_dry"] = int(v[32])
        d["wave_ends"] = [int(x) for x in v[33:33 + min(int(v[29]), 8192)]]      # (waves: row 29)
        return d

    def selftest_index(self, cell_count, inv_cell_count, n_cells, shift=0):
        """(mismatches, shape_ok) of the per-cell x-index thresholds vs the literal formula: every f in [0,1) x every cell < n_cells."""
        n, ok = ctypes.c_uint64(0), ctypes.c_int(0)
        self.check(lib().tdt_selftest_index(self.h, int(cell_count), float(inv_cell_count), int(n_cells), int(shift), ctypes.byref(n), ctypes.byref(ok)))
        return n.value, bool(ok.value)

    def close(self):
        if self.h:
            lib().tdt_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class VertexBufferObject:
    """VertexBufferObject::new::<T>(Vec<T>, ..): glGenBuffers + glBufferData (copies) — vbo.rs:32-55."""

    def __init__(self, ctx, data):
        a = np.ascontiguousarray(data)
        h = ctypes.c_void_p()
        ctx.check(lib().tdt_buffer_create(ctx.h, a.ctypes.data if a.size else None, a.nbytes, ctypes.byref(h)))
        self.ctx, self.h, self.nbytes = ctx, h, a.nbytes

    @classmethod
    def _adopt(cls, ctx, h, nbytes):
        b = cls.__new__(cls)
        b.ctx, b.h, b.nbytes = ctx, h, nbytes
        return b

    def sub_data(self, offset, data):
        a = np.ascontiguousarray(data)
        self.ctx.check(lib().tdt_buffer_sub_data(self.h, offset, a.nbytes, a.ctypes.data))

    def read(self, dtype=np.uint32):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype)
        self.ctx.check(lib().tdt_buffer_read(self.h, 0, out.nbytes, out.ctypes.data))
        return out


class Texture:
    """Texture::new_2d(TEXTURE0, 0, RGBA32F, RGBA, w, h) + BindImageTexture(unit 0) — texture.rs:47-75."""

    def __init__(self, ctx, h, width, height):
        self.ctx, self.h, self._w, self._h = ctx, h, width, height

    @classmethod
    def new_2d(cls, ctx, width, height, bind=True):
        h = ctypes.c_void_p()
        ctx.check(lib().tdt_image_create_rgba32f(ctx.h, width, height, ctypes.byref(h)))
        t = cls(ctx, h, width, height)
        if bind:
            t.bind()
        return t

    @classmethod
    def wrap_device(cls, ctx, device_ptr, width, height, bind=True):
        h = ctypes.c_void_p()
        ctx.check(lib().tdt_image_wrap_device(ctx.h, ctypes.c_void_p(device_ptr), width, height, ctypes.byref(h)))
        t = cls(ctx, h, width, height)
        if bind:
            t.bind()
        return t

    def bind(self):
        self.ctx.check(lib().tdt_bind_image(self.ctx.h, 0, self.h))

    def width(self):
        return self._w

    def height(self):
        return self._h

    def depth(self):
        return 1

    def read(self):
        img = np.empty((self._h, self._w, 4), np.float32)
        self.ctx.check(lib().tdt_image_read(self.h, img.ctypes.data))
        return img

    def read_rgba8(self, top_down=True):
        """The frame as the reference's quad pass presents it (quad.frag:10, main.rs:582-600): (H, W, 4) uint8, converted on
        the GPU; top_down=True puts the top scan-line first (image-file order)."""
        img = np.empty((self._h, self._w, 4), np.uint8)
        self.ctx.check(lib().tdt_image_read_rgba8(self.h, 1 if top_down else 0, img.ctypes.data))
        return img


class Program:
    """Uniform-by-name setters of program.rs:35-83 (the program object itself is the kernel)."""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def set_i32(self, name, value):
        self.ctx.check(lib().tdt_set_i32(self.h, name.encode(), int(value)))

    def set_f32(self, name, value):
        self.ctx.check(lib().tdt_set_f32(self.h, name.encode(), float(value)))

    def set_vector3_f32(self, name, v):
        self.ctx.check(lib().tdt_set_vec3f(self.h, name.encode(), float(v[0]), float(v[1]), float(v[2])))

    def set_vector3_i32(self, name, v):
        self.ctx.check(lib().tdt_set_vec3i(self.h, name.encode(), int(v[0]), int(v[1]), int(v[2])))


class ComputeShader:
    """ComputeShader::new(program) / dispatch_compute(w, h, d) — compute_shader.rs:15-38."""

    def __init__(self, ctx, kind=PROGRAM_RAYTRACER):
        h = ctypes.c_void_p()
        ctx.check(lib().tdt_compute_create(ctx.h, kind, ctypes.byref(h)))
        self.ctx, self.h = ctx, h
        self.program = Program(ctx, h)
        gs = (ctypes.c_int * 3)()
        ctx.check(lib().tdt_compute_group_size(h, gs))
        self.group_size = list(gs)

    def dispatch_compute(self, width, height, depth):
        self.ctx.check(lib().tdt_dispatch_compute(self.h, width, height, depth))

    # --- extensions (no reference counterpart) ---
    def set_partition(self, rank, world):
        self.ctx.check(lib().tdt_set_partition(self.h, rank, world))

    def dispatch_accumulate(self, width, height, depth, spp_begin, spp_count, carry_ptr=None):
        self.ctx.check(lib().tdt_dispatch_accumulate(self.h, width, height, depth, spp_begin, spp_count,
                                                     ctypes.c_void_p(carry_ptr) if carry_ptr else None))

    def dispatch_resolve(self, width, height, depth, total_spp):
        self.ctx.check(lib().tdt_dispatch_resolve(self.h, width, height, depth, total_spp))

    def covered_pixels(self, width, height, depth=1):
        return int(lib().tdt_covered_pixels(self.h, width, height, depth))

    def owned_tiles(self, width, height, depth=1):
        """(owned, tiles_per_row, total) 32x32 work-groups of a dispatch under the current partition."""
        tx, tot = ctypes.c_int(), ctypes.c_int()
        n = lib().tdt_owned_tiles(self.h, width, height, depth, ctypes.byref(tx), ctypes.byref(tot))
        return int(n), tx.value, tot.value

    def assemble_tiles(self, gathered_ptr, world, tiles_per_rank, dst_texture, width, height, depth=1):
        self.ctx.check(lib().tdt_assemble_tiles(self.h, ctypes.c_void_p(gathered_ptr), world, tiles_per_rank,
                                                dst_texture.h, width, height, depth))

    COUNT_FIELDS = ("pixels", "octree_hit_calls", "iterations", "node_loads", "lambertian", "metal", "dielectric",
                    "unknown_material")

    DEBUG_FIELDS = ("trav_slots", "trav_active", "level_slots", "level_active", "event_slots", "event_active",
                    "scatter_slots", "scatter_active", "memo_miss", "leaf_records")

    def debug_counters(self):
        c = (ctypes.c_uint64 * 32)()
        self.ctx.check(lib().tdt_debug_counters(self.ctx.h, c))
        d = dict(zip(self.DEBUG_FIELDS, [int(v) for v in c[8:18]]))
        d.update(first_start=int(c[18]), last_end=int(c[19]), sum_wave_cycles=int(c[20]), waves=int(c[21]), queue_empty=int(c[22]))
        d["region_cycles"] = dict(zip(("traverse", "gate", "hit_scatter", "end", "fetch", "primary", "newray"), [int(v) for v in c[24:31]]))
        return d

    def debug_wave_ends(self, n):
        c = (ctypes.c_uint64 * n)()
        self.ctx.check(lib().tdt_debug_wave_ends(self.ctx.h, c, n))
        return np.array(c, dtype=np.uint64)

    def debug_pixel_log(self, slots):
        """(slots, 8) uint32 per-pixel log of the last dispatch_counted; needs TDT_PIXEL_LOG in the environment."""
        log = np.zeros((slots, 8), np.uint32)
        self.ctx.check(lib().tdt_debug_pixel_log(self.ctx.h, log.ctypes.data, log.size))
        return log

    def dispatch_counted(self, width, height, depth=1):
        """Instrumented dispatch: event totals that define the algorithmic bytes."""
        c = (ctypes.c_uint64 * 8)()
        self.ctx.check(lib().tdt_dispatch_counted(self.h, width, height, depth, c))
        return dict(zip(self.COUNT_FIELDS, [int(v) for v in c]))

    def dispatch_counted_range(self, width, height, depth, spp_begin, spp_count, carry_ptr=None):
        """dispatch_accumulate, instrumented: the events of ONE launch of a progressive / two-phase frame."""
        c = (ctypes.c_uint64 * 8)()
        self.ctx.check(lib().tdt_dispatch_counted_range(self.h, width, height, depth, spp_begin, spp_count,
                                                        ctypes.c_void_p(carry_ptr) if carry_ptr else None, c))
        return dict(zip(self.COUNT_FIELDS, [int(v) for v in c]))


def octree_build_cells(ctx, voxels_xyzm, depth):
    """SURVEY §8f-1 on the GPU (tdt_octree_build_cells): (n, 4) int32 voxels {x, y, z, material + 1} in grid coordinates ->
    (cells VertexBufferObject, number of cells)."""
    v = np.ascontiguousarray(voxels_xyzm, np.int32).reshape(-1, 4)
    h, n = ctypes.c_void_p(), ctypes.c_uint32(0)
    ctx.check(lib().tdt_octree_build_cells(ctx.h, v.ctypes.data if v.size else None, v.shape[0], depth, ctypes.byref(h), ctypes.byref(n)))
    return VertexBufferObject._adopt(ctx, h, int(n.value) * 64), int(n.value)


def octree_build_from_points(ctx, voxels_xyzk, min_point, palette_keys, palette_rgb, z_up=True, max_iter=256):
    """tdt_scene_from_ply on the GPU (tdt_octree_build_from_points): PlyFileContent{voxels, albedos, min_point}
    (ply_point_loader.rs:84-93) -> ({slot: VertexBufferObject} for slots 0,1,2,3,4,6,7 — not bound —, max_depth, cell_count)."""
    v = np.ascontiguousarray(voxels_xyzk, np.int32).reshape(-1, 4)
    keys = np.ascontiguousarray(palette_keys, np.uint32)
    rgb = np.ascontiguousarray(palette_rgb, np.uint8).reshape(-1, 3)
    mp = (ctypes.c_int32 * 3)(*[int(x) for x in min_point])
    out = (ctypes.c_void_p * 8)()
    depth, cc = ctypes.c_int32(0), ctypes.c_int32(0)
    ctx.check(lib().tdt_octree_build_from_points(ctx.h, v.ctypes.data, v.shape[0], mp, keys.ctypes.data, rgb.ctypes.data, keys.size,
                                                 1 if z_up else 0, max_iter, out, ctypes.byref(depth), ctypes.byref(cc)))
    vbos = {}
    for slot in (0, 1, 2, 3, 4, 6, 7):
        vbos[slot] = VertexBufferObject._adopt(ctx, ctypes.c_void_p(out[slot]), 0)
    return vbos, int(depth.value), int(cc.value)


def initial_uniforms(camera, program):
    """camera.rs:241-253: sends all eight camera uniforms."""
    program.set_i32("camera.image_width", camera.image_width)
    program.set_i32("camera.image_height", camera.image_height)
    program.set_vector3_f32("camera.horizontal", camera.horizontal)
    program.set_vector3_f32("camera.vertical", camera.vertical)
    program.set_vector3_f32("camera.lower_left_corner", camera.lower_left_corner)
    program.set_vector3_f32("camera.origin", camera.origin)
    program.set_i32("camera.samples_per_pixel", camera.samples_per_pixel)
    program.set_i32("camera.max_bounce", camera.max_bounce)


def update_vbo(ctx, delta_vbo, delta, length, update_compute):
    """Octree::update_vbo (octree.rs:170-183): BufferSubData of `length` floats, then the oddly shaped
    dispatch — (0, n, 0) unless n / 1024 is integral, n = (length as f32 * 0.2) as i32."""
    d = np.ascontiguousarray(delta, np.float32)[:length]
    delta_vbo.sub_data(0, d)
    x_schedule = np.float32(length) * np.float32(0.2)
    dispatch_count = int(x_schedule)
    q = x_schedule / np.float32(32.0 * 32.0)
    if q - np.trunc(q) != 0:
        update_compute.dispatch_compute(0, dispatch_count, 0)
    else:
        update_compute.dispatch_compute(dispatch_count, 1, 1)


def upload_scene(ctx, scene):
    """What main.rs:343-450 and Octree::init_global_buffers (octree.rs:44-100) do: one buffer per
    payload, bound to its shader-storage slot.  Returns the buffers (keep them alive)."""
    vbos = {}
    for slot in (0, 1, 2, 3, 4, 6, 7):
        vbos[slot] = VertexBufferObject(ctx, scene.blobs[slot])
        ctx.bind_buffer_base(SHADER_STORAGE_BUFFER, slot, vbos[slot])
    return vbos


class Renderer:
    """Convenience wrapper used by tests, smoke() and bench.py: a context with one scene, one
    camera and one image, i.e. the state main.rs has built when it reaches its render loop."""

    def __init__(self, scene, camera, device=0, stream=None, rank=0, world=1, image_ptr=None, tile_buffer_tiles=None, devices=None):
        self.ctx = Context(device, stream, devices=devices)
        self.shader = ComputeShader(self.ctx)
        self.vbos = upload_scene(self.ctx, scene)
        self.camera = camera
        initial_uniforms(camera, self.shader.program)
        if devices is None:
            self.shader.set_partition(rank, world)
        if tile_buffer_tiles is not None:      # this rank's tile buffer [k][32][32] RGBA
            w, rows = 32, 32 * tile_buffer_tiles
        else:
            w, rows = camera.image_width, camera.image_height
        if image_ptr is not None:
            self.texture = Texture.wrap_device(self.ctx, image_ptr, w, rows)
        else:
            self.texture = Texture.new_2d(self.ctx, w, rows)

    def dispatch(self, width=None, height=None):
        """main.rs:579: dispatch_compute(texture.width() + 1, texture.height() + 1, 1)."""
        w = self.camera.image_width + 1 if width is None else width
        h = self.camera.image_height + 1 if height is None else height
        self.shader.dispatch_compute(w, h, 1)

    def render(self, width=None, height=None):
        self.dispatch(width, height)
        return self.texture.read()

    def close(self):
        self.ctx.close()
