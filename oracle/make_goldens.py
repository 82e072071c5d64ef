#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE shader itself (unmodified, read from
$REF_DIR at run time) on Mesa llvmpipe through oracle/_ref/libglref.so.

Run in the build container only (needs /root/reference and the swrast DRI driver):
    make -C oracle && python oracle/make_goldens.py
Fixtures are data: scene parameters, camera uniforms and the RGBA32F image the reference wrote.
Scenes are regenerated from their parameters by libtdthost.so (byte-identical everywhere; a
sha256 of every payload is stored and re-checked by the tests).
"""
import hashlib
import json
import os
import platform
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import oracle_py  # noqa: E402
from tdt4230_project_raytracing_amd import host  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# name -> (scene spec, W, H, spp, max_bounce, crop)   crop = (x0, y0, w, h) stored region or None (whole image)
CASES = {
    "demo_256x256_spp1_b6": (("config", 0), 256, 256, 1, 6, None),
    "demo_160x96_spp4_b6": (("config", 0), 160, 96, 4, 6, None),
    "demo_320x180_spp2_b3": (("config", 0), 320, 180, 2, 3, None),          # floor-div dispatch leaves rows 160..179 unwritten
    "config1_256x256_spp1_b1": (("config", 1), 256, 256, 1, 1, None),      # BASELINE configs[0]
    "config2_240x135_spp4_b8": (("config", 2), 240, 135, 4, 8, None),      # configs[1] scene, 1/8 resolution
    "config2_1920x1080_spp2_b8_crop": (("config", 2), 1920, 1080, 2, 8, (896, 480, 128, 96)),   # full-size uniforms, cropped
    "terrain7_192x108_spp2_b8": (("generate", 1, 7, 1 << 18, 200, 0x77), 192, 108, 2, 8, None),   # > LDS-table cells
    "shells7_192x108_spp2_b8": (("generate", 2, 7, 1 << 18, 256, 0x99), 192, 108, 2, 8, None),
    "config3_3840x2160_spp1_b16_crop": (("config", 3), 3840, 2160, 1, 16, (1856, 1000, 96, 64)),
    "config5_1920x1080_spp1_b8_crop": (("config", 5), 1920, 1080, 1, 8, (900, 500, 96, 64)),
    # cameras OUTSIDE / on the boundary of the octree: root-test misses, stale out-parameter records, NaN-free edge paths
    "demo_outside_side_160x96_spp3_b6": (("config", 0), 160, 96, 3, 6, None, (-1.2, 0.0, -0.5)),
    "config2_on_boundary_160x96_spp3_b6": (("config", 2), 160, 96, 3, 6, None, (0.0, -0.1, 0.0)),
    "config1_outside_corner_160x96_spp3_b6": (("config", 1), 160, 96, 3, 6, None, (0.5, 0.5, 0.0)),
    "config2_outside_front_160x96_spp3_b6": (("config", 2), 160, 96, 3, 6, None, (0.1, 0.05, 0.6)),
    # the reference's own MagicaVoxel model (assets/models/monu1_point.ply, 156 942 voxels) through the PLY loader
    # restatement and the octree builder; the built payloads (14 KB compressed) are stored in the fixture
    "monu1_ply_320x240_spp2_b6": (("ply", "monu1_point.ply"), 320, 240, 2, 6, None),
    # the reference host's own conventions on bigger trees: cell_count = PRE_ALLOCATED_CELLS = 100000 (main.rs:459: not a power of two —
    # for many cells the x index of treeLookup reads the upper half of the PREVIOUS cell at the bottom of the cell) and a pre-allocated
    # buffer with a tail of zero nodes (main.rs:339-341); ("config_cc", config, cell_count, zero nodes appended)
    "config2_cc100000_160x96_spp3_b6": (("config_cc", 2, 100000, 30000), 160, 96, 3, 6, None),
    "config2_cc100000_on_axes_160x96_spp3_b6": (("config_cc", 2, 100000, 30000), 160, 96, 3, 6, None, (0.0, 0.0, -0.5)),   # rays along cell boundaries: coordinates with f = 0
    "config3_cc100000_200x120_spp2_b6": (("config_cc", 3, 100000, 0), 200, 120, 2, 6, None),               # outside the LDS table
    "config2_cc12345_outside_160x96_spp3_b6": (("config_cc", 2, 12345, 0), 160, 96, 3, 6, None, (0.1, 0.05, 0.6)),
    # round 4 — TURNED cameras: the uniforms the reference's controller sends after turn_yaw / turn_pitch (camera.rs:68-82 ->
    # raytracer.comp:304-307), all nine components of horizontal / vertical / lower_left_corner non-zero.  The pose is a dict
    # (origin, yaw, pitch in degrees, vertical fov); the uniforms come from the host mirror of the controller (host.Camera) and are
    # STORED in the fixture, which the tests feed to the oracle and the kernel as they are ("camera_explicit": the controller's own
    # arithmetic is not what these fixtures pin — the shader's use of whatever uniforms it is sent is)
    "demo_turned_inside_160x96_spp4_b6": (("config", 0), 160, 96, 4, 6, None, {"origin": (0.2, -0.1, 0.25), "yaw": 23.0, "pitch": 20.0, "fov": 60.0}),
    "config2_turned_outside_in_160x96_spp16_b6": (("config", 2), 160, 96, 16, 6, None, {"origin": (0.58, 0.25, 0.66), "yaw": 43.0, "pitch": -25.0, "fov": 60.0}),   # the miss pre-pass decides most pixels; 16 spp: a two-phase frame
    "config3_cc100000_turned_grazing_200x120_spp3_b8": (("config_cc", 3, 100000, 0), 200, 120, 3, 8, None, {"origin": (-0.5004, 0.21, 0.37), "yaw": -17.0, "pitch": -25.0, "fov": 60.0}),   # along a face of the root cube, tree outside the LDS table
    "config5_turned_inside_192x108_spp2_b8": (("config", 5), 192, 108, 2, 8, None, {"origin": (0.05, -0.1, 0.3), "yaw": 23.0, "pitch": 8.0, "fov": 60.0}),
    # ... and one the controller cannot reach (its right vector stays horizontal: horizontal.y is always 0, camera.rs:70) but the shader
    # takes like any other uniforms: the first pose rolled about its view axis — all NINE components non-zero
    "demo_rolled_inside_160x96_spp4_b6": (("config", 0), 160, 96, 4, 6, None, {"origin": (0.2, -0.1, 0.25), "yaw": 23.0, "pitch": 20.0, "fov": 60.0, "roll": 31.0}),
}


def make_camera(W, H, spp, bounce, origin=None):
    if origin is None:
        return host.camera_reference_pose(W, H, spp, bounce)
    if isinstance(origin, dict):                          # a turned camera: the controller's uniforms (host.Camera)
        c = host.Camera(origin["fov"], W, aspect_ratio=float(np.float32(W) / np.float32(H)), viewport_height=2.0, origin=origin["origin"],
                        samples_per_pixel=spp, max_bounce=bounce)
        # the controller turns by 2 * angle * turn_rate radians (camera.rs:46-62; turn_rate 0.025, camera.rs:167)
        c.turn_yaw(float(np.radians(origin["yaw"])) / 0.05)
        c.turn_pitch(float(np.radians(origin["pitch"])) / 0.05)
        u = c.uniforms()
        if origin.get("roll"):                            # (fp32 throughout; the fixture stores what comes out)
            f32 = np.float32
            h, v, o3 = np.array(u.horizontal, f32), np.array(u.vertical, f32), np.array(u.origin, f32)
            fwd = o3 - h * f32(0.5) - v * f32(0.5) - np.array(u.lower_left_corner, f32)
            cr, sr = f32(np.cos(np.radians(origin["roll"]))), f32(np.sin(np.radians(origin["roll"])))
            wh, wv = np.linalg.norm(h).astype(f32), np.linalg.norm(v).astype(f32)
            hn, vn = h / wh, v / wv
            h2, v2 = (hn * cr + vn * sr) * wh, (vn * cr - hn * sr) * wv
            u.horizontal[:] = [float(x) for x in h2]
            u.vertical[:] = [float(x) for x in v2]
            u.lower_left_corner[:] = [float(x) for x in (o3 - h2 * f32(0.5) - v2 * f32(0.5) - fwd)]
        return u
    aspect = float(np.float32(W) / np.float32(H))
    return host.camera_build(90.0, W, aspect_ratio=aspect, viewport_height=2.0, origin=origin,
                             samples_per_pixel=spp, max_bounce=bounce)


def make_scene(spec):
    if spec[0] == "config":
        return host.Scene.config(spec[1])
    if spec[0] == "ply":
        path = os.path.join(oracle_py.REF_DIR, "assets", "models", spec[1])
        return host.Ply(open(path, "rb").read(), strict_crlf=False).to_scene(max_iter=256)
    if spec[0] == "config_cc":
        return host.scene_with_cell_count(host.Scene.config(spec[1]), spec[2], spec[3])
    return host.Scene.generate(*spec[1:])


def scene_digest(scene):
    return {str(k): hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in sorted(scene.blobs.items())}


def main():
    os.makedirs(OUT, exist_ok=True)
    gl = oracle_py.GLRef.get()
    flags = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                fl = set(line.split(":")[1].split())
                flags = " ".join(sorted(fl & {"fma", "avx2", "avx512f", "sse4_1"}))
                break
    except OSError:
        pass
    meta_common = {"renderer": gl.renderer(), "cpu_flags": flags, "machine": platform.machine(),
                   "shader": "assets/shaders/raytracer.comp (unmodified, loaded from $REF_DIR at run time)"}
    only = [a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--only=")]
    for name, case in CASES.items():
        if only and name not in only[0]:
            continue
        spec, W, H, spp, bounce, crop = case[:6]
        origin = case[6] if len(case) > 6 else None
        scene = make_scene(spec)
        cam = make_camera(W, H, spp, bounce, origin)
        img = gl.render(scene, cam)                       # dispatch_compute(W+1, H+1, 1), main.rs:579
        if crop:
            x0, y0, w, h = crop
            data = img[y0:y0 + h, x0:x0 + w].copy()
        else:
            data = img
        meta = dict(meta_common)
        meta.update({"scene": list(spec), "W": W, "H": H, "spp": spp, "max_bounce": bounce, "crop": crop, "origin": origin,
                     "camera_explicit": isinstance(origin, dict),
                     "scene_sha256": scene_digest(scene),
                     "camera": {k: (list(getattr(cam, k)) if hasattr(getattr(cam, k), "__len__") else getattr(cam, k))
                                for k, _ in host.CameraUniforms._fields_}})
        extra = {f"blob_{k}": v for k, v in scene.blobs.items()} if spec[0] == "ply" else {}
        np.savez_compressed(os.path.join(OUT, name + ".npz"), image=data, meta=json.dumps(meta), **extra)
        print(f"{name}: {data.shape} written px {(img[..., 3] == 1).sum()} nan {np.isnan(img).sum()}")


def math_table():
    """sin / cos / pow(x,5) / ... of llvmpipe on a fixed input set."""
    rng = np.random.default_rng(0xA2B)
    xs = np.concatenate([
        np.linspace(0, 2 * np.pi, 1024, dtype=np.float32), np.linspace(-200, 200, 2048, dtype=np.float32),
        rng.uniform(-2, 2, 1024).astype(np.float32), rng.uniform(-9000, 9000, 2048).astype(np.float32),
        rng.uniform(0, 1, 2048).astype(np.float32)]).astype(np.float32)
    tab = oracle_py.glref_math_table(xs)
    np.savez_compressed(os.path.join(OUT, "math_table.npz"), x=xs, table=tab,
                        columns=json.dumps(["sin", "cos", "pow5", "inversesqrt", "rcp", "sqrt", "fract", "fract_sin_43758"]))
    print("math_table:", tab.shape)


# voxel edits: name -> (scene spec, extra zeroed cells, counter before, deltas [pos xyz, type, value], dispatch)
EDITS = {
    "edit_demo_click": (("config", 0), 0, 19, [[0.3, 0.6, 0.2, 2.0, 5.0]], (0, 1, 0)),            # main.rs:561-568 -> update_vbo(delta, 5)
    "edit_demo_three": (("config", 0), 0, 19, [[0.51, 0.45, 0.31, 2.0, 3.0], [0.52, 0.45, 0.31, 2.0, 4.0], [0.9, 0.1, 0.8, 0.0, 0.0]], (0, 3, 0)),
    "edit_config2_two": (("config", 2), 64, 4209, [[0.5, 0.42, 0.62, 2.0, 13.0], [0.47, 0.40, 0.60, 2.0, 18.0]], (0, 2, 0)),
    "edit_demo_out_of_room": (("config", 0), -(100144 - 21 * 16), 19, [[0.3, 0.6, 0.2, 2.0, 5.0]], (0, 1, 0)),   # cells buffer cut to 21 cells
}


def edits():
    """octree_update.comp run by the reference on llvmpipe: cells / counter after the edit, and a render of the edited tree."""
    gl = oracle_py.GLRef.get()
    for name, (spec, pad, counter, deltas, dispatch) in EDITS.items():
        scene = make_scene(spec)
        if pad > 0:
            scene.blobs[0] = np.concatenate([scene.blobs[0], np.zeros(16 * pad, np.uint32)])
        elif pad < 0:
            scene.blobs[0] = scene.blobs[0][:pad].copy()
        delta = np.zeros((len(deltas), 8), np.float32)          # DeltaNode stride 32 bytes
        delta[:, :5] = np.array(deltas, np.float32)
        before = scene.blobs[0].copy()
        cells, cnt = oracle_py.glref_octree_update(scene, delta, counter, dispatch)
        changed = np.nonzero(cells != before)[0].astype(np.uint32)
        scene.blobs[0] = cells
        cam = host.camera_reference_pose(128, 96, 2, 6)
        img = gl.render(scene, cam)
        meta = {"scene": list(spec), "pad": pad, "counter_before": counter, "dispatch": list(dispatch),
                "counter_after": cnt, "scene_sha256_before": hashlib.sha256(before.tobytes()).hexdigest()}
        np.savez_compressed(os.path.join(OUT, name + ".npz"), delta=delta, changed_index=changed, changed_value=cells[changed],
                            image=img, meta=json.dumps(meta))
        print(f"{name}: {changed.size} dwords changed, counter {counter} -> {cnt}")


def present_input():
    """The float frame of the presentation fixtures: every value class the UNORM8 conversion distinguishes."""
    rng = np.random.default_rng(0x8F4)
    W, H = 96, 40
    img = rng.random((H, W, 4)).astype(np.float32)
    img[1] = rng.uniform(-0.5, 1.5, (W, 4)).astype(np.float32)                       # out-of-range values
    k = np.arange(W * 4, dtype=np.float32).reshape(W, 4)
    img[2] = ((k % 255) + 0.5) / 255                                                 # ties (x*255 = k + 1/2, up to rounding)
    img[3] = np.nextafter(img[2], np.float32(2), dtype=np.float32)                   # just above / below the ties
    img[4] = np.nextafter(img[2], np.float32(-1), dtype=np.float32)
    img[5] = (k % 256) / 255                                                         # exact codes
    special = np.array([0.0, -0.0, 1.0, np.nan, np.inf, -np.inf, 1e-30, -1e-30, 0.999999, 1.000001, 0.5, 0.25], np.float32)
    img[6, :len(special), 0] = special
    img[6, :len(special), 3] = special[::-1]
    return img


def present():
    """quad.vert + quad.frag run by llvmpipe into an RGBA8 colour buffer: bytes for crafted values and for a
    rendered frame of the demo scene (1280x720 window scaled down to 160x96: 3 work-group rows, all written)."""
    img = present_input()
    out = oracle_py.glref_present(img)
    np.savez_compressed(os.path.join(OUT, "present_values.npz"), image=img.view(np.uint32), rgba8=out,
                        meta=json.dumps({"rows": "image and rgba8: row 0 = bottom scan-line (glReadPixels order)"}))
    gl = oracle_py.GLRef.get()
    scene = make_scene(("config", 0))
    cam = host.camera_reference_pose(160, 100, 4, 6)
    frame = gl.render(scene, cam)
    out2 = oracle_py.glref_present(frame)
    np.savez_compressed(os.path.join(OUT, "present_demo.npz"), image=frame.view(np.uint32), rgba8=out2,
                        meta=json.dumps({"scene": ["config", 0], "camera": [160, 100, 4, 6],
                                         "note": "rows 96..99 are never written by the floor-div dispatch: texture zeros -> 0,0,0,0"}))
    print("present:", out.shape, out2.shape, "distinct bytes", len(np.unique(out2)))


if __name__ == "__main__":
    main()
    if not any(a.startswith("--only=") for a in sys.argv[1:]):
        math_table()
        edits()
        present()
