#!/usr/bin/env python3
"""How well does last frame's cost order serve a MOVING camera?  Renders a walk (translate + yaw per frame, camera.ron
rates, 60 fps time step) and prints per-frame kernel time next to the static steady state and to image order."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tdt4230_project_raytracing_amd import host, rt
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H, spp, bounce = 1920, 1080, 64, {2: 8, 3: 16, 5: 8}.get(cfg, 8)
scene = host.Scene.config(cfg)


def walk(frames, step):
    cam = host.Camera(90.0, W, aspect_ratio=W / H, origin=(0.0, -0.1, -0.3), viewport_height=2.0, samples_per_pixel=spp,
                      max_bounce=bounce, turn_rate=0.05, normal_speed=0.03, sprint_speed=0.15)
    r = rt.Renderer(scene, cam.uniforms())
    out = []
    for f in range(frames):
        if f and step:
            cam.translate("Front", step / 60.0); cam.turn_yaw(0.2 * step)
            rt.initial_uniforms(cam.uniforms(), r.shader.program)
        r.ctx.finish(); t = time.perf_counter()
        r.dispatch(); r.ctx.finish()
        out.append((time.perf_counter() - t) * 1e3)
    r.close()
    return out


static = walk(5, 0)
print("static camera   :", " ".join(f"{t:.1f}" for t in static), "ms  (frame 0 = image order)")
for step in (1, 4, 16):
    print(f"moving, step x{step:<2d}:", " ".join(f"{t:.1f}" for t in walk(8, step)), "ms")
os.environ["TDT_NO_COST_ORDER"] = "1"
print("image order     :", " ".join(f"{t:.1f}" for t in walk(3, 1)), "ms")
