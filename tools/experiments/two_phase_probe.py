#!/usr/bin/env python3
"""Would a history-free two-phase frame pay?  Per frame of a camera walk: k probe samples per pixel in the order of the
previous dispatch (useless under motion), costs recorded -> remaining samples in the cost order of THIS frame's probe
(existing accumulate / carry / resolve path; generic MODE-1 kernels, so compare only among the rows printed here)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tdt4230_project_raytracing_amd import host, rt
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H, spp, bounce = 1920, 1080, 64, {2: 8, 3: 16, 5: 8}.get(cfg, 8)
scene = host.Scene.config(cfg)


def walk(frames, step, k):
    cam = host.Camera(90.0, W, aspect_ratio=W / H, origin=(0.0, -0.1, -0.3), viewport_height=2.0, samples_per_pixel=spp,
                      max_bounce=bounce, turn_rate=0.05, normal_speed=0.03, sprint_speed=0.15)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    carry = torch.zeros((H, W, 16), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    r = rt.Renderer(scene, cam.uniforms(), image_ptr=acc.data_ptr())
    out = []
    for f in range(frames):
        if f and step:
            cam.translate("Front", step / 60.0); cam.turn_yaw(0.2 * step)
            rt.initial_uniforms(cam.uniforms(), r.shader.program)
        acc.zero_(); carry.zero_(); torch.cuda.synchronize()
        r.ctx.finish(); t = time.perf_counter()
        if k:
            r.shader.dispatch_accumulate(W + 1, H + 1, 1, 0, k, carry.data_ptr())
            r.shader.dispatch_accumulate(W + 1, H + 1, 1, k, spp - k, carry.data_ptr())
        else:
            r.shader.dispatch_accumulate(W + 1, H + 1, 1, 0, spp, carry.data_ptr())
        r.shader.dispatch_resolve(W + 1, H + 1, 1, spp)
        r.ctx.finish()
        out.append((time.perf_counter() - t) * 1e3)
    r.close()
    return out


for step in (0, 1):
    for k in (0, 1, 2, 4, 8):
        print(f"camera step x{step}, probe samples {k}:", " ".join(f"{t:.1f}" for t in walk(6, step, k)), "ms")
os.environ["TDT_NO_COST_ORDER"] = "1"
print("image order, one phase:", " ".join(f"{t:.1f}" for t in walk(4, 1, 0)), "ms")
