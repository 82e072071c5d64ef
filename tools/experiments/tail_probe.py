#!/usr/bin/env python3
"""How much of a frame is end-of-kernel tail?  Same frustum, k x the pixel rows: time per pixel row should stay flat."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tdt4230_project_raytracing_amd import host, rt
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scene = host.Scene.config(cfg)
bounce = {2: 8, 3: 16, 5: 8}.get(cfg, 8)
for k in (1, 2, 4):
    cam = host.camera_reference_pose(1920, 1080, 64, bounce); cam.image_height = 1080 * k
    r = rt.Renderer(scene, cam)
    r.dispatch(); r.dispatch(); r.ctx.finish()     # image order, then one frame in cost order
    t = time.perf_counter()
    for _ in range(3): r.dispatch()
    r.ctx.finish(); dt = (time.perf_counter() - t) / 3
    print(f"rows x{k}: {dt*1e3:.2f} ms  ({dt*1e3/k:.2f} ms per 1080 rows)")
    r.close()
