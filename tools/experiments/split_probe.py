#!/usr/bin/env python3
"""The gain to expect from making every sample (or pair of samples) of a low-spp frame its own queue item: the same frustum with k x the
pixel rows at spp / k — the same samples, each an independent item ("measured and rejected in round 3")."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tdt4230_project_raytracing_amd import host, rt
def run(scene, W, H, spp, b, factor, label):
    cam = host.camera_reference_pose(W, H, spp // factor, b)
    cam.image_height = H * factor
    r = rt.Renderer(scene, cam)
    for fresh in (True, False):
        for _ in range(5):
            if fresh: r.ctx.forget_costs()
            r.dispatch()
        r.ctx.finish(); t = time.perf_counter()
        for _ in range(50):
            if fresh: r.ctx.forget_costs()
            r.dispatch()
        r.ctx.finish(); dt = (time.perf_counter() - t) / 50
        print(f"{label}: {W}x{H*factor} spp {spp//factor} {'history-free' if fresh else 'replay'}: {dt*1e3:.3f} ms", flush=True)
    r.close()
scene = host.Scene.demo()
run(scene, 1280, 720, 4, 6, 1, "demo as is")
run(scene, 1280, 720, 4, 6, 4, "demo, every sample its own pixel (4x rows, same frustum)")
run(scene, 1280, 720, 4, 6, 2, "demo, 2 samples per item")
