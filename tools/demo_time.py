#!/usr/bin/env python3
"""The reference's own workload on the GPU: demo scene (main.rs:235-463, cell_count 100000), 1280x720 (main.rs:26), 4 spp /
bounce 6 (assets/settings/camera.ron:2-3), dispatched as main.rs:579 does.  Prints history-free and replay frame times and
which kernel variants the environment switches select.  usage: demo_time.py [steps] [W H spp bounce]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdt4230_project_raytracing_amd import host, rt
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
W, H, spp, b = (1280, 720, 4, 6) if len(sys.argv) < 6 else map(int, sys.argv[2:6])
scene = host.Scene.demo()
cam = host.camera_reference_pose(W, H, spp, b)
r = rt.Renderer(scene, cam)
px = r.shader.covered_pixels(W + 1, H + 1)
for fresh in (True, False):
    for _ in range(5):
        if fresh: r.ctx.forget_costs()
        r.dispatch()
    r.ctx.finish(); t = time.perf_counter()
    for _ in range(steps):
        if fresh: r.ctx.forget_costs()
        r.dispatch()
    r.ctx.finish(); dt = (time.perf_counter() - t) / steps
    print(f"demo {W}x{H} spp {spp} bounce {b} {'history-free' if fresh else 'replay'}: {dt*1e3:.3f} ms  {px*spp/dt/1e6:.1f} Msamples/s", flush=True)
r.close()
