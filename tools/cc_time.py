#!/usr/bin/env python3
"""Frame time of a BASELINE scene under another cell_count uniform (the reference's own host passes 100000, not a power of two).
usage: cc_time.py config cell_count [W H spp bounce]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tdt4230_project_raytracing_amd import host, rt
import numpy as np
cfg, cc = int(sys.argv[1]), int(sys.argv[2])
W, H, spp, b = (1920, 1080, 16, 8) if len(sys.argv) < 7 else map(int, sys.argv[3:7])
scene = host.Scene.config(cfg)
if cc:
    blobs = {k: v.copy() for k, v in scene.blobs.items()}
    blobs[6][6] = np.float32(1.0) / np.float32(cc); blobs[7][2] = cc
    scene = host.Scene(blobs, scene.counts, f"config{cfg}_cc{cc}")
cam = host.camera_reference_pose(W, H, spp, b)
r = rt.Renderer(scene, cam)
for fresh in (True, False):
    for _ in range(3):
        if fresh: r.ctx.forget_costs()
        r.dispatch()
    r.ctx.finish(); t = time.perf_counter()
    for _ in range(5):
        if fresh: r.ctx.forget_costs()
        r.dispatch()
    r.ctx.finish(); dt = (time.perf_counter() - t) / 5
    print(f"config {cfg} cell_count {scene.cell_count} cells {scene.counts['cells']} {W}x{H} spp {spp} {'history-free' if fresh else 'replay'}: {dt*1e3:.2f} ms", flush=True)
r.close()
