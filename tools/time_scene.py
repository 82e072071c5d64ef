#!/usr/bin/env python3
"""Steady-state frame time of a generated scene.  usage: time_scene.py kind depth log2_cells max_iter [W H spp bounce]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdt4230_project_raytracing_amd import host, rt
kind, depth, lc, mi = map(int, sys.argv[1:5])
W, H, spp, b = (1920, 1080, 16, 8) if len(sys.argv) < 9 else map(int, sys.argv[5:9])
scene = host.Scene.generate(kind, depth, 1 << lc, mi, 0x5EED0007)
cam = host.camera_reference_pose(W, H, spp, b)
r = rt.Renderer(scene, cam)
for _ in range(3): r.dispatch()
r.ctx.finish(); t = time.perf_counter()
for _ in range(3): r.dispatch()
r.ctx.finish(); dt = (time.perf_counter() - t) / 3
print(f"kind {kind} depth {depth} cells {scene.counts['cells']} ({scene.nbytes()} B): {dt*1e3:.2f} ms")
r.close()
