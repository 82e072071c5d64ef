#!/usr/bin/env python3
"""End-of-frame helping: frame times with and without it (history-free and replay), and how much of it happened.
usage: help_probe.py [config] [W H spp bounce]   (TDT_NO_HELP=1 in the environment: the run without)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdt4230_project_raytracing_amd import host, rt
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H, spp, b = (1920, 1080, 64, 8) if len(sys.argv) < 6 else map(int, sys.argv[2:6])
scene = host.Scene.config(cfg); cam = host.camera_reference_pose(W, H, spp, b)
r = rt.Renderer(scene, cam)
r.dispatch(); r.ctx.finish()
for label, fresh in (("history-free", True), ("replay", False)):
    ts = []
    for _ in range(6):
        if fresh:
            r.ctx.forget_costs()
        r.ctx.finish(); t = time.perf_counter(); r.dispatch(); r.ctx.finish(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"config {cfg} {label}: " + " ".join(f"{t:.2f}" for t in ts) + f" ms; help {r.ctx.help_stats()}")
r.close()
