#!/usr/bin/env python3
"""ONE well-defined last frame for a rocprofv3 --pmc pass: a few warm frames, then a final frame that is either one the scheduler
has not seen (--mode fresh: probe launch + main launch + resolve; the main launch is the last dispatch of the plain trace kernel)
or a replay of the frame before it (--mode replay).  usage: pmc_frame.py [--config 2] [--mode fresh|replay]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tdt4230_project_raytracing_amd import host, rt
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--mode", default="fresh", choices=("fresh", "replay"))
a = ap.parse_args()
W, H, spp, bounce, desc, scene_cfg = bench.WORKLOADS[a.config]
scene = host.Scene.config(scene_cfg); cam = host.camera_reference_pose(W, H, spp, bounce)
r = rt.Renderer(scene, cam)
for _ in range(2):
    r.dispatch()
if a.mode == "fresh":
    r.ctx.forget_costs()
r.dispatch()
r.ctx.finish()
r.close()
