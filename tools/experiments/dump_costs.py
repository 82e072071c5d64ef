#!/usr/bin/env python3
"""Per-pixel step / event counts of the probe's sample range and of the whole frame (instrumented kernel, TDT_PIXEL_LOG=1), saved for the
hand-out order study on the CPU (tools/sim/order_sim.py).  usage: dump_costs.py <config> <out.npz> [W H spp bounce]"""
import sys, os
os.environ["TDT_PIXEL_LOG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tdt4230_project_raytracing_amd import host, rt
cfg, out = int(sys.argv[1]), sys.argv[2]
W, H, spp, b = (1920, 1080, 64, 8) if len(sys.argv) < 7 else map(int, sys.argv[3:7])
scene = host.Scene.config(cfg); cam = host.camera_reference_pose(W, H, spp, b)
r = rt.Renderer(scene, cam)
tiles = (W // 32) * (H // 32)
res = {}
for name, (s0, n) in (("probe", (0, max(1, spp // 16))), ("frame", (0, spp))):
    r.shader.dispatch_counted_range(W + 1, H + 1, 1, s0, n)
    log = r.shader.debug_pixel_log(tiles * 1024)
    res[name + "_S"], res[name + "_E"] = log[:, 0].copy(), log[:, 1].copy()
    res[name + "_t0"], res[name + "_t1"] = log[:, 3].copy(), log[:, 4].copy()
    print(name, "pixels", int((log[:, 1] > 0).sum()), "steps", int(log[:, 0].sum()), "events", int(log[:, 1].sum()), flush=True)
np.savez_compressed(out, W=W, H=H, spp=spp, **res)
r.close()
