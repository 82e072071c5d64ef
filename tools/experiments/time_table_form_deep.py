#!/usr/bin/env python3
"""history-free / replay frame time of deep trees outside the LDS table that have NO bricks: the config-3 and config-5 scenes under the reference's own
cell_count 100000 (per-cell-threshold form), 1080p, 16 spp.  usage: [TDT_LIB=...] python tools/experiments/time_table_form_deep.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tdt4230_project_raytracing_amd import host, rt
for cfg in (3, 5):
    scene = host.scene_with_cell_count(host.Scene.config(cfg), 100000 if cfg == 3 else 1000000, 0)
    cam = host.camera_reference_pose(1920, 1080, 16, 8)
    r = rt.Renderer(scene, cam)
    r.dispatch(); r.ctx.finish()
    v = r.ctx.last_variant()
    res = []
    for mode in ("history-free", "replay"):
        ts = []
        for _ in range(4):
            if mode == "history-free":
                r.ctx.forget_costs()
            r.ctx.finish(); t = time.perf_counter(); r.dispatch(); r.ctx.finish(); ts.append((time.perf_counter() - t) * 1e3)
        res.append(min(ts))
    print("config %d scene, cell_count %d: variant %s  history-free %.2f ms  replay %.2f ms" % (cfg, int(scene.blobs[7][2]), v, res[0], res[1]), flush=True)
    r.close()
