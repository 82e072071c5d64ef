#!/usr/bin/env python3
"""When did each pixel of a PRODUCT launch start and end, on which wave, and what did it cost?  (-DTDT_STATS build with TDT_PIXEL_LOG=1 and
TDT_STATS_SKIP_PROBE=1: the main launch of a history-free frame, then the replay of the same frame.)
usage: TDT_LIB=build_ab/lib_stats.so python tools/experiments/pixel_times.py <config> <out.npz>"""
import os, sys
os.environ["TDT_PIXEL_LOG"] = "1"; os.environ["TDT_STATS_SKIP_PROBE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from tdt4230_project_raytracing_amd import host, rt
cfg, out = int(sys.argv[1]), sys.argv[2]
W, H, spp, bounce, desc, scene_cfg = bench.WORKLOADS[cfg]
scene = host.Scene.config(scene_cfg); cam = host.camera_reference_pose(W, H, spp, bounce)
r = rt.Renderer(scene, cam)
r.ctx.stats()
tiles = (W // 32) * (H // 32)
res = {}
for _ in range(2):
    r.dispatch()
for mode in ("fresh", "replay"):
    if mode == "fresh":
        r.ctx.forget_costs()
    r.ctx.finish(); r.ctx.stats(reset=True)
    r.dispatch(); r.ctx.finish()
    st = r.ctx.stats(reset=True)
    log = r.shader.debug_pixel_log(tiles * 1024)
    t0, t1 = log[:, 3].astype(np.int64), log[:, 4].astype(np.int64)
    ok = log[:, 0] != 0
    base = t0[ok].min()
    res[mode + "_cost"], res[mode + "_t0"], res[mode + "_t1"], res[mode + "_wave"] = log[:, 0].copy(), ((t0 - base) & 0xFFFFFFFF).astype(np.uint32), ((t1 - base) & 0xFFFFFFFF).astype(np.uint32), log[:, 5].astype(np.uint16)
    span = (t1[ok].max() - base) / 100.0
    print(mode, "pixels logged", int(ok.sum()), "span %.1f us" % span, "queue dry share", (st["t_queue_dry"] - st["t_first"]) / max(st["t_last"] - st["t_first"], 1), flush=True)
np.savez_compressed(out, W=W, H=H, spp=spp, **res)
r.close()
