#!/usr/bin/env python3
"""Where does a history-free (two-phase) frame lose against a replay?  Instrumented main launch (samples [spp/16, spp)) once in
the order of this frame's probe and once in the order of a full previous frame: wave end-time percentiles, when the queue ran
dry, lane utilisation of the traversal / event / scatter regions.  usage: phase_probe.py [config] [W H spp bounce]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tdt4230_project_raytracing_amd import host, rt
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H, spp, b = (1920, 1080, 64, 8) if len(sys.argv) < 6 else map(int, sys.argv[2:6])
scene = host.Scene.config(cfg); cam = host.camera_reference_pose(W, H, spp, b)
acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
carry = torch.zeros((H, W, 16), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
r = rt.Renderer(scene, cam, image_ptr=acc.data_ptr())
probe = spp // 16


def report(tag):
    d = r.shader.debug_counters()
    span = d["last_end"] - d["first_start"]
    e = r.shader.debug_wave_ends(d["waves"]).astype(np.int64); e = (e - d["first_start"]) / 100.0
    q = np.percentile(e, [1, 25, 50, 75, 95, 100])
    util = {k: d[k + "_active"] / max(1, d[k + "_slots"]) for k in ("trav", "event", "scatter")}
    print(f"{tag}: span {span/100:.0f} us, queue dry at {(d['queue_empty'] - d['first_start'])/span*100:.1f} % of it, mean wave busy "
          f"{d['sum_wave_cycles']/max(1, d['waves'])/span*100:.1f} %, wave ends (us) p1/25/50/75/95/100 {np.round(q).tolist()}, "
          f"lane util trav {util['trav']:.3f} event {util['event']:.3f} scatter {util['scatter']:.3f}")


def frame(order):
    acc.zero_(); carry.zero_(); torch.cuda.synchronize()
    if order == "probe":
        r.ctx.forget_costs()
    r.shader.dispatch_accumulate(W + 1, H + 1, 1, 0, probe, carry.data_ptr())
    r.shader.dispatch_counted_range(W + 1, H + 1, 1, probe, spp - probe, carry.data_ptr())


frame("probe"); report("main launch, order of this frame's probe      ")
frame("replay"); frame("replay"); report("main launch, order of the previous full frame")
os.environ["TDT_NO_COST_ORDER"] = "1"
r.close()
r = rt.Renderer(scene, cam, image_ptr=acc.data_ptr())
frame("probe"); report("main launch, image order                       ")
r.close()
