#!/usr/bin/env python3
"""What would a cheap tile-level cost prior buy a low-spp frame the scheduler has not seen?  Upper bound, with the machinery that
exists: a 1-spp frame of the same view first (its per-pixel costs become 8x8-tile sums, because the next frame's inputs — spp —
differ), then the 4-spp frame in that tile order; only the second is timed (device events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tdt4230_project_raytracing_amd import host, rt
scene = host.Scene.demo()
W, H, spp, b = 1280, 720, 4, 6
cam = host.camera_reference_pose(W, H, spp, b)
stream = torch.cuda.Stream()
r = rt.Renderer(scene, cam, stream=stream.cuda_stream)
def timed(fn, n=50):
    tot = 0.0
    for _ in range(n):
        pre = fn()                                   # untimed part
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); r.dispatch(); e1.record(stream); stream.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n
def fresh():
    r.ctx.forget_costs()
def prior():
    r.ctx.forget_costs()
    r.shader.program.set_i32("camera.samples_per_pixel", 1); r.dispatch()
    r.shader.program.set_i32("camera.samples_per_pixel", spp)
def replay():
    pass
for _ in range(3): r.dispatch()
print(f"4-spp frame, no history (image order): {timed(fresh):.3f} ms")
print(f"4-spp frame in the tile order of a 1-spp frame of the same view: {timed(prior):.3f} ms (+ the 1-spp frame itself)")
r.dispatch(); r.dispatch()
print(f"4-spp frame, replay: {timed(replay):.3f} ms")
r.shader.program.set_i32("camera.samples_per_pixel", 1)
print(f"1-spp frame, no history: {timed(fresh):.3f} ms")
r.close()
