#!/usr/bin/env python3
"""Long run of the bench frame: per-dispatch kernel times over N dispatches (cost sums restart every 256)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tdt4230_project_raytracing_amd import host, rt
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
scene = host.Scene.config(2); cam = host.camera_reference_pose(1920, 1080, 64, 8)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    r = rt.Renderer(scene, cam, stream=stream.cuda_stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record(stream)
    for i in range(N):
        r.dispatch(); ev[i + 1].record(stream)
torch.cuda.synchronize()
t = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(N)])
print("first 3:", t[:3].round(2), " median %.2f  p99 %.2f  max %.2f (at %d)" % (np.median(t), np.percentile(t, 99), t.max(), t.argmax()))
print("around restarts:", {k: t[k - 1:k + 3].round(2).tolist() for k in (256, 257, 512, 513) if k + 3 < N})
r.close()
