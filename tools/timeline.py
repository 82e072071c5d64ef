#!/usr/bin/env python3
"""Wave timeline of the instrumented kernel: how much of the kernel's span is the average wave busy?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdt4230_project_raytracing_amd import host, rt
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H, spp, b = (1920, 1080, 64, 8) if len(sys.argv) < 6 else map(int, sys.argv[2:6])
scene = host.Scene.config(cfg); cam = host.camera_reference_pose(W, H, spp, b)
r = rt.Renderer(scene, cam)
for i in range(3):
    r.shader.dispatch_counted(W + 1, H + 1, 1); d = r.shader.debug_counters()
    span = d["last_end"] - d["first_start"]; mean = d["sum_wave_cycles"] / max(1, d["waves"])
    print(f"dispatch {i}: span {span/100:.1f} us, mean wave lifetime {mean/100:.1f} us = {100*mean/span:.1f}% of span, waves {d['waves']}")
import numpy as np
e = r.shader.debug_wave_ends(d['waves']).astype(np.int64); e = (e - d['first_start']) / 100.0
q = np.percentile(e, [0, 1, 5, 25, 50, 75, 95, 99, 100]); print('wave end times (us) percentiles 0/1/5/25/50/75/95/99/100:', np.round(q).tolist())
print(f"queue ran dry at {(d['queue_empty'] - d['first_start'])/100:.1f} us")
hh = r.shader.debug_wave_ends(16384 + 256)[16384:].astype(np.int64)
for name, h in (("pixels started while the queue had work", hh[:128]), ("pixels of waves that saw the queue's end", hh[128:])):
    c = np.cumsum(h); tot = max(1, c[-1])
    pct = [int(np.searchsorted(c, tot * f)) / 10 for f in (0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0)]
    print(f"{name}: n={tot}, duration ms at 1/10/25/50/75/90/99/100 %: {pct}")
r.close()
