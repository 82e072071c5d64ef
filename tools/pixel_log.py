#!/usr/bin/env python3
"""Per-pixel log of the instrumented kernel (TDT_PIXEL_LOG=1): how well do a pixel's own step / event counts
predict the time it occupies its lane?  usage: pixel_log.py <config> [W H spp bounce]"""
import sys, os, ctypes
os.environ["TDT_PIXEL_LOG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tdt4230_project_raytracing_amd import host, rt
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H, spp, b = (1920, 1080, 16, 8) if len(sys.argv) < 6 else map(int, sys.argv[2:6])
scene = host.Scene.config(cfg); cam = host.camera_reference_pose(W, H, spp, b)
r = rt.Renderer(scene, cam)
tiles = (W // 32) * (H // 32)
for i in range(3):
    r.shader.dispatch_counted(W + 1, H + 1, 1); d = r.shader.debug_counters()
    log = r.shader.debug_pixel_log(tiles * 1024)
    S, E, passes = log[:, 0].astype(np.float64), log[:, 1].astype(np.float64), log[:, 2].astype(np.float64)
    t0, t1 = log[:, 3].astype(np.float64) / 100, log[:, 4].astype(np.float64) / 100     # us
    ok = E > 0; S, E, passes, t0, t1 = S[ok], E[ok], passes[ok], t0[ok], t1[ok]
    dur = t1 - t0
    span = (d["last_end"] - d["first_start"]) / 100
    print(f"dispatch {i}: span {span:.0f} us, queue dry at {(d['queue_empty'] - d['first_start'])/100:.0f} us; pixels {ok.sum()}")
    A = np.stack([S, E], 1)
    for name, y in (("passes", passes), ("us", dur)):
        coef, *_ = np.linalg.lstsq(A, y, rcond=None)
        pred = A @ coef; rel = np.abs(pred - y) / np.maximum(y, 1)
        print(f"  {name} ~ {coef[0]:.3f}*S + {coef[1]:.3f}*E  (w = {coef[1]/coef[0]:.1f}); median rel err {np.median(rel):.3f}, p90 {np.percentile(rel, 90):.3f}")
    print("  S/E (steps per event) percentiles 10/50/90:", np.percentile(S / E, [10, 50, 90]).round(1).tolist(),
          " us per pass p10/50/90:", np.percentile(dur / np.maximum(passes, 1), [10, 50, 90]).round(3).tolist())
    evp, thr = log[:, 6].astype(np.float64)[ok], log[:, 7].astype(np.float64)[ok]
    late = t1 > np.percentile(t1, 99.5)
    print(f"  latest 0.5% of pixel ends: start us p10/50/90 {np.percentile(t0[late], [10,50,90]).round(0).tolist()}, dur {np.percentile(dur[late], [10,50,90]).round(0).tolist()}, S {np.percentile(S[late],[10,50,90]).round(0).tolist()}, E {np.percentile(E[late],[10,50,90]).round(0).tolist()}, passes {np.percentile(passes[late],[10,50,90]).round(0).tolist()}, event passes {np.percentile(evp[late],[10,50,90]).round(0).tolist()}, threshold {np.percentile(thr[late],[10,50,90]).round(0).tolist()}")
    for lo, hi in ((0, 20), (20, 40), (40, 60), (60, 80), (80, 100)):
        a, bq = np.percentile(t1, [lo, hi]); m = (t1 >= a) & (t1 <= bq)
        print(f"  pixels ending in {lo}-{hi}% of time: S {np.median(S[m]):.0f} E {np.median(E[m]):.0f} passes {np.median(passes[m]):.0f} evpasses {np.median(evp[m]):.0f} thr {np.median(thr[m]):.0f} dur {np.median(dur[m]):.0f} us -> {np.median(dur[m]/passes[m]):.2f} us/pass")
r.close()
