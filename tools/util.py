#!/usr/bin/env python3
"""Lane-utilisation diagnostics of the trace kernel on a config (instrumented dispatch)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdt4230_project_raytracing_amd import host, rt
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H, spp, b = (1920, 1080, 16, 8) if len(sys.argv) < 6 else map(int, sys.argv[2:6])
scene = host.Scene.config(cfg); cam = host.camera_reference_pose(W, H, spp, b)
r = rt.Renderer(scene, cam)
for _ in range(int(os.environ.get("UTIL_DISPATCHES", "1"))):   # > 1: the later ones run in cost-feedback order
    c = r.shader.dispatch_counted(W + 1, H + 1, 1); d = r.shader.debug_counters()
r.close()
print(json.dumps(c)); print(json.dumps(d))
for k in ("trav", "level", "event", "scatter"):
    print(f"{k:8s} util {d[k+'_active']/max(1,d[k+'_slots']):.3f}  slots {d[k+'_slots']:.4g}")
print("levels/iter %.2f  memo miss rate %.3f  iters/ray %.2f  rays/sample %.2f" % (c["node_loads"]/c["iterations"], d["memo_miss"]/c["node_loads"], c["iterations"]/c["octree_hit_calls"], c["octree_hit_calls"]/(c["pixels"]*spp)))
rc = d.get("region_cycles", {})
tot = sum(rc.values()) or 1
print("time per region (wave cycles, instrumented build):", "  ".join(f"{k} {100*v/tot:.1f}%" for k, v in rc.items()))
