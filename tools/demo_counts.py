#!/usr/bin/env python3
"""Event counts of the reference's own frame (demo scene, 1280x720, 4 spp): rays per sample, traversal steps per ray, tree levels per step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdt4230_project_raytracing_amd import host, rt
scene = host.Scene.demo(); cam = host.camera_reference_pose(1280, 720, 4, 6)
r = rt.Renderer(scene, cam)
c = r.shader.dispatch_counted(1281, 721, 1)
print(c)
print("levels/step", c["node_loads"]/c["iterations"], "steps/ray", c["iterations"]/c["octree_hit_calls"], "rays/sample", c["octree_hit_calls"]/(c["pixels"]*4))
r.close()
