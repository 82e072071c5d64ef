#!/bin/bash
# GPU session 37: the fixed cost of a trace launch — tiny frames of the demo scene (per-cell-threshold build), the 64^3 scene (whole-depth table) and the 256^3 scene (bricks)
O=gpurun_out/r04aj; mkdir -p $O
python3 - > $O/fixed_cost.txt 2>&1 <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
from tdt4230_project_raytracing_amd import host, rt
for name, scene in (("demo (per-cell thresholds, depth 10)", host.Scene.demo()), ("64^3 (whole-depth table)", host.Scene.config(2)), ("256^3 (bricks)", host.Scene.config(3)), ("512^3 (bricks)", host.Scene.config(5))):
    for W, H, spp in ((64, 64, 1), (256, 256, 1), (1280, 720, 1)):
        cam = host.camera_reference_pose(W, H, spp, 6)
        r = rt.Renderer(scene, cam)
        for mode in ("history-free", "replay"):
            for _ in range(5):
                if mode == "history-free": r.ctx.forget_costs()
                r.dispatch()
            r.ctx.finish(); t = time.perf_counter()
            n = 200
            for _ in range(n):
                if mode == "history-free": r.ctx.forget_costs()
                r.dispatch()
            r.ctx.finish(); dt = (time.perf_counter() - t) / n
            print("%-40s %4dx%-4d spp %d %-12s %.4f ms" % (name, W, H, spp, mode, dt * 1e3), flush=True)
        r.close()
PY
cat $O/fixed_cost.txt | grep -v amdgpu
