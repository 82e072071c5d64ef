#!/bin/bash
# GPU session 38: the new guard test (and its margin)
O=gpurun_out/r04ak; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stop_advancing" > $O/test.txt 2>&1; tail -3 $O/test.txt
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from tdt4230_project_raytracing_amd import host, rt
scene = host.Scene.config(5); cam = host.camera_reference_pose(480, 270, 16, 8)
for walk in (False, True):
    if walk: os.environ["TDT_NO_BRICKS"] = "1"
    r = rt.Renderer(scene, cam); r.dispatch(); r.ctx.finish(); r.ctx.phase_timing(True)
    ts = []
    for _ in range(3):
        r.ctx.forget_costs(); r.dispatch(); r.ctx.finish(); ts.append(sum(r.ctx.phase_timing(True)))
    print("walk" if walk else "brick", [round(t, 3) for t in ts]); r.close()
PY
