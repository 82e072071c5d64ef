import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
from tdt4230_project_raytracing_amd import host, rt
import oracle_py
orc = oracle_py.Oracle()
scene = host.Scene.config(2)
W, H = 256, 144
def diff(a, b): return int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum())
for spp in (4, 16):
    cam = host.camera_reference_pose(W, H, spp, 8)
    ref = orc.render(scene, cam, threads=16)
    r = rt.Renderer(scene, cam); a = r.render(); b = r.render(); r.close()
    print(f"spp {spp}: first frame {diff(a, ref)} px differ, replay {diff(b, ref)}")
    bad0 = (a.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    ys0, xs0 = np.nonzero(bad0)
    for x, y in list(zip(xs0.tolist(), ys0.tolist()))[:6]:
        print("     px", x, y, "got", a[y, x].tolist(), "want", ref[y, x].tolist())
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0"); carry = torch.zeros((H, W, 16), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    r = rt.Renderer(scene, cam, image_ptr=acc.data_ptr())
    r.shader.dispatch_accumulate(W + 1, H + 1, 1, 0, spp, carry.data_ptr()); r.shader.dispatch_resolve(W + 1, H + 1, 1, spp)
    one = r.texture.read()
    acc.zero_(); carry.zero_(); torch.cuda.synchronize()
    r.shader.dispatch_accumulate(W + 1, H + 1, 1, 0, 1, carry.data_ptr()); r.shader.dispatch_accumulate(W + 1, H + 1, 1, 1, spp - 1, carry.data_ptr()); r.shader.dispatch_resolve(W + 1, H + 1, 1, spp)
    two = r.texture.read(); r.close()
    print(f"   accumulate one launch {diff(one, ref)}, split 1+{spp-1}: {diff(two, ref)}")
    bad = (two.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    if bad.any():
        ys, xs = np.nonzero(bad); print("   first bad pixels:", list(zip(xs[:8].tolist(), ys[:8].tolist())), "max abs diff", float(np.abs(two - ref).max()))
