#!/usr/bin/env python3
"""GPU octree builder and parallel voxel edits: times beside their host / ordered counterparts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from octree_util import expand_cells
from tdt4230_project_raytracing_amd import host, rt

ctx = rt.Context(0)
rng = np.random.default_rng(1)
cases = []
for cfg in (2, 3, 5):
    t = time.perf_counter(); scene = host.Scene.config(cfg); th = time.perf_counter() - t
    cases.append((f"config {cfg} scene ({scene.max_depth} levels)", expand_cells(scene.blobs[0], scene.max_depth), scene.max_depth, scene.blobs[0], th))
for depth, n in ((8, 4_000_000), (9, 16_000_000)):
    v = np.concatenate([rng.integers(0, 1 << depth, size=(n, 3), dtype=np.int32), rng.integers(1, 200, size=(n, 1), dtype=np.int32)], axis=1)
    cases.append((f"{n} random voxels in {1 << depth}^3", v, depth, None, None))
for name, vox, depth, want, th in cases:
    vox = np.ascontiguousarray(vox[rng.permutation(len(vox))])
    rt.octree_build_cells(ctx, vox[:1000], depth)            # warm the code object
    ts = []
    for _ in range(3):
        ctx.finish(); t = time.perf_counter(); vbo, n_cells = rt.octree_build_cells(ctx, vox, depth); ctx.finish(); ts.append(time.perf_counter() - t)
    ok = "" if want is None else (" identical to the host builder" if np.array_equal(vbo.read(np.uint32), np.ascontiguousarray(want).view(np.uint32)) else " DIFFERS")
    extra = "" if th is None else f"; host generator + builder {th * 1e3:.1f} ms"
    print(f"{name}: {len(vox)} voxels -> {n_cells} cells in {min(ts) * 1e3:.2f} ms incl. the 16-B/voxel upload ({len(vox) / min(ts) / 1e6:.0f} Mvoxel/s){ok}{extra}")

# edits: n non-colliding edits, parallel plan vs the ordered one-lane walk
from octree_util import distinct_deltas, edit_setup as setup
scene = host.Scene.config(3)
used, depth = scene.counts["cells"], scene.max_depth
for n in (1, 64, 4096, 32768):
    base = host.Scene.config(3)
    base.blobs[0] = np.concatenate([base.blobs[0], np.zeros(16 * (n * depth + 8), np.uint32)])
    delta = distinct_deltas(np.random.default_rng(3), n, depth, base.blobs[0])
    n = len(delta)
    out = []
    for mode in (0, 1):
        r, upd, counter = setup(base, used, delta)
        r.ctx.edit_mode(mode)
        upd.dispatch_compute(n, 1, 1); r.ctx.finish()                       # warm: scratch allocation, code object
        r.vbos[0].sub_data(0, base.blobs[0]); counter.sub_data(0, np.array([used], np.uint32)); r.ctx.finish()
        t = time.perf_counter(); upd.dispatch_compute(n, 1, 1); r.ctx.finish(); dt = time.perf_counter() - t
        out.append((dt, r.ctx.last_edit_path(), r.vbos[0].read(np.uint32)))
        r.close()
    same = np.array_equal(out[0][2], out[1][2])
    print(f"{n} edits: planned {out[0][0] * 1e3:.3f} ms (path {out[0][1]}), ordered walk {out[1][0] * 1e3:.3f} ms (path {out[1][1]}), same bytes: {same}")
ctx.close()
