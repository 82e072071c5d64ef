#!/usr/bin/env python3
"""Randomised parity sweep: generated scenes x cameras (inside / outside the octree, axis-aligned views that put rays exactly on
cell boundaries, odd image sizes, spp above and below the two-phase limit), GPU frames (first = two-phase or image order, then a
cost-ordered replay; a random progressive split; a 2-4 rank work-group partition; a 2-5 share multi-device context) against the
oracle, bit for bit.
usage: fuzz_parity.py [seconds] [seed]; tests/test_gpu_fuzz.py runs a seeded, time-boxed sweep of it in the -m gpu suite."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import torch
from tdt4230_project_raytracing_amd import host, rt
import oracle_py



def with_cell_count(scene, cc):
    """The same tree under another cell_count uniform: Octree::init_global_buffers' floats[6] = 1.0 / cell_count as f32
    (octree.rs:49), ints[2] = cell_count."""
    blobs = {k: v.copy() for k, v in scene.blobs.items()}
    blobs[6][6] = np.float32(1.0) / np.float32(cc)
    blobs[7][2] = cc
    return host.Scene(blobs, scene.counts, scene.name + f"_cc{cc}")


def with_zero_tail(scene, nodes):
    blobs = {k: v.copy() for k, v in scene.blobs.items()}
    blobs[0] = np.concatenate([blobs[0], np.zeros(2 * nodes, np.uint32)])
    return host.Scene(blobs, scene.counts, scene.name + f"_tail{nodes}")


def run(budget=120.0, seed=1, log=print):
    """Sweep for `budget` seconds; returns (cases, mismatches)."""
    rng = np.random.default_rng(seed)
    orc = oracle_py.Oracle()
    t0 = time.time(); n = 0; bad = 0
    while time.time() - t0 < budget:
        kind = int(rng.integers(0, 3)); depth = int(rng.integers(3, 10))
        cells_log = int(rng.integers(12, 21)); max_iter = int(rng.choice([40, 100, 256]))
        try:
            scene = host.Scene.generate(kind, depth, 1 << cells_log, max_iter, int(rng.integers(1, 1 << 30)))
        except RuntimeError:
            continue                                                   # scene needs more cells than cell_count
        flavour = int(rng.integers(0, 4))
        if os.environ.get("FUZZ_BIAS") and rng.random() < 0.6:
            flavour = 1                                                # FUZZ_BIAS=1: mostly non-power-of-two counts ...
        if flavour == 1:                                               # a cell_count that is not a power of two (the reference's own is 100000)
            scene = with_cell_count(scene, int(rng.choice([100000, 99999, 12345, 65537, 3000, 1000003])) if rng.random() < 0.7 else int(rng.integers(scene.counts["cells"] + 1, 1 << 21)))
        if flavour == 2 or (flavour == 1 and rng.random() < 0.5):      # a pre-allocated cells buffer: a tail of zero nodes (main.rs:339-341)
            scene = with_zero_tail(scene, int(rng.integers(1, 60000)))
        W = int(rng.choice([32, 64, 96, 100, 131])); H = int(rng.choice([32, 64, 70, 97]))
        spp = int(rng.choice([1, 2, 5, 16, 33])); bounce = int(rng.choice([1, 3, 8]))
        mode = int(rng.integers(0, 4))
        if os.environ.get("FUZZ_BIAS") and rng.random() < 0.4:
            mode = 2                                                   # ... and cameras anywhere (mostly outside: the miss pre-pass)
        if mode == 0:
            cam = host.camera_reference_pose(W, H, spp, bounce)
        else:
            origin = {1: (float(rng.uniform(-0.45, 0.45)), float(rng.uniform(-0.45, 0.45)), float(rng.uniform(-0.95, -0.05))),   # inside
                      2: (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), float(rng.uniform(-3, 1))),                    # anywhere
                      3: (0.0, 0.0, -0.5)}[mode]                                                                               # on cell boundaries
            c = host.Camera(float(rng.choice([40.0, 90.0, 120.0])), W, aspect_ratio=np.float32(W) / np.float32(H), origin=origin,
                            viewport_height=2.0, samples_per_pixel=spp, max_bounce=bounce)
            if mode != 3:
                c.turn_yaw(float(rng.uniform(-60, 60))); c.turn_pitch(float(rng.uniform(-20, 20)))
            cam = c.uniforms()
        ref = orc.render(scene, cam, threads=16)
        r = rt.Renderer(scene, cam)
        try:
            for k in range(2):
                got = r.render()
                if not (got.view(np.uint32) == ref.view(np.uint32)).all():
                    bad += 1
                    log(f"MISMATCH {scene.name} kind {kind} depth {depth} cells 2^{cells_log} iter {max_iter} {W}x{H} spp {spp} bounce {bounce} cam mode {mode} frame {k}: "
                          f"{int((got.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())} px")
        finally:
            r.close()
        how = int(rng.integers(0, 5))
        if how == 1 and spp >= 2:                                      # progressive: running sums + carry over a random split, then resolve
            a = int(rng.integers(1, spp))
            acc = torch.zeros((cam.image_height, W, 4), dtype=torch.float32, device="cuda:0")
            carry = torch.zeros((cam.image_height, W, 16), dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            r = rt.Renderer(scene, cam, image_ptr=acc.data_ptr())
            try:
                r.shader.dispatch_accumulate(W + 1, cam.image_height + 1, 1, 0, a, carry.data_ptr())
                r.shader.dispatch_accumulate(W + 1, cam.image_height + 1, 1, a, spp - a, carry.data_ptr())
                r.shader.dispatch_resolve(W + 1, cam.image_height + 1, 1, spp)
                got = r.texture.read()
            finally:
                r.close()
            if not (got.view(np.uint32) == ref.view(np.uint32)).all():
                bad += 1; log(f"MISMATCH progressive split {a}/{spp} kind {kind} depth {depth} {W}x{H}")
        elif how == 2:                                                 # work-group partition over 2-4 ranks into one image
            world = int(rng.integers(2, 5))
            full = torch.zeros((cam.image_height, W, 4), dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            for rank in range(world):
                r = rt.Renderer(scene, cam, rank=rank, world=world, image_ptr=full.data_ptr())
                try:
                    r.dispatch(); r.ctx.finish()
                finally:
                    r.close()
            got = full.cpu().numpy()
            if not (got.view(np.uint32) == ref.view(np.uint32)).all():
                bad += 1; log(f"MISMATCH partition world {world} kind {kind} depth {depth} {W}x{H}")
        if how == 3:                                                   # one multi-device context, 2-5 shares on this GPU
            shares = int(rng.integers(2, 6))
            r = rt.Renderer(scene, cam, devices=[0] * shares)
            try:
                for k in range(2):
                    got = r.render()
                    if not (got.view(np.uint32) == ref.view(np.uint32)).all():
                        bad += 1; log(f"MISMATCH multi-device context shares {shares} frame {k} kind {kind} depth {depth} {W}x{H}")
            finally:
                r.close()
        n += 1
        if n % 20 == 0:
            log(f"{n} cases, {bad} mismatches, {time.time() - t0:.0f} s")
    return n, bad


if __name__ == "__main__":
    n, bad = run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                 log=lambda m: print(m, flush=True))
    print(f"done: {n} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)
