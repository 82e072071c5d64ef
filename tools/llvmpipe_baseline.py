#!/usr/bin/env python3
"""BUILD-CONTAINER ONLY: time the reference's own raytracer.comp on Mesa llvmpipe for bench.py's workloads.

The reference GL path (assets/shaders/raytracer.comp dispatched as main.rs:579 does) is run, unmodified, through
oracle/_ref/libglref.so on the CPU cores of THIS container, on exactly the scene / camera / dispatch size bench.py uses, and
the result is written to profiles/llvmpipe_baseline.json.  bench.py QUOTES that file beside the GPU number
(cpu_baseline.reference_llvmpipe) — nothing of the reference travels to the GPU box, only these numbers.

    python tools/llvmpipe_baseline.py [--config 2] [--spp 64] [--warm 3]
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1, help="LP_NUM_THREADS (Mesa caps its pool at 16)")
    args = ap.parse_args()
    os.environ["LP_NUM_THREADS"] = str(args.threads)      # read by llvmpipe when the screen is created

    import numpy as np
    import oracle_py
    import bench
    from tdt4230_project_raytracing_amd import host

    if not oracle_py.glref_available():
        raise SystemExit("needs oracle/_ref/libglref.so and the reference checkout (build container only)")
    W, H, spp, bounce, desc = bench.WORKLOADS[args.config][:5]
    if args.spp:
        spp = args.spp
    scene = host.Scene.config(args.config)
    cam = host.camera_reference_pose(W, H, spp, bounce)
    g = oracle_py.GLRef.get()
    written = min(max((W + 1) // 32, 1) * 32, W) * min(max((H + 1) // 32, 1) * 32, H)
    times = []
    t0 = time.perf_counter()
    img, t = g.render(scene, cam, want_time=True)                 # cold: includes the JIT of the shader variant
    cold = time.perf_counter() - t0
    for _ in range(args.warm):
        img, t = g.render(scene, cam, want_time=True, upload=False)
        times.append(t)
    # how many rasteriser threads Mesa really started (it names them "llvmpipe-N")
    n_lp = 0
    for tid in os.listdir("/proc/self/task"):
        try:
            if open(f"/proc/self/task/{tid}/comm").read().startswith("llvmpipe"):
                n_lp += 1
        except OSError:
            pass
    med = sorted(times)[len(times) // 2]
    oc = oracle_py.Oracle()
    # the frame it rendered is the frame the oracle and the GPU kernel produce (a few bands; full frames are in tests/)
    ref = np.zeros_like(img)
    bad = 0
    for y0 in (0, H // 2 // 8 * 8, (H - 40) // 8 * 8):
        oc.render(scene, cam, rows=(y0, y0 + 4), threads=args.threads, image=ref)
        bad += int((img[y0:y0 + 4].view(np.uint32) != ref[y0:y0 + 4].view(np.uint32)).any(axis=2).sum())
    out_path = os.path.join(ROOT, "profiles", "llvmpipe_baseline.json")
    table = json.load(open(out_path)) if os.path.exists(out_path) else {}
    table[f"config{args.config}_spp{spp}"] = {
        "kind": "reference-llvmpipe (quoted, build container)",
        "value": round(written * spp / med / 1e6, 3), "unit": "Mray-samples/s",
        "cores": os.cpu_count(), "lp_num_threads": args.threads, "llvmpipe_threads_seen": n_lp,
        "renderer": g.renderer(), "cpu": platform.processor() or open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
        "workload": desc, "image": [W, H], "dispatch": [W + 1, H + 1, 1], "written_pixels": written, "spp": spp, "max_bounce": bounce,
        "cold_s": round(cold, 3), "warm_s": [round(x, 3) for x in times], "median_warm_s": round(med, 3),
        "timing": "glFinish-bracketed glDispatchCompute + glMemoryBarrier inside glref_dispatch_compute",
        "oracle_mismatched_pixels_in_3_bands": bad,
    }
    json.dump(table, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(table[f"config{args.config}_spp{spp}"]))


if __name__ == "__main__":
    main()
