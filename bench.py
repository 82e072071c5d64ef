#!/usr/bin/env python3
"""Benchmark of the hot path: one step = one ComputeShader::dispatch_compute(W+1, H+1, 1) of the
voxel path trace over a synthetic scene already resident in HBM (plus, for N > 1 GPUs, the single
RCCL gather of per-rank tile buffers and their de-interleave on rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|5] [--spp S] [--scaling weak|strong]

N = 1 workload (default): BASELINE.json's metric configuration — 1920x1080, 64 spp, max_bounce 8,
the 64^3-octree synthetic scene of configs[1] — dispatched exactly as the reference does
(main.rs:579), so 1056 of the 1080 rows are written (compute_shader.rs:30-32) and only written
pixels are counted.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md chip table

WORKLOADS = {
    # config id -> (W, H, spp, max_bounce, description)
    2: (1920, 1080, 64, 8, "1920x1080, 64 spp, max_bounce 8, 64^3 octree terrain+spheres (BASELINE configs[1] scene at the metric's 64 spp)"),
    3: (3840, 2160, 64, 16, "3840x2160, 64 spp, max_bounce 16, 256^3 octree (BASELINE configs[2])"),
    5: (1920, 1080, 64, 8, "1920x1080, 64 spp pass of the progressive config, max_bounce 8, 512^3 sparse octree (BASELINE configs[4])"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N>1: weak = N x the pixel rows over the same frustum (per-GPU work fixed); strong = same image")
    ap.add_argument("--passes", type=int, default=1,
                    help="N=1 only: a step = one PROGRESSIVE frame of passes x spp samples per pixel (configs[4] is 16 x 64): "
                         "running sums and hit-record carry through HBM, one resolve at the end (bit-identical to one pass)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend of the N>1 gather; gloo (through host memory) lets several ranks share "
                         "one GPU to rehearse the multi-rank path on a single-GPU box")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path (tile buffer, RCCL gather, de-interleave) even with one rank: a self-test")
    args = ap.parse_args()

    # the contract is ONE JSON line on stdout: libraries that chat on fd 1 (RCCL prints a host / library banner when a
    # communicator is created) are sent to stderr for the duration of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from tdt4230_project_raytracing_amd import host, rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace has no CPU path")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()              # rehearsal: ranks may share a device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_dist
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    W, H, spp, bounce, desc = WORKLOADS[args.config]
    if args.spp:
        spp = args.spp
    rows_factor = world if (world > 1 and args.scaling == "weak") else 1
    scene = host.Scene.config(args.config)
    cam = host.camera_reference_pose(W, H, spp, bounce)
    if rows_factor > 1:
        # same frustum, rows_factor x the pixel rows: every rank keeps one N=1 frame's worth of work
        cam.image_height = H * rows_factor
    IW, IH = cam.image_width, cam.image_height
    dw, dh = IW + 1, IH + 1                      # main.rs:579

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        full = torch.zeros((IH, IW, 4), dtype=torch.float32, device=dev) if rank == 0 else None
        if not sharded:
            r = rt.Renderer(scene, cam, device=local_rank, stream=stream.cuda_stream, image_ptr=full.data_ptr())
            tiles_per_rank = 0
            tile_buf = gathered = full_tex = None
        else:
            cover_w = min(max(dw // 32, 1) * 32, IW)
            cover_h = min(max(dh // 32, 1) * 32, IH)
            total_tiles = -(-cover_w // 32) * -(-cover_h // 32)
            tiles_per_rank = -(-total_tiles // world)                # what rank 0 owns: the most
            tile_buf = torch.zeros((tiles_per_rank, 32, 32, 4), dtype=torch.float32, device=dev)
            r = rt.Renderer(scene, cam, device=local_rank, stream=stream.cuda_stream, rank=rank, world=world,
                            image_ptr=tile_buf.data_ptr(), tile_buffer_tiles=tiles_per_rank)
            gathered = torch.zeros((world, tiles_per_rank, 32, 32, 4), dtype=torch.float32, device=dev) if rank == 0 else None
            full_tex = rt.Texture.wrap_device(r.ctx, full.data_ptr(), IW, IH, bind=False) if rank == 0 else None
    carry = None
    if args.passes > 1:
        if sharded:
            raise SystemExit("--passes is a single-GPU option")
        with torch.cuda.stream(stream):
            carry = torch.zeros((IH, IW, 16), dtype=torch.float32, device=dev)
        args.no_cpu_baseline = True
    my_pixels = r.shader.covered_pixels(dw, dh)
    if sharded:
        assert r.shader.owned_tiles(dw, dh)[0] <= tiles_per_rank

    ev_pairs = []

    first_pair = []

    def step(timed):
        with torch.cuda.stream(stream):
            if not timed and not first_pair:             # the context's very first dispatch: image order, no cost history
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                first_pair.extend([e0, e1])
                timed_first = True
            else:
                timed_first = False
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            if carry is None:
                r.shader.dispatch_compute(dw, dh, 1)
            else:
                full.zero_(); carry.zero_()
                r.shader.dispatch_accumulate(dw, dh, 1, 0, spp, carry.data_ptr())
            if timed:
                e1.record(stream)
                ev_pairs.append((e0, e1))
            if timed_first:
                first_pair[1].record(stream)
            if carry is not None:
                for k in range(1, args.passes):
                    r.shader.dispatch_accumulate(dw, dh, 1, k * spp, spp, carry.data_ptr())
                r.shader.dispatch_resolve(dw, dh, 1, spp * args.passes)
            if sharded:
                if args.backend == "nccl":
                    glist = list(gathered.unbind(0)) if rank == 0 else None
                    dist.gather(tile_buf, glist, dst=0)
                else:                                        # rehearsal path: gloo gathers host tensors
                    stream.synchronize()
                    host_buf = tile_buf.cpu()
                    hlist = [torch.empty_like(host_buf) for _ in range(world)] if rank == 0 else None
                    dist.gather(host_buf, hlist, dst=0)
                    if rank == 0:
                        gathered.copy_(torch.stack(hlist))
                if rank == 0:
                    r.shader.assemble_tiles(gathered.data_ptr(), world, tiles_per_rank, full_tex, dw, dh)

    def fence():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(my_pixels)], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if sharded:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, total_pixels = float(tmax[0]), float(tsum[1])
    else:
        total_pixels = float(my_pixels)
    ms_per_step = dt / args.steps * 1e3
    samples_per_step = total_pixels * spp * args.passes
    value = samples_per_step / (dt / args.steps) / 1e6

    # --- roofline of the dominant kernel (the trace kernel), measured on rank 0 -----------------
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev_pairs) / max(1, len(ev_pairs))
    counts = r.shader.dispatch_counted(dw, dh, 1)          # instrumented, untimed: algorithmic events of ONE launch
    # 8 B per Node load + material / albedo / attribute reads + 40 B of octree uniforms
    read_bytes = (8 * counts["node_loads"] + 24 * counts["lambertian"] + 28 * counts["metal"]
                  + 16 * counts["dielectric"] + 40)
    write_bytes = 16 * counts["pixels"]
    achieved = read_bytes / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            key = f"config{args.config}_spp{spp}_gpus{world}"
            traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "kernel": "tdt::trace_kernel<false,...> (COUNT = false: the product build)", "kernel_ms": round(kernel_ms, 4),
                "algorithmic_read_bytes": int(read_bytes), "algorithmic_write_bytes": int(write_bytes),
                "node_loads": counts["node_loads"], "rays": counts["octree_hit_calls"]}

    # --- CPU baseline: the oracle (a port, not the product) on a bounded sample ----------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py
        orc = oracle_py.Oracle()
        cores = min(os.cpu_count() or 1, 32)
        bands, band_rows = 96, 8
        ys = [int(i * (IH - 32) / bands) // 8 * 8 for i in range(bands)]
        img = np.zeros((IH, IW, 4), np.float32)
        t0 = time.perf_counter()
        px = 0
        for y0 in ys:
            orc.render(scene, cam, (dw, dh), rows=(y0, y0 + band_rows), threads=cores, image=img)
            px += IW * band_rows
        tc = time.perf_counter() - t0
        cpu_baseline = {"value": round(px * spp / tc / 1e6, 3), "unit": "Mray-samples/s", "cores": cores, "kind": "port",
                        "sample": f"{bands} bands of {band_rows} rows ({px} px x {spp} spp) of the same frame, {tc:.1f} s"}
        # spot-check the product against the checker on the sampled rows (never the other way round)
        got = full.cpu().numpy()
        bad = 0
        for y0 in ys:
            bad += int((got[y0:y0 + band_rows].view(np.uint32) != img[y0:y0 + band_rows].view(np.uint32)).any(axis=2).sum())
        cpu_baseline["mismatched_pixels_in_sample"] = bad

    if rank == 0:
        out = {
            "metric": "Mray-samples/sec at 1080p/64spp/depth-8" if args.config == 2 and spp == 64 and args.passes == 1 else f"Mray-samples/sec (config {args.config}, {spp * args.passes} spp" + (f" as {args.passes} progressive passes" if args.passes > 1 else "") + ")",
            "value": round(value, 2), "unit": "Mray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc + (f"; x{rows_factor} pixel rows over the same frustum (weak scaling)" if rows_factor > 1 else ""),
                       "image": [IW, IH], "dispatch": [dw, dh, 1], "written_pixels": int(total_pixels), "spp": spp * args.passes,
                       "passes": args.passes,
                       "max_bounce": bounce, "octree_max_depth": scene.max_depth, "octree_cells": scene.counts["cells"],
                       "scene_bytes": scene.nbytes(),
                       "schedule": "global pixel queue; pixels handed out most-expensive-first from the work counts (tree levels, "
                                   "steps, path events) the previous dispatch recorded per pixel (a bench step repeats the same frame; "
                                   "when the camera or scene changed, 8x8 tiles are ordered instead); the first dispatch of a "
                                   "context runs in image order (TDT_NO_COST_ORDER=1: always) — first_dispatch_ms is that dispatch, timed "
                                   "during warm-up",
                       "first_dispatch_ms": round(first_pair[0].elapsed_time(first_pair[1]), 4) if first_pair else None,
                       "partition": f"32x32 work-groups dealt round-robin over {world} rank(s)" + ("; one RCCL gather + de-interleave per step" if sharded else "")},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    r.close()
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
