#!/usr/bin/env python3
"""Benchmark of the hot path: one step = one ComputeShader::dispatch_compute(W+1, H+1, 1) of the voxel path trace over a
synthetic scene already resident in HBM (plus, for N > 1 GPUs, the single RCCL gather of per-rank tile buffers and their
de-interleave on rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5] [--spp S] [--scaling weak|strong]

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its N ranks itself (child processes,
started before this process touches a GPU); under `python -m torch.distributed.run --nproc-per-node N` it is one of the ranks.

N = 1 workload (default): BASELINE.json's metric configuration — 1920x1080, 64 spp, max_bounce 8, the 64^3-octree synthetic
scene of configs[1] — dispatched exactly as the reference does (main.rs:579), so 1056 of the 1080 rows are written
(compute_shader.rs:30-32) and only written pixels are counted.  Every timed step is a frame the scheduler has NOT seen
(tdt_forget_costs before it: two-phase schedule, probe in image order); the replay of an identical frame is reported beside
it as config.replay_ms.  N > 1: weak scaling of that frame (N x the pixel rows over the same frustum).  Every line also
carries a `strong` block — BASELINE configs[3]: ONE 7680x4320 / 64 spp / 256^3 frame sharded over the N ranks, per-rank trace,
gather and assemble times separated — and a `single_process` block: the same 8K frame driven through the C ABI's
multi-device context (tdt_ctx_create_multi) from one process.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md chip table

WORKLOADS = {
    # config id -> (W, H, spp, max_bounce, description, scene config)
    0: (1280, 720, 4, 6, "the reference's own workload: demo scene (main.rs:235-463, cell_count 100000), 1280x720 window (main.rs:26), 4 spp / max_bounce 6 "
                         "(assets/settings/camera.ron:2-3)", 0),
    2: (1920, 1080, 64, 8, "1920x1080, 64 spp, max_bounce 8, 64^3 octree terrain+spheres (BASELINE configs[1] scene at the metric's 64 spp)", 2),
    3: (3840, 2160, 64, 16, "3840x2160, 64 spp, max_bounce 16, 256^3 octree (BASELINE configs[2])", 3),
    4: (7680, 4320, 64, 8, "7680x4320, 64 spp, max_bounce 8, 256^3 octree, ONE frame tile-sharded over the ranks (BASELINE configs[3])", 4),
    5: (1920, 1080, 64, 8, "1920x1080, 64 spp pass of the progressive config, max_bounce 8, 512^3 sparse octree (BASELINE configs[4])", 5),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="N>1: weak = N x the pixel rows over the same frustum (per-GPU work fixed; default); strong = same image "
                         "(always strong for --config 4)")
    ap.add_argument("--passes", type=int, default=1,
                    help="N=1 only: a step = one PROGRESSIVE frame of passes x spp samples per pixel (configs[4] is 16 x 64): "
                         "running sums and hit-record carry through HBM, one resolve at the end (bit-identical to one pass)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the weak-scaled side block (N x the pixel rows)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaled 8K block (BASELINE configs[3])")
    ap.add_argument("--no-target", action="store_true", help="skip the target_4k block (BASELINE configs[2], the north-star roofline config)")
    ap.add_argument("--no-reference-default", action="store_true", help="skip the reference_default block (demo scene, 1280x720, 4 spp)")
    ap.add_argument("--target-steps", type=int, default=5)
    ap.add_argument("--reference-steps", type=int, default=50)
    ap.add_argument("--no-single-process", action="store_true", help="skip the multi-device-context (one process, N GPUs) block")
    ap.add_argument("--strong-steps", type=int, default=3)
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend of the N>1 gather; gloo (through host memory) lets several ranks share "
                         "one GPU to rehearse the multi-rank path on a single-GPU box")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path (tile buffer, RCCL gather, de-interleave) even with one rank: a self-test")
    ap.add_argument("--replay", action="store_true", help="headline = replay of an identical frame (round-1 behaviour)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="self-check of the launch path without a GPU: the ranks meet over gloo, sum their ranks, rank 0 prints a JSON line")
    ap.add_argument("--single-process-leg", type=int, default=0, metavar="N",
                    help="(internal) run ONLY the multi-device-context measurement over N devices and print its JSON")
    ap.add_argument("--share-device", action="store_true",
                    help="single-process leg: all N shares on device 0 (one-GPU box rehearsal; peer-copy transport)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ self-launch -----
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_environments(n, port, base=None):
    """The environment torch.distributed.run would give each of n ranks on one node."""
    envs = []
    for r in range(n):
        e = dict(base if base is not None else os.environ)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=e.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        envs.append(e)
    return envs


def self_launch(argv, n):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as fresh child processes (this process has not
    touched a GPU and never will), pass rank 0's JSON line through, exit with the worst return code."""
    envs = rank_environments(n, free_port())
    procs = []
    for r, e in enumerate(envs):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        try:
            p.wait(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = rc or 1
        rc = rc or p.returncode
    line = ""
    for ln in (out or "").splitlines():
        if ln.startswith("{"):
            line = ln
    if line:
        print(line, flush=True)
    elif not rc:
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------------ helpers ---------
def algorithmic_bytes(counts):
    """8 B per Node load + material / albedo / attribute reads + 40 B of octree uniforms; 16 B written per pixel."""
    read = 8 * counts["node_loads"] + 24 * counts["lambertian"] + 28 * counts["metal"] + 16 * counts["dielectric"] + 40
    return int(read), int(16 * counts["pixels"])


def quoted(path, key):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", path))).get(key)
    except Exception:
        return None


def quoted_valu(wkey):
    """(SQ-counter summary of the workload's dominant launch, the file it came from): the newest round that has one."""
    for name in ("r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json"):
        v = quoted(name, wkey)
        if v:
            return v, "profiles/" + name
    return None, None


NOMINAL_BOUND = "algorithmic-bytes/HBM (nominal)"
NOMINAL_NOTE = ("NOMINAL, not a physical roofline: algorithmic bytes (what a cache-less implementation would fetch: 8 B per tree level visited + the "
                "material records, SURVEY §8d) over the HBM peak.  The tree is served from LDS / L2 and one table or brick load answers several "
                "levels, so measured HBM traffic is a small fraction of the algorithmic bytes and this figure may exceed 1; the kernel is "
                "physically bound by VALU issue x live lanes: see `physical`")


def physical_roofline(wkey):
    """The bound the kernel really runs against: share of the SIMDs' VALU issue slots used x share of the 64 lanes live per issued
    instruction = fraction of peak useful lane throughput (rocprofv3 SQ counters of the workload's dominant launch, quoted from
    profiles/: PMC passes cannot run inside a timed bench)."""
    v, src = quoted_valu(wkey)
    if not v or v.get("issue_util") is None or v.get("lane_util") is None:
        return None
    return {"bound": "valu", "frac": round(v["issue_util"] * v["lane_util"], 4), "issue_util": v["issue_util"], "lane_util": v["lane_util"],
            "unit": "fraction of peak VALU lane throughput (issue-slot use x live lanes)", "salu_per_valu": round(v["salu_instructions"] / v["valu_wave_instructions"], 4)
            if v.get("salu_instructions") and v.get("valu_wave_instructions") else None,
            "kernel_ms_under_pmc": v.get("kernel_ms_under_pmc"), "source": src, "formula": v.get("formula"),
            "note": ("utilisation of a NOMINAL issue peak (one wave64 VALU instruction per 2 cycles per SIMD).  On gfx950 only add / sub / mul / fma / logic / "
                     "mov issue at that rate; compares, min / max, converts, shifts left, three-operand integer forms and every VALU instruction with "
                     "an SGPR operand take about 4.3 cycles, transcendentals 8.2 (tools/micro/pipe_model.hip), and the traversal step is about half "
                     "such instructions — round 4 made the frame faster by REMOVING instructions, which lowers this figure; see DESIGN.md §4")}


def single_process_leg(args):
    """One process, N devices, through tdt_ctx_create_multi: the 8K frame of BASELINE configs[3] (or --config)."""
    import numpy as np
    import torch
    from tdt4230_project_raytracing_amd import host, rt
    n = args.single_process_leg
    cfg = args.config if args.config != 2 else 4
    W, H, spp, bounce, desc, scene_cfg = WORKLOADS[cfg]
    if args.spp:
        spp = args.spp
    have = torch.cuda.device_count()
    devices = [0] * n if args.share_device else list(range(n))
    if not args.share_device and have < n:
        raise SystemExit(f"single-process leg over {n} devices, but {have} are visible")
    scene = host.Scene.config(scene_cfg)
    cam = host.camera_reference_pose(W, H, spp, bounce)
    r = rt.Renderer(scene, cam, devices=devices)
    dw, dh = W + 1, H + 1
    px = r.shader.covered_pixels(dw, dh)
    for _ in range(max(1, args.warmup)):
        r.ctx.forget_costs()
        r.dispatch()
    r.ctx.finish()
    rows = []
    t0 = time.perf_counter()
    for _ in range(args.strong_steps):
        r.ctx.forget_costs()
        r.dispatch()
        rows.append(r.ctx.multi_timing())                 # blocks until the frame is assembled
    dt = (time.perf_counter() - t0) / args.strong_steps
    # the assembled frame, checked on three 4-row bands against a single-device trace of the same rows' work-groups is
    # the job of tests/test_gpu_multi.py; here: the whole covered image was written
    img = r.texture.read()
    written = int((img[..., 3] == 1).sum())
    out = {"devices": devices, "transport": r.ctx.multi_transport(), "rccl_ranks": r.ctx.multi_rccl_ranks(), "workload": desc, "image": [W, H], "spp": spp,
           "written_pixels": written, "covered_pixels": px, "steps": args.strong_steps, "ms_per_frame": round(dt * 1e3, 3),
           "value": round(px * spp / dt / 1e6, 2), "unit": "Mray-samples/s",
           "trace_ms_per_device": [round(float(np.mean([row[0][i] for row in rows])), 3) for i in range(n)],
           "gather_ms": round(float(np.mean([row[1] for row in rows])), 3),
           "assemble_ms": round(float(np.mean([row[2] for row in rows])), 3),
           "schedule": "every frame history-free (tdt_forget_costs): two-phase on each device"}
    r.close()
    print(json.dumps(out), flush=True)


def run_single_process_child(args, n, timeout_s=240):
    """The multi-device-context leg in a fresh process (isolated: a failure there is reported, not fatal)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--single-process-leg", str(n), "--strong-steps", str(args.strong_steps),
           "--warmup", "1"]
    if args.config == 4 and args.spp:
        cmd += ["--spp", str(args.spp)]
    if args.backend == "gloo":
        cmd += ["--share-device"]                          # the one-GPU rehearsal: all shares on device 0
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                           "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout_s} s"}
    for ln in reversed(p.stdout.splitlines()):
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                break
    return {"error": f"rc {p.returncode}: " + (p.stderr or p.stdout)[-400:]}


def strong_summary(ms_per_frame, trace_ms_per_rank, one_gpu_ms, one_gpu_source, backend, ranks_seen):
    """The fields that make a strong-scaling line self-explaining: speed-up over one GPU, load imbalance of the per-rank traces,
    the collective's backend and how many ranks it really saw."""
    mean = sum(trace_ms_per_rank) / max(1, len(trace_ms_per_rank))
    return {"speedup_vs_1gpu": round(one_gpu_ms / ms_per_frame, 3) if (one_gpu_ms and ms_per_frame) else None,
            "one_gpu_ms": round(one_gpu_ms, 3) if one_gpu_ms else None, "one_gpu_source": one_gpu_source,
            "imbalance": round(max(trace_ms_per_rank) / mean, 4) if mean > 0 else None,
            "imbalance_note": "max / mean of trace_ms_per_rank (1.0 = perfectly balanced shares)",
            "backend": backend, "ranks_seen": int(ranks_seen)}


def headline_summary(value_n, value_1, world):
    """The N > 1 headline against the same frame on one GPU: value(N) / value(1) IS the strong-scaling speed-up."""
    return {"scaling": "strong", "n_gpus": int(world), "value": round(value_n, 2), "one_gpu_value": round(value_1, 2),
            "speedup_vs_1gpu": round(value_n / value_1, 3) if value_1 else None,
            "efficiency_note": "speedup_vs_1gpu / n_gpus is the strong-scaling efficiency (the driver computes its own from the per-N lines)"}


def rendezvous_only():
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank), 1.0], dtype=torch.float64)
    dist.all_reduce(t)
    # the strong block's summary fields, from numbers every rank contributes over the same process group (self-check of the
    # N>1 line's arithmetic without a GPU): rank r "traced" for 10 (r + 1) ms, the frame took as long as the slowest
    tr = torch.zeros(world, dtype=torch.float64)
    tr[rank] = 10.0 * (rank + 1)
    dist.all_reduce(tr)
    seen = dist.get_world_size()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        trace = [float(v) for v in tr]
        # the N > 1 headline's arithmetic on the same synthetic times: a frame of 1000 "samples" takes sum(trace) ms on one GPU and
        # max(trace) ms sharded, so value(N) / value(1) must equal the strong block's speed-up
        value_1, value_n = 1000.0 / sum(trace), 1000.0 / max(trace)
        print(json.dumps({"rendezvous": world, "rank_sum": t[0].item(), "ranks_seen": int(t[1].item()),
                          "strong": strong_summary(max(trace), trace, sum(trace), "synthetic (sum of the per-rank times)", "gloo", seen),
                          "headline_scaling": headline_summary(value_n, value_1, world)}), flush=True)


# ------------------------------------------------------------------------------------------------ a rank ----------
class Rank:
    """One rank's state: device, stream, process group."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the trace has no CPU path")
        if args.backend == "gloo":
            local_rank %= torch.cuda.device_count()              # rehearsal: ranks may share a device
        torch.cuda.set_device(local_rank)
        self.local_rank = local_rank
        self.dev = torch.device("cuda", local_rank)
        self.backend = args.backend
        self.sharded = self.world > 1 or args.force_dist
        if self.sharded:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
        self.stream = torch.cuda.Stream(device=self.dev)

    def fence(self):
        if self.sharded:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def reduce(self, values, op):
        """all-reduce a short list of floats over the ranks (max / sum)."""
        if not self.sharded:
            return list(values)
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return [float(v) for v in t]

    def gather_floats(self, value):
        """every rank's float, on every rank"""
        if not self.sharded:
            return [float(value)]
        t = self.torch.zeros(self.world, dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        t[self.rank] = value
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t]

    def close(self):
        if self.sharded:
            self.dist.destroy_process_group()


class Workload:
    """One scene + camera + image on a rank: a whole image (unsharded) or this rank's tile buffer + gather + assemble."""

    def __init__(self, R, cfg, spp=None, rows_factor=1, passes=1, alone=False):
        """alone: this rank traces the whole image by itself, whatever the world size (the 1-GPU reference of the strong block)."""
        import numpy as np  # noqa: F401
        from tdt4230_project_raytracing_amd import host, rt
        torch = R.torch
        self.R, self.rt = R, rt
        self.sharded = R.sharded and not alone
        W, H, wspp, bounce, desc, scene_cfg = WORKLOADS[cfg]
        self.cfg, self.desc, self.bounce = cfg, desc, bounce
        self.spp = spp or wspp
        self.passes = passes
        self.scene = host.Scene.config(scene_cfg)
        self.cam = host.camera_reference_pose(W, H, self.spp, bounce)
        self.rows_factor = rows_factor
        if rows_factor > 1:
            # same frustum, rows_factor x the pixel rows: every rank keeps one N=1 frame's worth of work
            self.cam.image_height = H * rows_factor
        self.IW, self.IH = self.cam.image_width, self.cam.image_height
        self.dw, self.dh = self.IW + 1, self.IH + 1                      # main.rs:579
        IW, IH, dw, dh = self.IW, self.IH, self.dw, self.dh
        with torch.cuda.stream(R.stream):
            self.full = torch.zeros((IH, IW, 4), dtype=torch.float32, device=R.dev) if (R.rank == 0 or alone) else None
            if not self.sharded:
                self.r = rt.Renderer(self.scene, self.cam, device=R.local_rank, stream=R.stream.cuda_stream, image_ptr=self.full.data_ptr())
                self.tiles_per_rank = 0
                self.tile_buf = self.gathered = self.full_tex = None
            else:
                cover_w = min(max(dw // 32, 1) * 32, IW)
                cover_h = min(max(dh // 32, 1) * 32, IH)
                total_tiles = -(-cover_w // 32) * -(-cover_h // 32)
                self.tiles_per_rank = -(-total_tiles // R.world)                # what rank 0 owns: the most
                self.tile_buf = torch.zeros((self.tiles_per_rank, 32, 32, 4), dtype=torch.float32, device=R.dev)
                self.r = rt.Renderer(self.scene, self.cam, device=R.local_rank, stream=R.stream.cuda_stream, rank=R.rank, world=R.world,
                                     image_ptr=self.tile_buf.data_ptr(), tile_buffer_tiles=self.tiles_per_rank)
                self.gathered = torch.zeros((R.world, self.tiles_per_rank, 32, 32, 4), dtype=torch.float32, device=R.dev) if R.rank == 0 else None
                self.full_tex = rt.Texture.wrap_device(self.r.ctx, self.full.data_ptr(), IW, IH, bind=False) if R.rank == 0 else None
            self.carry = torch.zeros((IH, IW, 16), dtype=torch.float32, device=R.dev) if passes > 1 else None
        self.my_pixels = self.r.shader.covered_pixels(dw, dh)
        if self.sharded:
            assert self.r.shader.owned_tiles(dw, dh)[0] <= self.tiles_per_rank
        self.trace_events, self.gather_events, self.assemble_events = [], [], []

    def step(self, timed, fresh):
        """One frame.  fresh: the scheduler's cost history is dropped first (a frame it has not seen)."""
        R, torch, r = self.R, self.R.torch, self.r
        ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
        with torch.cuda.stream(R.stream):
            if fresh:
                r.ctx.forget_costs()
            e0, e1 = ev(), ev()
            e0.record(R.stream)
            if self.carry is None:
                r.shader.dispatch_compute(self.dw, self.dh, 1)
            else:
                self.full.zero_(); self.carry.zero_()
                for k in range(self.passes):
                    r.shader.dispatch_accumulate(self.dw, self.dh, 1, k * self.spp, self.spp, self.carry.data_ptr())
                r.shader.dispatch_resolve(self.dw, self.dh, 1, self.spp * self.passes)
            e1.record(R.stream)
            if timed:
                self.trace_events.append((e0, e1))
            if self.sharded:
                e2 = ev()
                if R.backend == "nccl":
                    glist = list(self.gathered.unbind(0)) if R.rank == 0 else None
                    R.dist.gather(self.tile_buf, glist, dst=0)
                else:                                        # rehearsal path: gloo gathers host tensors
                    R.stream.synchronize()
                    host_buf = self.tile_buf.cpu()
                    hlist = [torch.empty_like(host_buf) for _ in range(R.world)] if R.rank == 0 else None
                    R.dist.gather(host_buf, hlist, dst=0)
                    if R.rank == 0:
                        self.gathered.copy_(torch.stack(hlist))
                e2.record(R.stream)
                if R.rank == 0:
                    e3 = ev()
                    r.shader.assemble_tiles(self.gathered.data_ptr(), R.world, self.tiles_per_rank, self.full_tex, self.dw, self.dh)
                    e3.record(R.stream)
                    if timed:
                        self.assemble_events.append((e2, e3))
                if timed:
                    self.gather_events.append((e1, e2))

    def run(self, steps, warmup, fresh):
        """(seconds per step, max over ranks; total written pixels over ranks)"""
        R = self.R
        self.trace_events, self.gather_events, self.assemble_events = [], [], []
        for _ in range(warmup):
            self.step(False, fresh)
        if not self.sharded:                                 # (an unsharded workload is this rank's own business: no barrier)
            R.torch.cuda.synchronize(R.dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(True, fresh)
            R.torch.cuda.synchronize(R.dev)
            return (time.perf_counter() - t0) / steps, float(self.my_pixels)
        R.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(True, fresh)
        R.fence()
        dt = time.perf_counter() - t0
        dt = R.reduce([dt], "max")[0]
        total_pixels = R.reduce([float(self.my_pixels)], "sum")[0]
        return dt / steps, total_pixels

    @staticmethod
    def mean_ms(pairs):
        return sum(a.elapsed_time(b) for a, b in pairs) / max(1, len(pairs))

    def close(self):
        self.r.close()


def measure_phases(wl, n=5):
    """[probe, main, resolve] ms of a history-free frame, from events recorded inside the library on the launch stream."""
    import numpy as np
    r = wl.r
    r.ctx.phase_timing(True)
    acc = np.zeros(3)
    for _ in range(n):
        wl.step(False, True)
        acc += np.array(r.ctx.phase_timing(True))
    r.ctx.phase_timing(False)
    return [float(v) for v in acc / n]


def library_probe_samples(spp):
    """Probe samples of a two-phase frame, by the rule tdt_ctx_create / dispatch_frame apply (csrc/tdt_rt.hip): max(spp / TDT_PROBE_DIV, 1),
    the divisor 16 unless the environment holds a value in [2, 64] — so that the counted sample range is the one whose launch is timed."""
    try:
        div = int(os.environ.get("TDT_PROBE_DIV", "16"))
    except ValueError:
        div = 16
    if not 2 <= div <= 64:
        div = 16
    return max(spp // div, 1)


def nominal_roofline(R, wl, cfg, phase, frame_kernel_ms):
    """The `roofline` object for a workload's dominant launch: algorithmic bytes (counted by the instrumented build, untimed) over
    its duration, against the HBM peak — nominal — with the physical (VALU) bound quoted beside it."""
    torch = R.torch
    r, spp, dw, dh, IW, IH = wl.r, wl.spp, wl.dw, wl.dh, wl.IW, wl.IH
    counts_frame = r.shader.dispatch_counted(dw, dh, 1)        # instrumented, untimed: algorithmic events of the whole frame
    read_bytes, write_bytes = algorithmic_bytes(counts_frame)
    kernel_ms, kernel_note, main_read = frame_kernel_ms, "whole dispatch (one launch)", read_bytes
    if phase is not None and phase[0] > 0:
        # a two-phase frame: the dominant launch is the main one (samples [spp/16, spp)); count ITS algorithmic events
        probe = library_probe_samples(spp)
        with torch.cuda.stream(R.stream):
            c = torch.zeros((IH, IW, 16), dtype=torch.float32, device=R.dev)
            wl.full.zero_()
        R.stream.synchronize()
        r.shader.dispatch_counted_range(dw, dh, 1, 0, probe, c.data_ptr())      # (instrumented too: keeps the product kernel's profile rows to whole launches)
        counts_main = r.shader.dispatch_counted_range(dw, dh, 1, probe, spp - probe, c.data_ptr())
        main_read, _ = algorithmic_bytes(counts_main)
        kernel_ms, kernel_note = phase[1], f"main launch of the two-phase frame: samples [{probe}, {spp}) of every pixel in the cost order of this frame's probe"
        del c
    achieved = main_read / (kernel_ms * 1e-3) / 1e9
    wkey = f"config{cfg}_spp{spp}_gpus{R.world}"
    traffic = (quoted("traffic.json", wkey) or {}).get("hbm_bytes_per_launch")
    valu, _ = quoted_valu(wkey)
    return {"bound": NOMINAL_BOUND, "bound_note": NOMINAL_NOTE,
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "frac_physical": (physical_roofline(wkey) or {}).get("frac"),      # beside the nominal figure: VALU issue x lanes, the bound the kernel runs against (see `physical`)
            "traffic": traffic, "traffic_note": "measured HBM bytes of the same launch (rocprofv3 TCC counters, profiles/traffic.json): << algorithmic",
            "physical": physical_roofline(wkey),
            "kernel": "tdt::trace_kernel<false, ...> (COUNT = false: the product build)", "kernel_launch": kernel_note,
            "kernel_ms": round(kernel_ms, 4), "algorithmic_read_bytes": int(main_read),
            "frame": {"dispatch_ms": round(frame_kernel_ms, 4), "algorithmic_read_bytes": int(read_bytes), "algorithmic_write_bytes": int(write_bytes),
                      "node_loads": counts_frame["node_loads"], "rays": counts_frame["octree_hit_calls"],
                      "frac_of_hbm_peak": round(read_bytes / (frame_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "phases_ms": {"probe": round(phase[0], 4), "main": round(phase[1], 4), "resolve": round(phase[2], 4)} if phase else None,
            "valu": valu}


def oracle_bands(wl, bands, band_rows, cores):
    """Rows [y0, y0 + band_rows) of `bands` bands of the workload's frame on the CPU oracle (the checker; a port, not the
    product): (seconds, pixels, mismatching pixels of the GPU frame in those rows)."""
    import numpy as np
    R, r = wl.R, wl.r
    r.ctx.forget_costs()
    r.shader.dispatch_compute(wl.dw, wl.dh, 1)
    R.stream.synchronize()
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    orc = oracle_py.Oracle()
    IW, IH = wl.IW, wl.IH
    covered_h = min(max(wl.dh // 32, 1) * 32, IH)
    ys = [int(i * (covered_h - band_rows) / max(1, bands - 1)) // 8 * 8 for i in range(bands)]
    img = np.zeros((IH, IW, 4), np.float32)
    t0 = time.perf_counter()
    px = 0
    for y0 in ys:
        orc.render(wl.scene, wl.cam, (wl.dw, wl.dh), rows=(y0, y0 + band_rows), threads=cores, image=img)
        px += IW * band_rows
    tc = time.perf_counter() - t0
    got = wl.full.cpu().numpy()
    bad = 0
    for y0 in ys:
        bad += int((got[y0:y0 + band_rows].view(np.uint32) != img[y0:y0 + band_rows].view(np.uint32)).any(axis=2).sum())
    return tc, px, bad


def target_block(R, args):
    """BASELINE configs[2] — 3840x2160 / 64 spp / max_bounce 16 / 256^3: the configuration north_star quotes its roofline target on."""
    wl = Workload(R, 3)
    sec, px = wl.run(args.target_steps, 2, True)
    frame_ms = Workload.mean_ms(wl.trace_events)
    phase = measure_phases(wl, 3)
    wl.step(False, False)
    rsec, _ = wl.run(args.target_steps, 1, False)
    roof = nominal_roofline(R, wl, 3, phase, frame_ms)
    out = {"workload": WORKLOADS[3][4], "image": [wl.IW, wl.IH], "dispatch": [wl.dw, wl.dh, 1], "spp": wl.spp, "max_bounce": wl.bounce,
           "written_pixels": int(px), "octree_max_depth": wl.scene.max_depth, "octree_cells": wl.scene.counts["cells"],
           "steps": args.target_steps, "history_free_ms": round(sec * 1e3, 3), "replay_ms": round(rsec * 1e3, 3),
           "value": round(px * wl.spp / sec / 1e6, 2), "unit": "Mray-samples/s",
           "schedule": "every timed step history-free (tdt_forget_costs before it), as the headline; replay_ms = the identical frame again",
           "roofline": roof, "target": "north_star: >= 40 % of the HBM-read roofline (nominal definition) on this configuration"}
    wl.close()
    return out


def reference_default_block(R, args):
    """What the reference's own main.rs renders every frame: its demo scene at its window size and camera.ron settings.
    cell_count = 100000 is not a power of two: the per-cell threshold builds (FORM_TABLE) trace it; the literal-formula kernel
    (TDT_NO_TABLE_FORM=1: what every such scene ran before round 3) is timed beside it on the same frame."""
    steps = args.reference_steps
    wl = Workload(R, 0)
    sec, px = wl.run(steps, 5, True)
    wl.step(False, False)
    rsec, _ = wl.run(steps, 1, False)
    cores = min(os.cpu_count() or 1, 32)
    tc, opx, bad = oracle_bands(wl, 8, 8, cores)
    spp = wl.spp
    out = {"workload": WORKLOADS[0][4], "image": [wl.IW, wl.IH], "dispatch": [wl.dw, wl.dh, 1], "spp": spp, "max_bounce": wl.bounce,
           "written_pixels": int(px), "octree_max_depth": wl.scene.max_depth, "cell_count": wl.scene.cell_count, "steps": steps,
           "history_free_ms": round(sec * 1e3, 4), "replay_ms": round(rsec * 1e3, 4), "value": round(px * spp / sec / 1e6, 2), "unit": "Mray-samples/s",
           "schedule": "spp < 16: a frame the scheduler has not seen is ONE launch in image order (no probe); replay_ms = the identical frame again in its recorded cost order",
           "oracle_check": {"bands": 8, "rows_per_band": 8, "pixels": opx, "mismatched_pixels": bad, "oracle_s": round(tc, 2), "cores": cores,
                            "oracle_value": round(opx * spp / tc / 1e6, 3)},
           "reference_llvmpipe": quoted("llvmpipe_baseline.json", f"config0_spp{spp}")}
    # ... and the way the reference really drives it (main.rs:486-601): a camera walk, every frame a dispatch with moved uniforms —
    # W held down at normal speed plus a mouse turn, 60 fps time step, camera.ron's rates.  Such frames reuse the previous frame's
    # costs by 8x8 tile (the per-pixel order of a still camera would be stale); no frame is forgotten, none is a replay.
    from tdt4230_project_raytracing_amd import host, rt
    walker = host.Camera(90.0, wl.IW, aspect_ratio=wl.IW / wl.IH, origin=(0.0, -0.1, -0.3), viewport_height=2.0, samples_per_pixel=spp,
                         max_bounce=wl.bounce, turn_rate=0.05, normal_speed=0.03, sprint_speed=0.15)
    torch = R.torch
    with torch.cuda.stream(R.stream):
        def walk_frames(n):
            for _ in range(n):
                walker.translate("Front", 1.0 / 60.0); walker.turn_yaw(0.2)
                rt.initial_uniforms(walker.uniforms(), wl.r.shader.program)
                wl.r.shader.dispatch_compute(wl.dw, wl.dh, 1)
        walk_frames(5)
        R.stream.synchronize()
        t0 = time.perf_counter()
        walk_frames(steps)
        R.stream.synchronize()
        walk_ms = (time.perf_counter() - t0) / steps * 1e3
    # stand still (the order-reuse state: from the third identical frame on the sort is skipped), then move: the first moved frame
    # must not fall back to image order (round 3's advice: the reuse state used to leave all-zero costs behind)
    with torch.cuda.stream(R.stream):
        ms_moved = []
        for _ in range(max(steps // 10, 3)):
            for _ in range(5):
                wl.r.shader.dispatch_compute(wl.dw, wl.dh, 1)
            R.stream.synchronize()
            t0 = time.perf_counter()
            walk_frames(1)
            R.stream.synchronize()
            ms_moved.append((time.perf_counter() - t0) * 1e3)
    out["still_then_move_ms"] = round(sorted(ms_moved)[len(ms_moved) // 2], 4)
    out["still_then_move_note"] = "median of the first MOVED frame after five frames of a still camera (which reuse their hand-out order): ordered by the tile sums of the still frames' costs"
    out["camera_walk_ms"] = round(walk_ms, 4)
    out["camera_walk_note"] = (f"{steps} consecutive frames of a walking, turning camera (main.rs's render loop: uniforms updated, then dispatch_compute), "
                               "host-side uniform updates included: the reference's interactive case")
    wl.close()
    os.environ["TDT_NO_TABLE_FORM"] = "1"                    # read when a context is created
    try:
        gl = Workload(R, 0)
        gsec, _ = gl.run(steps, 5, True)
        gl.step(False, False)
        grsec, _ = gl.run(steps, 1, False)
        gl.close()
    finally:
        del os.environ["TDT_NO_TABLE_FORM"]
    out["literal_kernel"] = {"history_free_ms": round(gsec * 1e3, 4), "replay_ms": round(grsec * 1e3, 4), "value": round(px * spp / gsec / 1e6, 2),
                             "note": "trace_kernel<false, FORM_LITERAL>: float treeLookup, no tables (TDT_NO_TABLE_FORM=1)"}
    out["speedup_vs_literal_kernel"] = round(gsec / sec, 3)
    return out


def main():
    args = parse_args()
    if args.single_process_leg:
        return single_process_leg(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    if args.rendezvous_only:
        return rendezvous_only()

    # the contract is ONE JSON line on stdout: libraries that chat on fd 1 (RCCL prints a host / library banner when a
    # communicator is created) are sent to stderr for the duration of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    R = Rank(args)
    world, rank = R.world, R.rank
    args.gpus = world
    # N > 1: the headline is STRONG scaling of the metric's own frame (1080p / 64 spp sharded over the ranks; N = 1 is that frame on one
    # GPU, so a scaling curve read from `value` starts at the single-GPU line); the weak-scaled figure (N x the pixel rows) and the 8K
    # frame of BASELINE configs[3] are side blocks (`weak`, `strong`).  --scaling weak restores round 3's headline.
    scaling = "strong" if args.config == 4 else (args.scaling or ("strong" if world > 1 else "weak"))
    rows_factor = world if (world > 1 and scaling == "weak") else 1
    if args.passes > 1 and R.sharded:
        raise SystemExit("--passes is a single-GPU option")
    if args.passes > 1:
        args.no_cpu_baseline = True

    wl = Workload(R, args.config, spp=args.spp, rows_factor=rows_factor, passes=args.passes)
    spp, IW, IH, dw, dh, r, scene, cam = wl.spp, wl.IW, wl.IH, wl.dw, wl.dh, wl.r, wl.scene, wl.cam
    progressive = args.passes > 1

    # the context's very first dispatch (includes the one-time LDS-table build and scan): reported, not the headline
    torch = R.torch
    with torch.cuda.stream(R.stream):
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record(R.stream)
        if not progressive:
            r.shader.dispatch_compute(dw, dh, 1)
        f1.record(R.stream)
    R.fence()
    first_dispatch_ms = f0.elapsed_time(f1) if not progressive else None

    # ---- the headline: frames the scheduler has not seen (unless --replay) ---------------------------------------------
    fresh = not args.replay and not progressive
    sec, total_pixels = wl.run(args.steps, args.warmup, fresh)
    ms_per_step = sec * 1e3
    samples_per_step = total_pixels * spp * args.passes
    value = samples_per_step / sec / 1e6
    frame_kernel_ms = Workload.mean_ms(wl.trace_events)
    gather_ms = Workload.mean_ms(wl.gather_events) if R.sharded else None
    assemble_ms = Workload.mean_ms(wl.assemble_events) if (R.sharded and rank == 0) else None
    rank_trace_ms = R.gather_floats(frame_kernel_ms)

    # ---- N > 1, strong headline: the same frame on ONE GPU in this run (rank 0 alone), and the weak-scaled figure as a side block ----
    headline_scaling = weak = None
    if world > 1 and scaling == "strong" and args.config != 4 and not progressive:
        one_sec = None
        if rank == 0:
            ow = Workload(R, args.config, spp=args.spp, alone=True)
            one_sec, one_px = ow.run(max(args.steps // 2, 1), 1, fresh)
            ow.close()
        R.fence()
        if rank == 0:
            headline_scaling = headline_summary(value, one_px * spp / one_sec / 1e6, world)
            headline_scaling["one_gpu_source"] = "rank 0 alone, same run, same schedule"
        if not args.no_weak:
            ww = Workload(R, args.config, spp=args.spp, rows_factor=world)
            wsec, wpx = ww.run(max(args.steps // 2, 1), 1, fresh)
            weak = {"workload": ww.desc + f"; x{world} pixel rows over the same frustum", "image": [ww.IW, ww.IH], "written_pixels": int(wpx), "ms_per_step": round(wsec * 1e3, 4),
                    "value": round(wpx * spp / wsec / 1e6, 2), "unit": "Mray-samples/s", "scaling": "weak",
                    "note": "every rank keeps one N = 1 frame of work (round 3's headline at N > 1)"}
            ww.close()

    # phases of a history-free frame (probe / main / resolve) from events recorded inside the library, a few more frames
    phase = measure_phases(wl, 5) if (fresh and rank == 0 and not R.sharded) else None

    # ---- the other schedule, for the record -----------------------------------------------------------------------------
    other_ms = None
    if not progressive:
        wl.step(False, False)                                # (a replay needs a recorded frame)
        other_sec, _ = wl.run(args.steps, 1, not fresh)
        other_ms = other_sec * 1e3

    # ---- roofline of the dominant kernel: the main launch of the frame, measured on rank 0 ------------------------------
    roofline = nominal_roofline(R, wl, args.config, phase, frame_kernel_ms) if rank == 0 else None

    # ---- CPU baseline: the oracle (a port, not the product) on a bounded sample; the reference's own llvmpipe figure quoted ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cores = min(os.cpu_count() or 1, 32)
        bands, band_rows = 96, 8
        tc, px, bad = oracle_bands(wl, bands, band_rows, cores)
        cpu_baseline = {"value": round(px * spp / tc / 1e6, 3), "unit": "Mray-samples/s", "cores": cores, "kind": "port",
                        "sample": f"{bands} bands of {band_rows} rows ({px} px x {spp} spp) of the same frame, {tc:.1f} s",
                        # the product spot-checked against the checker on the sampled rows (never the other way round)
                        "mismatched_pixels_in_sample": bad,
                        "reference_llvmpipe": quoted("llvmpipe_baseline.json", f"config{args.config}_spp{spp}")}
    wl.close()

    # ---- BASELINE configs[2] (the north-star roofline configuration) and the reference's own workload: N = 1 lines only ------
    target_4k = reference_default = None
    if world == 1 and not R.sharded and not progressive and args.config == 2:
        if not args.no_target:
            target_4k = target_block(R, args)
        if not args.no_reference_default:
            reference_default = reference_default_block(R, args)

    # ---- BASELINE configs[3]: ONE 8K frame, strong-sharded over the ranks -------------------------------------------------
    strong = None
    if not args.no_strong and args.config != 4 and not progressive:
        sw = Workload(R, 4, rows_factor=1)
        ssec, spx = sw.run(args.strong_steps, 1, True)
        strace = R.gather_floats(Workload.mean_ms(sw.trace_events))
        strong = {"workload": WORKLOADS[4][4], "image": [sw.IW, sw.IH], "spp": sw.spp, "written_pixels": int(spx),
                  "steps": args.strong_steps, "ms_per_frame": round(ssec * 1e3, 3), "value": round(spx * sw.spp / ssec / 1e6, 2),
                  "unit": "Mray-samples/s", "scaling": "strong", "trace_ms_per_rank": [round(v, 3) for v in strace],
                  "gather_ms": round(Workload.mean_ms(sw.gather_events), 3) if R.sharded else 0.0,
                  "gather_note": "from the end of rank 0's trace to the end of the gather on rank 0: includes waiting for the slowest rank",
                  "assemble_ms": round(Workload.mean_ms(sw.assemble_events), 3) if (R.sharded and rank == 0) else 0.0,
                  "tile_buffer_bytes_per_rank": int(sw.tiles_per_rank) * 32 * 32 * 16,
                  "schedule": "every frame history-free (tdt_forget_costs): two-phase on each rank"}
        sw.close()
        # the same frame on ONE GPU, in this run: rank 0 alone (the others wait at the fence below); at N = 1 it is the measurement above
        one_ms, one_src = ssec * 1e3, "this measurement (one rank)"
        if world > 1 and rank == 0:
            ow = Workload(R, 4, alone=True)
            osec, _ = ow.run(1, 1, True)
            ow.close()
            one_ms, one_src = osec * 1e3, "rank 0 alone, same run: 1 warm-up + 1 timed history-free frame"
        strong.update(strong_summary(ssec * 1e3, strace, one_ms, one_src, R.backend if R.sharded else "none (one rank: no collective)",
                                     R.dist.get_world_size() if R.sharded else 1))

    R.fence()
    R.close()
    if rank != 0:
        return

    # ---- the same 8K frame through the C ABI's multi-device context, one process (a fresh one) ------------------------------
    single = None
    if not args.no_single_process and not progressive:
        torch.cuda.synchronize()
        single = run_single_process_child(args, world)

    desc = wl.desc
    out = {
        "metric": "Mray-samples/sec at 1080p/64spp/depth-8" if args.config == 2 and spp == 64 and args.passes == 1 else f"Mray-samples/sec (config {args.config}, {spp * args.passes} spp" + (f" as {args.passes} progressive passes" if args.passes > 1 else "") + ")",
        "value": round(value, 2), "unit": "Mray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": scaling if world > 1 else "weak",      # (N = 1: nothing is sharded; the contract's default label)
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc + (f"; x{rows_factor} pixel rows over the same frustum (weak scaling)" if rows_factor > 1 else "") +
                               (f"; ONE frame sharded over {world} ranks (strong scaling)" if (world > 1 and rows_factor == 1) else ""),
                   "image": [IW, IH], "dispatch": [dw, dh, 1], "written_pixels": int(total_pixels), "spp": spp * args.passes,
                   "passes": args.passes,
                   "max_bounce": wl.bounce, "octree_max_depth": scene.max_depth, "octree_cells": scene.counts["cells"],
                   "scene_bytes": scene.nbytes(),
                   "schedule": ("every timed step is a frame the scheduler has not seen (tdt_forget_costs before it): spp/16 probe samples per pixel in "
                                "image order, then the rest most-expensive-pixel-first from the probe's work counts, then the resolve"
                                if fresh else "replay: every timed step repeats the frame before it, pixels handed out most-expensive-first from "
                                "the work counts that frame recorded") if not progressive else "progressive passes (running sums + carry through HBM)",
                   "replay_ms": round(other_ms if fresh else ms_per_step, 4) if other_ms is not None else None,
                   "history_free_ms": round(ms_per_step if fresh else other_ms, 4) if other_ms is not None else None,
                   "first_dispatch_ms": round(first_dispatch_ms, 4) if first_dispatch_ms is not None else None,
                   "first_dispatch_note": "the context's very first frame: includes the one-time LDS-table build and scan of the cells buffer",
                   "trace_ms_per_rank": [round(v, 4) for v in rank_trace_ms],
                   "gather_ms": round(gather_ms, 4) if gather_ms is not None else None,
                   "assemble_ms": round(assemble_ms, 4) if assemble_ms is not None else None,
                   "partition": f"32x32 work-groups dealt round-robin over {world} rank(s)" + ("; one RCCL gather + de-interleave per step" if R.sharded else "")},
        "headline_scaling": headline_scaling,
        "weak": weak,
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
        "target_4k": target_4k,
        "reference_default": reference_default,
        "strong": strong,
        "single_process": single,
    }
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)


if __name__ == "__main__":
    main()
