#!/usr/bin/env python3
"""Where the SIMD lane-cycles of a trace launch go -> profiles/r04_loss_budget.json.

Three sources, one frame each (the main launch of a history-free frame, or a replay):
  * pass statistics of the PRODUCT kernels from a -DTDT_STATS build of the library (`collect`, on the GPU box, run with
    TDT_LIB=build_ab/lib_stats.so TDT_STATS_SKIP_PROBE=1): traversal / event passes per wave and live lanes per code region;
  * the SQ counters of the same frame under the product library (tools/profile_r0x.sh -> profiles/r0x_pmc_summary.json);
  * static VALU counts of the event regions (tools/isa_regions.py -> profiles/r04_isa_regions.json).
`report` combines them:  VALU wave-instructions of the event regions = passes x size; the traversal step's = SQ_INSTS_VALU minus
those (per loop pass: cross-checked against the hand count of the fast path); active lane-instructions likewise from
SQ_THREAD_CYCLES_VALU.  The budget is in units of SIMD lane-slots: 1024 SIMDs x launch time x clock / 2 cycles x 64 lanes.

    TDT_LIB=build_ab/lib_stats.so TDT_STATS_SKIP_PROBE=1 python tools/loss_budget.py collect --config 2 --mode fresh > stats.json
    python tools/loss_budget.py report --stats-dir gpurun_out/r04/ --pmc profiles/r04_pmc_summary.json
"""
import argparse, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EVENT_REGIONS = (("hit_prologue", "hit"), ("lambert", "lamb"), ("metal", "metal"), ("dielectric", "diel"), ("hit_epilogue", "hit"), ("end_of_path", "end"),
                 ("pixel_end", "pixel_end"), ("fetch", "fetch"), ("primary", "primary"), ("threshold", "event"), ("newray", "newray"))
ISA_KEY = {2: "config2_64cube_full", 3: "config3_256cube_brick8", 5: "config5_512cube_brick9", 0: "config0_demo_table10"}


def collect(a):
    import bench
    from tdt4230_project_raytracing_amd import host, rt
    W, H, spp, bounce, desc, scene_cfg = bench.WORKLOADS[a.config]
    scene = host.Scene.config(scene_cfg)
    cam = host.camera_reference_pose(W, H, spp, bounce)
    r = rt.Renderer(scene, cam)
    r.ctx.stats()                                   # switches the collection on
    for _ in range(2):
        r.dispatch()
    if a.mode == "fresh":
        r.ctx.forget_costs()
    r.ctx.finish()
    r.ctx.stats(reset=True)
    r.dispatch()
    r.ctx.finish()
    st = r.ctx.stats(reset=True)
    ends = sorted(e - st["t_first"] for e in st.pop("wave_ends") if e)
    span = max(st["t_last"] - st["t_first"], 1)
    pct = lambda f: ends[min(len(ends) - 1, int(f * len(ends)))] / span if ends else None
    st["wave_end_share_of_span_p1_p5_p25_p50_p75_p95"] = [round(pct(f), 4) for f in (0.01, 0.05, 0.25, 0.5, 0.75, 0.95)] if ends else None
    st["queue_dry_share_of_span"] = round((st["t_queue_dry"] - st["t_first"]) / span, 4) if st["t_queue_dry"] < (1 << 63) else None
    st["span_ms"] = span / 1e5
    variant = r.ctx.last_variant()
    r.close()
    print(json.dumps({"config": a.config, "mode": a.mode, "workload": desc, "variant": variant, "stats": st,
                      "note": "main launch of a history-free frame" if a.mode == "fresh" and spp >= 16 else ("replay of an identical frame" if a.mode == "replay" else "the one launch of a history-free frame")}))


def report(a):
    isa = json.load(open(a.isa))["kernels"]
    pmc = json.load(open(a.pmc))
    out = {"units": "share of the launch's SIMD lane-slots (1024 SIMDs x kernel time x clock / 2 cycles per wave64 VALU instruction x 64 lanes)",
           "sources": {"pass_statistics": "tools/loss_budget.py collect (-DTDT_STATS build)", "sq_counters": os.path.basename(a.pmc), "region_sizes": os.path.basename(a.isa)},
           "frames": {}}
    for f in sorted(glob.glob(os.path.join(a.stats_dir, "stats_c*_*.json"))):
        rec = json.loads(open(f).read().strip().splitlines()[-1])
        c, mode, st = rec["config"], rec["mode"], rec["stats"]
        key = "config%d_spp%d_gpus1%s" % (c, 4 if c == 0 else 64, "_replay" if mode == "replay" else "")
        if key not in pmc:
            continue
        q = pmc[key]; cnt = q["counters"]
        reg = isa[ISA_KEY[c]]["regions"]
        t = q["kernel_ms_under_pmc"] * 1e-3
        slots = 1024 * t * a.clock_ghz * 1e9 / 2          # wave-instruction issue slots of the launch
        valu = cnt["SQ_INSTS_VALU"]
        lanes_per_inst = cnt["SQ_THREAD_CYCLES_VALU"] / cnt["SQ_ACTIVE_INST_VALU"]          # active lanes per VALU instruction (of 64)
        active = valu * lanes_per_inst
        ev_inst = ev_act = 0.0
        rows = {}
        for name, stat in EVENT_REGIONS:
            size = reg["newray" if name == "newray" and c == 0 else name]["valu"]
            if name == "newray" and c == 0:
                size = isa[ISA_KEY[2]]["regions"]["newray"]["valu"]      # (the demo kernel's rotated loop counts its cold blocks there)
            passes, lanes = st[stat + "_pass"], st[stat + "_lanes"]
            rows[name] = {"valu_per_pass": size, "passes": passes, "lanes_per_pass": round(lanes / max(passes, 1), 2)}
            ev_inst += passes * size; ev_act += lanes * size
        gate = reg["gate"]["valu"] * st["loop_pass"]
        trav_inst = valu - ev_inst - gate
        trav_act = active - ev_act - gate * 64.0
        alive = st["wave_ticks"] / max(st["waves"], 1) / max(st["t_last"] - st["t_first"], 1)
        total = slots * 64.0
        b = {
            "traversal_useful_lanes": trav_act / total,
            "traversal_idle_lanes": (trav_inst * 64.0 - trav_act) / total,       # parked at the gate, outside the octree this step, or retired
            "event_useful_lanes": ev_act / total,
            "event_idle_lanes": (ev_inst * 64.0 - ev_act) / total,               # lanes not in the state (or of the material) a region serves
            "gate": gate * 64.0 / total,
        }
        wc = cnt.get("SQ_WAVE_CYCLES")
        extra = {}
        if wc:
            # per wave, of its lifetime (quad-cycles): issuing anything, parked at s_waitcnt / barrier, ready but not issued
            extra = {"per_wave_issuing": cnt["SQ_ACTIVE_INST_ANY"] / wc, "per_wave_waitcnt": cnt["SQ_WAIT_ANY"] / wc, "per_wave_issue_stall": cnt["SQ_WAIT_INST_ANY"] / wc,
                     "per_wave_issuing_valu": cnt["SQ_ACTIVE_INST_VALU"] / wc,
                     "waves_alive_share_of_launch": wc * 4 / (cnt["SQ_WAVES"] * t * a.clock_ghz * 1e9)}
            dead = 1.0 - extra["waves_alive_share_of_launch"]
            b["no_valu_issued_dead_waves"] = dead                                # wave slots whose wave has ended (the launch's tail) or not started
            b["no_valu_issued_alive"] = 1.0 - valu / slots - dead                # waves alive: s_waitcnt on the table / LDS loads, scalar + branch issue, dependency and arbitration stalls
        else:
            b["no_valu_issued"] = 1.0 - valu / slots
        out["frames"][key] = {
            "workload": rec["workload"], "variant": rec["variant"], "kernel_ms": q["kernel_ms_under_pmc"], "issue_util": q["issue_util"], "lane_util": q["lane_util"],
            "budget": {k: round(v, 4) for k, v in b.items()},
            "traversal": {"loop_passes": st["loop_pass"], "passes_with_a_traversing_lane": st["trav_pass"], "valu_per_loop_pass": round(trav_inst / max(st["loop_pass"], 1), 1),
                          "traversing_lanes_per_pass": round(st["trav_lanes"] / max(st["trav_pass"], 1), 2), "lanes_in_the_lookup_per_pass": round(st["inside_lanes"] / max(st["trav_pass"], 1), 2),
                          "lanes_parked_at_the_gate_per_pass": round(st["gate_wait_lanes"] / max(st["loop_pass"], 1), 2), "lanes_alive_per_pass": round(st["alive_lanes"] / max(st["loop_pass"], 1), 2),
                          "lane_util": round(trav_act / max(trav_inst * 64.0, 1), 3)},
            "events": {"passes": st["event_pass"], "lanes_served_per_pass": round(st["event_lanes"] / max(st["event_pass"], 1), 2), "valu_per_event_pass": round(ev_inst / max(st["event_pass"], 1), 1),
                       "lane_util": round(ev_act / max(ev_inst * 64.0, 1), 3), "share_of_valu_instructions": round(ev_inst / valu, 3), "regions": rows},
            "after_the_queue_ran_dry": {"traversal_passes": st["drained_trav_pass"], "event_passes": st["drained_event_pass"],
                                        "share_of_loop_passes": round(st["drained_trav_pass"] / max(st["loop_pass"], 1), 3)},
            "waves": {"alive_share_of_launch_stats_build": round(alive, 3), "queue_dry_share_of_span": st.get("queue_dry_share_of_span"),
                      "wave_end_share_of_span_p1_p5_p25_p50_p75_p95": st.get("wave_end_share_of_span_p1_p5_p25_p50_p75_p95"), **{k: round(v, 3) for k, v in extra.items()}},
        }
    json.dump(out, open(a.out, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    c = sub.add_parser("collect"); c.add_argument("--config", type=int, default=2); c.add_argument("--mode", default="fresh", choices=("fresh", "replay"))
    r = sub.add_parser("report"); r.add_argument("--stats-dir", required=True); r.add_argument("--pmc", default=os.path.join(ROOT, "profiles", "r04_pmc_summary.json"))
    r.add_argument("--isa", default=os.path.join(ROOT, "profiles", "r04_isa_regions.json")); r.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_loss_budget.json"))
    r.add_argument("--clock-ghz", type=float, default=2.4)
    a = ap.parse_args()
    collect(a) if a.cmd == "collect" else report(a)
