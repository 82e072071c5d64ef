#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files for the trace kernel: the LAST dispatch of the product trace
kernel of each run (tools/pmc_frame.py makes that the frame of interest), its duration, and the derived VALU figures:

  issue_util = SQ_INSTS_VALU x 2 cycles (a wave64 instruction on a SIMD32 datapath) / (CUs x 4 SIMDs x clock x duration)
  lane_util  = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64): active lanes per VALU instruction issued

usage: pmc_summary.py [--json out.json --key KEY --note TEXT] dir [dir ...]"""
import argparse, csv, glob, json, os, re, sys
ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--json"); ap.add_argument("--key"); ap.add_argument("--note", default="")
ap.add_argument("--cus", type=int, default=256); ap.add_argument("--clock-ghz", type=float, default=2.4)
a = ap.parse_args()
out, dur = {}, []
for d in a.dirs:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            m = re.search(r"trace_kernel<([^>]*)>", k)
            targs = [t.strip() for t in m.group(1).split(",")] if m else []
            if targs and targs[0] == "false":     # the product build (COUNT = false); the LAST such dispatch is the frame's main launch (a probe launch precedes it)
                acc.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, v in acc.items():
            last = max(i for i, _, _ in v)
            out[k] = sum(x for i, x, _ in v if i == last)
            dur.append([t for i, _, t in v if i == last][0])
g = lambda k: out.get(k, float("nan"))
res = {"counters": out}
if dur:
    res["kernel_ms_under_pmc"] = round(sum(dur) / len(dur) / 1e6, 4)
if "SQ_INSTS_VALU" in out and dur:
    t = sum(dur) / len(dur) * 1e-9
    res["issue_util"] = round(g("SQ_INSTS_VALU") * 2 / (a.cus * 4 * a.clock_ghz * 1e9 * t), 4)
    res["lane_util"] = round(g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64), 4) if g("SQ_ACTIVE_INST_VALU") else None
    res["valu_wave_instructions"] = g("SQ_INSTS_VALU"); res["salu_instructions"] = g("SQ_INSTS_SALU")
    res["formula"] = "issue_util = SQ_INSTS_VALU * 2 cycles / (%d CUs * 4 SIMDs * %.1f GHz * kernel time); lane_util = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)" % (a.cus, a.clock_ghz)
if a.note:
    res["note"] = a.note
print(json.dumps(res, indent=1))
if a.json and a.key:
    table = json.load(open(a.json)) if os.path.exists(a.json) else {}
    table[a.key] = res
    json.dump(table, open(a.json, "w"), indent=1, sort_keys=True)
