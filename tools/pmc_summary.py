#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files for the trace kernel: the LAST dispatch of the run (the
steady state: the first dispatch of a context runs in image order, the later ones in cost-feedback order)."""
import csv, glob, json, sys
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "trace_kernel<false" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for k, v in acc.items():
            last = max(i for i, _ in v)
            out[k] = sum(x for i, x in v if i == last)
d = out
def g(k): return d.get(k, float("nan"))
print(json.dumps(out, indent=1))
if "SQ_INSTS_VALU" in d:
    print("VALU wave-instr %.3g  SALU %.3g  lane util %.1f%%" % (g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), 100 * g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64 / 4 * 4) if g("SQ_ACTIVE_INST_VALU") else 0))
