#!/usr/bin/env python3
"""Static instruction counts of the trace kernel's code regions (CPU only: hipcc cross-compiles).

Compiles csrc/tdt_rt.hip with -DTDT_MARKERS (assembler comments at the region boundaries, see TDT_MARK in trace_device.hpp), cuts
the compiler's assembly of the builds the bench frames run at those comments and counts VALU / transcendental / SALU / LDS / VMEM
instructions per region.  Blocks the compiler laid out after the main loop (the `__builtin_expect`-unlikely paths: band fallbacks, the
IEEE forms of rcp / sqrt outside the exponent window, the literal normal) are reported as `cold`.  The marker build is not the
product build (an `asm volatile` is a scheduling boundary), but the counts per region are the source's, to a few instructions.

    python tools/isa_regions.py [out.json]         # default: profiles/r04_isa_regions.json
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tdt4230_project_raytracing_amd import build as b   # noqa: E402

# the builds the bench frames run (tests/test_gpu_variants.py pins them): <COUNT, FORM, DEPTH, RESIDENT, SAFEV, FULL, UNIT, BRICK>
KERNELS = {
    "config2_64cube_full": "_ZN3tdt12trace_kernelILb0ELi1ELi6ELb1ELb1ELb1ELb1ELb0EEEv11TraceParams",
    "config3_256cube_brick8": "_ZN3tdt12trace_kernelILb0ELi1ELi8ELb0ELb1ELb0ELb1ELb1EEEv11TraceParams",
    "config5_512cube_brick9": "_ZN3tdt12trace_kernelILb0ELi1ELi9ELb0ELb1ELb0ELb1ELb1EEEv11TraceParams",
    "config0_demo_table10": "_ZN3tdt12trace_kernelILb0ELi2ELi10ELb1ELb1ELb0ELb1ELb0EEEv11TraceParams",
}
TRANS = re.compile(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_")


def regions(asm, symbol):
    out, cur, on = {}, "prologue", False
    for line in asm.splitlines():
        if line.startswith(symbol + ":"):
            on = True
            continue
        if not on:
            continue
        t = line.strip()
        m = re.match(r";\s*TDT_MARK (\w+)", t)
        if m:
            cur = {"after_loop": "epilogue", "loop_tail": "cold"}.get(m.group(1), m.group(1))
            continue
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        r = out.setdefault(cur, dict(valu=0, trans=0, salu=0, lds=0, vmem=0))
        if op.startswith("v_"):
            r["valu"] += 1
            if TRANS.match(op):
                r["trans"] += 1
        elif op.startswith("s_"):
            r["salu"] += 1
            if op == "s_endpgm":
                break
        elif op.startswith("ds_"):
            r["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            r["vmem"] += 1
    return out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_isa_regions.json")
    with tempfile.TemporaryDirectory() as tmp:
        s_path = os.path.join(tmp, "k.s")
        flags = [f for f in b.HIP_FLAGS if f not in ("-shared", "-fPIC")]
        cmd = [b.HIPCC] + flags + ["-DTDT_MARKERS", "-I", b.INCLUDE, "-I", b.CSRC, "--cuda-device-only", "-S", os.path.join(b.CSRC, "tdt_rt.hip"), "-o", s_path]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit("compile failed:\n" + r.stderr[-3000:])
        asm = open(s_path).read()
    res = {"note": "static instruction counts per code region of the -DTDT_MARKERS build (tools/isa_regions.py); `cold` = blocks laid out at the end of the loop (unlikely paths; in the demo kernel, whose loop the compiler rotated, they follow `newray` and are counted there); `walk` = treeLookup level by level",
           "kernels": {}}
    for name, sym in KERNELS.items():
        reg = regions(asm, sym)
        if not reg:
            raise SystemExit("kernel not found in the assembly: " + sym)
        res["kernels"][name] = {"symbol": sym, "regions": reg}
        print(name)
        for k, v in reg.items():
            print("   %-14s VALU %4d (trans %2d)  SALU %4d  LDS %3d  VMEM %3d" % (k, v["valu"], v["trans"], v["salu"], v["lds"], v["vmem"]))
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("->", out_path)


if __name__ == "__main__":
    main()
