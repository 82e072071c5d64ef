#!/usr/bin/env python3
"""Registers, scratch, occupancy and LDS of every kernel of csrc/tdt_rt.hip as the compiler reports them
(`-Rpass-analysis=kernel-resource-usage`, the build's own flags; cross-compiles without a GPU, ~25 s).

    python tools/kernel_resources.py [out.txt]      # default: profiles/r04_kernel_resources.txt

tests/test_kernel_resources.py asserts on the same rows: no trace kernel may spill to scratch, drop below 4 waves per SIMD or
outgrow the CU's 160 KiB of LDS — a feature that costs registers has to fail a test, not silently lose 15 %."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tdt4230_project_raytracing_amd import build as b   # noqa: E402

FIELDS = ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]")
CXXFILT = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"


def collect(unit="tdt_rt.hip"):
    """[{name, mangled, TotalSGPRs, VGPRs, ...}] for every kernel of one translation unit."""
    with tempfile.TemporaryDirectory() as tmp:
        flags = [f for f in b.HIP_FLAGS if f != "-shared"]
        cmd = [b.HIPCC] + flags + ["-I", b.INCLUDE, "-I", b.CSRC, "-c", os.path.join(b.CSRC, unit), "-o", os.path.join(tmp, "o.o"),
               "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed:\n" + r.stderr[-4000:])
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"mangled": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]+\])?): (\S+) \[-Rpass-analysis", line)
        if m and cur is not None:
            key, val = m.group(1).strip(), m.group(2)
            if key in FIELDS:
                cur[key] = int(val)
    names = subprocess.run([CXXFILT] + [row["mangled"] for row in rows], capture_output=True, text=True).stdout.splitlines() if os.path.exists(CXXFILT) else []
    for i, row in enumerate(rows):
        row["name"] = re.sub(r"\(TraceParams\)$", "", names[i].replace("void ", "", 1)) if i < len(names) else row["mangled"]
    return rows


def table(rows):
    out = ["%-92s %5s %5s %7s %5s %8s %7s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS B", "spill s/v")]
    for r in rows:
        out.append("%-92s %5d %5d %7d %5d %8d %4d/%d" % (r["name"][:92], r["VGPRs"], r["TotalSGPRs"], r["ScratchSize [bytes/lane]"], r["Occupancy [waves/SIMD]"],
                                                        r["LDS Size [bytes/block]"], r["SGPRs Spill"], r["VGPRs Spill"]))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_kernel_resources.txt")
    rows = collect()
    head = ("# hipcc -Rpass-analysis=kernel-resource-usage over csrc/tdt_rt.hip with the build's flags (tools/kernel_resources.py).\n"
            "# trace_kernel<COUNT, FORM (0 literal / 1 pow2 / 2 table), DEPTH, RESIDENT, SAFEV, FULL, UNIT, BRICK>; 1024-thread blocks: 128 VGPRs = 4 waves/SIMD is the cap.\n")
    open(dst, "w").write(head + table(rows))
    print(table(rows))
