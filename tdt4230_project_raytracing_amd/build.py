"""Build recipes for the two in-tree shared libraries (explicit compiler invocations, outputs
stay in-tree so they travel to the GPU box with the snapshot)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# -ffp-contract=off and IEEE divide/sqrt are PARITY flags, not tuning knobs: the reference's
# arithmetic is unfused and correctly rounded (DESIGN.md §4 "Parity flags").
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wextra",
             # measured: SLP-packed v_pk_*_f32 costs 6 % here (register-pair shuffles around scalar vec3 code)
             "-fno-slp-vectorize"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-ffp-contract=off"]


def _newer(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return r.stdout + r.stderr


def build_host(force=False):
    out = os.path.join(_HERE, "libtdthost.so")
    srcs = [os.path.join(CSRC, "host_scene.cpp"), os.path.join(CSRC, "host_view.cpp"), os.path.join(INCLUDE, "tdt_host.h"),
            os.path.abspath(__file__)]
    if force or _newer(out, srcs):
        _run(["g++"] + HOST_FLAGS + ["-I", INCLUDE, srcs[0], srcs[1], "-o", out, "-lz"])      # zlib: the PNG writer (§8f-4)
    return out


DEVICE_UNITS = ("tdt_rt.hip", "tdt_multi.hip", "tdt_build.hip", "tdt_edit.hip")
DEVICE_HEADERS = ("trace_device.hpp", "trace_params.h", "tdt_internal.hpp", "device_scan.hpp")


def build_device(force=False, extra_flags=(), out=None):
    """libtdtrt.so: one object per translation unit (compiled side by side: the trace kernels take ~15 s, the rest
    seconds), then one link.  Objects live in csrc/obj/ (git- and gpurun-ignored); only the .so travels."""
    out = out or os.path.join(_HERE, "libtdtrt.so")
    hdrs = [os.path.join(CSRC, f) for f in DEVICE_HEADERS] + [os.path.join(INCLUDE, "tdt_rt.h"), os.path.abspath(__file__)]
    srcs = [os.path.join(CSRC, f) for f in DEVICE_UNITS]
    if not (force or extra_flags or _newer(out, srcs + hdrs)):
        return out
    objdir = os.path.join(CSRC, "obj" + ("_" + str(abs(hash(tuple(extra_flags))) % 100000) if extra_flags else ""))
    os.makedirs(objdir, exist_ok=True)
    compile_flags = [f for f in HIP_FLAGS if f != "-shared"] + list(extra_flags)
    jobs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or extra_flags or _newer(obj, [src] + hdrs):
            jobs.append((subprocess.Popen([HIPCC] + compile_flags + ["-I", INCLUDE, "-I", CSRC, "-c", src, "-o", obj],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), src))
    for p, src in jobs:
        log, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("build failed: " + src + "\n" + log)
    objs = [os.path.join(objdir, os.path.basename(src) + ".o") for src in srcs]
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out, "-ldl"])
    return out


def build_demo(force=False):
    """tdt_demo: the reference's main.rs, headless, written against include/renderer.hpp (C++ host mirror)."""
    out = os.path.join(_HERE, "tdt_demo")
    srcs = [os.path.join(CSRC, "demo_main.cpp"), os.path.join(INCLUDE, "renderer.hpp"), os.path.join(INCLUDE, "tdt_rt.h"),
            os.path.join(INCLUDE, "tdt_host.h"), os.path.abspath(__file__)]
    if force or _newer(out, srcs):
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(HIPCC))), "lib")
        _run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-I", INCLUDE, srcs[0], "-o", out, "-L", _HERE, "-ltdtrt", "-ltdthost",
              "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + rocm_lib])
    return out


def build_all(force=False):
    return build_host(force), build_device(force), build_demo(force)
