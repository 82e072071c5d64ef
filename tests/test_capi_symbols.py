"""The C-ABI libraries load on a machine without a GPU, export every symbol their headers declare,
and the product fails LOUDLY (no CPU fallback) when no HIP device is present."""
import ctypes
import os
import re

import pytest

from tdt4230_project_raytracing_amd import host, rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tdt_[a-z0-9_]+)\s*\(", text)))


def test_rt_library_exports_every_declared_symbol():
    L = ctypes.CDLL(rt.LIB_PATH)
    names = declared("tdt_rt.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"libtdtrt.so does not export {n}"
    assert sorted(n for n, _, _ in rt.SYMBOLS) == names      # the Python binding covers the whole ABI


def test_host_library_exports_every_declared_symbol():
    L = host.lib()
    for n in declared("tdt_host.h"):
        assert hasattr(L, n), f"libtdthost.so does not export {n}"


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.TdtError) as e:
        rt.Context(0)
    assert e.value.code == rt.ERR_NO_DEVICE


def test_strerror_matches_reference_messages():
    L = rt.lib()
    assert L.tdt_strerror(rt.ERR_INVALID_ENUM) == b"gl error: invalid enum"          # renderer/mod.rs:47
    assert L.tdt_strerror(rt.ERR_INVALID_VALUE) == b"gl error: invalid value"        # :48
    assert L.tdt_strerror(rt.ERR_INVALID_OPERATION) == b"gl error: invalid operation"  # :49
    assert L.tdt_strerror(rt.ERR_VARIABLE_NOT_FOUND).startswith(b"failed to locate uniform")  # :53
