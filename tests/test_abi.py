"""C-ABI checks that need no GPU: the shared library loads, exports every function include/avsep.h
declares, and the Python binding lists exactly the same set."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from av_separation import _native


def header_functions():
    text = open(os.path.join(ROOT, "include", "avsep.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(avsep_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_what_binding_lists():
    assert header_functions() == sorted(_native.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_native.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in header_functions():
        assert hasattr(lib, name), name


def test_version_and_error_string_without_gpu():
    lib = _native.load()
    assert lib.avsep_abi_version() == 1
    # argument validation happens before any HIP call
    assert lib.avsep_create(None, None) == -1
    assert b"null" in lib.avsep_last_error()
    cfg = _native.AvsepConfig(257, 250, 4, 2, 2, 2)          # d_model not a multiple of 32
    ctx = ctypes.c_void_p()
    assert lib.avsep_create(ctypes.byref(cfg), ctypes.byref(ctx)) == -1
    assert b"multiple of 32" in lib.avsep_last_error()
    cfg = _native.AvsepConfig(257, 256, 3, 2, 2, 2)          # d % nhead != 0, like nn.MultiheadAttention's assert
    assert lib.avsep_create(ctypes.byref(cfg), ctypes.byref(ctx)) == -1
    assert lib.avsep_workspace_bytes(None, 1, 1, 1, 1, 1) == 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libavsep_hip.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        _native.load()
