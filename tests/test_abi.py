"""C-ABI checks that need no GPU: the shared library loads, exports every function include/avsep.h
declares, and the Python binding lists exactly the same set."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from av_separation import _native


def header_functions(dev_only=False):
    """Functions include/avsep.h declares for the product (default) or only inside its `#ifdef AVSEP_DEV` sections."""
    text = open(os.path.join(ROOT, "include", "avsep.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    dev = "".join(re.findall(r"#ifdef AVSEP_DEV(.*?)#endif", text, flags=re.S))
    if dev_only:
        text = dev
    else:
        text = re.sub(r"#ifdef AVSEP_DEV.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(avsep_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_what_binding_lists():
    assert header_functions() == sorted(_native.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_native.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in header_functions():
        assert hasattr(lib, name), name


def test_developer_library_is_a_superset_and_the_product_has_no_developer_entry_points():
    assert os.path.exists(_native.DEV_LIB_PATH), "run __graft_entry__.build() first"
    dev, prod = ctypes.CDLL(_native.DEV_LIB_PATH), ctypes.CDLL(_native.LIB_PATH)
    extra = header_functions(dev_only=True)
    assert extra, "the header lists the developer build's extra entry points"
    for name in header_functions() + extra:
        assert hasattr(dev, name), name
    for name in extra:
        assert not hasattr(prod, name), name
    assert dev.avsep_abi_version() == prod.avsep_abi_version()


def test_product_library_reads_no_environment_variable():
    """`nm -D --undefined-only`: the product library does not even import getenv / secure_getenv."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--undefined-only", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in out
    out = subprocess.run(["nm", "-D", "--undefined-only", _native.DEV_LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" in out


def test_product_package_reads_no_environment_variable_but_the_library_selector():
    """like the product library (no getenv import, above), the Python package decides nothing from the
    environment -- the one read is AVSEP_LIB in _native.py, which selects WHICH library file is opened (the developer build
    for tools/ and the bit-identity tests).  A/B switches of the training path are module attributes set by tests and tools."""
    import glob
    import os
    import re
    pkg = os.path.dirname(os.path.abspath(_native.__file__))
    hits = []
    for path in sorted(glob.glob(os.path.join(pkg, "*.py"))):
        for no, line in enumerate(open(path), 1):
            code = line.split("#", 1)[0]
            if re.search(r"environ|getenv", code):
                hits.append((os.path.basename(path), no, code.strip()))
    assert hits == [("_native.py", hits[0][1], '_want = os.environ.get("AVSEP_LIB")')], hits


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
