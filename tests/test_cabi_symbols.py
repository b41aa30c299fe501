"""CPU-only: the C-ABI library loads and exports every symbol include/rgbm.h declares (no compute calls)."""
import os
import re

from rgbmanip_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "rgbm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rgbm_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_bound_and_exported():
    names = _declared()
    assert len(names) >= 15
    lib = _lib.load()
    for n in names:
        assert n in _lib.SIGNATURES, f"{n} declared in rgbm.h but not bound in _lib.SIGNATURES"
        assert getattr(lib, n) is not None
    for n in _lib.SIGNATURES:
        assert n in names, f"{n} bound but not declared in include/rgbm.h"
    assert lib.rgbm_version() == 100


def test_errors_are_reported_not_swallowed():
    import ctypes as C
    lib = _lib.load()
    n = C.c_size_t()
    rc = lib.rgbm_adapose_workspace_bytes(None, 1, C.byref(n))
    assert rc != 0
    assert b"workspace_bytes" in lib.rgbm_last_error()


import pytest  # noqa: E402


@pytest.mark.parametrize("src", ["conv0_sweep_x3.hip", "conv0_sweep.hip"])
def test_sweep_asm_gathers_keep_their_registers(src):
    """hipcc must not copy or re-home a register whose inline-asm load is still in flight (tools/check_asm_gathers.py explains how it
    did while conv0_sweep_x3.hip's cooperative producers were written); the check reads the ISA of the current sources."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH: the check compiles the kernel to ISA")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_asm_gathers.py"),
                        os.path.join(root, "rgbmanip_amd", "csrc", src)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "plane loop lines" in r.stdout
