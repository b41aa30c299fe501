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


def test_shipped_kernels_hold_no_packed_fp32_instructions(tmp_path):
    """Round 4: hipcc's v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 on registers a global_load had just delivered computed from other
    values in half of the lane rows when two forwards overlapped on the device; the library is built without packed-fp32 instructions
    (build.sh: -target-feature -packed-fp32-ops).  The code objects inside the built .so must hold none."""
    import glob
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "rgbmanip_amd", "librgbm_hip.so")
    if not (os.path.exists(objdump) and os.path.exists(so)):
        pytest.skip("needs llvm-objdump and the built library")
    work = os.path.join(str(tmp_path), "librgbm_hip.so")
    shutil.copy(so, work)
    subprocess.run([objdump, "--offloading", work], check=True, capture_output=True, timeout=600)
    objs = glob.glob(work + ".*gfx950")
    assert len(objs) >= 10, objs
    packed, mfma = 0, 0
    for o in objs:
        dis = subprocess.run([objdump, "-d", o], check=True, capture_output=True, text=True, timeout=600).stdout
        packed += sum(1 for line in dis.splitlines() if "v_pk_" in line and "_f32" in line)
        mfma += dis.count("v_mfma")
    assert mfma > 1000                         # the disassembly is the real thing
    assert packed == 0, f"{packed} packed-fp32 instructions in the shipped kernels"
