"""AddressSanitizer over the library's HOST side (packers, planner, launchers) — GPU ASan is not available on this pool, so the
host code is compiled --cuda-host-only with -fsanitize=address and linked against a HIP stand-in whose device memory is host
memory (tests/asan/hip_stub.cpp, tools/build_asan_host.sh); tests/asan/drive_host.py then creates the network in every storage
type and norm mode, plans workspaces up to batch 512 and runs forwards (kernel launches are no-ops) through every option."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang"


@pytest.mark.skipif(shutil.which("hipcc") is None or not os.path.exists(CLANG), reason="needs hipcc / the ROCm clang (ASan runtime)")
def test_host_side_is_clean_under_address_sanitizer():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "build_asan_host.sh")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               RGBM_HIP_LIB=os.path.join(ROOT, "rgbmanip_amd", "librgbm_hip_asan_host.so"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan", "drive_host.py")], capture_output=True, text=True,
                       timeout=900, env=env)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "ASAN_HOST_OK" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
    # canary: a weight buffer half as long as its descriptor says must be reported (the sanitizer is live in this set-up)
    c = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan", "drive_host.py"), "canary"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert c.returncode != 0 and "AddressSanitizer: heap-buffer-overflow" in c.stderr and "CANARY_NOT_CAUGHT" not in c.stdout, \
        c.stdout[-500:] + c.stderr[-2000:]
