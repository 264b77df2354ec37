"""Host-side sanitizer run: the label / plan-builder half of libwagg and the C
oracle under AddressSanitizer + UBSan.  CPU only; GPU sanitizers are not available on this pool."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_half_is_clean_under_asan_ubsan():
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt:
        pytest.skip("clang AddressSanitizer runtime not found")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "climate_toolbox_amd", "csrc"), "hostsan"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"])
    env = dict(os.environ, LD_PRELOAD=rt[0], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostsan_check.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "hostsan ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
