"""csrc/glibc_libm.h (the device's sinf / cosf / tanf / acosf / atanf / atan2f) against the host's libm,
GNU libc 2.35 — the library the reference CPU integrator calls.  Bit for bit, NaN == NaN.

The CPU suite sweeps every 61st float bit pattern (70 M arguments per function, all exponents and both
signs, a stride coprime to every power of two); `tests/libm/libm_check <function> 1` sweeps all 2^32 —
profiles/r02_libm_exhaustive.txt holds that run's output."""
import ctypes
import os
import platform
import subprocess

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libm")


def _glibc():
    try:
        return platform.libc_ver()[1]
    except Exception:
        return ""


@pytest.fixture(scope="module")
def checker():
    if "fma" not in open("/proc/cpuinfo").read():
        pytest.skip("host without FMA: its libm selects the non-fused sinf/cosf variant")
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    lib = ctypes.CDLL(os.path.join(HERE, "libmcpt_libm_check.so"))
    lib.mcpt_libm_check.restype = ctypes.c_longlong
    lib.mcpt_libm_check.argtypes = [ctypes.c_char_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_int]
    return lib


@pytest.mark.skipif(_glibc() != "2.35", reason="the restatement is of GNU libc 2.35")
@pytest.mark.parametrize("name", ["sinf", "cosf", "tanf", "acosf", "atanf", "atan2f"])
def test_matches_host_libm(checker, name):
    first = np.zeros(16, np.uint32)
    bad = checker.mcpt_libm_check(name.encode(), 61, first.ctypes.data, 16)
    assert bad == 0, f"{name}: {bad} arguments differ, first {[hex(v) for v in first[:min(bad, 16)]]}"
