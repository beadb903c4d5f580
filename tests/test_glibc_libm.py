"""csrc/glibc_libm.h (the device's sinf / cosf / tanf / acosf / atanf / atan2f and the medium code's double exp / log) against the host's libm,
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


def _wrong_host_libm():
    """A host whose libm is not GNU libc 2.35: the device code restates THAT library, so on such a box "the CPU reference image
    of the same box" is not bit-equal to the GPU frame.  On a box without a GPU this is a skipped check; on a GPU box (where
    bench.py and the parity tests compare GPU frames with CPU frames made here) it is a FAILURE, with what to do about it."""
    if _glibc() == "2.35":
        return
    message = (f"this host's C library is glibc {_glibc() or '?'}, not 2.35: csrc/glibc_libm.h restates glibc 2.35's sinf / cosf / tanf / "
               "acosf / atanf / atan2f / exp / log bit for bit, so CPU frames rendered on THIS host (oracle, compiled reference, "
               "mcpt_cli --cpu) can differ from the GPU's in the last bits.  Either run the suite in the project's image (Ubuntu "
               "22.04, glibc 2.35), or regenerate csrc/glibc_libm_tables.inc for this library with tools/extract_libm_tables.py "
               "and re-run tests/libm/libm_check <function> 1 until every function reports 0 differences.")
    if os.path.exists("/dev/kfd"):
        pytest.fail(message)
    pytest.skip(message)


@pytest.mark.parametrize("name", ["sinf", "cosf", "tanf", "acosf", "atanf", "atan2f", "exp", "log"])
def test_matches_host_libm(checker, name):
    _wrong_host_libm()
    first = np.zeros(16, np.uint32)
    bad = checker.mcpt_libm_check(name.encode(), 61, first.ctypes.data, 16)
    assert bad == 0, f"{name}: {bad} arguments differ, first {[hex(v) for v in first[:min(bad, 16)]]}"


@pytest.mark.gpu
def test_device_evaluation_equals_the_host_libm_on_every_argument():
    """The header as the GPU evaluates it: tests/libm/libm_device_sweep runs sinf cosf tanf acosf atanf for ALL 2^32
    float bit patterns (atan2f for 2^26 pairs; double exp / log for three arguments per float bit pattern: the float as a
    double, a neighbour that is not a float, a hashed 64-bit pattern) in a gfx950 kernel and compares, chunk by chunk, with what this
    host's libm returns.  "device == host" is thereby tested, not inferred from frames.  It FAILS — not skips — on a
    host whose libm is not the restated one (GNU libc 2.35 on an FMA-capable CPU): on such a box "the CPU reference
    image of the same box" is not bit-equal to the GPU frame, and the suite must say so."""
    import json
    subprocess.run(["make", "-s", "-C", HERE, "libm_device_sweep"], check=True)
    r = subprocess.run([os.path.join(HERE, "libm_device_sweep")], capture_output=True, text=True, timeout=900)
    assert r.returncode in (0, 1), r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    print(json.dumps(rec))
    host = f"glibc {_glibc() or '?'}, FMA {'yes' if 'fma' in open('/proc/cpuinfo').read() else 'NO'}"
    bad = {k: v for k, v in rec.items() if isinstance(v, dict) and v["differing_chunks"]}
    assert not bad, (f"device evaluation of csrc/glibc_libm.h differs from this host's libm ({host}; the restatement is of "
                     f"glibc 2.35, FMA variant): {bad}")
    for name in ("sinf", "cosf", "tanf", "acosf", "atanf", "exp", "log"):
        assert rec[name]["arguments"] == 2 ** 32  # (exp / log: three double arguments per float bit pattern)
