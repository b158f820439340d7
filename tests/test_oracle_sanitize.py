"""Runs the C oracle under AddressSanitizer + UBSan (the build's equivalent of race/memory
checking for the checker itself) in a subprocess, on every multiplier regime."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r"""
import ctypes, sys, numpy as np
L = ctypes.CDLL(sys.argv[1])
class P(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int) for k in ("log2n","data_width","twdl_width","format","rndmode","xser","use_fly")]
L.orc_exec.argtypes = [ctypes.POINTER(P)] + [ctypes.c_int]*3 + [ctypes.c_void_p]*2 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
rng = np.random.default_rng(0)
for (l, dw, tw, fmt, rnd, new) in [(6,16,16,0,0,1),(6,16,16,0,1,1),(6,16,16,1,0,1),(5,30,16,1,0,0),(5,44,16,1,0,1),
                                   (6,32,24,1,0,1),(4,14,24,1,0,1),(12,16,16,0,0,1)]:
    n = 1 << l
    x = rng.integers(-(1 << (dw-1)), 1 << (dw-1), size=(3, n, 2)).astype(np.int64)
    for d in (0, 1, 2):
        p = P(l, dw, tw, fmt, rnd, new, 1)
        y = np.empty_like(x)
        for form in (0, 1):
            rc = L.orc_exec(ctypes.byref(p), d, 0, 0, x.ctypes.data, y.ctypes.data, 3, form, 2)
            assert rc in (0, -1), rc
print("asan-ok")
"""


def test_oracle_under_asan_ubsan(tmp_path):
    so = os.path.join(ROOT, "oracle", "libintfft_oracle_asan.so")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libintfft_oracle_asan.so"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-200:])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="2")
    drv = tmp_path / "drv.py"
    drv.write_text(DRIVER)
    out = subprocess.run([sys.executable, str(drv), so], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "asan-ok" in out.stdout, out.stderr[-2000:]
