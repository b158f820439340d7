"""CPU-side checks of the drop-in boundary: libintfft.so loads, exports every symbol include/intfft.h
declares, validates generics like RTL elaboration, and refuses to compute without a HIP device."""
import ctypes
import os
import re

import pytest

from intfftk_amd import _capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "intfft.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(intfft_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    declared = _declared_symbols()
    assert set(declared) == set(capi.SYMBOLS)
    for sym in declared:
        assert getattr(L, sym) is not None


def test_no_oracle_in_product():
    """The product must not route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "intfftk_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle_c" not in src and "oracle_py" not in src and "intfft_oracle" not in src, f
    # and the shared object does not link the oracle
    import subprocess
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def _p(**kw):
    d = dict(log2n=10, data_width=16, twdl_width=16, format=0, rndmode=0, xser=1, direction=0, use_fly=1,
             in_order=0, out_order=0)
    d.update(kw)
    return capi.Params(*[d[k] for k, _ in capi.Params._fields_])


@pytest.mark.parametrize("kw,bits,cont", [
    (dict(), (16, 16), (2, 2)),
    (dict(format=1), (16, 26), (2, 4)),
    (dict(format=1, direction=2), (16, 36), (2, 8)),
    (dict(log2n=16, data_width=24, twdl_width=24, format=1), (24, 40), (4, 8)),
    (dict(log2n=20), (16, 16), (2, 2)),
    (dict(log2n=10, data_width=60, twdl_width=12, format=1), (60, 70), (8, 16)),  # results beyond 64 bits: 16-byte containers
    (dict(log2n=17, data_width=32, twdl_width=15, format=1, direction=2), (32, 66), (4, 16)),  # (the trpl18 tail, int_cmult_dsp48.vhd:267-303)
])
def test_io_widths(kw, bits, cont):
    a, b, c, d = (ctypes.c_int() for _ in range(4))
    p = _p(**kw)
    assert capi.lib().intfft_io_widths(ctypes.byref(p), a, b, c, d) == capi.OK
    assert (a.value, b.value) == bits and (c.value, d.value) == cont


@pytest.mark.parametrize("kw,code", [
    (dict(log2n=2), capi.ERR_INVALID),
    (dict(log2n=21), capi.ERR_INVALID),
    (dict(direction=3), capi.ERR_INVALID),
    (dict(in_order=4), capi.ERR_INVALID),
    (dict(format=1, rndmode=1), capi.ERR_UNSUPPORTED),          # int_dif2_fly.vhd:339-346
    (dict(twdl_width=28), capi.ERR_UNSUPPORTED),                # find_delay -> 0
    (dict(twdl_width=26, xser=0), capi.ERR_UNSUPPORTED),
    (dict(data_width=60, twdl_width=24), capi.ERR_UNSUPPORTED),  # no cmult regime for w >= 53 at t > 18
    (dict(log2n=10, data_width=57, format=1), capi.ERR_UNSUPPORTED),  # a 65 x 16 multiplier: its product slice leaves the 79 bits of P (int_cmult_trpl18_dsp48.vhd:151-152)
    (dict(log2n=17, data_width=32, format=1, direction=2), capi.ERR_UNSUPPORTED),  # the pair's last DIT multiplier: 65 x 16
    (dict(log2n=20, data_width=64, format=1, direction=2), capi.ERR_UNSUPPORTED),  # 104-bit results
])
def test_plan_create_rejects_like_elaboration(kw, code):
    plan = ctypes.c_void_p()
    p = _p(**kw)
    assert capi.lib().intfft_plan_create(ctypes.byref(plan), ctypes.byref(p), 0) == code
    assert not plan.value


def test_no_device_means_error_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    plan = ctypes.c_void_p()
    p = _p()
    rc = capi.lib().intfft_plan_create(ctypes.byref(plan), ctypes.byref(p), 0)
    assert rc == capi.ERR_NO_DEVICE and "no CPU fallback" in capi.strerror(rc)
    from intfftk_amd import IntFFTCore

    with pytest.raises(RuntimeError):
        IntFFTCore(10, 16, 16, 0, 0)


def test_null_arguments():
    L = capi.lib()
    assert L.intfft_plan_create(None, None, 0) == capi.ERR_NULL
    assert L.intfft_plan_destroy(None) == capi.ERR_NULL
    assert L.intfft_exec(None, None, None, 1, None) == capi.ERR_NULL


def test_plan_create_2d_validates_like_elaboration():
    """intfft_plan_create_2d: both factors must be native core lengths; elaboration errors come before the device check."""
    L = capi.lib()
    plan = ctypes.c_void_p()
    for l1, want in [(0, capi.ERR_INVALID), (2, capi.ERR_INVALID), (20, capi.ERR_INVALID), (19, capi.ERR_INVALID)]:
        assert L.intfft_plan_create_2d(ctypes.byref(plan), ctypes.byref(_p(log2n=21)), l1, 0) == want, l1
    assert L.intfft_plan_create_2d(ctypes.byref(plan), ctypes.byref(_p(log2n=25)), 12, 0) == capi.ERR_INVALID
    assert L.intfft_plan_create_2d(ctypes.byref(plan), ctypes.byref(_p(log2n=21, use_fly=0)), 10, 0) == capi.ERR_INVALID
    assert L.intfft_plan_create_2d(ctypes.byref(plan), ctypes.byref(_p(log2n=21, format=1, rndmode=1)), 10, 0) == capi.ERR_UNSUPPORTED
    assert L.intfft_plan_create_2d(ctypes.byref(plan), ctypes.byref(_p(log2n=21, data_width=50, format=1)), 10, 0) == capi.ERR_UNSUPPORTED
    assert L.intfft_plan_create_2d(None, ctypes.byref(_p(log2n=21)), 10, 0) == capi.ERR_NULL
    rc = L.intfft_plan_create_2d(ctypes.byref(plan), ctypes.byref(_p(log2n=21)), 10, 0)
    assert rc in (capi.OK, capi.ERR_NO_DEVICE)  # no CPU fallback: without a HIP device nothing is planned
    if rc == capi.OK:
        L.intfft_plan_destroy(plan)


def test_diagnostic_switches_need_the_master_switch():
    """Every INTFFT_* read of the library goes through diag_env() (intfft_internal.hpp): honoured only under INTFFT_DIAG=1, so a
    stray variable in a production environment cannot change which kernels a plan uses.  The only plain getenv is the master switch."""
    import glob
    import re

    src = os.path.join(ROOT, "intfftk_amd", "csrc")
    hits = []
    for f in glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.hpp")):
        for i, line in enumerate(open(f), 1):
            for m in re.finditer(r"(?<![\w])getenv\(\s*\"([A-Z_0-9]+)\"", line):
                if "diag_env" not in line[: m.start()][-12:]:
                    hits.append((os.path.basename(f), i, m.group(1)))
    assert [h for h in hits if h[2] != "INTFFT_DIAG"] == [], hits
    hdr = open(os.path.join(ROOT, "include", "intfft.h")).read()
    assert "INTFFT_DIAG=1" in hdr
