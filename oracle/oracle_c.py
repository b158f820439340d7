"""ctypes binding of oracle/libintfft_oracle.so (TEST INFRASTRUCTURE ONLY -- the checker).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FWD, INV, PAIR = 0, 1, 2
NATURAL, BITREV, HALVES, BITREV_LANES = 0, 1, 2, 3
REGIMES = {0: "sngl", 1: "dbl18", 2: "trpl18", 3: "sngl25", 4: "dbl35", 5: "trpl52", -1: None}


class Params(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int) for k in
                ("log2n", "data_width", "twdl_width", "format", "rndmode", "xser", "use_fly")]


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (oracle/Makefile).  Building the checker is not using it."""
    so = os.path.join(_HERE, "libintfft_oracle.so")
    src = os.path.join(_HERE, "intfft_oracle.c")
    hdr = os.path.join(_HERE, "intfft_oracle.h")
    stale = (not os.path.exists(so)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libintfft_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.orc_wrap.restype = ctypes.c_int64
        L.orc_wrap.argtypes = [ctypes.c_int64, ctypes.c_int]
        L.orc_cmult_regime.argtypes = [ctypes.c_int] * 3
        L.orc_cmult.argtypes = [ctypes.c_int64] * 4 + [ctypes.c_int] * 3 + [i64p, i64p]
        L.orc_twiddles.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_validate.argtypes = [ctypes.POINTER(Params), ctypes.c_int]
        L.orc_out_width.argtypes = [ctypes.POINTER(Params), ctypes.c_int]
        L.orc_order_index.restype = ctypes.c_size_t
        L.orc_order_index.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t]
        L.orc_exec.argtypes = [ctypes.POINTER(Params), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                               ctypes.c_int]
        L.orc_exec_i16.argtypes = L.orc_exec.argtypes
        L.orc_exec_2d.argtypes = [ctypes.POINTER(Params), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
        L.orc_validate_2d.argtypes = [ctypes.POINTER(Params), ctypes.c_int, ctypes.c_int]
        L.orc_twiddle_2d.restype = None
        L.orc_twiddle_2d.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, i64p, i64p]
        _LIB = L
    return _LIB


def make_params(log2n, data_width=16, twdl_width=16, fmt=0, rnd=0, new=True, use_fly=1) -> Params:
    return Params(log2n, data_width, twdl_width, fmt, rnd, 1 if new else 0, use_fly)


def cmult(d_re, d_im, wr, wi, w, t, new=True):
    o_re, o_im = ctypes.c_int64(), ctypes.c_int64()
    rc = lib().orc_cmult(d_re, d_im, wr, wi, w, t, 1 if new else 0,
                         ctypes.byref(o_re), ctypes.byref(o_im))
    if rc:
        raise ValueError("unsupported widths w=%d t=%d" % (w, t))
    return o_re.value, o_im.value


def cmult_regime(w, t, new=True):
    return REGIMES[lib().orc_cmult_regime(w, t, 1 if new else 0)]


def twiddles(stage, t, new=True):
    n = 1 << stage
    re = np.zeros(n, dtype=np.int64)
    im = np.zeros(n, dtype=np.int64)
    rc = lib().orc_twiddles(stage, t, 1 if new else 0, re.ctypes.data, im.ctypes.data)
    if rc:
        raise ValueError("bad twiddle request")
    return re, im


def out_width(p: Params, direction: int) -> int:
    return lib().orc_out_width(ctypes.byref(p), direction)


def execute(x: np.ndarray, p: Params, direction=FWD, in_order=NATURAL, out_order=NATURAL,
            form=1, threads=0) -> np.ndarray:
    """x: integer array [batch, N, 2] (any int dtype) -> int64 array [batch, N, 2].
    threads = 0: a team sized to the work (about 2^21 stage-samples per thread, at most one per frame; see pick_threads in
    intfft_oracle.c) -- not "every core" as before round 3; an explicit count is capped at the number of frames."""
    n = 1 << p.log2n
    a = np.ascontiguousarray(x, dtype=np.int64).reshape(-1, n, 2)
    out = np.empty_like(a)
    rc = lib().orc_exec(ctypes.byref(p), direction, in_order, out_order, a.ctypes.data,
                        out.ctypes.data, a.shape[0], form, threads)
    if rc:
        raise ValueError("orc_exec failed rc=%d" % rc)
    return out


def execute_i16(x: np.ndarray, p: Params, direction=FWD, in_order=NATURAL, out_order=NATURAL,
                form=1, threads=0) -> np.ndarray:
    """int16 container fast path (headline config): [batch, N, 2] int16 -> int16."""
    n = 1 << p.log2n
    a = np.ascontiguousarray(x, dtype=np.int16).reshape(-1, n, 2)
    out = np.empty_like(a)
    rc = lib().orc_exec_i16(ctypes.byref(p), direction, in_order, out_order, a.ctypes.data,
                            out.ctypes.data, a.shape[0], form, threads)
    if rc:
        raise ValueError("orc_exec_i16 failed rc=%d" % rc)
    return out


def execute_2d(x: np.ndarray, p: Params, log2_n1: int, direction=FWD, in_order=NATURAL, out_order=NATURAL,
               form=1, threads=0) -> np.ndarray:
    """The N > 512K "2D-FFT scheme" extension (see intfft_oracle.c): p.log2n = log2 N, N1 = 2^log2_n1.
    form 0 structural (stream cores), 1 flat, 2 structural (in-place cores)."""
    n = 1 << p.log2n
    a = np.ascontiguousarray(x, dtype=np.int64).reshape(-1, n, 2)
    out = np.empty_like(a)
    rc = lib().orc_exec_2d(ctypes.byref(p), log2_n1, direction, in_order, out_order, a.ctypes.data, out.ctypes.data,
                           a.shape[0], form, threads)
    if rc:
        raise ValueError("orc_exec_2d failed rc=%d" % rc)
    return out


def twiddle_2d(log2n: int, t: int, m: int):
    re, im = ctypes.c_int64(), ctypes.c_int64()
    lib().orc_twiddle_2d(log2n, t, m, ctypes.byref(re), ctypes.byref(im))
    return re.value, im.value


def num_threads() -> int:
    return lib().orc_num_threads()
