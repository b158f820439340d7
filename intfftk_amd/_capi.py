"""ctypes view of include/intfft.h (libintfft.so).  No compute happens in Python."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("INTFFT_LIB") or os.path.join(_HERE, "lib", "libintfft.so")

OK = 0
ERR_INVALID, ERR_UNSUPPORTED, ERR_NULL, ERR_NO_DEVICE, ERR_ALLOC, ERR_TRANSPORT = -1, -2, -3, -4, -5, -6
FWD, INV, PAIR = 0, 1, 2
TRANSPORT_PEER, TRANSPORT_RCCL = 0, 1
ORDER_NATURAL, ORDER_BITREV, ORDER_HALVES, ORDER_BITREV_LANES = 0, 1, 2, 3
ORDERS = {"NATURAL": 0, "BITREV": 1, "HALVES": 2, "BITREV_LANES": 3}
DIRECTIONS = {"FWD": 0, "INV": 1, "PAIR": 2}

# every symbol include/intfft.h declares
SYMBOLS = ("intfft_io_widths", "intfft_plan_create", "intfft_plan_create_2d", "intfft_plan_destroy", "intfft_plan_get_info",
           "intfft_exec", "intfft_plan_workspace_bytes", "intfft_exec_ws", "intfft_plan_release_scratch", "intfft_exec_host",
           "intfft_shard_prepare", "intfft_exec_sharded", "intfft_exec_sharded_async", "intfft_shard_set_transport", "intfft_reorder",
           "intfft_twiddles", "intfft_strerror", "intfft_version", "intfft_stream_open", "intfft_stream_push", "intfft_stream_flush",
           "intfft_stream_pull", "intfft_stream_pending", "intfft_stream_close")


class Params(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in
                ("log2n", "data_width", "twdl_width", "format", "rndmode", "xser", "direction",
                 "use_fly", "in_order", "out_order")]


class PlanInfo(ctypes.Structure):
    _fields_ = [("in_bits", ctypes.c_int32), ("out_bits", ctypes.c_int32),
                ("in_container", ctypes.c_int32), ("out_container", ctypes.c_int32),
                ("n_passes", ctypes.c_int32), ("compute_word", ctypes.c_int32),
                ("fast_path", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("scratch_bytes", ctypes.c_uint64), ("kernel_name", ctypes.c_char * 64)]


class IntFFTError(RuntimeError):
    def __init__(self, status: int, what: str):
        super().__init__("%s: %s (status %d)" % (what, strerror(status), status))
        self.status = status


_lib = None


def lib():
    """Loads libintfft.so; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "intfftk_amd: %s is missing -- build it with `python -m intfftk_amd.build` "
                "(there is no CPU fallback)" % LIB_PATH)
        # libintfft.so must share ONE HIP runtime with torch (device pointers and streams cross the
        # boundary): load torch's bundled libamdhip64 first so the library binds to it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        pp = ctypes.POINTER(Params)
        ip = ctypes.POINTER(ctypes.c_int)
        L.intfft_io_widths.argtypes = [pp, ip, ip, ip, ip]
        L.intfft_plan_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), pp, ctypes.c_int]
        L.intfft_plan_create_2d.argtypes = [ctypes.POINTER(ctypes.c_void_p), pp, ctypes.c_int, ctypes.c_int]
        L.intfft_plan_destroy.argtypes = [ctypes.c_void_p]
        L.intfft_plan_get_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(PlanInfo)]
        L.intfft_exec.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                  ctypes.c_void_p]
        L.intfft_plan_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.intfft_exec_ws.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                     ctypes.c_void_p]
        L.intfft_plan_release_scratch.argtypes = [ctypes.c_void_p]
        L.intfft_exec_sharded_async.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.intfft_exec_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.c_size_t]
        L.intfft_exec_sharded.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_size_t]
        L.intfft_shard_prepare.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_size_t]
        L.intfft_shard_set_transport.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.intfft_reorder.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        L.intfft_twiddles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.POINTER(ctypes.c_size_t)]
        szp = ctypes.POINTER(ctypes.c_size_t)
        L.intfft_stream_open.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.intfft_stream_push.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, szp]
        L.intfft_stream_flush.argtypes = [ctypes.c_void_p]
        L.intfft_stream_pull.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, szp, ctypes.c_int]
        L.intfft_stream_pending.argtypes = [ctypes.c_void_p, szp, szp]
        L.intfft_stream_close.argtypes = [ctypes.c_void_p]
        L.intfft_strerror.restype = ctypes.c_char_p
        L.intfft_strerror.argtypes = [ctypes.c_int]
        L.intfft_version.restype = ctypes.c_char_p
        for fn in ("intfft_io_widths", "intfft_plan_create", "intfft_plan_create_2d", "intfft_plan_destroy",
                   "intfft_plan_get_info", "intfft_exec", "intfft_exec_host", "intfft_exec_sharded", "intfft_exec_sharded_async",
                   "intfft_plan_workspace_bytes", "intfft_exec_ws", "intfft_plan_release_scratch",
                   "intfft_shard_prepare", "intfft_shard_set_transport", "intfft_reorder", "intfft_twiddles", "intfft_stream_open",
                   "intfft_stream_push", "intfft_stream_flush", "intfft_stream_pull", "intfft_stream_pending", "intfft_stream_close"):
            getattr(L, fn).restype = ctypes.c_int
        _lib = L
    return _lib


def strerror(status: int) -> str:
    return lib().intfft_strerror(status).decode()


def check(status: int, what: str):
    if status != OK:
        raise IntFFTError(status, what)
