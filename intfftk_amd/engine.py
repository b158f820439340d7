"""Host-side mirror of the reference's entity interface, on top of the C-ABI (libintfft.so).

The reference has no software API; its public interface is the generic/port list of four VHDL
entities.  Each gets a constructor here with the generics under their RTL names:

    int_fftNk            src/vhdl/fft/int_fftNk.vhd:72-103          HALVES in  -> BITREV out
    int_ifftNk           src/vhdl/fft/int_ifftNk.vhd:71-102         BITREV in  -> HALVES out
    int_fft_single_path  src/vhdl/main/int_fft_single_path.vhd:85-113   NATURAL -> NATURAL
    int_fft_ifft_pair    src/vhdl/main/int_fft_ifft_pair.vhd:74-107     NATURAL -> NATURAL, FFT then IFFT

A core is called with a device tensor [batch, N, 2] (re, im) in the container dtype of its input
width and returns a new tensor in the container dtype of its output width.  Torch is used for device
memory and streams only; all arithmetic happens in the HIP kernels behind intfft_exec.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import _capi as capi

_MODES = {"UNSCALED": (1, 0), "TRUNCATE": (0, 0), "ROUNDING": (0, 1)}  # fft_signle_test.vhd:80-112
_NP_DT = {2: np.int16, 4: np.int32, 8: np.int64, 16: np.int64}  # 16: two int64 words per component (low, high)


def wide_to_int(a) -> np.ndarray:
    """[..., 2] int64 words (low, high) of 16-byte containers -> object array of Python ints (results beyond 64 bits:
    include/intfft.h, "containers").  Little-endian two's complement: value = high * 2^64 + (low mod 2^64)."""
    a = np.asarray(a)
    if a.dtype != np.int64 or a.shape[-1] != 2:
        raise ValueError("expected [..., 2] int64 words")
    lo = a[..., 0].astype(object) & ((1 << 64) - 1)
    return a[..., 1].astype(object) * (1 << 64) + lo


def _torch():
    import torch

    return torch


def _torch_dtype(nbytes: int):
    torch = _torch()
    return {2: torch.int16, 4: torch.int32, 8: torch.int64, 16: torch.int64}[nbytes]


def set_mode(mode: str):
    """tb helper set_mode(): "UNSCALED" | "ROUNDING" | "TRUNCATE" -> (FORMAT, RNDMODE)."""
    try:
        return _MODES[mode.upper()]
    except KeyError:
        raise ValueError("MODE must be UNSCALED, ROUNDING or TRUNCATE") from None


class IntFFTCore:
    """One elaborated core (an `intfft_plan`).  Immutable; call it on batches of frames."""

    def __init__(self, NFFT: int, DATA_WIDTH: int = 16, TWDL_WIDTH: int = 16, FORMAT: int = 1,
                 RNDMODE: int = 0, XSER: str = "NEW", direction: str = "FWD", in_order: str = "NATURAL",
                 out_order: str = "NATURAL", USE_FLY: int = 1, device: Optional[int] = None,
                 RAMB_TYPE: str = "WRAP", USE_MLT: bool = False, NFFT1: int = 0):
        # NFFT1 != 0: the N > 512K "2D-FFT scheme" (include/intfft.h: intfft_plan_create_2d): N = 2^NFFT = 2^NFFT1 * 2^(NFFT-NFFT1)
        # RAMB_TYPE (strobe tolerance, int_fftNk.vhd:23-37) and USE_MLT (row_twiddle_tay.vhd:206-240)
        # do not change values; accepted for interface compatibility.
        if XSER not in ("NEW", "OLD"):
            raise ValueError('XSER must be "NEW" or "OLD"')
        if RAMB_TYPE not in ("WRAP", "CONT"):
            raise ValueError('RAMB_TYPE must be "WRAP" or "CONT"')
        self.params = capi.Params(NFFT, DATA_WIDTH, TWDL_WIDTH, FORMAT, RNDMODE, 1 if XSER == "NEW" else 0,
                                  capi.DIRECTIONS[direction], USE_FLY, capi.ORDERS[in_order],
                                  capi.ORDERS[out_order])
        self.n = 1 << NFFT
        L = capi.lib()
        self.nfft1 = int(NFFT1)
        if self.nfft1:  # lengths beyond the 1-D cores: widths follow the same rule (DATA_WIDTH + FORMAT * NFFT)
            growth = NFFT * FORMAT * (2 if direction == "PAIR" else 1)
            self.in_bits, self.out_bits = DATA_WIDTH, DATA_WIDTH + growth
            cb = lambda b: 2 if b <= 16 else 4 if b <= 32 else 8  # noqa: E731
            self.in_container, self.out_container = cb(self.in_bits), cb(self.out_bits)
        else:
            ib, ob, ic, oc = (ctypes.c_int() for _ in range(4))
            capi.check(L.intfft_io_widths(ctypes.byref(self.params), ib, ob, ic, oc), "intfft_io_widths")
            self.in_bits, self.out_bits = ib.value, ob.value
            self.in_container, self.out_container = ic.value, oc.value
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("intfftk_amd needs a HIP device: there is no CPU execution path")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self._plan = ctypes.c_void_p()
        if self.nfft1:
            capi.check(L.intfft_plan_create_2d(ctypes.byref(self._plan), ctypes.byref(self.params), self.nfft1, self.device),
                       "intfft_plan_create_2d")
        else:
            capi.check(L.intfft_plan_create(ctypes.byref(self._plan), ctypes.byref(self.params), self.device),
                       "intfft_plan_create")
        info = capi.PlanInfo()
        capi.check(L.intfft_plan_get_info(self._plan, ctypes.byref(info)), "intfft_plan_get_info")
        self.info = {k: getattr(info, k) for k, _ in info._fields_ if k not in ("reserved", "kernel_name")}
        self.info["kernel_name"] = info.kernel_name.decode()

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_plan", None) is not None and self._plan.value:
            capi.lib().intfft_plan_destroy(self._plan)
            self._plan = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- execution --------------------------------------------------------------------------
    @property
    def in_dtype(self):
        return _torch_dtype(self.in_container)

    @property
    def out_dtype(self):
        return _torch_dtype(self.out_container)

    def out_shape(self, batch: int):
        """[batch, N, 2], or [batch, N, 2, 2] int64 words (low, high) when the results need 16-byte containers
        (wide_to_int() turns those into Python ints)."""
        return (batch, self.n, 2) + ((2,) if self.out_container == 16 else ())

    def __call__(self, x, out=None):
        torch = _torch()
        if not (x.is_cuda and x.device.index == self.device):
            raise ValueError("input must live on cuda:%d" % self.device)
        if x.dtype != self.in_dtype:
            raise TypeError("input dtype must be %s for DATA_WIDTH=%d" % (self.in_dtype, self.in_bits))
        if x.dim() != 3 or x.shape[1] != self.n or x.shape[2] != 2:
            raise ValueError("input must be [batch, %d, 2]" % self.n)
        x = x.contiguous()
        batch = x.shape[0]
        if out is None:
            out = torch.empty(self.out_shape(batch), dtype=self.out_dtype, device=x.device)
        elif out.dtype != self.out_dtype or tuple(out.shape) != self.out_shape(batch) or not out.is_contiguous():
            raise ValueError("bad `out` tensor")
        stream = torch.cuda.current_stream(x.device).cuda_stream
        capi.check(capi.lib().intfft_exec(self._plan, x.data_ptr(), out.data_ptr(), batch, stream),
                   "intfft_exec")
        return out

    def exec_raw(self, in_ptr: int, out_ptr: int, batch: int, stream: int = 0):
        """Straight intfft_exec on raw device pointers (bench loop: no tensor bookkeeping)."""
        capi.check(capi.lib().intfft_exec(self._plan, in_ptr, out_ptr, batch, stream), "intfft_exec")

    def workspace_bytes(self, batch: int) -> int:
        """intfft_plan_workspace_bytes: the caller-supplied workspace with which exec_ws(batch frames) runs like the plan-owned scratch
        (0 for single-launch plans)."""
        n = ctypes.c_size_t()
        capi.check(capi.lib().intfft_plan_workspace_bytes(self._plan, batch, ctypes.byref(n)), "intfft_plan_workspace_bytes")
        return n.value

    def exec_ws(self, x, workspace=None, out=None):
        """The transform on a caller-supplied workspace (intfft_exec_ws): re-entrant for every plan -- one core may run on several
        streams at once, each call with its own workspace (a uint8 / any-dtype device tensor of >= workspace_bytes(1) bytes, 256-byte
        aligned as torch allocations are; None for single-launch plans)."""
        torch = _torch()
        if not (x.is_cuda and x.device.index == self.device) or x.dtype != self.in_dtype or x.dim() != 3 or x.shape[1] != self.n or x.shape[2] != 2:
            raise ValueError("input must be a [batch, %d, 2] %s tensor on cuda:%d" % (self.n, self.in_dtype, self.device))
        x = x.contiguous()
        batch = x.shape[0]
        if out is None:
            out = torch.empty(self.out_shape(batch), dtype=self.out_dtype, device=x.device)
        elif out.dtype != self.out_dtype or tuple(out.shape) != self.out_shape(batch) or not out.is_contiguous():
            raise ValueError("bad `out` tensor")
        ws_ptr, ws_bytes = (0, 0) if workspace is None else (workspace.data_ptr(), workspace.numel() * workspace.element_size())
        stream = torch.cuda.current_stream(x.device).cuda_stream
        capi.check(capi.lib().intfft_exec_ws(self._plan, x.data_ptr(), out.data_ptr(), batch, ws_ptr, ws_bytes, stream), "intfft_exec_ws")
        return out

    def release_scratch(self):
        """intfft_plan_release_scratch: frees the plan-owned scratch; afterwards only exec_ws runs this core."""
        capi.check(capi.lib().intfft_plan_release_scratch(self._plan), "intfft_plan_release_scratch")
        info = capi.PlanInfo()
        capi.check(capi.lib().intfft_plan_get_info(self._plan, ctypes.byref(info)), "intfft_plan_get_info")
        self.info["scratch_bytes"] = info.scratch_bytes

    def exec_host(self, x: np.ndarray, chunk_frames: int = 0) -> np.ndarray:
        """Host-resident frames through intfft_exec_host: chunked, double-buffered H2D / transform / D2H
        on three streams (every frame is still transformed on the GPU)."""
        x = np.asarray(x)
        if x.dtype != _NP_DT[self.in_container]:  # same rule as __call__: no silent narrowing of wider / float arrays
            raise TypeError("input dtype must be %s for DATA_WIDTH=%d" % (np.dtype(_NP_DT[self.in_container]).name, self.in_bits))
        x = np.ascontiguousarray(x)
        if x.ndim != 3 or x.shape[1] != self.n or x.shape[2] != 2:
            raise ValueError("input must be [batch, %d, 2]" % self.n)
        out = np.empty(self.out_shape(x.shape[0]), dtype=_NP_DT[self.out_container])
        capi.check(capi.lib().intfft_exec_host(self._plan, x.ctypes.data, out.ctypes.data, x.shape[0], chunk_frames),
                   "intfft_exec_host")
        return out

    def twiddles(self, stage: int) -> np.ndarray:
        """[2^stage, 2] int32 (re, im): what rom_twiddle_int emits for cnt = 0..2^stage-1
        (2-D scheme plans: stage = -1 returns the inter-pass table W_N^m, [N, 2])."""
        cnt = ctypes.c_size_t()
        L = capi.lib()
        capi.check(L.intfft_twiddles(self._plan, stage, None, ctypes.byref(cnt)), "intfft_twiddles")
        out = np.empty((cnt.value, 2), dtype=np.int32)
        capi.check(L.intfft_twiddles(self._plan, stage, out.ctypes.data, ctypes.byref(cnt)), "intfft_twiddles")
        return out


class FrameStream:
    """The frame-queue form of the streaming interface (include/intfft.h: intfft_stream_*): frames are pushed as they arrive -- any
    number per call, with gaps -- and results are pulled in push order; upload, transform and download overlap across calls (the
    software analogue of the RTL's valid strobes, int_fftNk.vhd:23-37).  One producer thread (push / flush) and one consumer thread
    (pull) may use an object concurrently.  The core must outlive the stream."""

    def __init__(self, core: IntFFTCore, slot_frames: int = 0, n_slots: int = 0):
        self.core = core
        self._s = ctypes.c_void_p()
        capi.check(capi.lib().intfft_stream_open(core._plan, slot_frames, n_slots, ctypes.byref(self._s)), "intfft_stream_open")

    def push(self, x: np.ndarray) -> int:
        """frames accepted (fewer than len(x) when every slot is in flight or waits to be pulled); never waits for the device"""
        c = self.core
        x = np.asarray(x)
        if x.dtype != _NP_DT[c.in_container]:
            raise TypeError("input dtype must be %s for DATA_WIDTH=%d" % (np.dtype(_NP_DT[c.in_container]).name, c.in_bits))
        x = np.ascontiguousarray(x)
        if x.ndim != 3 or x.shape[1] != c.n or x.shape[2] != 2:
            raise ValueError("input must be [frames, %d, 2]" % c.n)
        acc = ctypes.c_size_t()
        capi.check(capi.lib().intfft_stream_push(self._s, x.ctypes.data, x.shape[0], ctypes.byref(acc)), "intfft_stream_push")
        return acc.value

    def flush(self):
        capi.check(capi.lib().intfft_stream_flush(self._s), "intfft_stream_flush")

    def pull(self, max_frames: int, wait: bool = False, out: Optional[np.ndarray] = None) -> np.ndarray:
        """up to max_frames finished frames in push order ([0, N, 2] when nothing is ready); `out`: a C-contiguous array of the output
        dtype with room for max_frames frames to receive them (a view of its first rows is returned)"""
        c = self.core
        if out is None:
            out = np.empty(c.out_shape(max_frames), dtype=_NP_DT[c.out_container])
        elif out.dtype != _NP_DT[c.out_container] or not out.flags.c_contiguous or out.shape[0] < max_frames or tuple(out.shape[1:]) != c.out_shape(1)[1:]:
            raise ValueError("bad `out` array")
        got = ctypes.c_size_t()
        capi.check(capi.lib().intfft_stream_pull(self._s, out.ctypes.data, max_frames, ctypes.byref(got), 1 if wait else 0),
                   "intfft_stream_pull")
        return out[:got.value]

    def pending(self):
        """(frames pushed and not pulled yet, frames in the slot that is still filling)"""
        a, b = ctypes.c_size_t(), ctypes.c_size_t()
        capi.check(capi.lib().intfft_stream_pending(self._s, ctypes.byref(a), ctypes.byref(b)), "intfft_stream_pending")
        return a.value, b.value

    def close(self):
        if getattr(self, "_s", None) is not None and self._s.value:
            capi.lib().intfft_stream_close(self._s)
            self._s = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def int_fftNk(NFFT=10, DATA_WIDTH=16, TWDL_WIDTH=16, FORMAT=1, RNDMODE=0, XSER="NEW", USE_FLY=1,
              RAMB_TYPE="WRAP", USE_MLT=False, device=None) -> IntFFTCore:
    """Forward DIF core with its native stream orders (int_fftNk.vhd:15-21)."""
    return IntFFTCore(NFFT, DATA_WIDTH, TWDL_WIDTH, FORMAT, RNDMODE, XSER, "FWD", "HALVES", "BITREV",
                      USE_FLY, device, RAMB_TYPE, USE_MLT)


def int_ifftNk(NFFT=10, DATA_WIDTH=16, TWDL_WIDTH=16, FORMAT=1, RNDMODE=0, XSER="NEW", USE_FLY=1,
               RAMB_TYPE="WRAP", USE_MLT=False, device=None) -> IntFFTCore:
    """Inverse DIT core with its native stream orders (int_ifftNk.vhd:15-21)."""
    return IntFFTCore(NFFT, DATA_WIDTH, TWDL_WIDTH, FORMAT, RNDMODE, XSER, "INV", "BITREV", "HALVES",
                      USE_FLY, device, RAMB_TYPE, USE_MLT)


def int_fft_single_path(NFFT=10, DATA_WIDTH=16, TWDL_WIDTH=16, FORMAT=1, RNDMODE=0, XSERIES="NEW",
                        USE_FLY=1, RAMB_TYPE="CONT", USE_MLT=False, device=None) -> IntFFTCore:
    """inbuf_half_path -> int_fftNk -> outbuf_half_path -> int_bitrev_order: natural in, natural out
    (int_fft_single_path.vhd:157-268)."""
    return IntFFTCore(NFFT, DATA_WIDTH, TWDL_WIDTH, FORMAT, RNDMODE, XSERIES, "FWD", "NATURAL", "NATURAL",
                      USE_FLY, device, RAMB_TYPE, USE_MLT)


def int_fft_ifft_pair(NFFT=10, DATA_WIDTH=16, TWDL_WIDTH=16, FORMAT=1, RNDMODE=0, XSERIES="NEW",
                      USE_FLY=1, RAMB_TYPE="WRAP", USE_MLT=False, device=None) -> IntFFTCore:
    """iobuf -> int_fftNk -> int_ifftNk -> iobuf (int_fft_ifft_pair.vhd:161-330); outputs the correct
    (re, im) per lane, not the reference's mis-wired Q0_IM/Q1_RE (:332-335)."""
    return IntFFTCore(NFFT, DATA_WIDTH, TWDL_WIDTH, FORMAT, RNDMODE, XSERIES, "PAIR", "NATURAL", "NATURAL",
                      USE_FLY, device, RAMB_TYPE, USE_MLT)


def int_fft_2d(NFFT=20, NFFT1=10, DATA_WIDTH=16, TWDL_WIDTH=16, FORMAT=0, RNDMODE=0, XSER="NEW", direction="FWD",
               in_order="NATURAL", out_order="NATURAL", device=None) -> IntFFTCore:
    """N > 512K: the "2D-FFT scheme" int_fftNk.vhd:11-13 points to, as this library defines it (include/intfft.h,
    DESIGN.md section 4.5): 2^NFFT1-point cores over the columns, inter-pass twiddle, 2^(NFFT-NFFT1)-point cores over the rows."""
    return IntFFTCore(NFFT, DATA_WIDTH, TWDL_WIDTH, FORMAT, RNDMODE, XSER, direction, in_order, out_order, 1, device,
                      NFFT1=NFFT1)


def exec_sharded(cores, x, root: int = 0, transport: str = None, asynchronous: bool = False):
    """Single-process multi-GPU transform through intfft_exec_sharded: `cores` are IntFFTCore objects with identical
    generics (normally one per HIP device), `x` a [batch, N, 2] tensor on the device of cores[root].  Contiguous
    shards, remainder to the last cores, no collective.  transport: None (leave the plan set as it is: peer copies unless
    changed earlier), "peer" (hipMemcpyPeerAsync) or "rccl" (grouped ncclSend / ncclRecv over xGMI, one group each way;
    raises if RCCL cannot serve the set, e.g. two cores on one device).  Blocking -- or, with asynchronous=True,
    intfft_exec_sharded_async on torch's current stream of the root device: no host synchronisation, the result is ordered on that
    stream like any other kernel's."""
    import torch

    if not cores or not 0 <= root < len(cores):
        raise ValueError("bad cores/root")
    c0 = cores[root]
    if not (x.is_cuda and x.device.index == c0.device):
        raise ValueError("input must live on cuda:%d (the device of cores[root])" % c0.device)
    if len({id(c) for c in cores}) != len(cores):
        raise ValueError("every shard needs its own core (a plan owns its scratch)")
    if x.dtype != c0.in_dtype or x.dim() != 3 or x.shape[1] != c0.n or x.shape[2] != 2 or not x.is_contiguous():
        raise ValueError("input must be a contiguous [batch, %d, 2] %s tensor" % (c0.n, c0.in_dtype))
    y = torch.empty(c0.out_shape(x.shape[0]), dtype=c0.out_dtype, device=x.device)
    arr = (ctypes.c_void_p * len(cores))(*[c._plan for c in cores])
    if transport is not None:
        if transport not in ("peer", "rccl"):
            raise ValueError("transport must be 'peer' or 'rccl'")
        capi.check(capi.lib().intfft_shard_set_transport(arr, len(cores), root, capi.TRANSPORT_RCCL if transport == "rccl" else capi.TRANSPORT_PEER),
                   "intfft_shard_set_transport")
    if asynchronous:
        stream = torch.cuda.current_stream(x.device).cuda_stream
        capi.check(capi.lib().intfft_exec_sharded_async(arr, len(cores), root, x.data_ptr(), y.data_ptr(), x.shape[0], stream),
                   "intfft_exec_sharded_async")
        return y
    torch.cuda.synchronize(x.device)
    capi.check(capi.lib().intfft_exec_sharded(arr, len(cores), root, x.data_ptr(), y.data_ptr(), x.shape[0]),
               "intfft_exec_sharded")
    return y

