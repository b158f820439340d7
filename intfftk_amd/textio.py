"""Text-file compatibility with the reference testbenches, so that the owner of
a Vivado install can cross-check GPU results against a real RTL simulation -- the only external pin this
path can ever get.  Pure host-side I/O; no arithmetic beyond the bit slicing the testbench itself does.

  di_single.dat   two integers per line, (re, im), one sample per line in serial (natural) order
                  -- read by src/vhdl/tb/fft_signle_test.vhd:139-166
  di_double.dat   four integers per line: D0_RE D1_RE D0_IM D1_IM, one 2-lane beat per line; the pair
                  wrapper takes the interleave-2 stream (lane 0 = even samples, lane 1 = odd samples,
                  src/vhdl/buffers/iobuf_flow_int2.vhd:18-40) -- read by src/vhdl/tb/fft_double_test.vhd:127-165
  dout_pair.dat   four integers per line: the TOP 17 BITS of Q0_RE Q1_RE Q0_IM Q1_IM
                  -- written by src/vhdl/tb/fft_double_test.vhd:200-217
"""
from __future__ import annotations

import numpy as np


def read_di_single(path: str, n: int | None = None) -> np.ndarray:
    """-> int64 [frames, n, 2] (or [samples, 2] when n is None)."""
    a = np.loadtxt(path, dtype=np.int64, ndmin=2)
    if a.shape[1] != 2:
        raise ValueError("di_single.dat holds two integers per line")
    if n is None:
        return a
    if a.shape[0] % n:
        raise ValueError("%d samples are not a whole number of %d-point frames" % (a.shape[0], n))
    return a.reshape(-1, n, 2)


def _ints(a) -> np.ndarray:
    """int64, or an object array of Python ints left as it is (results beyond 64 bits: engine.wide_to_int)."""
    a = np.asarray(a)
    return a if a.dtype == object else a.astype(np.int64)


def write_di_single(path: str, frames: np.ndarray) -> None:
    np.savetxt(path, _ints(frames).reshape(-1, 2), fmt="%d")


def read_di_double(path: str, n: int) -> np.ndarray:
    """-> int64 [frames, n, 2] in natural order (beat i carries samples 2i and 2i+1)."""
    a = np.loadtxt(path, dtype=np.int64, ndmin=2)
    if a.shape[1] != 4:
        raise ValueError("di_double.dat holds four integers per line")
    if a.shape[0] % (n // 2):
        raise ValueError("%d beats are not a whole number of %d-point frames" % (a.shape[0], n))
    d0 = np.stack([a[:, 0], a[:, 2]], axis=-1)  # (re, im) of lane 0
    d1 = np.stack([a[:, 1], a[:, 3]], axis=-1)
    x = np.stack([d0, d1], axis=1).reshape(-1, 2)  # beat-major interleave = natural order
    return x.reshape(-1, n, 2)


def write_di_double(path: str, frames: np.ndarray) -> None:
    x = _ints(frames).reshape(-1, 2, 2)  # [beat, lane, (re, im)]
    np.savetxt(path, np.stack([x[:, 0, 0], x[:, 1, 0], x[:, 0, 1], x[:, 1, 1]], axis=-1), fmt="%d")


def top_bits(v: np.ndarray, width: int, keep: int = 17) -> np.ndarray:
    """q(width-1 downto width-keep) read as a signed integer (fft_double_test.vhd:208-214)."""
    v = _ints(v)
    if width <= keep:
        return v
    return (v >> (width - keep)).astype(np.int64)  # 17 bits: fits whatever the width was


def dout_pair_lines(frames: np.ndarray, width: int, reference_wiring: bool = False) -> np.ndarray:
    """[beats, 4] = Q0_RE Q1_RE Q0_IM Q1_IM (top 17 bits) of natural-order pair outputs.
    reference_wiring=True reproduces the slice mix-up of int_fft_ifft_pair.vhd:332-335 (Q0_IM carries the
    REAL part of lane 0, Q1_RE the IMAGINARY part of lane 1) for comparison with a dump of the unfixed RTL."""
    x = _ints(frames).reshape(-1, 2, 2)  # [beat, lane, (re, im)]
    q0_re, q0_im, q1_re, q1_im = x[:, 0, 0], x[:, 0, 1], x[:, 1, 0], x[:, 1, 1]
    if reference_wiring:
        q0_im, q1_re = q0_re, q1_im
    return np.stack([top_bits(q0_re, width), top_bits(q1_re, width), top_bits(q0_im, width),
                     top_bits(q1_im, width)], axis=-1)


def write_dout_pair(path: str, frames: np.ndarray, width: int, reference_wiring: bool = False) -> None:
    np.savetxt(path, dout_pair_lines(frames, width, reference_wiring), fmt="%d", delimiter="    ")


def read_dout_pair(path: str) -> np.ndarray:
    return np.loadtxt(path, dtype=np.int64, ndmin=2)
