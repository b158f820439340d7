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
  *.hex           the full-width form of the two layouts above for the cross-check kit's hex testbenches
                  (tools/vivado_crosscheck/tb_single_hex.vhd / tb_pair_hex.vhd): the same columns, every value a two's-complement
                  word of 4 * ceil(width / 4) bits in upper-case hex -- what ieee.std_logic_textio hread / hwrite (the package
                  fft_signle_test.vhd:73 imports) read and write, with no 32-bit limit of a VHDL integer
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


# ---- full-width hexadecimal text (cross-check kit, widths beyond a VHDL integer) ------------------------------------------------
def hex_digits(width: int) -> int:
    """hex digits of one word: hread / hwrite work on vectors whose length is a multiple of 4"""
    return (width + 3) // 4


def write_hex(path: str, table, width: int) -> None:
    """table: [lines, columns] of integers (int64 or Python ints) that fit `width` bits two's complement -> one line per row,
    every value sign-extended to 4 * ceil(width / 4) bits and printed as upper-case hex"""
    nd = hex_digits(width)
    mask = (1 << (4 * nd)) - 1
    lo, hi = -(1 << (width - 1)), (1 << (width - 1)) - 1
    with open(path, "w") as fh:
        for row in table:
            vals = [int(v) for v in row]
            if any(v < lo or v > hi for v in vals):
                raise ValueError("value outside %d bits" % width)
            fh.write(" ".join("%0*X" % (nd, v & mask) for v in vals) + "\n")


def read_hex(path: str, width: int) -> np.ndarray:
    """-> int64 [lines, columns] (widths up to 64 bits): every word read as two's complement of its OWN digit count
    (a dump sign-extended to the digit boundary, like tb_*_hex.vhd writes it), then checked against `width`"""
    rows = []
    with open(path) as fh:
        for line in fh:
            f = line.split()
            if not f:
                continue
            vals = []
            for w in f:
                v = int(w, 16)
                bits = 4 * len(w)
                if v >> (bits - 1):
                    v -= 1 << bits
                if v < -(1 << (width - 1)) or v >= (1 << (width - 1)):
                    raise ValueError("%s: %s does not fit %d bits" % (path, w, width))
                vals.append(v)
            rows.append(vals)
    return np.array(rows, dtype=np.int64)


def single_to_table(frames) -> np.ndarray:
    return _ints(frames).reshape(-1, 2)


def double_to_table(frames) -> np.ndarray:
    x = _ints(frames).reshape(-1, 2, 2)  # [beat, lane, (re, im)]
    return np.stack([x[:, 0, 0], x[:, 1, 0], x[:, 0, 1], x[:, 1, 1]], axis=-1)


def table_to_double(a: np.ndarray, n: int) -> np.ndarray:
    """[beats, 4] (lane-0 re, lane-1 re, lane-0 im, lane-1 im) -> [frames, n, 2] in natural order"""
    d0 = np.stack([a[:, 0], a[:, 2]], axis=-1)
    d1 = np.stack([a[:, 1], a[:, 3]], axis=-1)
    return np.stack([d0, d1], axis=1).reshape(-1, n, 2)
