"""The N > 512K "2D-FFT scheme" extension (int_fftNk.vhd:11-13 names it, this project defines it: DESIGN.md section 4.5) in the
checker: the three C forms agree, C == the independent Python twin, the result is as close to the exact DFT as the
1-D core of the same length, and the inter-pass twiddle table has the documented structure."""
import numpy as np
import pytest

from oracle import oracle_c as C
from oracle import oracle_py as P
from tests.helpers import to_complex, to_list, uniform_frames

DIRS = {"FWD": (C.FWD, P.FWD), "INV": (C.INV, P.INV), "PAIR": (C.PAIR, P.PAIR)}
# (log2n, l1, dw, tw, fmt, rnd, new)
CASES = [(6, 3, 16, 16, 0, 0, True), (7, 3, 16, 16, 0, 1, True), (7, 4, 16, 16, 1, 0, True), (8, 4, 24, 24, 1, 0, True),
         (8, 5, 16, 16, 0, 0, False), (9, 3, 30, 16, 1, 0, True), (9, 6, 12, 10, 0, 0, True), (10, 5, 16, 18, 0, 0, True)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("direction", list(DIRS))
def test_three_forms_agree(case, direction):
    log2n, l1, dw, tw, fmt, rnd, new = case
    p = C.make_params(log2n, dw, tw, fmt, rnd, new)
    if C.lib().orc_validate_2d(p, l1, DIRS[direction][0]):
        pytest.skip("not elaboratable")
    x = uniform_frames(4, 1 << log2n, dw, 1000 + log2n)
    outs = [C.execute_2d(x, p, l1, DIRS[direction][0], form=f) for f in (0, 1, 2)]
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


@pytest.mark.parametrize("case", CASES[:6])
@pytest.mark.parametrize("direction", list(DIRS))
def test_c_equals_python_twin(case, direction):
    log2n, l1, dw, tw, fmt, rnd, new = case
    p = C.make_params(log2n, dw, tw, fmt, rnd, new)
    cd, pd = DIRS[direction]
    if C.lib().orc_validate_2d(p, l1, cd):
        pytest.skip("not elaboratable")
    x = uniform_frames(2, 1 << log2n, dw, 77)
    for io, oo in [(C.NATURAL, C.NATURAL), (C.HALVES, C.BITREV), (C.BITREV, C.BITREV_LANES)]:
        got = C.execute_2d(x, p, l1, cd, io, oo)
        for f in range(2):
            want = P.execute_2d(to_list(x[f]), log2n, l1, dw, tw, fmt, rnd, new, pd, io, oo)
            assert to_list(got[f]) == want, (case, direction, io, oo)


def test_inter_pass_twiddle_table():
    for log2n, t in [(6, 16), (8, 24), (10, 18), (12, 16)]:
        n = 1 << log2n
        mg = 2 ** (t - 1) - 1 if t < 18 else 2 ** (t - 2) - 1
        tab = np.array([C.twiddle_2d(log2n, t, m) for m in range(n)], dtype=np.float64)
        assert [tuple(map(int, r)) for r in tab] == [P.twiddle_2d(log2n, t, m) for m in range(n)]
        ideal = mg * np.exp(-2j * np.pi * np.arange(n) / n)
        assert np.abs(tab[:, 0] + 1j * tab[:, 1] - ideal).max() <= 0.7072  # each component within half an LSB
        assert tuple(tab[0]) == (mg, 0) and tuple(tab[n // 4]) == (0, -mg) and tuple(tab[n // 2]) == (-mg, 0)
        # the first half circle at N = 2 * 2^s resolution is the stage-s ROM table wherever that table has no Taylor step
        if log2n - 1 < 11:
            re, im = C.twiddles(log2n - 1, t)
            assert np.array_equal(tab[: n // 2, 0], re) and np.array_equal(tab[: n // 2, 1], im)


@pytest.mark.parametrize("log2n,l1", [(10, 5), (12, 4), (14, 7), (16, 8)])
def test_close_to_exact_dft_like_the_1d_core(log2n, l1):
    """Scaled 16/16: max |X - fft(x)/N| of the 2-D scheme stays within log2N + 2 LSB (the 1-D core: log2N + 1,
    tests/test_oracle.py); unscaled: relative rms error below 3e-4 like the 1-D core."""
    n = 1 << log2n
    x = uniform_frames(2, n, 15, 5 + log2n)
    ref = np.fft.fft(to_complex(x), axis=1)
    p = C.make_params(log2n, 16, 16, 0, 0, True)
    y2 = to_complex(C.execute_2d(x, p, l1, C.FWD))
    y1 = to_complex(C.execute(x, p, C.FWD))
    e2, e1 = np.abs(y2 - ref / n).max(), np.abs(y1 - ref / n).max()
    assert e2 <= log2n + 2 and e1 <= log2n + 1
    pu = C.make_params(log2n, 16, 16, 1, 0, True)
    yu = to_complex(C.execute_2d(x, pu, l1, C.FWD))
    assert np.sqrt(np.mean(np.abs(yu - ref) ** 2)) / np.sqrt(np.mean(np.abs(ref) ** 2)) < 3e-4
    # pair: x -> X/N -> x/N (scaled): the same truncation floor as the 1-D pair (the IFFT's first stages see X/N ~ a few LSB)
    back = to_complex(C.execute_2d(x, p, l1, C.PAIR))
    back1 = to_complex(C.execute(x, p, C.PAIR))
    assert np.abs(back - to_complex(x) / n).max() <= max(log2n + 2, np.abs(back1 - to_complex(x) / n).max() + 2)
