"""CPU tests of the oracle itself (no GPU): the C restatement against its independent Python twin,
stream form against in-place form, the reference model's dataflow against numpy.fft, and the
algebraic identities SURVEY.md section 8c lists as the only things that pin this path."""
import numpy as np
import pytest

from oracle import oracle_c as C
from oracle import oracle_py as P
from tests.helpers import chirp_frame, edge_frames, to_complex, to_list, uniform_frames

# (log2n, data_width, twdl_width, format, rndmode, new) -- chosen to visit every multiplier regime
CONFIGS = [
    (3, 16, 16, 0, 0, True),
    (4, 16, 16, 0, 1, True),
    (5, 16, 16, 1, 0, True),
    (6, 16, 16, 0, 0, False),
    (5, 12, 10, 0, 0, True),
    (5, 16, 18, 0, 0, True),    # t = 18: half-amplitude twiddles (rom_twiddle_int.vhd:143-147)
    (5, 16, 17, 1, 0, True),
    (5, 24, 24, 1, 0, True),    # sngl25 impossible (w >= 25) -> dbl35
    (4, 14, 24, 1, 0, True),    # sngl25 (w < 19)
    (6, 32, 24, 1, 0, True),    # dbl35 -> trpl52
    (6, 32, 24, 1, 0, False),
    (5, 27, 16, 1, 0, True),    # sngl -> dbl18 at w = 28
    (5, 25, 16, 1, 0, False),   # OLD: dbl18 from w = 26
    (5, 30, 16, 0, 1, True),    # dbl18, scaled round
    (5, 44, 16, 1, 0, True),    # trpl18
    (4, 42, 16, 1, 0, False),   # OLD: trpl18 from w = 43
    (5, 8, 8, 0, 0, True),
    (5, 64, 16, 0, 1, True),    # 64-bit data, round mode: the exact 65-bit sums (the C oracle's int64 rhu2 overflowed here until round 3)
    (4, 64, 16, 0, 0, True),
]


def _regimes(cfg):
    log2n, dw, tw, fmt, rnd, new = cfg
    return {C.cmult_regime(dw + ii * fmt + fmt, tw, new) for ii in range(log2n - 2)}


def test_configs_visit_every_regime():
    seen = set()
    for cfg in CONFIGS:
        seen |= _regimes(cfg)
    assert {"sngl", "dbl18", "trpl18", "sngl25", "dbl35", "trpl52"} <= seen


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("direction", [P.FWD, P.INV, P.PAIR])
def test_c_oracle_equals_python_twin(cfg, direction):
    log2n, dw, tw, fmt, rnd, new = cfg
    if direction == P.PAIR and dw + 2 * fmt * log2n > 64:
        pytest.skip("pair output wider than 64 bits")
    if direction == P.PAIR and fmt and C.cmult_regime(dw + 2 * log2n - 1, tw, new) is None:
        pytest.skip("inverse core not elaboratable at these widths")
    n = 1 << log2n
    x = np.concatenate([uniform_frames(2, n, dw, 1234 + log2n), edge_frames(n, dw)[[1, 3, 5]]])
    p = C.make_params(log2n, dw, tw, fmt, rnd, new)
    for in_o, out_o in [(P.NATURAL, P.NATURAL), (P.HALVES, P.BITREV), (P.BITREV_LANES, P.HALVES)]:
        got = C.execute(x, p, direction, in_o, out_o, form=0)
        got_ip = C.execute(x, p, direction, in_o, out_o, form=1)
        assert np.array_equal(got, got_ip), "stream form != in-place form"
        for f in range(x.shape[0]):
            want = P.execute(to_list(x[f]), log2n, dw, tw, fmt, rnd, new, direction, in_o, out_o)
            assert to_list(got[f]) == want, (cfg, direction, in_o, out_o, f)


@pytest.mark.parametrize("log2n", [3, 7, 10, 12, 13])
@pytest.mark.parametrize("mode", [(0, 0), (0, 1), (1, 0)])
def test_stream_form_equals_inplace_form(log2n, mode):
    """The commutator network (int_delay_line == fn_rev2rdx) only re-orders: the flat in-place
    indexing of SURVEY.md section 9.1 must give identical bits (includes Taylor stages at N = 4096+)."""
    fmt, rnd = mode
    n = 1 << log2n
    x = np.concatenate([uniform_frames(3, n, 16, 99), edge_frames(n, 16)])
    p = C.make_params(log2n, 16, 16, fmt, rnd, True)
    for direction in (C.FWD, C.INV, C.PAIR):
        a = C.execute(x, p, direction, form=0)
        b = C.execute(x, p, direction, form=1)
        assert np.array_equal(a, b)


def test_reference_float_model_is_the_dft():
    """fn_radix2.m restated in double precision equals fft() / N*ifft() (test_fft_radix2.m:89-110,
    which only plots).  This pins the ordering conventions the integer oracle shares."""
    for n in (8, 128, 1024):
        x = to_complex(chirp_frame(n))
        fwd = P.fn_radix2_float(list(x), n, "FWD")
        assert np.allclose(fwd, np.fft.fft(x), atol=1e-7 * n)
        inv = P.fn_radix2_float(list(fwd), n, "INV")
        assert np.allclose(inv, n * x, atol=1e-6 * n)


@pytest.mark.parametrize("log2n", [7, 10])
def test_c1_chirp_close_to_fft(log2n):
    """BASELINE config 1: the chirp of test_fft_radix2.m through the 16/16 scaled core, against
    numpy.fft.fft(x)/N.  Tolerance: < 1 LSB of rounding noise per stage (log2n LSB worst case;
    the survey's scratch figure is 7.5 LSB max at N = 1024)."""
    n = 1 << log2n
    x = chirp_frame(n) * 64
    p = C.make_params(log2n, 16, 16, 0, 0, True)
    got = to_complex(C.execute(x[None], p)[0])
    want = np.fft.fft(to_complex(x)) / n
    err = np.abs(got - want)
    assert err.max() < log2n + 1, err.max()
    assert np.sqrt((err ** 2).mean()) < 3.0
    # unscaled: relative error of the full-growth result
    p1 = C.make_params(log2n, 16, 16, 1, 0, True)
    got1 = to_complex(C.execute(x[None], p1)[0])
    want1 = np.fft.fft(to_complex(x))
    rel = np.sqrt((np.abs(got1 - want1) ** 2).mean()) / np.sqrt((np.abs(want1) ** 2).mean())
    assert rel < 1e-3, rel


def test_roundtrip_pair_close_to_identity():
    n, log2n = 256, 8
    x = uniform_frames(4, n, 14, 7)
    ps = C.make_params(log2n, 16, 16, 0, 0, True)
    y = C.execute(x, ps, C.PAIR)                     # scaled: ~ x / N
    assert np.abs(y - x / n).max() < log2n + 2
    pu = C.make_params(log2n, 16, 16, 1, 0, True)
    z = C.execute(x, pu, C.PAIR)                     # unscaled: ~ N * x (twiddle gain (1-2^-15)^2 per pair)
    rel = np.abs(z - n * x).max() / (n * (1 << 13))
    assert rel < 2e-3, rel


def test_impulse_reads_out_twiddle_table():
    """x = delta[n-1] through the unscaled DIF core: X[k] = amp * W_N^k up to per-stage floor --
    exactly reproduced by the first-stage table for k < N/2 (first butterfly: D = -amp ... )."""
    log2n, n, amp = 6, 64, 1 << 13
    x = np.zeros((1, n, 2), dtype=np.int64)
    x[0, 1, 0] = amp
    p = C.make_params(log2n, 16, 16, 1, 0, True)
    got = to_complex(C.execute(x, p)[0])
    want = amp * np.exp(-2j * np.pi * np.arange(n) / n)
    assert np.abs(got - want).max() < 2.5


def test_dc_goes_to_bin_zero():
    log2n, n = 7, 128
    x = np.zeros((1, n, 2), dtype=np.int64)
    x[0, :, 0] = 1000
    x[0, :, 1] = -300
    pu = C.make_params(log2n, 16, 16, 1, 0, True)
    y = C.execute(x, pu)[0]
    assert tuple(y[0]) == (1000 * n, -300 * n)
    assert np.abs(y[1:]).max() == 0
    ps = C.make_params(log2n, 16, 16, 0, 0, True)
    y = C.execute(x, ps)[0]
    assert tuple(y[0]) == (1000, -300)
    assert np.abs(y[1:]).max() == 0


@pytest.mark.parametrize("fmt", [0, 1])
def test_use_fly_zero_is_a_pure_permutation(fmt):
    """USE_FLY = '0' (int_fftNk.vhd:260-277): only the commutators act.  FFT: memory in BITREV
    order equals the input (X[rev n] = x[n]); unscaled mode zero-extends."""
    log2n, n = 6, 64
    x = uniform_frames(2, n, 16, 5)
    p = C.make_params(log2n, 16, 16, fmt, 0, True, use_fly=0)
    for form in (0, 1):
        y = C.execute(x, p, C.FWD, C.NATURAL, C.BITREV, form=form)
        want = x if not fmt else (x & 0xFFFF)
        assert np.array_equal(y, want)
        z = C.execute(x, p, C.PAIR, form=form)
        assert np.array_equal(z, want)


def test_negation_quirk_visible():
    """STAGE = 1 multiplies by -j with not(x) for negative x (int_dif2_fly.vhd:280-304):
    an impulse of -1000 gives X[N/4] = (0, 999), not (0, 1000) (SURVEY.md section 8c KAT)."""
    x = np.zeros((1, 16, 2), dtype=np.int64)
    x[0, 1, 0] = -1000
    p = C.make_params(4, 16, 16, 1, 0, True)
    y = C.execute(x, p)[0]
    assert tuple(y[4]) == (0, 999) and tuple(y[12]) == (0, -999)


def test_validate_rejects_non_elaboratable():
    assert C.lib().orc_validate(C.make_params(10, 16, 16, 1, 1, True), C.FWD) != 0   # FORMAT=1+RND
    assert C.lib().orc_validate(C.make_params(10, 16, 28, 0, 0, True), C.FWD) != 0   # TWD >= 28
    assert C.lib().orc_validate(C.make_params(10, 16, 26, 0, 0, False), C.FWD) != 0  # OLD: TWD >= 26
    assert C.lib().orc_validate(C.make_params(10, 60, 24, 0, 0, True), C.FWD) != 0   # w >= 53 at t > 18
    assert C.lib().orc_validate(C.make_params(10, 16, 16, 0, 0, True), C.FWD) == 0
    assert C.lib().orc_validate(C.make_params(19, 16, 16, 1, 0, True), C.PAIR) == 0


def test_i16_entry_matches_i64_entry():
    log2n, n = 10, 1024
    x = uniform_frames(8, n, 16, 3)
    p = C.make_params(log2n, 16, 16, 0, 0, True)
    a = C.execute(x, p)
    b = C.execute_i16(x.astype(np.int16), p)
    assert np.array_equal(a, b.astype(np.int64))


def test_trpl18_a_port_and_product_slice():
    """int_cmult_trpl18_dsp48: the A port is SXT(M_AA, 61 | 59) -- an operand beyond that is CUT to its low bits (:161-162) -- and
    the product slice P(MAW+MBW-2 downto MBW-1) must lie inside the 79 | 77 bits of P (:151-152), i.e. MAW + MBW <= 80 | 78.
    Both restatements agree on the bound and on the cut (full-range operands differ from the uncut product)."""
    assert C.cmult_regime(64, 16, True) == P.cmult_regime(64, 16, True) == "trpl18"
    assert C.cmult_regime(65, 16, True) is None and P.cmult_regime(65, 16, True) is None
    assert C.cmult_regime(62, 16, False) == P.cmult_regime(62, 16, False) == "trpl18"
    assert C.cmult_regime(63, 16, False) is None and P.cmult_regime(63, 16, False) is None
    assert C.cmult_regime(72, 8, True) == P.cmult_regime(72, 8, True) == "trpl18"
    rng = np.random.default_rng(61)
    differs = 0
    for new, w, t in ((True, 64, 16), (True, 62, 18), (False, 62, 16), (False, 60, 18), (True, 61, 16), (False, 59, 17)):
        awd = 61 if new else 59
        for _ in range(200):
            d = [int(v) for v in rng.integers(-(1 << (w - 1)), (1 << (w - 1)) - 1, size=2, endpoint=True)]
            ww = [int(v) for v in rng.integers(-(1 << (t - 1)) + 1, (1 << (t - 1)) - 1, size=2, endpoint=True)]
            got = C.cmult(d[0], d[1], ww[0], ww[1], w, t, new)
            assert got == P.cmult(d[0], d[1], ww[0], ww[1], w, t, new)
            exact = (P.sgn((d[0] * ww[0] >> (t - 1)) - (d[1] * ww[1] >> (t - 1)), w), P.sgn((d[0] * ww[1] >> (t - 1)) + (d[1] * ww[0] >> (t - 1)), w))
            if w <= awd:
                assert got == exact  # inside the port: the plain truncated product
            else:
                differs += got != exact
    assert differs > 100
