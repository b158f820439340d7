"""int_fftNk with FORMAT = 1 (full bit growth) at N = 2^17 .. 2^20 (round 5, csrc/intfft_widelong.hip): 16-bit ADC data through a long
unscaled core -- 33 .. 36-bit results in int64 containers -- and wider data while DATA_WIDTH + NFFT <= 40.  Three launches (k_wide_pre:
STAGE NFFT-1 .. 16 on int32; k_wide16_p1 / p2 on 2^16-point blocks, the second pass taking its rows across the blocks) against the CPU
oracle, bit for bit, and against the generic k_pass<int64> plan they replace (src/vhdl/fft/int_fftNk.vhd:184-342)."""
import numpy as np
import pytest

from oracle import oracle_c as C
from tests.helpers import edge_frames, uniform_frames
from tests.test_gpu_parity import check, run_gpu

pytestmark = pytest.mark.gpu

NAME = "k_wide_pre+k_wide16_p1+p2"


@pytest.mark.parametrize("log2n,dw,tw,batch", [(17, 16, 16, 3), (17, 16, 24, 1), (18, 16, 16, 2), (18, 15, 18, 1), (19, 16, 16, 1), (19, 14, 16, 2),
                                               (20, 16, 16, 1), (20, 13, 24, 1), (17, 20, 16, 2), (17, 23, 24, 1), (18, 22, 16, 1), (19, 21, 18, 1),
                                               (20, 20, 16, 1), (17, 18, 16, 5)])
def test_long_unscaled_three_launches(log2n, dw, tw, batch, monkeypatch):
    n = 1 << log2n
    x = np.concatenate([uniform_frames(batch, n, dw, 700 + log2n + dw), edge_frames(n, dw)[[0, 4]]])[:batch + (1 if log2n < 19 else 0)]
    info = check(x, log2n, dw, tw, 1, 0, True)
    assert info["kernel_name"] == NAME and info["n_passes"] == 3 and info["out_container"] == 8 and info["out_bits"] == dw + log2n, info
    assert info["in_container"] == (2 if dw <= 16 else 4), info
    if log2n <= 18:
        a, _ = run_gpu(x, log2n, dw, tw, 1, 0, True)
        monkeypatch.setenv("INTFFT_NO_WIDELONG", "1")
        b, ib = run_gpu(x, log2n, dw, tw, 1, 0, True)
        assert ib["kernel_name"] != NAME and np.array_equal(a, b), ib


def test_long_unscaled_old_twiddle_series_and_full_scale():
    """XSER = OLD tables, and frames of full-scale samples (every guard bit in use at every stage)."""
    log2n, dw, tw = 17, 16, 16
    n = 1 << log2n
    rng = np.random.default_rng(5)
    x = rng.choice(np.array([-(1 << 15), (1 << 15) - 1], dtype=np.int64), size=(2, n, 2))
    if C.lib().orc_validate(C.make_params(log2n, dw, tw, 1, 0, False), C.FWD) == 0:
        info = check(x, log2n, dw, tw, 1, 0, False)
        assert info["kernel_name"] == NAME, info
    info = check(x, log2n, dw, tw, 1, 0, True)
    assert info["kernel_name"] == NAME, info


def test_long_unscaled_chunks_on_two_streams(monkeypatch):
    """Several scratch chunks per call (alternating between the caller's stream and the pooled side stream) = one chunk = the oracle."""
    log2n, dw, tw = 17, 16, 16
    n = 1 << log2n
    x = uniform_frames(7, n, dw, 41)
    monkeypatch.setenv("INTFFT_SCRATCH_MB", "4")  # 2 frames per chunk
    info = check(x, log2n, dw, tw, 1, 0, True)
    assert info["kernel_name"] == NAME, info
    a, _ = run_gpu(x, log2n, dw, tw, 1, 0, True)
    monkeypatch.delenv("INTFFT_SCRATCH_MB")
    b, _ = run_gpu(x, log2n, dw, tw, 1, 0, True)
    assert np.array_equal(a, b)


def test_long_unscaled_class_boundaries():
    """Outside the class the generic passes serve the plan: results of at most 32 bits (the int32 class), the BITREV_LANES order on either core."""
    _, info = run_gpu(uniform_frames(1, 1 << 17, 12, 3), 17, 12, 16, 1, 0, True)  # 29-bit results
    assert info["kernel_name"] != NAME, info
    x = uniform_frames(1, 1 << 17, 16, 4)
    info = check(x, 17, 16, 16, 1, 0, True, out_order="BITREV_LANES")
    assert info["kernel_name"] != NAME, info
    info = check(x, 17, 16, 16, 1, 0, True, direction="INV", in_order="BITREV_LANES")
    assert info["kernel_name"] not in (NAME, "k_wide16_q1+q2+k_wide_post"), info


@pytest.mark.parametrize("log2n,dw,tw", [(17, 16, 16), (18, 15, 16), (20, 16, 16), (17, 19, 16)])
def test_long_unscaled_int32_first_round_of_the_last_pass(log2n, dw, tw, monkeypatch):
    """STAGE 7 .. 4 on the int32 butterflies where their widths allow (DATA_WIDTH + NFFT - 4 <= 32: k_wide16_p2<.., R32>) = the same stages on
    64-bit words (INTFFT_NO_WIDELONG_R32) = the oracle.  (17, 19): 32-bit outputs of STAGE 4 exactly; wider data takes the 64-bit round by itself."""
    n = 1 << log2n
    x = np.concatenate([uniform_frames(1, n, dw, 900 + log2n + dw), edge_frames(n, dw)[[4]]])
    info = check(x, log2n, dw, tw, 1, 0, True)
    assert info["kernel_name"] == NAME, info
    a, _ = run_gpu(x, log2n, dw, tw, 1, 0, True)
    monkeypatch.setenv("INTFFT_NO_WIDELONG_R32", "1")
    b, ib = run_gpu(x, log2n, dw, tw, 1, 0, True)
    assert ib["kernel_name"] == NAME and np.array_equal(a, b), ib


NAME64 = "k_wide_pre+k_wide64_p1+k_wide16_p2"


@pytest.mark.parametrize("log2n,dw,tw,batch", [(17, 24, 24, 2), (17, 28, 16, 1), (18, 24, 16, 1), (19, 22, 12, 1), (20, 24, 16, 1), (17, 20, 12, 3),
                                               (20, 28, 16, 1), (18, 30, 16, 1), (17, 31, 8, 1)])
def test_long_unscaled_64_bit_first_pass(log2n, dw, tw, batch, monkeypatch):
    """DATA_WIDTH + NFFT - 8 > 32 (BASELINE config 3's 24-bit data at N = 2^17 .. 2^20) or twiddles below 16 bits: STAGE 15 .. 8 on 64-bit words
    too (k_wide64_p1 on the blocks, 16-byte scratch samples), results up to 48 bits."""
    n = 1 << log2n
    x = np.concatenate([uniform_frames(batch, n, dw, 800 + log2n + dw), edge_frames(n, dw)[[0, 4]]])[:batch + (1 if log2n < 19 else 0)]
    info = check(x, log2n, dw, tw, 1, 0, True)
    assert info["kernel_name"] == NAME64 and info["n_passes"] == 3 and info["out_container"] == 8 and info["in_container"] == 4, info
    if log2n <= 17:
        a, _ = run_gpu(x, log2n, dw, tw, 1, 0, True)
        monkeypatch.setenv("INTFFT_NO_WIDELONG", "1")
        b, ib = run_gpu(x, log2n, dw, tw, 1, 0, True)
        assert ib["kernel_name"] != NAME64 and np.array_equal(a, b), ib


def test_long_unscaled_64_bit_first_pass_chunks_on_two_streams(monkeypatch):
    log2n, dw, tw = 17, 24, 24
    x = uniform_frames(5, 1 << log2n, dw, 43)
    monkeypatch.setenv("INTFFT_SCRATCH_MB", "6")  # 2 frames per chunk (24 bytes of scratch per sample)
    info = check(x, log2n, dw, tw, 1, 0, True)
    assert info["kernel_name"] == NAME64, info


def test_long_unscaled_products_beyond_64_bits_stay_generic():
    """DATA_WIDTH 24 under 24-bit twiddles at N = 2^20: the multiplier inputs reach 43 bits, 43 + 24 > 64 -- no exact 64-bit product; the generic
    kernels (96-bit products) serve the plan, bit-exact."""
    x = uniform_frames(1, 1 << 20, 24, 6)
    info = check(x, 20, 24, 24, 1, 0, True)
    assert info["kernel_name"] not in (NAME, NAME64), info


# ---- widths within 32 bits (csrc/intfft_bigwlong.hip): k_bigw_pre + k_bigw_a<16> in place on the blocks + k_bigw_b ----------------------------------------
NAMEW = "k_bigw_pre+k_bigw_a/b"


@pytest.mark.parametrize("log2n,dw,tw,fmt,rnd,batch", [(17, 18, 18, 0, 0, 3), (17, 24, 24, 0, 1, 2), (18, 32, 24, 0, 0, 1), (18, 18, 16, 0, 1, 2),
                                                       (19, 24, 16, 0, 0, 1), (19, 20, 25, 0, 1, 1), (20, 18, 18, 0, 0, 1), (20, 32, 16, 0, 1, 1),
                                                       (17, 12, 16, 1, 0, 3), (18, 14, 16, 1, 0, 1), (19, 13, 24, 1, 0, 1), (20, 12, 16, 1, 0, 1),
                                                       (17, 8, 16, 0, 0, 2), (17, 6, 12, 0, 1, 2), (18, 17, 26, 0, 0, 1), (17, 15, 16, 1, 0, 1)])
def test_long_frames_within_32_bits(log2n, dw, tw, fmt, rnd, batch, monkeypatch):
    """Every mode (scaled-truncate, scaled-round, unscaled) whose widths stay within 32 bits at N = 2^17 .. 2^20: int16 and int32 containers on either
    side, single- and multi-DSP multiplier regimes, both twiddle series; bit-exact to the oracle and equal to the generic passes it replaces."""
    n = 1 << log2n
    x = np.concatenate([uniform_frames(batch, n, dw, 600 + log2n + dw), edge_frames(n, dw)[[0, 4]]])[:batch + (1 if log2n < 19 else 0)]
    for new in (True, False):
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, new), C.FWD) != 0:
            continue
        info = check(x, log2n, dw, tw, fmt, rnd, new)
        assert info["kernel_name"] == NAMEW and info["n_passes"] == 3, info
        if log2n >= 19:
            break
    if log2n <= 18:
        a, _ = run_gpu(x, log2n, dw, tw, fmt, rnd, True)
        monkeypatch.setenv("INTFFT_NO_BIGWLONG", "1")
        b, ib = run_gpu(x, log2n, dw, tw, fmt, rnd, True)
        assert ib["kernel_name"] != NAMEW and np.array_equal(a, b), ib


def test_long_frames_within_32_bits_several_chunks(monkeypatch):
    log2n, dw, tw = 17, 18, 18
    x = uniform_frames(7, 1 << log2n, dw, 44)
    monkeypatch.setenv("INTFFT_SCRATCH_MB", "2")  # 2 frames per chunk
    info = check(x, log2n, dw, tw, 0, 0, True)
    assert info["kernel_name"] == NAMEW, info


def test_long_frames_16_bit_scaled_keep_the_packed_kernels():
    _, info = run_gpu(uniform_frames(1, 1 << 17, 16, 2), 17, 16, 16, 0, 0, True)
    assert info["kernel_name"] != NAMEW and "k_big2p_a" in info["kernel_name"], info


NAMEWI = "k_bigw_qb/qa+k_bigw_post"


@pytest.mark.parametrize("log2n,dw,tw,fmt,rnd,batch", [(17, 18, 18, 0, 0, 3), (17, 24, 24, 0, 1, 2), (18, 32, 24, 0, 0, 1), (18, 18, 16, 0, 1, 2),
                                                       (19, 24, 16, 0, 0, 1), (19, 20, 25, 0, 1, 1), (20, 18, 18, 0, 0, 1), (20, 32, 16, 0, 1, 1),
                                                       (17, 12, 16, 1, 0, 3), (18, 14, 16, 1, 0, 1), (19, 13, 24, 1, 0, 1), (20, 12, 16, 1, 0, 1),
                                                       (17, 8, 16, 0, 0, 2), (17, 6, 12, 0, 1, 2), (18, 17, 26, 0, 0, 1), (17, 15, 16, 1, 0, 1)])
def test_long_frames_within_32_bits_inverse(log2n, dw, tw, fmt, rnd, batch, monkeypatch):
    """int_ifftNk on the same class: k_bigw_qb at L = NFFT (bit-reversed gather + DIT STAGE 0 .. 7), k_bigw_qa<16> in place on the 2^16-point blocks
    (STAGE 8 .. 15), k_bigw_post (STAGE 16 .. NFFT-1, natural order out)."""
    n = 1 << log2n
    x = np.concatenate([uniform_frames(batch, n, dw, 650 + log2n + dw), edge_frames(n, dw)[[0, 4]]])[:batch + (1 if log2n < 19 else 0)]
    for new in (True, False):
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, new), C.INV) != 0:
            continue
        info = check(x, log2n, dw, tw, fmt, rnd, new, direction="INV")
        assert info["kernel_name"] == NAMEWI and info["n_passes"] == 3, info
        if log2n >= 19:
            break
    if log2n <= 18:
        a, _ = run_gpu(x, log2n, dw, tw, fmt, rnd, True, direction="INV")
        monkeypatch.setenv("INTFFT_NO_BIGWLONG", "1")
        b, ib = run_gpu(x, log2n, dw, tw, fmt, rnd, True, direction="INV")
        assert ib["kernel_name"] != NAMEWI and np.array_equal(a, b), ib


@pytest.mark.parametrize("log2n,dw,tw,fmt,rnd", [(17, 18, 18, 0, 0), (18, 24, 16, 0, 1), (19, 12, 16, 1, 0), (20, 18, 18, 0, 0), (17, 8, 16, 0, 0), (18, 32, 24, 0, 0)])
@pytest.mark.parametrize("direction,time_o,freq_o", [("FWD", "HALVES", "BITREV"), ("FWD", "HALVES", "NATURAL"), ("FWD", "NATURAL", "BITREV"),
                                                     ("INV", "HALVES", "BITREV"), ("INV", "NATURAL", "BITREV"), ("INV", "HALVES", "NATURAL")])
def test_long_frames_within_32_bits_cores_own_orders(log2n, dw, tw, fmt, rnd, direction, time_o, freq_o):
    """int_fftNk takes HALVES beats and emits BITREV order, int_ifftNk the reverse (int_fftNk.vhd:15-21): HALVES on k_bigw_pre / k_bigw_post (the blocks
    b and b + B/2 of one position are adjacent samples: one access), BITREV on the NAT instantiations of k_bigw_b / k_bigw_qb at L = NFFT."""
    if log2n >= 19 and (time_o, freq_o) != ("HALVES", "BITREV"):
        pytest.skip("the mixed forms are covered at N = 2^17 / 2^18")
    n = 1 << log2n
    x = np.concatenate([uniform_frames(1, n, dw, 660 + log2n + dw), edge_frames(n, dw)[[4]]])[:2 if log2n < 19 else 1]
    in_o, out_o = (time_o, freq_o) if direction == "FWD" else (freq_o, time_o)
    info = check(x, log2n, dw, tw, fmt, rnd, True, direction=direction, in_order=in_o, out_order=out_o)
    assert info["kernel_name"] == (NAMEW if direction == "FWD" else NAMEWI), info


@pytest.mark.parametrize("log2n,dw,tw", [(17, 16, 16), (18, 16, 24), (19, 14, 16), (20, 16, 16), (17, 24, 24), (18, 28, 16), (20, 24, 16), (17, 20, 16)])
@pytest.mark.parametrize("in_o,out_o", [("HALVES", "BITREV"), ("HALVES", "NATURAL"), ("NATURAL", "BITREV")])
def test_long_unscaled_core_own_orders(log2n, dw, tw, in_o, out_o):
    """int_fftNk(NFFT = 17 .. 20, FORMAT = 1) as the RTL instantiates it -- HALVES beats in, BITREV order out (int_fftNk.vhd:15-21) -- on both width classes:
    one access per block pair in k_wide_pre, the NAT instantiations of k_wide16_p2 (rows of a unit across the blocks)."""
    if log2n >= 19 and (in_o, out_o) != ("HALVES", "BITREV"):
        pytest.skip("the mixed forms are covered at N = 2^17 / 2^18")
    n = 1 << log2n
    x = np.concatenate([uniform_frames(1, n, dw, 670 + log2n + dw), edge_frames(n, dw)[[4]]])[:2 if log2n < 19 else 1]
    info = check(x, log2n, dw, tw, 1, 0, True, in_order=in_o, out_order=out_o)
    assert info["kernel_name"] in (NAME, NAME64), info


NAMEI = "k_wide16_q1+q2+k_wide_post"


@pytest.mark.parametrize("log2n,dw,tw,batch", [(17, 16, 16, 3), (17, 16, 24, 1), (18, 16, 16, 2), (18, 15, 18, 1), (19, 16, 16, 1), (19, 17, 16, 2),
                                               (20, 16, 16, 1), (20, 20, 24, 1), (17, 20, 16, 2), (17, 23, 24, 1), (18, 22, 16, 1), (17, 18, 16, 5)])
def test_long_unscaled_inverse(log2n, dw, tw, batch, monkeypatch):
    """int_ifftNk with FORMAT = 1 at N = 2^17 .. 2^20 and results of 33 .. 40 bits: k_wide16_q1<16, ., XS> (the gather at L = NFFT, rows of a unit across the
    blocks; int16 or int32 containers in), k_wide16_q2<16> on the blocks, k_wide_post (STAGE 16 .. NFFT-1 on 64-bit words)."""
    n = 1 << log2n
    x = np.concatenate([uniform_frames(batch, n, dw, 750 + log2n + dw), edge_frames(n, dw)[[0, 4]]])[:batch + (1 if log2n < 19 else 0)]
    for new in (True, False):
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, 1, 0, new), C.INV) != 0:
            continue
        info = check(x, log2n, dw, tw, 1, 0, new, direction="INV")
        assert info["kernel_name"] == NAMEI and info["n_passes"] == 3 and info["out_container"] == 8, info
        if log2n >= 19:
            break
    if log2n <= 17:
        a, _ = run_gpu(x, log2n, dw, tw, 1, 0, True, direction="INV")
        monkeypatch.setenv("INTFFT_NO_WIDELONG", "1")
        b, ib = run_gpu(x, log2n, dw, tw, 1, 0, True, direction="INV")
        assert ib["kernel_name"] != NAMEI and np.array_equal(a, b), ib


def test_long_unscaled_inverse_chunks_on_two_streams(monkeypatch):
    log2n, dw, tw = 17, 16, 16
    x = uniform_frames(5, 1 << log2n, dw, 45)
    monkeypatch.setenv("INTFFT_SCRATCH_MB", "6")  # 2 frames per chunk (24 bytes of scratch per sample)
    info = check(x, log2n, dw, tw, 1, 0, True, direction="INV")
    assert info["kernel_name"] == NAMEI, info


NAMEI64 = "k_wide64_q1+k_wide16_q2+k_wide_post"


@pytest.mark.parametrize("log2n,dw,tw,batch", [(17, 24, 24, 2), (17, 28, 16, 1), (18, 24, 16, 1), (19, 25, 12, 1), (20, 24, 16, 1), (17, 20, 12, 3),
                                               (20, 28, 16, 1), (18, 30, 16, 1), (17, 31, 8, 1)])
def test_long_unscaled_inverse_64_bit_first_pass(log2n, dw, tw, batch, monkeypatch):
    """The inverse beyond class 1 (24-bit data at N = 2^17 .. 2^20, twiddles below 16 bits): every stage on 64-bit words -- k_wide64_q1<16, ., XS> +
    k_wide16_q2<16, IN64> on the blocks + k_wide_post; results up to 48 bits."""
    n = 1 << log2n
    x = np.concatenate([uniform_frames(batch, n, dw, 850 + log2n + dw), edge_frames(n, dw)[[0, 4]]])[:batch + (1 if log2n < 19 else 0)]
    info = check(x, log2n, dw, tw, 1, 0, True, direction="INV")
    assert info["kernel_name"] == NAMEI64 and info["n_passes"] == 3 and info["out_container"] == 8 and info["in_container"] == 4, info
    if log2n <= 17:
        a, _ = run_gpu(x, log2n, dw, tw, 1, 0, True, direction="INV")
        monkeypatch.setenv("INTFFT_NO_WIDELONG", "1")
        b, ib = run_gpu(x, log2n, dw, tw, 1, 0, True, direction="INV")
        assert ib["kernel_name"] != NAMEI64 and np.array_equal(a, b), ib


@pytest.mark.parametrize("log2n,dw,tw", [(17, 16, 16), (18, 16, 24), (19, 15, 16), (20, 16, 16), (17, 24, 24), (18, 28, 16), (20, 24, 16), (17, 20, 16)])
@pytest.mark.parametrize("in_o,out_o", [("BITREV", "HALVES"), ("NATURAL", "HALVES"), ("BITREV", "NATURAL")])
def test_long_unscaled_inverse_core_own_orders(log2n, dw, tw, in_o, out_o):
    """int_ifftNk(NFFT = 17 .. 20, FORMAT = 1) as the RTL instantiates it -- BITREV order in, HALVES beats out (int_ifftNk.vhd:15-21) -- on both width classes:
    the NAT instantiations of k_wide16_q1 / k_wide64_q1 at XS > 0 (int16 containers: one plane of packed samples through the exchange), one 32-byte access
    per block pair in k_wide_post."""
    if log2n >= 19 and (in_o, out_o) != ("BITREV", "HALVES"):
        pytest.skip("the mixed forms are covered at N = 2^17 / 2^18")
    n = 1 << log2n
    x = np.concatenate([uniform_frames(1, n, dw, 680 + log2n + dw), edge_frames(n, dw)[[4]]])[:2 if log2n < 19 else 1]
    info = check(x, log2n, dw, tw, 1, 0, True, direction="INV", in_order=in_o, out_order=out_o)
    assert info["kernel_name"] in (NAMEI, NAMEI64), info
