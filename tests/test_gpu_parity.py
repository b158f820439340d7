"""GPU parity tests proper: the HIP path, called through the C-ABI (libintfft.so), against the CPU
oracle on the same seeded inputs.  Bit-exact (integer path): np.array_equal, no tolerance."""
import numpy as np
import pytest

from oracle import oracle_c as C
from tests.helpers import chirp_frame, edge_frames, uniform_frames

pytestmark = pytest.mark.gpu

ORD = {"NATURAL": C.NATURAL, "BITREV": C.BITREV, "HALVES": C.HALVES, "BITREV_LANES": C.BITREV_LANES}
DIR = {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}


def run_gpu(x, log2n, dw, tw, fmt, rnd, new, direction="FWD", in_order="NATURAL", out_order="NATURAL",
            use_fly=1):
    import torch

    from intfftk_amd import IntFFTCore

    core = IntFFTCore(log2n, dw, tw, fmt, rnd, "NEW" if new else "OLD", direction, in_order, out_order, use_fly)
    xin = torch.from_numpy(np.ascontiguousarray(x.astype({2: np.int16, 4: np.int32, 8: np.int64}[core.in_container])))
    y = core(xin.cuda())
    torch.cuda.synchronize()
    info = core.info
    core.close()
    return y.cpu().numpy().astype(np.int64), info


def run_ref(x, log2n, dw, tw, fmt, rnd, new, direction="FWD", in_order="NATURAL", out_order="NATURAL",
            use_fly=1):
    p = C.make_params(log2n, dw, tw, fmt, rnd, new, use_fly)
    return C.execute(x, p, DIR[direction], ORD[in_order], ORD[out_order], form=1)


def check(x, *cfg, **kw):
    got, info = run_gpu(x, *cfg, **kw)
    want = run_ref(x, *cfg, **kw)
    assert got.shape == want.shape
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError("GPU != oracle for %r %r: %d mismatches, first at %r: got %r want %r"
                             % (cfg, kw, len(bad), bad[0], got[tuple(bad[0])], want[tuple(bad[0])]))
    return info


# (log2n, dw, tw, fmt, rnd, new): every multiplier regime, every rounding mode, both XSER
CONFIGS = [
    (3, 16, 16, 0, 0, True), (4, 16, 16, 0, 1, True), (5, 16, 16, 1, 0, True), (6, 16, 16, 0, 0, False),
    (5, 12, 10, 0, 0, True), (5, 16, 18, 0, 0, True), (5, 16, 17, 1, 0, True), (5, 24, 24, 1, 0, True),
    (4, 14, 24, 1, 0, True), (6, 32, 24, 1, 0, True), (6, 32, 24, 1, 0, False), (5, 27, 16, 1, 0, True),
    (5, 25, 16, 1, 0, False), (5, 30, 16, 0, 1, True), (5, 44, 16, 1, 0, True), (4, 42, 16, 1, 0, False),
    (5, 8, 8, 0, 0, True), (7, 32, 16, 0, 0, True), (7, 32, 16, 0, 1, True), (8, 40, 24, 0, 0, True),
]


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("direction", ["FWD", "INV", "PAIR"])
def test_every_regime_and_mode(cfg, direction):
    log2n, dw, tw, fmt, rnd, new = cfg
    p = C.make_params(log2n, dw, tw, fmt, rnd, new)
    if C.lib().orc_validate(p, DIR[direction]) != 0:
        pytest.skip("not elaboratable")
    n = 1 << log2n
    x = np.concatenate([uniform_frames(37, n, dw, 77 + log2n), edge_frames(n, dw)])
    check(x, *cfg, direction=direction)


@pytest.mark.parametrize("in_order", list(ORD))
@pytest.mark.parametrize("out_order", list(ORD))
@pytest.mark.parametrize("direction", ["FWD", "INV", "PAIR"])
def test_io_orders(in_order, out_order, direction):
    x = uniform_frames(5, 256, 16, 11)
    check(x, 8, 16, 16, 0, 0, True, direction=direction, in_order=in_order, out_order=out_order)


@pytest.mark.parametrize("log2n", list(range(3, 14)))
@pytest.mark.parametrize("mode", [(0, 0), (0, 1), (1, 0)])
def test_all_single_pass_lengths(log2n, mode):
    fmt, rnd = mode
    n = 1 << log2n
    x = np.concatenate([uniform_frames(9, n, 16, log2n), edge_frames(n, 16)])
    for direction in ("FWD", "INV", "PAIR"):
        check(x, log2n, 16, 16, fmt, rnd, True, direction=direction)


@pytest.mark.parametrize("log2n,dw,tw,fmt", [(15, 16, 16, 0), (15, 16, 16, 1), (16, 24, 24, 1), (16, 24, 16, 1),
                                              (13, 24, 24, 1), (17, 16, 16, 0), (14, 20, 16, 0), (17, 20, 16, 1), (18, 18, 24, 1)])
@pytest.mark.parametrize("direction", ["FWD", "INV", "PAIR"])
def test_multi_pass_lengths(log2n, dw, tw, fmt, direction):
    """N beyond one LDS tile: strided + contiguous passes through plan scratch; Taylor twiddles."""
    if C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, 0, True), DIR[direction]) != 0:
        pytest.skip("not elaboratable")
    n = 1 << log2n
    x = uniform_frames(3, n, dw, 1000 + log2n)
    for in_o, out_o in [("NATURAL", "NATURAL"), ("HALVES", "BITREV")]:
        info = check(x, log2n, dw, tw, fmt, 0, True, direction=direction, in_order=in_o, out_order=out_o)
        assert info["n_passes"] >= 2


@pytest.mark.parametrize("env", ["INTFFT_NO_MIXED_WORDS", "INTFFT_NO_NARROW_MUL"])
def test_wide_plan_variants_agree(env, monkeypatch):
    """64-bit plans: int32 leading passes / single-int64 products are optimisations of the same arithmetic --
    switching either off gives the same bits (and both equal the oracle)."""
    x = uniform_frames(3, 1 << 16, 24, 4242)
    a, _ = run_gpu(x, 16, 24, 24, 1, 0, True)
    monkeypatch.setenv(env, "1")
    b, _ = run_gpu(x, 16, 24, 24, 1, 0, True)
    assert np.array_equal(a, b)
    assert np.array_equal(a, run_ref(x, 16, 24, 24, 1, 0, True))


@pytest.mark.parametrize("log2n,batch", [(13, 515), (13, 1024), (14, 259), (15, 3), (15, 130), (16, 5), (16, 64), (17, 3),
                                         (17, 9), (18, 5), (19, 3), (19, 2)])
def test_three_pass_kernels_n8192_to_n524288(log2n, batch):
    """N = 2^13 .. 2^19, 16-bit scaled: the three-pass kernels of BASELINE config 4 with frame groups forming
    virtual 2^16 / 2^20-point frames in pass 1 (partial last groups included), Taylor twiddles for STAGE >= 11."""
    n = 1 << log2n
    x = uniform_frames(batch, n, 15, 2000 + log2n)
    x[0] = uniform_frames(1, n, 16, 7)[0]  # one full-scale frame: exact extraction in its tiles
    info = check(x, log2n, 16, 16, 0, 0, True)
    if log2n <= 14:  # one pass since round 4
        assert info["kernel_name"] == "k_fft16k_i16" and info["n_passes"] == 1, info
    else:
        assert info["kernel_name"].startswith("k_big2") and info["n_passes"] == 2
    if batch <= 9:
        check(x, log2n, 16, 13, 0, 0, False)  # narrower twiddles, XSER "OLD"


@pytest.mark.parametrize("log2n,batch", [(19, 1), (19, 5), (19, 34), (20, 1), (20, 3), (20, 17)])
def test_two_pass_n2pow19_n2pow20(log2n, batch, monkeypatch):
    """N = 2^19, 2^20 forward (BASELINE config 4 is N = 2^20): 2^(L-10) rows x 1024 columns in two ten-stage passes on half
    lines (k_big2x_a / k_big2x_b, XCD-paired blocks), natural or HALVES order in; against the oracle and against the three-pass
    plan it replaces (INTFFT_NO_BIG2X); batches that do not fill the frame groups of the launch, a full-scale frame (exact
    extraction in its tiles), 13-bit twiddles with XSER "OLD", 12-bit data."""
    n = 1 << log2n
    x = uniform_frames(batch, n, 15, 2100 + log2n + batch)
    x[0] = uniform_frames(1, n, 16, 7)[0]
    got, info = run_gpu(x, log2n, 16, 16, 0, 0, True)
    assert info["kernel_name"] == "k_big2x_a/k_big2x_b" and info["n_passes"] == 2, info
    assert np.array_equal(got, run_ref(x, log2n, 16, 16, 0, 0, True))
    with monkeypatch.context() as m:
        m.setenv("INTFFT_NO_BIG2X", "1")
        got3, info3 = run_gpu(x, log2n, 16, 16, 0, 0, True)
        assert info3["kernel_name"] == "k_big20_p1/p2/p3" and info3["n_passes"] == 3, info3
    assert np.array_equal(got, got3)
    if batch <= 5:
        info = check(x, log2n, 16, 16, 0, 0, True, in_order="HALVES")
        assert info["n_passes"] == 2
        check(x[:2], log2n, 16, 13, 0, 0, False)
        info = check(x[:2] >> 4, log2n, 12, 16, 0, 0, True)
        assert info["kernel_name"] == "k_big2x_a/k_big2x_b", info


@pytest.mark.parametrize("direction", ["FWD", "INV"])
def test_two_pass_cores_own_orders_chunked_n2pow20(direction, monkeypatch):
    """N = 2^20 in the cores' own orders (int_fftNk: HALVES in / BITREV out; int_ifftNk: BITREV in / HALVES out) on the two-pass tiles,
    with a batch beyond one scratch half: the chunks alternate between two streams (32-frame halves) -- frames around the chunk borders
    against the oracle, the whole batch against the one-stream run and against the three-pass plan; a full-scale frame included."""
    n = 1 << 20
    x = uniform_frames(70, n, 15, 4100)
    x[33] = uniform_frames(1, n, 16, 17)[0]
    kw = dict(in_order="HALVES", out_order="BITREV") if direction == "FWD" else dict(direction="INV", in_order="BITREV", out_order="HALVES")
    got, info = run_gpu(x, 20, 16, 16, 0, 0, True, **kw)
    assert info["n_passes"] == 2 and "k_big2x" in info["kernel_name"], info
    sel = [0, 31, 32, 33, 63, 64, 69]
    assert np.array_equal(got[sel], run_ref(x[sel], 20, 16, 16, 0, 0, True, **kw))
    with monkeypatch.context() as m:
        m.setenv("INTFFT_ONE_STREAM", "1")
        got1, _ = run_gpu(x, 20, 16, 16, 0, 0, True, **kw)
    assert np.array_equal(got, got1)
    with monkeypatch.context() as m:
        m.setenv("INTFFT_NO_TWOPASS", "1")
        got3, info3 = run_gpu(x[:40], 20, 16, 16, 0, 0, True, **kw)
        assert info3["n_passes"] == 3, info3
    assert np.array_equal(got[:40], got3)


@pytest.mark.parametrize("case", [(18, 16, 16, 0, 0, "PAIR", 300), (18, 16, 16, 0, 0, "INV", 300), (16, 18, 24, 0, 0, "FWD", 1100)])
def test_two_stream_switch_other_multi_pass_families(case, monkeypatch):
    """INTFFT_TWO_STREAMS=1 (include/intfft.h): the chunk alternation between the caller's stream and the plan's side stream for the
    multi-pass families that keep one stream by default (k_big20_* / k_big2p_* / k_bigw_*).  A batch beyond one scratch half (128 MiB),
    so both streams carry chunks: same bits as the default plan, frames around the chunk border against the oracle."""
    log2n, dw, tw, fmt, rnd, direction, batch = case
    if dw != 16:
        monkeypatch.setenv("INTFFT_NO_WIDE16", "1")
    n = 1 << log2n
    x = uniform_frames(batch, n, dw, 5200 + log2n)
    got, info = run_gpu(x, log2n, dw, tw, fmt, rnd, True, direction=direction)
    assert info["n_passes"] >= 2, info
    with monkeypatch.context() as m:
        m.setenv("INTFFT_TWO_STREAMS", "1")
        got2, info2 = run_gpu(x, log2n, dw, tw, fmt, rnd, True, direction=direction)
    assert info2["kernel_name"] == info["kernel_name"] and info2["scratch_bytes"] >= info["scratch_bytes"], (info, info2)
    assert np.array_equal(got, got2)
    half = (128 << 20) // (n * 2 * (2 if dw == 16 else 4))
    sel = [0, half - 1, half, min(2 * half, batch - 1), batch - 1]
    assert np.array_equal(got2[sel], run_ref(x[sel], log2n, dw, tw, fmt, rnd, True, direction=direction))


@pytest.mark.parametrize("log2n", [17, 18])
@pytest.mark.parametrize("out_order", ["NATURAL", "BITREV"])
def test_two_pass_32_register_first_pass(log2n, out_order, monkeypatch):
    """N = 2^17, 2^18 forward from natural order: k_big2p_a (9 / 10 stages, 32 samples per thread, quarter-turn twiddle
    sharing) + the 256-point second pass.  Bit-exact to the oracle incl. full-scale frames (exact extraction in their tiles),
    odd batches, both XSER / a narrower twiddle width, and equal to the three-pass plan it replaces (INTFFT_NO_TWOPASS)."""
    n = 1 << log2n
    x = uniform_frames(7, n, 15, 900 + log2n)
    x[2] = uniform_frames(1, n, 16, 8)[0]
    x[5] = edge_frames(n, 16)[3]
    info = check(x, log2n, 16, 16, 0, 0, True, out_order=out_order)
    assert info["kernel_name"].startswith("k_big2p_a") and info["n_passes"] == 2, info
    check(x[:3], log2n, 16, 16, 0, 0, False, out_order=out_order)
    check(x[:3], log2n, 16, 12, 0, 0, True, out_order=out_order)
    monkeypatch.setenv("INTFFT_NO_TWOPASS", "1")
    got3, info3 = run_gpu(x, log2n, 16, 16, 0, 0, True, out_order=out_order)
    monkeypatch.delenv("INTFFT_NO_TWOPASS")
    got2, _ = run_gpu(x, log2n, 16, 16, 0, 0, True, out_order=out_order)
    assert info3["n_passes"] == 3 and np.array_equal(got2, got3)


@pytest.mark.parametrize("log2n", [17, 18])
@pytest.mark.parametrize("in_order,out_order", [("NATURAL", "NATURAL"), ("BITREV", "NATURAL"), ("NATURAL", "HALVES"), ("BITREV", "HALVES")])
def test_two_pass_32_register_inverse(log2n, in_order, out_order, monkeypatch):
    """N = 2^17, 2^18 inverse: k_mid_q1 | k_mid_c (STAGE 0..7) + k_big2p_q (STAGE 8..L-1, 32 samples per thread, quarter-turn
    sharing on DIT-packed twiddles).  Bit-exact to the oracle incl. full-scale and edge frames (exact extraction in their
    tiles), XSER "OLD", a narrower twiddle width, and equal to the three-pass plan it replaces (INTFFT_NO_TWOPASS)."""
    n = 1 << log2n
    x = uniform_frames(7, n, 15, 950 + log2n)
    x[1] = uniform_frames(1, n, 16, 18)[0]
    x[4] = edge_frames(n, 16)[3]
    x[6] = edge_frames(n, 16)[1]
    kw = dict(direction="INV", in_order=in_order, out_order=out_order)
    info = check(x, log2n, 16, 16, 0, 0, True, **kw)
    assert info["kernel_name"].endswith("k_big2p_q") and info["n_passes"] == 2, info
    check(x[:3], log2n, 16, 16, 0, 0, False, **kw)
    check(x[:3], log2n, 16, 12, 0, 0, True, **kw)
    got2, _ = run_gpu(x, log2n, 16, 16, 0, 0, True, **kw)
    monkeypatch.setenv("INTFFT_NO_TWOPASS", "1")
    got3, info3 = run_gpu(x, log2n, 16, 16, 0, 0, True, **kw)
    assert info3["n_passes"] == 3 and np.array_equal(got2, got3)


@pytest.mark.parametrize("log2n", [13, 14])
@pytest.mark.parametrize("direction", ["FWD", "INV", "PAIR"])
def test_single_pass_n8192_n16384(log2n, direction, monkeypatch):
    """N = 8192 / 16384, 16-bit scaled-truncate, natural order: ONE pass (k_fft16k_i16: a workgroup per frame, three register rounds
    around two LDS transposes, int_fftNk.vhd:75,184-207 / int_ifftNk.vhd:183-341) against the oracle and against the two-pass plan it
    replaces (INTFFT_NO_FAST16K); batches of one frame, fewer frames than workgroups, more frames than the resident grid; a
    full-scale frame (exact extraction), edge frames, XSER "OLD" with narrower twiddles, narrow data (DATA_WIDTH 12 and 9)."""
    n = 1 << log2n
    for batch in (1, 3, 1500 if log2n == 13 else 700):
        x = uniform_frames(batch, n, 15, 4100 + log2n + batch)
        x[0] = uniform_frames(1, n, 16, 13)[0]
        if batch > 2:
            x[1] = -(1 << 15)
            x[2] = 0
            x[2, 1] = (1 << 14, -(1 << 14))
        a, ia = run_gpu(x, log2n, 16, 16, 0, 0, True, direction=direction)
        assert ia["kernel_name"] == "k_fft16k_i16" and ia["n_passes"] == 1 and ia["fast_path"] == 1, ia
        sel = [0, 1, 2, batch - 1] if batch > 2 else list(range(batch))
        assert np.array_equal(a[sel], run_ref(x[sel], log2n, 16, 16, 0, 0, True, direction=direction))
        with monkeypatch.context() as m:
            m.setenv("INTFFT_NO_FAST16K", "1")
            b, ib = run_gpu(x, log2n, 16, 16, 0, 0, True, direction=direction)
            assert ib["n_passes"] == (3 if direction == "PAIR" else 2), ib
        assert np.array_equal(a, b)
    x = uniform_frames(5, n, 15, 4200 + log2n)
    check(x, log2n, 16, 13, 0, 0, False, direction=direction)  # exact extraction (t != 16), XSER "OLD"
    check(x, log2n, 16, 9, 0, 0, True, direction=direction)
    for dw in (12, 9):
        xs = uniform_frames(4, n, dw - 1, 4300 + dw)
        xs[3] = uniform_frames(1, n, 16, 5)[0]  # containers beyond DATA_WIDTH: wrapped on load like conv_std_logic_vector
        info = check(xs, log2n, dw, 16, 0, 0, True, direction=direction)
        assert info["kernel_name"] == "k_fft16k_i16", info


@pytest.mark.parametrize("log2n", [13, 14])
@pytest.mark.parametrize("rnd", [0, 1])
@pytest.mark.parametrize("direction,in_order,out_order", [("FWD", "HALVES", "BITREV"), ("FWD", "NATURAL", "BITREV"), ("FWD", "HALVES", "NATURAL"),
                                                          ("INV", "BITREV", "HALVES"), ("INV", "NATURAL", "HALVES"), ("INV", "BITREV", "NATURAL")])
def test_single_pass_n8192_n16384_native_orders(log2n, rnd, direction, in_order, out_order, monkeypatch):
    """The cores' own beat orders on the one-pass kernel (round 4): int_fftNk HALVES in / BITREV out, int_ifftNk BITREV in / HALVES out and the
    mixed forms -- HALVES beats as 8-byte accesses of layout-A register pairs, BITREV order through the staging rows of the transpose region -- in
    both rounding modes, ragged batches, a full-scale frame; against the oracle and against the two-pass plan in the same orders."""
    n = 1 << log2n
    kw = dict(direction=direction, in_order=in_order, out_order=out_order)
    for batch in (1, 5, 600 if log2n == 13 else 300):
        x = uniform_frames(batch, n, 15, 4700 + log2n + batch)
        x[0] = uniform_frames(1, n, 16, 17)[0]
        a, ia = run_gpu(x, log2n, 16, 16, 0, rnd, True, **kw)
        assert ia["kernel_name"] == "k_fft16k_i16" and ia["n_passes"] == 1, ia
        sel = sorted({0, 1 % batch, batch // 2, batch - 1})
        assert np.array_equal(a[sel], run_ref(x[sel], log2n, 16, 16, 0, rnd, True, **kw))
        with monkeypatch.context() as m:
            m.setenv("INTFFT_NO_FAST16K", "1")
            b, ib = run_gpu(x, log2n, 16, 16, 0, rnd, True, **kw)
            assert ib["n_passes"] == 2, ib
        assert np.array_equal(a, b)
    xs = np.concatenate([edge_frames(n, 12), uniform_frames(3, n, 11, 4800 + log2n)])
    info = check(xs, log2n, 12, 14, 0, rnd, True, **kw)  # narrow data, narrower twiddles (exact extraction)
    assert info["kernel_name"] == "k_fft16k_i16", info


@pytest.mark.parametrize("log2n", [13, 14])
@pytest.mark.parametrize("direction", ["FWD", "INV", "PAIR"])
def test_single_pass_n8192_n16384_round_mode(log2n, direction, monkeypatch):
    """RNDMODE = 1 (the testbench's "ROUNDING" UUT, int_dif2_fly.vhd:167-219 / int_dit2_fly.vhd:164-217) on the one-pass kernel's ROUND instantiations
    (round 4): full-scale frames (the rhu2 wrap at the maximum), edge frames, ragged batches, both XSER / a narrower twiddle width, narrow data (the
    w-bit wraps), against the oracle and against the two-pass round-mode plan (INTFFT_NO_FAST16K)."""
    n = 1 << log2n
    for batch in (1, 3, 700 if log2n == 13 else 330):
        x = uniform_frames(batch, n, 16, 4400 + log2n + batch)
        if batch > 2:
            x[1] = -(1 << 15)
            x[2] = (1 << 15) - 1
            x[2, ::2] = -(1 << 15)
        a, ia = run_gpu(x, log2n, 16, 16, 0, 1, True, direction=direction)
        assert ia["kernel_name"] == "k_fft16k_i16" and ia["n_passes"] == 1, ia
        sel = [0, 1, 2, batch - 1] if batch > 2 else list(range(batch))
        assert np.array_equal(a[sel], run_ref(x[sel], log2n, 16, 16, 0, 1, True, direction=direction))
        with monkeypatch.context() as m:
            m.setenv("INTFFT_NO_FAST16K", "1")
            b, ib = run_gpu(x, log2n, 16, 16, 0, 1, True, direction=direction)
            assert ib["n_passes"] == (3 if direction == "PAIR" else 2), ib
        assert np.array_equal(a, b)
    x = np.concatenate([edge_frames(n, 16), uniform_frames(3, n, 16, 4500 + log2n)])
    check(x, log2n, 16, 13, 0, 1, False, direction=direction)
    check(x, log2n, 16, 10, 0, 1, True, direction=direction)
    for dw in (14, 9):
        xs = np.concatenate([edge_frames(n, dw), uniform_frames(3, n, dw, 4600 + dw)])
        xs[-1] = uniform_frames(1, n, 16, 7)[0]  # containers beyond DATA_WIDTH
        info = check(xs, log2n, dw, 16, 0, 1, True, direction=direction)
        assert info["kernel_name"] == "k_fft16k_i16", info
    with monkeypatch.context() as m:
        m.setenv("INTFFT_NO_PACKED_ROUND", "1")
        _, ig = run_gpu(x[:2], log2n, 16, 16, 0, 1, True, direction=direction)
        assert ig["kernel_name"] != "k_fft16k_i16", ig


@pytest.mark.parametrize("log2n,batch", [(13, 515), (13, 1027), (14, 259), (15, 130), (16, 5), (16, 64)])
@pytest.mark.parametrize("direction,time_order,freq_order", [("FWD", "NATURAL", "NATURAL"), ("FWD", "HALVES", "NATURAL"),
                                                             ("FWD", "HALVES", "BITREV"), ("FWD", "NATURAL", "BITREV"),
                                                             ("INV", "NATURAL", "NATURAL"), ("INV", "HALVES", "NATURAL"),
                                                             ("INV", "HALVES", "BITREV"), ("INV", "NATURAL", "BITREV")])
def test_two_pass_vs_three_pass_split(log2n, batch, direction, time_order, freq_order, monkeypatch):
    """N = 2^13 .. 2^16: the two-pass split (forward: k_big20_p1<., ., 8> = stages L-1..8 on virtual 2^16-point frames, then
    k_mid_p2 = stages 7..0 + bit-reversed store, or k_mid_c for BITREV output; inverse: k_mid_q1 / k_mid_c, then
    k_big20_q1<., ., 8>) against the three-pass split of the same plan (INTFFT_NO_TWOPASS) and the oracle; partial last
    frame groups, HALVES beats on the time side and one full-scale frame included."""
    n = 1 << log2n
    x = uniform_frames(batch, n, 15, 2100 + log2n)
    x[batch // 2] = uniform_frames(1, n, 16, 9)[0]
    kw = (dict(direction=direction, in_order=time_order, out_order=freq_order) if direction == "FWD"
          else dict(direction=direction, in_order=freq_order, out_order=time_order))
    # N = 8192 / 16384 is ONE pass since round 4 (k_fft16k_i16, test_single_pass_n8192_n16384 and ..._native_orders): the two-pass
    # split is then the A/B form behind INTFFT_NO_FAST16K
    if log2n <= 14:
        monkeypatch.setenv("INTFFT_NO_FAST16K", "1")
    a, ia = run_gpu(x, log2n, 16, 16, 0, 0, True, **kw)
    two = {("FWD", "NATURAL"): "k_big20_p1/k_mid_p2", ("FWD", "BITREV"): "k_big20_p1/k_mid_c",
           ("INV", "NATURAL"): "k_mid_q1/k_big20_q1", ("INV", "BITREV"): "k_mid_c/k_big20_q1"}[(direction, freq_order)]
    assert ia["kernel_name"] == two and ia["n_passes"] == 2
    monkeypatch.setenv("INTFFT_NO_TWOPASS", "1")
    b, ib = run_gpu(x, log2n, 16, 16, 0, 0, True, **kw)
    assert ib["kernel_name"] == ("k_big20_p1/p2/p3" if direction == "FWD" else "k_big20_q3/q2/q1") and ib["n_passes"] == 3
    assert np.array_equal(a, b)
    assert np.array_equal(a, run_ref(x, log2n, 16, 16, 0, 0, True, **kw))


@pytest.mark.parametrize("log2n", [13, 14, 15])
@pytest.mark.parametrize("batch", [1, 2, 7, 9])
def test_multi_pass_kernels_small_batches(log2n, batch):
    """A few frames only (partial virtual 2^16-point frames, grids of a handful of workgroups): the dedicated multi-pass
    kernels serve every batch size -- a lone N = 8192 frame takes 10 us through them, 21-53 us as one workgroup of the
    generic pass kernel."""
    n = 1 << log2n
    x = uniform_frames(batch, n, 15, 2500 + log2n + batch)
    x[0] = uniform_frames(1, n, 16, 11)[0]
    for direction in ("FWD", "INV", "PAIR"):
        info = check(x, log2n, 16, 16, 0, 0, True, direction=direction)
        assert ("k_big20" if log2n > 14 else "k_fft16k_i16") in info["kernel_name"], info
    for kw in (dict(in_order="HALVES", out_order="BITREV"), dict(direction="INV", in_order="BITREV", out_order="HALVES")):
        info = check(x, log2n, 16, 16, 0, 0, True, **kw)
        assert ("k_big20" in info["kernel_name"] or "k_mid" in info["kernel_name"]) if log2n > 14 else info["kernel_name"] == "k_fft16k_i16", info
    for direction in ("FWD", "INV"):  # general widths: unscaled 16-bit, 12-bit scaled-round
        info = check(x, log2n, 16, 16, 1, 0, True, direction=direction)
        assert info["kernel_name"].startswith("k_bigw"), info
        check(x >> 4, log2n, 12, 16, 0, 1, True, direction=direction)


@pytest.mark.parametrize("log2n,batch", [(13, 515), (14, 259), (15, 3), (15, 130), (16, 5), (17, 3), (17, 9), (18, 5), (19, 3),
                                         (20, 2)])
def test_three_pass_pair_n8192_to_n2pow20(log2n, batch, monkeypatch):
    """int_fft_ifft_pair for N >= 8192 in three passes: DIF STAGE L-1..12, the whole pair of STAGE 11..0 / 0..11 on every
    4096-point block (the bit reversal between the cores cancels), DIT STAGE 12..L-1.  N <= 2^18 split 2^(L-8) x 256 instead
    (DIF L-1..8, the pair of 7..0 / 0..7 on every 256-point group in k_mid_pair, DIT 8..L-1; N = 2^17, 2^18 through the
    32-register passes k_big2p_a / k_big2p_q); both splits are checked."""
    n = 1 << log2n
    x = uniform_frames(batch, n, 15, 3000 + log2n)
    x[0] = uniform_frames(1, n, 16, 8)[0]
    if log2n <= 14:  # one launch since round 4 (k_fft16k_i16<PAIR>); the three-pass plans are its A/B forms
        info = check(x, log2n, 16, 16, 0, 0, True, direction="PAIR")
        assert info["kernel_name"] == "k_fft16k_i16" and info["n_passes"] == 1, info
        monkeypatch.setenv("INTFFT_NO_FAST16K", "1")
    info = check(x, log2n, 16, 16, 0, 0, True, direction="PAIR")
    assert info["kernel_name"] == ("k_big20_p1/k_mid_pair/q1" if log2n <= 16 else "k_big2p_a/k_mid_pair/k_big2p_q" if log2n <= 18
                                   else "k_big20_p1/k_fft4096_i16<MID>/q1")
    assert info["n_passes"] == 3
    if log2n <= 18:
        with monkeypatch.context() as m:
            m.setenv("INTFFT_NO_TWOPASS", "1")
            info = check(x, log2n, 16, 16, 0, 0, True, direction="PAIR")
            assert info["kernel_name"] == "k_big20_p1/k_fft4096_i16<MID>/q1"
    if batch <= 9 and log2n < 20:
        check(x, log2n, 16, 13, 0, 0, False, direction="PAIR")


@pytest.mark.parametrize("log2n,batch", [(13, 515), (14, 259), (15, 3), (15, 130), (16, 5), (17, 3), (17, 9), (18, 5), (19, 3),
                                         (20, 2)])
def test_three_pass_inverse_n8192_to_n2pow20(log2n, batch):
    """int_ifftNk for N >= 8192: the three passes mirrored (bit-reversed load + DIT 0..3, DIT 4..11 per 4096-point block,
    DIT 12..L-1 on frame groups)."""
    n = 1 << log2n
    x = uniform_frames(batch, n, 15, 4000 + log2n)
    x[0] = uniform_frames(1, n, 16, 9)[0]
    info = check(x, log2n, 16, 16, 0, 0, True, direction="INV")
    if log2n <= 14:  # one pass since round 4 (test_single_pass_n8192_n16384)
        assert info["kernel_name"] == "k_fft16k_i16" and info["n_passes"] == 1, info
    else:
        assert ("k_big2" in info["kernel_name"]) and info["n_passes"] == 2
    if batch <= 9 and log2n < 20:
        check(x, log2n, 16, 13, 0, 0, False, direction="INV")


@pytest.mark.parametrize("log2n,batch", [(19, 1), (19, 5), (19, 70), (20, 1), (20, 3), (20, 35)])
def test_two_pass_inverse_n2pow19_n2pow20(log2n, batch, monkeypatch):
    """int_ifftNk at N = 2^19, 2^20 from natural order: k_big2x_qb (bit-reversed gather as 64-byte pieces, STAGE 0..9 along the rows)
    + k_big2x_qa (STAGE 10..L-1 down the columns, natural or HALVES order out) against the oracle and the three-pass plan they replace
    (INTFFT_NO_BIG2X); several scratch chunks (two streams), a full-scale frame, 13-bit twiddles / XSER OLD, 12-bit data."""
    n = 1 << log2n
    x = uniform_frames(batch, n, 15, 4100 + log2n + batch)
    x[0] = uniform_frames(1, n, 16, 9)[0]
    got, info = run_gpu(x, log2n, 16, 16, 0, 0, True, direction="INV")
    assert info["kernel_name"] == "k_big2x_qb/k_big2x_qa" and info["n_passes"] == 2, info
    sel = list(range(batch)) if batch <= 5 else [0, 1, batch // 2, batch - 1]
    assert np.array_equal(got[sel], run_ref(x[sel], log2n, 16, 16, 0, 0, True, direction="INV"))
    with monkeypatch.context() as m:
        m.setenv("INTFFT_NO_BIG2X", "1")
        got3, info3 = run_gpu(x, log2n, 16, 16, 0, 0, True, direction="INV")
        assert info3["kernel_name"] == "k_big20_q3/q2/q1" and info3["n_passes"] == 3, info3
    assert np.array_equal(got, got3)
    if batch <= 5:
        info = check(x, log2n, 16, 16, 0, 0, True, direction="INV", out_order="HALVES")
        assert info["n_passes"] == 2
        check(x[:2], log2n, 16, 13, 0, 0, False, direction="INV")
        check(x[:2] >> 4, log2n, 12, 16, 0, 0, True, direction="INV")


@pytest.mark.parametrize("log2n,batch", [(13, 259), (13, 1030), (14, 131), (15, 3), (15, 70), (16, 5), (16, 33)])
@pytest.mark.parametrize("case", [(16, 16, 1, 0), (16, 16, 0, 1), (12, 16, 0, 0), (18, 24, 0, 0), (10, 18, 1, 0), (32, 16, 0, 0)])
def test_general_width_three_pass_kernels(log2n, batch, case, monkeypatch):
    """N = 2^13 .. 2^16 with widths within 32 bits (the unscaled 16-bit transform reaches exactly 32 bits at N = 65536):
    int32 pairs, frame groups as virtual 2^16-point frames in the first pass; forward = the two-pass split k_bigw_a/b,
    compared with the three passes of the same plan (INTFFT_NO_TWOPASS) and the oracle."""
    dw, tw, fmt, rnd = case
    if dw + fmt * log2n > 32:
        pytest.skip("results exceed 32 bits")
    monkeypatch.setenv("INTFFT_NO_NARROW16", "1")  # 12-bit truncate-mode data would otherwise take the packed int16 kernels
    monkeypatch.setenv("INTFFT_NO_PACKED_ROUND", "1")  # and 16-bit round mode the packed multi-pass kernels
    n = 1 << log2n
    x = uniform_frames(batch, n, dw, 6000 + log2n + dw)
    x[0] = edge_frames(n, dw)[4]
    info = check(x, log2n, dw, tw, fmt, rnd, True)
    assert info["kernel_name"] == "k_bigw_a/b" and info["n_passes"] == 2, info
    with monkeypatch.context() as m:
        m.setenv("INTFFT_NO_TWOPASS", "1")
        info = check(x, log2n, dw, tw, fmt, rnd, True)
        assert info["kernel_name"] == "k_bigw_p1/p2/p3" and info["n_passes"] == 3, info
    if batch in (259, 131, 3, 5, 1030):  # the inverse through the mirrored passes
        info = check(x, log2n, dw, tw, fmt, rnd, True, direction="INV")
        assert info["kernel_name"] == "k_bigw_qb/qa" and info["n_passes"] == 2, info
        with monkeypatch.context() as m:
            m.setenv("INTFFT_NO_TWOPASS", "1")
            info = check(x, log2n, dw, tw, fmt, rnd, True, direction="INV")
            assert info["kernel_name"] == "k_bigw_q3/q2/q1" and info["n_passes"] == 3, info


@pytest.mark.parametrize("log2n,batch", [(13, 259), (14, 37), (15, 3), (16, 5)])
@pytest.mark.parametrize("case", [(16, 16, 1, 0), (16, 16, 0, 1), (18, 24, 0, 0), (10, 18, 1, 0), (32, 16, 0, 0)])
def test_general_width_two_pass_native_orders(log2n, batch, case, monkeypatch):
    """The cores' own beat orders on the general-width two-pass kernels (round 5: k_bigw_a/b, k_bigw_qb/qa NAT instantiations; int16 and int32 containers
    on either side): int_fftNk HALVES in -> BITREV out, int_ifftNk BITREV in -> HALVES out and the mixed forms with natural order, against the oracle (ragged
    batches: partial virtual frames) and equal to the generic kernels (INTFFT_NO_FASTW32)."""
    dw, tw, fmt, rnd = case
    if dw + fmt * log2n > 32:
        pytest.skip("results exceed 32 bits")
    monkeypatch.setenv("INTFFT_NO_NARROW16", "1")
    monkeypatch.setenv("INTFFT_NO_PACKED_ROUND", "1")
    n = 1 << log2n
    x = uniform_frames(batch, n, dw, 6100 + log2n + dw)
    x[0] = edge_frames(n, dw)[4]
    for direction, orders, name in (("FWD", [("HALVES", "BITREV"), ("HALVES", "NATURAL"), ("NATURAL", "BITREV")], "k_bigw_a/b"),
                                    ("INV", [("BITREV", "HALVES"), ("BITREV", "NATURAL"), ("NATURAL", "HALVES")], "k_bigw_qb/qa")):
        for in_o, out_o in orders:
            info = check(x, log2n, dw, tw, fmt, rnd, True, direction=direction, in_order=in_o, out_order=out_o)
            assert info["kernel_name"] == name and info["n_passes"] == 2, (info, in_o, out_o)
        a, _ = run_gpu(x, log2n, dw, tw, fmt, rnd, True, direction=direction, in_order=orders[0][0], out_order=orders[0][1])
        with monkeypatch.context() as m:
            m.setenv("INTFFT_NO_FASTW32", "1")
            b, ib = run_gpu(x, log2n, dw, tw, fmt, rnd, True, direction=direction, in_order=orders[0][0], out_order=orders[0][1])
        assert ib["kernel_name"].startswith("k_pass") and np.array_equal(a, b)


@pytest.mark.parametrize("log2n,batch", [(13, 515), (14, 259), (15, 5), (16, 3), (17, 3), (18, 5), (19, 3), (20, 1)])
@pytest.mark.parametrize("direction,in_order,out_order", [("FWD", "HALVES", "BITREV"), ("FWD", "NATURAL", "BITREV"),
                                                          ("FWD", "HALVES", "NATURAL"), ("INV", "BITREV", "HALVES"),
                                                          ("INV", "NATURAL", "HALVES"), ("INV", "BITREV", "NATURAL")])
def test_three_pass_native_orders(log2n, batch, direction, in_order, out_order, monkeypatch):
    """The cores' own beat orders for N >= 8192: HALVES beats as 8-byte loads / stores of register pairs in pass 1,
    BITREV order = the core index, so the last (first) four stages run on 16 consecutive samples (k_big_c)."""
    if log2n <= 14:  # one pass since round 4 (test_single_pass_n8192_n16384_native_orders): this test keeps the multi-pass plans covered
        monkeypatch.setenv("INTFFT_NO_FAST16K", "1")
    n = 1 << log2n
    x = uniform_frames(batch, n, 15, 7000 + log2n)
    x[0] = uniform_frames(1, n, 16, 10)[0]
    info = check(x, log2n, 16, 16, 0, 0, True, direction=direction, in_order=in_order, out_order=out_order)
    # every length in two passes: N = 2^17, 2^18 on the 32-register passes, N = 2^19, 2^20 on the half-line tiles (BITREV side: the
    # thread's 32 consecutive core positions as eight 16-byte accesses)
    assert "k_big2" in info["kernel_name"] and info["n_passes"] == 2, info


def test_config4_n_2pow20_taylor_extension():
    """BASELINE config 4 shape at a reduced batch: N = 2^20, 16-bit scaled, Taylor ii = 8 extension."""
    x = uniform_frames(2, 1 << 20, 15, 0xC0FFEE04)
    check(x, 20, 16, 16, 0, 0, True, direction="FWD")
    check(x[:1], 20, 16, 16, 0, 0, True, direction="FWD", out_order="BITREV")


def test_config3_shape_reduced_batch():
    """BASELINE config 3: N = 65536, 24-bit unscaled (40-bit results in int64), both twiddle widths."""
    x = np.concatenate([uniform_frames(3, 1 << 16, 23, 0xC0FFEE03), edge_frames(1 << 16, 24)[[1, 4, 5]]])
    check(x, 16, 24, 24, 1, 0, True)
    check(x[:2], 16, 24, 16, 1, 0, True)


def test_config5_pair_4096():
    x = np.concatenate([uniform_frames(29, 4096, 15, 0xC0FFEE05), edge_frames(4096, 16)])
    check(x, 12, 16, 16, 0, 0, True, direction="PAIR")


@pytest.mark.parametrize("rnd", [0, 1])
@pytest.mark.parametrize("out_order", ["NATURAL", "BITREV"])
def test_config2_headline_shape(rnd, out_order):
    """BASELINE config 2 at a reduced batch incl. the 8 edge frames: N = 1024, 16/16 scaled DIF."""
    x = np.concatenate([edge_frames(1024, 16), uniform_frames(4096 - 8, 1024, 15, 0xC0FFEE02)])
    check(x, 10, 16, 16, 0, rnd, True, out_order=out_order)


@pytest.mark.parametrize("log2n", [8, 9, 10])
def test_headline_magnitude_votes(log2n):
    """k_fft1024_i16's fast path is chosen by magnitude votes (|re|, |im| < 23100 on the inputs, or on the values behind stage 6 for
    a frame that failed the first vote): frames AT the threshold from both sides, with the sample patterns that keep the complex
    magnitude at its bound through every stage (all four corners (+-A, +-A): constant, alternating with every period, random signs),
    70 % and 100 % full-scale noise, and mixtures -- all bit-exact to the oracle, whichever extraction a frame takes."""
    n = 1 << log2n
    rng = np.random.default_rng(4242 + log2n)
    A = 23099
    frames = []
    for a in (A, A + 1, -A - 1, -A - 2, 32767, -32768):
        f = np.full((n, 2), a, dtype=np.int64)
        frames.append(np.clip(f, -32768, 32767))
        for period in (1, 2, 4, 8, 16, 64, n // 2):  # +-a alternating in blocks: maximal differences at one stage each
            sgn = np.where((np.arange(n) // period) % 2 == 0, 1, -1)
            g = np.clip(np.stack([sgn * a, -sgn * a], -1), -32768, 32767)
            frames.append(g)
            frames.append(np.clip(np.stack([sgn * a, np.roll(sgn, period // 2 + 1) * a], -1), -32768, 32767))
    for a in (A, A + 1):
        for _ in range(6):
            frames.append(np.stack([rng.choice([-a, a], n), rng.choice([-a, a], n)], -1))
    x = np.stack(frames).astype(np.int64)
    x = np.concatenate([x, uniform_frames(24, n, 16, 99), rng.integers(-23100, 23100, size=(24, n, 2)),
                        rng.integers(-23101, 23101, size=(8, n, 2)), uniform_frames(16, n, 15, 98)])
    info = check(x, log2n, 16, 16, 0, 0, True)
    assert info["kernel_name"].startswith("k_fft1024_i16"), info
    check(x[::3], log2n, 16, 16, 0, 0, True, in_order="HALVES", out_order="BITREV")


W64_CASES = [(32, 16, 1, 0, True), (32, 24, 1, 0, True), (32, 24, 1, 0, False), (24, 24, 1, 0, True), (28, 18, 1, 0, True), (40, 16, 0, 0, True),
             (44, 16, 0, 1, True), (48, 24, 0, 0, True), (36, 16, 1, 0, False), (50, 16, 1, 0, True), (54, 10, 1, 0, True), (33, 26, 0, 1, True),
             (64, 16, 0, 0, True), (60, 16, 0, 1, False), (64, 16, 0, 1, True)]


@pytest.mark.parametrize("case", W64_CASES)
def test_wave_kernel_64_bit_results(case, monkeypatch):
    """N = 1024 forward with results of 33 .. 64 bits (32-bit unscaled data: int_fft_single_path.vhd:15 documents DATA_WIDTH 8-32;
    wide scaled data; every multiplier regime that fits 64-bit words, both XSER): the 64-bit wave kernel k_fft1024_w64 against the
    oracle and against the generic pass kernel it replaces (INTFFT_NO_FASTW64); ragged batches, edge frames, int32 and int64
    input containers."""
    dw, tw, fmt, rnd, new = case
    if C.lib().orc_validate(C.make_params(10, dw, tw, fmt, rnd, new), C.FWD) != 0:
        pytest.skip("not elaboratable")
    x = np.concatenate([edge_frames(1024, dw), uniform_frames(37, 1024, dw, 640 + dw), uniform_frames(6, 1024, max(2, dw - 3), 641 + dw)])
    got, info = run_gpu(x, 10, dw, tw, fmt, rnd, new)
    if info["out_bits"] <= 32 or info["out_bits"] > 64 or info["kernel_name"] == "k_fft1024_w32":
        pytest.skip("not a plan of the 64-bit wave kernel (33 / 34-bit unscaled results: k_fft1024_w32 with its 64-bit tail)")
    assert info["kernel_name"] == "k_fft1024_w64" and info["fast_path"] == 1 and info["compute_word"] == 8, info
    assert np.array_equal(got, run_ref(x, 10, dw, tw, fmt, rnd, new))
    with monkeypatch.context() as m:
        m.setenv("INTFFT_NO_FASTW64", "1")
        got_g, info_g = run_gpu(x[:9], 10, dw, tw, fmt, rnd, new)
        assert info_g["kernel_name"].startswith("k_pass"), info_g
    assert np.array_equal(got[:9], got_g)


W64_INV_CASES = [(26, 16, 1, 0, True), (34, 24, 1, 0, True), (34, 24, 1, 0, False), (30, 18, 1, 0, True), (40, 16, 0, 0, True), (44, 16, 0, 1, True),
                 (48, 24, 0, 0, True), (42, 16, 1, 0, False), (50, 10, 1, 0, True), (33, 26, 0, 1, True), (64, 16, 0, 0, True), (64, 16, 0, 1, True)]


@pytest.mark.parametrize("case", W64_INV_CASES)
def test_wave_kernel_64_bit_results_inverse(case, monkeypatch):
    """N = 1024 int_ifftNk with results of 33 .. 64 bits (the inverse half of an unscaled pair: its DATA_WIDTH is the forward core's
    output width, int_fft_ifft_pair.vhd:100-103): k_ifft1024_w64 against the oracle and against the generic pass kernel."""
    dw, tw, fmt, rnd, new = case
    if C.lib().orc_validate(C.make_params(10, dw, tw, fmt, rnd, new), C.INV) != 0:
        pytest.skip("not elaboratable")
    x = np.concatenate([edge_frames(1024, dw), uniform_frames(37, 1024, dw, 740 + dw), uniform_frames(6, 1024, max(2, dw - 3), 741 + dw)])
    got, info = run_gpu(x, 10, dw, tw, fmt, rnd, new, direction="INV")
    if info["out_bits"] <= 32 or info["out_bits"] > 64:
        pytest.skip("not a plan of the 64-bit wave kernel")
    assert info["kernel_name"] == "k_ifft1024_w64" and info["fast_path"] == 1 and info["compute_word"] == 8, info
    assert np.array_equal(got, run_ref(x, 10, dw, tw, fmt, rnd, new, direction="INV"))
    with monkeypatch.context() as m:
        m.setenv("INTFFT_NO_FASTW64", "1")
        got_g, info_g = run_gpu(x[:9], 10, dw, tw, fmt, rnd, new, direction="INV")
        assert info_g["kernel_name"].startswith("k_pass"), info_g
    assert np.array_equal(got[:9], got_g)


@pytest.mark.parametrize("direction", ["FWD", "INV"])
@pytest.mark.parametrize("case", [(32, 16, 1, 0, True), (32, 24, 1, 0, False), (28, 18, 1, 0, True), (40, 16, 0, 0, True), (44, 16, 0, 1, True), (48, 24, 0, 0, True),
                                  (50, 16, 1, 0, True), (64, 16, 0, 1, True), (60, 16, 0, 0, False)])
def test_wave_kernel_64_bit_native_orders(case, direction, monkeypatch):
    """N = 1024 with results of 33 .. 64 bits in the cores' own beat orders (round 5: the NAT instantiations of k_fft1024_w64 / k_ifft1024_w64, every rounding
    kind and multiplier form, int32 and int64 input containers): int_fftNk HALVES in -> BITREV out, int_ifftNk BITREV in -> HALVES out and the mixed forms
    with natural order, against the oracle and equal to the generic kernel (INTFFT_NO_FASTW64)."""
    dw, tw, fmt, rnd, new = case
    if C.lib().orc_validate(C.make_params(10, dw, tw, fmt, rnd, new), DIR[direction]) != 0:
        pytest.skip("not elaboratable")
    x = np.concatenate([edge_frames(1024, dw), uniform_frames(21, 1024, dw, 840 + dw), uniform_frames(3, 1024, max(2, dw - 3), 841 + dw)])
    orders = [("HALVES", "BITREV"), ("HALVES", "NATURAL"), ("NATURAL", "BITREV")] if direction == "FWD" else [("BITREV", "HALVES"), ("BITREV", "NATURAL"),
                                                                                                             ("NATURAL", "HALVES")]
    name = "k_ifft1024_w64" if direction == "INV" else "k_fft1024_w64"
    for in_o, out_o in orders:
        got, info = run_gpu(x, 10, dw, tw, fmt, rnd, new, direction=direction, in_order=in_o, out_order=out_o)
        if info["out_bits"] <= 32 or info["out_bits"] > 64 or info["kernel_name"] == "k_fft1024_w32":
            pytest.skip("not a plan of the 64-bit wave kernel")
        assert info["kernel_name"] == name and info["n_passes"] == 1, (info, in_o, out_o)
        assert np.array_equal(got, run_ref(x, 10, dw, tw, fmt, rnd, new, direction=direction, in_order=in_o, out_order=out_o)), (in_o, out_o)
    a, _ = run_gpu(x[:9], 10, dw, tw, fmt, rnd, new, direction=direction, in_order=orders[0][0], out_order=orders[0][1])
    with monkeypatch.context() as m:
        m.setenv("INTFFT_NO_FASTW64", "1")
        b, ib = run_gpu(x[:9], 10, dw, tw, fmt, rnd, new, direction=direction, in_order=orders[0][0], out_order=orders[0][1])
    assert ib["kernel_name"].startswith("k_pass") and np.array_equal(a, b)


@pytest.mark.parametrize("direction", ["FWD", "INV"])
@pytest.mark.parametrize("case", [(6, 32, 16, 1, 0), (7, 32, 16, 1, 0), (7, 40, 24, 1, 0), (8, 32, 24, 1, 0), (9, 30, 18, 1, 0), (7, 44, 16, 0, 1), (8, 48, 24, 0, 0),
                                  (9, 40, 16, 0, 0), (6, 50, 16, 1, 0)])
def test_wave_kernel_64_bit_short_frames(case, direction):
    """N = 64 .. 512 on the 64-bit wave kernels (2^(10 - NFFT) frames per wave): 32-bit unscaled data at the reference testbenches'
    N = 128 (fft_signle_test.vhd:70-92), ragged batches (the last wave's absent frames), int32 and int64 containers."""
    log2n, dw, tw, fmt, rnd = case
    n = 1 << log2n
    if C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, True), DIR[direction]) != 0:
        pytest.skip("not elaboratable")
    x = np.concatenate([edge_frames(n, dw), uniform_frames(77, n, dw, 940 + dw + log2n)])
    for nb in (len(x), 1, 5):
        got, info = run_gpu(x[:nb], log2n, dw, tw, fmt, rnd, True, direction=direction)
        # (short frames are instantiated for the one-int64-per-product multiplier form: beyond mw + TWDL_WIDTH = 64 the generic kernel)
        narrow = dw + ((log2n - 2 if direction == "FWD" else log2n - 1) if fmt else 0) + tw <= 64
        assert info["kernel_name"] == (("k_ifft1024_w64" if direction == "INV" else "k_fft1024_w64") if narrow else "k_pass<long>"), info
        assert np.array_equal(got, run_ref(x[:nb], log2n, dw, tw, fmt, rnd, True, direction=direction)), nb


@pytest.mark.parametrize("direction", ["FWD", "INV"])
@pytest.mark.parametrize("case", [(7, 32, 16, 1, 0), (7, 40, 24, 1, 0), (8, 32, 24, 1, 0), (9, 30, 18, 1, 0), (7, 44, 16, 0, 1), (8, 48, 24, 0, 0), (9, 40, 16, 0, 0)])
def test_wave_kernel_64_bit_short_frames_native_orders(case, direction):
    """N = 128 .. 512 with results of 33 .. 64 bits in the cores' own beat orders (round 5: NAT instantiations of the short-frame wave kernels; 32-bit unscaled
    data at the reference testbenches' N = 128, fft_signle_test.vhd:70-92, as int_fftNk itself takes and gives it): the three order pairs per direction, ragged
    batches (absent frames of the last wave), against the oracle."""
    log2n, dw, tw, fmt, rnd = case
    n = 1 << log2n
    if C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, True), DIR[direction]) != 0:
        pytest.skip("not elaboratable")
    x = np.concatenate([edge_frames(n, dw), uniform_frames(77, n, dw, 1240 + dw + log2n)])
    orders = [("HALVES", "BITREV"), ("HALVES", "NATURAL"), ("NATURAL", "BITREV")] if direction == "FWD" else [("BITREV", "HALVES"), ("BITREV", "NATURAL"),
                                                                                                             ("NATURAL", "HALVES")]
    narrow = dw + ((log2n - 2 if direction == "FWD" else log2n - 1) if fmt else 0) + tw <= 64
    for in_o, out_o in orders:
        for nb in (len(x), 1, 5):
            got, info = run_gpu(x[:nb], log2n, dw, tw, fmt, rnd, True, direction=direction, in_order=in_o, out_order=out_o)
            assert info["kernel_name"] == (("k_ifft1024_w64" if direction == "INV" else "k_fft1024_w64") if narrow else "k_pass<long>"), (info, in_o, out_o)
            assert np.array_equal(got, run_ref(x[:nb], log2n, dw, tw, fmt, rnd, True, direction=direction, in_order=in_o, out_order=out_o)), (nb, in_o, out_o)


@pytest.mark.parametrize("direction", ["FWD", "INV"])
@pytest.mark.parametrize("case", [(11, 32, 16, 1, 0, True), (12, 32, 16, 1, 0, True), (12, 32, 24, 1, 0, False), (11, 30, 18, 1, 0, True), (12, 40, 16, 0, 0, True),
                                  (11, 44, 16, 0, 1, True), (12, 48, 24, 0, 0, True), (11, 50, 10, 1, 0, True), (12, 64, 16, 0, 0, True), (12, 24, 24, 1, 0, True)])
def test_block_kernel_64_bit_native_orders(case, direction, monkeypatch):
    """N = 2048 / 4096 with results of 33 .. 64 bits in the cores' own beat orders (round 5: the NAT instantiations of k_fft4096_w64 / k_ifft4096_w64; HALVES pairs
    as 16- / 32-byte accesses, BITREV order one component at a time through the transpose region): the three order pairs per direction against the oracle,
    ragged batches (N = 2048: the absent second frame of the last workgroup), int32 and int64 containers, and equal to the generic kernel."""
    log2n, dw, tw, fmt, rnd, new = case
    n = 1 << log2n
    if C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, new), DIR[direction]) != 0:
        pytest.skip("not elaboratable")
    x = np.concatenate([edge_frames(n, dw), uniform_frames(13, n, dw, 1140 + dw + log2n), uniform_frames(2, n, max(2, dw - 3), 1141 + dw)])
    orders = [("HALVES", "BITREV"), ("HALVES", "NATURAL"), ("NATURAL", "BITREV")] if direction == "FWD" else [("BITREV", "HALVES"), ("BITREV", "NATURAL"),
                                                                                                             ("NATURAL", "HALVES")]
    mw_max = dw + ((log2n - 2 if direction == "FWD" else log2n - 1) if fmt else 0)
    for in_o, out_o in orders:
        got, info = run_gpu(x, log2n, dw, tw, fmt, rnd, new, direction=direction, in_order=in_o, out_order=out_o)
        if info["out_bits"] <= 32 or info["out_bits"] > 64:
            pytest.skip("not a 64-bit-word plan")
        if (rnd == 1 or mw_max > 63) and mw_max + tw > 64:
            assert info["kernel_name"] == "k_pass<long>", info
        else:
            assert info["kernel_name"] in (("k_ifft4096_w64",) if direction == "INV" else ("k_fft4096_w64", "k_fft4096_w32")) and info["n_passes"] == 1, (info, in_o, out_o)
        assert np.array_equal(got, run_ref(x, log2n, dw, tw, fmt, rnd, new, direction=direction, in_order=in_o, out_order=out_o)), (in_o, out_o)
        for nb in (1, 3):
            g, _ = run_gpu(x[:nb], log2n, dw, tw, fmt, rnd, new, direction=direction, in_order=in_o, out_order=out_o)
            assert np.array_equal(g, got[:nb]), (nb, in_o, out_o)
    a, _ = run_gpu(x[:5], log2n, dw, tw, fmt, rnd, new, direction=direction, in_order=orders[0][0], out_order=orders[0][1])
    with monkeypatch.context() as m:
        m.setenv("INTFFT_NO_FASTW64", "1")
        m.setenv("INTFFT_NO_FASTW32", "1")
        b, ib = run_gpu(x[:5], log2n, dw, tw, fmt, rnd, new, direction=direction, in_order=orders[0][0], out_order=orders[0][1])
    assert ib["kernel_name"].startswith("k_pass") and np.array_equal(a, b)


@pytest.mark.parametrize("direction", ["FWD", "INV"])
@pytest.mark.parametrize("case", [(11, 32, 16, 1, 0, True), (12, 32, 16, 1, 0, True), (12, 32, 24, 1, 0, False), (11, 30, 18, 1, 0, True), (12, 40, 16, 0, 0, True),
                                  (11, 44, 16, 0, 1, True), (12, 48, 24, 0, 0, True), (12, 36, 24, 1, 0, True), (11, 50, 10, 1, 0, True), (12, 33, 26, 0, 1, True),
                                  (12, 64, 16, 0, 0, True), (12, 28, 16, 1, 0, True), (11, 26, 24, 1, 0, False)])
def test_block_kernel_64_bit_results(case, direction, monkeypatch):
    """N = 2048 / 4096 with results of 33 .. 64 bits on the 64-bit block kernels k_fft4096_w64 / k_ifft4096_w64 (32-bit unscaled data:
    43 / 44-bit results; the inverse half of unscaled pairs; wide scaled data; narrow and three-dword products): against the oracle and
    the generic pass kernel, ragged batches (N = 2048: two frames per workgroup), int32 and int64 containers.  Plans the partly-32-bit
    block kernel k_fft4096_w32 serves (24-bit unscaled forward) keep it; round mode beyond mw + t = 64 stays generic."""
    log2n, dw, tw, fmt, rnd, new = case
    n = 1 << log2n
    if C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, new), DIR[direction]) != 0:
        pytest.skip("not elaboratable")
    x = np.concatenate([edge_frames(n, dw), uniform_frames(21, n, dw, 1040 + dw + log2n), uniform_frames(4, n, max(2, dw - 3), 1041 + dw)])
    got, info = run_gpu(x, log2n, dw, tw, fmt, rnd, new, direction=direction)
    if info["out_bits"] <= 32 or info["out_bits"] > 64:
        pytest.skip("not a 64-bit-word plan")
    mw_max = dw + ((log2n - 2 if direction == "FWD" else log2n - 1) if fmt else 0)
    if info["kernel_name"] == "k_fft4096_w32":
        assert direction == "FWD" and fmt == 1 and dw + log2n <= 40, info
    elif (rnd == 1 or mw_max > 63) and mw_max + tw > 64:  # neither the narrow nor the three-dword multiplier form
        assert info["kernel_name"] == "k_pass<long>", info
    else:
        assert info["kernel_name"] == ("k_ifft4096_w64" if direction == "INV" else "k_fft4096_w64") and info["fast_path"] == 1 and info["compute_word"] == 8, info
    assert np.array_equal(got, run_ref(x, log2n, dw, tw, fmt, rnd, new, direction=direction))
    for nb in (1, 2, 3):
        g, _ = run_gpu(x[:nb], log2n, dw, tw, fmt, rnd, new, direction=direction)
        assert np.array_equal(g, got[:nb]), nb
    if info["kernel_name"].endswith("4096_w64"):
        with monkeypatch.context() as m:
            m.setenv("INTFFT_NO_FASTW64", "1")
            got_g, info_g = run_gpu(x[:5], log2n, dw, tw, fmt, rnd, new, direction=direction)
            assert info_g["kernel_name"].startswith("k_pass"), info_g
        assert np.array_equal(got[:5], got_g)


@pytest.mark.parametrize("case", [(7, 32, 16, 1, 0, True), (12, 16, 16, 1, 0, True), (11, 24, 16, 1, 0, True), (10, 16, 16, 1, 0, True), (10, 24, 24, 1, 0, True), (10, 22, 16, 1, 0, False), (10, 40, 16, 0, 0, True), (10, 36, 18, 0, 1, True)])
def test_pair_of_dedicated_kernels_on_64_bit_words(case, monkeypatch):
    """FFT -> IFFT pairs whose results need 33 .. 64 bits (16-bit unscaled: 36 bits; 24-bit unscaled: 44): a forward sub-plan, a
    middle buffer and an inverse sub-plan, each on its dedicated kernel, against the oracle's pair and the generic pair kernel;
    the chunk loop of the middle buffer on a small INTFFT_SCRATCH_MB."""
    log2n, dw, tw, fmt, rnd, new = case
    if C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, new), C.PAIR) != 0:
        pytest.skip("not elaboratable")
    n = 1 << log2n
    x = np.concatenate([edge_frames(n, dw), uniform_frames(70, n, dw, 840 + dw)])
    got, info = run_gpu(x, log2n, dw, tw, fmt, rnd, new, direction="PAIR")
    assert info["kernel_name"].startswith("pair[") and info["n_passes"] == 2, info
    assert np.array_equal(got, run_ref(x, log2n, dw, tw, fmt, rnd, new, direction="PAIR"))
    with monkeypatch.context() as m:
        m.setenv("INTFFT_NO_PAIR_COMPOSITE", "1")
        got_g, info_g = run_gpu(x[:9], log2n, dw, tw, fmt, rnd, new, direction="PAIR")
        assert info_g["kernel_name"].startswith("k_pass"), info_g
    assert np.array_equal(got[:9], got_g)
    with monkeypatch.context() as m:
        m.setenv("INTFFT_SCRATCH_MB", "1")  # 1 MiB middle buffer: 64 / 128 frames per chunk
        got_c, info_c = run_gpu(np.concatenate([x] * 4), log2n, dw, tw, fmt, rnd, new, direction="PAIR")
        assert info_c["kernel_name"].startswith("pair["), info_c
    assert np.array_equal(got_c, np.concatenate([got] * 4))


@pytest.mark.parametrize("case", [(10, 18, 16, 0, 0, True), (10, 24, 24, 0, 1, True), (12, 24, 18, 0, 0, False), (11, 20, 24, 0, 1, True), (8, 12, 18, 1, 0, True),
                                  (7, 9, 12, 1, 0, False), (14, 18, 16, 0, 0, True), (13, 24, 24, 0, 1, True), (16, 32, 16, 0, 0, True), (9, 32, 26, 0, 0, True)])
def test_pair_of_dedicated_kernels_on_32_bit_words(case, monkeypatch):
    """FFT -> IFFT pairs of general widths within 32 bits (18 / 24 / 32-bit scaled data in both rounding modes, small unscaled pairs): since
    round 4 a forward sub-plan (k_fft1024_w32 / k_fft4096_w32 / k_bigw_a/b), a middle buffer and an inverse sub-plan (k_ifft*_w32 / k_bigw_qb/qa)
    instead of the generic pair kernel -- against the oracle's pair, the generic kernel, and through the chunk loop of the middle buffer."""
    log2n, dw, tw, fmt, rnd, new = case
    if C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, new), C.PAIR) != 0:
        pytest.skip("not elaboratable")
    monkeypatch.setenv("INTFFT_NO_NARROW16", "1")
    n = 1 << log2n
    x = np.concatenate([edge_frames(n, dw), uniform_frames(max(3, (1 << 16) // n), n, dw, 860 + dw + log2n)])
    got, info = run_gpu(x, log2n, dw, tw, fmt, rnd, new, direction="PAIR")
    assert info["kernel_name"].startswith("pair[") and "k_pass" not in info["kernel_name"], info
    assert np.array_equal(got, run_ref(x, log2n, dw, tw, fmt, rnd, new, direction="PAIR"))
    with monkeypatch.context() as m:
        m.setenv("INTFFT_NO_PAIR_COMPOSITE", "1")
        got_g, info_g = run_gpu(x[:9], log2n, dw, tw, fmt, rnd, new, direction="PAIR")
        assert info_g["kernel_name"].startswith("k_pass"), info_g
    assert np.array_equal(got[:9], got_g)
    if log2n <= 12:
        with monkeypatch.context() as m:
            m.setenv("INTFFT_SCRATCH_MB", "1")  # 1 MiB middle buffer: the chunk loop
            got_c, info_c = run_gpu(np.concatenate([x] * 3), log2n, dw, tw, fmt, rnd, new, direction="PAIR")
            assert info_c["kernel_name"].startswith("pair["), info_c
        assert np.array_equal(got_c, np.concatenate([got] * 3))


@pytest.mark.parametrize("log2n", [7, 8, 9, 10])
@pytest.mark.parametrize("in_order,out_order", [("HALVES", "BITREV"), ("NATURAL", "BITREV"), ("HALVES", "NATURAL")])
def test_unscaled_wave_kernel_native_orders(log2n, in_order, out_order, monkeypatch):
    """int_fftNk with FORMAT = 1 in its own beat orders (HALVES in / BITREV out, and the mixed forms) on k_fft1024_u32's native-order instantiation
    (round 4): HALVES beats as 8-byte loads of register pairs, BITREV order through the wave's LDS tile in memory order.  Ragged batches (partial
    last chunk of 2^(10-L) frames), full-scale frames (exact path), both XSER / a narrower twiddle width; against the oracle and the generic kernel."""
    n = 1 << log2n
    fp = 1 << (10 - log2n)
    kw = dict(in_order=in_order, out_order=out_order)
    for batch in (1, fp + 1, 5 * fp + 3, 300):
        x = uniform_frames(batch, n, 15, 5100 + log2n + batch)
        x[0] = uniform_frames(1, n, 16, 19)[0]
        got, info = run_gpu(x, log2n, 16, 16, 1, 0, True, **kw)
        assert info["kernel_name"] == "k_fft1024_u32" and info["n_passes"] == 1, info
        assert np.array_equal(got, run_ref(x, log2n, 16, 16, 1, 0, True, **kw))
    x = np.concatenate([edge_frames(n, 16), uniform_frames(fp + 1, n, 16, 5200 + log2n)])
    check(x, log2n, 16, 12, 1, 0, False, **kw)
    with monkeypatch.context() as m:
        m.setenv("INTFFT_NO_FAST1024U", "1")
        got_g, info_g = run_gpu(x, log2n, 16, 16, 1, 0, True, **kw)
        assert info_g["kernel_name"] != "k_fft1024_u32", info_g
    got, _ = run_gpu(x, log2n, 16, 16, 1, 0, True, **kw)
    assert np.array_equal(got, got_g)


@pytest.mark.parametrize("log2n", [7, 8, 9, 10])
@pytest.mark.parametrize("in_order,out_order", [("BITREV", "HALVES"), ("NATURAL", "HALVES"), ("BITREV", "NATURAL")])
def test_unscaled_inverse_wave_kernel_native_orders(log2n, in_order, out_order, monkeypatch):
    """int_ifftNk with FORMAT = 1 in its own beat orders (BITREV in / HALVES out, and the mixed forms) on k_fft1024ux_u32's native-order
    instantiation (round 4): the chunk loaded in memory order and handed to the LC lanes through the wave's LDS tile, HALVES beats as 16-byte
    stores of L1 register pairs.  Ragged batches, full-scale frames, XSER "OLD" with narrower twiddles; against the oracle and the generic kernel."""
    n = 1 << log2n
    fp = 1 << (10 - log2n)
    kw = dict(direction="INV", in_order=in_order, out_order=out_order)
    for batch in (1, fp + 1, 5 * fp + 3, 300):
        x = uniform_frames(batch, n, 15, 5300 + log2n + batch)
        x[0] = uniform_frames(1, n, 16, 23)[0]
        got, info = run_gpu(x, log2n, 16, 16, 1, 0, True, **kw)
        assert info["kernel_name"] == "k_fft1024ux_u32" and info["n_passes"] == 1, info
        assert np.array_equal(got, run_ref(x, log2n, 16, 16, 1, 0, True, **kw))
    x = np.concatenate([edge_frames(n, 16), uniform_frames(fp + 1, n, 16, 5400 + log2n)])
    check(x, log2n, 16, 12, 1, 0, False, **kw)
    with monkeypatch.context() as m:
        m.setenv("INTFFT_GENERIC_ONLY", "1")
        got_g, info_g = run_gpu(x, log2n, 16, 16, 1, 0, True, **kw)
        assert info_g["kernel_name"].startswith("k_pass"), info_g
    got, _ = run_gpu(x, log2n, 16, 16, 1, 0, True, **kw)
    assert np.array_equal(got, got_g)


@pytest.mark.parametrize("case", [(10, 18, 16, 0, 0), (8, 24, 18, 0, 1), (7, 24, 24, 1, 0), (9, 12, 16, 0, 0), (10, 32, 24, 0, 0), (7, 14, 16, 0, 1), (9, 20, 16, 1, 0),
                                  (10, 24, 24, 1, 0), (10, 23, 16, 1, 0)])  # the last two: 34- / 33-bit results in int64 containers (the 64-bit tail stages)
@pytest.mark.parametrize("in_order,out_order", [("HALVES", "BITREV"), ("NATURAL", "BITREV"), ("HALVES", "NATURAL")])
def test_general_width_wave_kernel_native_orders(case, in_order, out_order, monkeypatch):
    """int_fftNk in its own beat orders for general widths within 32 bits (k_fft1024_w32's native-order instantiation, round 4): int16 and int32
    containers on either side, the three sum / difference modes, both multiplier forms; ragged batches, edge frames; against the oracle and the
    generic kernel."""
    log2n, dw, tw, fmt, rnd = case
    monkeypatch.setenv("INTFFT_NO_NARROW16", "1")  # 12- / 14-bit data: the int16-container paths of this kernel
    n = 1 << log2n
    fp = 1 << (10 - log2n)
    kw = dict(in_order=in_order, out_order=out_order)
    for batch in (1, 5 * fp + 3, 200):
        x = uniform_frames(batch, n, dw, 5500 + log2n + batch + dw)
        got, info = run_gpu(x, log2n, dw, tw, fmt, rnd, True, **kw)
        assert info["kernel_name"] == "k_fft1024_w32" and info["n_passes"] == 1, info
        assert np.array_equal(got, run_ref(x, log2n, dw, tw, fmt, rnd, True, **kw))
    x = np.concatenate([edge_frames(n, dw), uniform_frames(fp + 1, n, dw, 5600 + log2n)])
    check(x, log2n, dw, tw, fmt, rnd, False, **kw)
    with monkeypatch.context() as m:
        m.setenv("INTFFT_GENERIC_ONLY", "1")
        got_g, info_g = run_gpu(x, log2n, dw, tw, fmt, rnd, True, **kw)
        assert info_g["kernel_name"].startswith("k_pass"), info_g
    got, _ = run_gpu(x, log2n, dw, tw, fmt, rnd, True, **kw)
    assert np.array_equal(got, got_g)


@pytest.mark.parametrize("case", [(10, 18, 16, 0, 0), (8, 24, 18, 0, 1), (7, 24, 24, 1, 0), (9, 12, 16, 0, 0), (10, 32, 24, 0, 0), (7, 14, 16, 0, 1), (9, 20, 16, 1, 0)])
@pytest.mark.parametrize("in_order,out_order", [("BITREV", "HALVES"), ("NATURAL", "HALVES"), ("BITREV", "NATURAL")])
def test_general_width_inverse_wave_kernel_native_orders(case, in_order, out_order, monkeypatch):
    """int_ifftNk in its own beat orders for general widths within 32 bits (k_ifft1024_w32's native-order instantiation, round 4): int16 and
    int32 containers on either side, the three sum / difference modes; ragged batches, edge frames; against the oracle and the generic kernel."""
    log2n, dw, tw, fmt, rnd = case
    monkeypatch.setenv("INTFFT_NO_NARROW16", "1")
    n = 1 << log2n
    fp = 1 << (10 - log2n)
    kw = dict(direction="INV", in_order=in_order, out_order=out_order)
    for batch in (1, 5 * fp + 3, 200):
        x = uniform_frames(batch, n, dw, 5700 + log2n + batch + dw)
        got, info = run_gpu(x, log2n, dw, tw, fmt, rnd, True, **kw)
        assert info["kernel_name"] == "k_ifft1024_w32" and info["n_passes"] == 1, info
        assert np.array_equal(got, run_ref(x, log2n, dw, tw, fmt, rnd, True, **kw))
    x = np.concatenate([edge_frames(n, dw), uniform_frames(fp + 1, n, dw, 5800 + log2n)])
    check(x, log2n, dw, tw, fmt, rnd, False, **kw)
    with monkeypatch.context() as m:
        m.setenv("INTFFT_GENERIC_ONLY", "1")
        got_g, info_g = run_gpu(x, log2n, dw, tw, fmt, rnd, True, **kw)
        assert info_g["kernel_name"].startswith("k_pass"), info_g
    got, _ = run_gpu(x, log2n, dw, tw, fmt, rnd, True, **kw)
    assert np.array_equal(got, got_g)


@pytest.mark.parametrize("case", [(12, 16, 16, 1, 0), (11, 16, 16, 1, 0), (12, 18, 16, 0, 0), (11, 24, 24, 0, 1), (12, 12, 16, 0, 0), (12, 20, 18, 1, 0), (11, 32, 16, 0, 0)])
@pytest.mark.parametrize("in_order,out_order", [("HALVES", "BITREV"), ("NATURAL", "BITREV"), ("HALVES", "NATURAL")])
def test_general_width_block_kernel_native_orders(case, in_order, out_order, monkeypatch):
    """int_fftNk in its own beat orders at N = 2048 / 4096 for general widths within 32 bits (k_fft4096_w32's native-order instantiation, round 4:
    the unscaled 16-bit transform, 18 / 24 / 32-bit scaled data, 12-bit data in int16 containers); ragged batches (N = 2048: two frames per
    workgroup), edge frames; against the oracle and the generic kernel."""
    log2n, dw, tw, fmt, rnd = case
    monkeypatch.setenv("INTFFT_NO_NARROW16", "1")
    n = 1 << log2n
    kw = dict(in_order=in_order, out_order=out_order)
    for batch in (1, 3, 70):
        x = uniform_frames(batch, n, dw, 5900 + log2n + batch + dw)
        got, info = run_gpu(x, log2n, dw, tw, fmt, rnd, True, **kw)
        assert info["kernel_name"] == "k_fft4096_w32" and info["n_passes"] == 1, info
        assert np.array_equal(got, run_ref(x, log2n, dw, tw, fmt, rnd, True, **kw))
    x = np.concatenate([edge_frames(n, dw), uniform_frames(3, n, dw, 6000 + log2n)])
    check(x, log2n, dw, tw, fmt, rnd, False, **kw)
    with monkeypatch.context() as m:
        m.setenv("INTFFT_GENERIC_ONLY", "1")
        got_g, info_g = run_gpu(x, log2n, dw, tw, fmt, rnd, True, **kw)
        assert info_g["kernel_name"].startswith("k_pass"), info_g
    got, _ = run_gpu(x, log2n, dw, tw, fmt, rnd, True, **kw)
    assert np.array_equal(got, got_g)


@pytest.mark.parametrize("case", [(12, 16, 16, 1, 0), (11, 16, 16, 1, 0), (12, 18, 16, 0, 0), (11, 24, 24, 0, 1), (12, 12, 16, 0, 0), (12, 20, 18, 1, 0), (11, 32, 16, 0, 0)])
@pytest.mark.parametrize("in_order,out_order", [("BITREV", "HALVES"), ("NATURAL", "HALVES"), ("BITREV", "NATURAL")])
def test_general_width_inverse_block_kernel_native_orders(case, in_order, out_order, monkeypatch):
    """int_ifftNk in its own beat orders at N = 2048 / 4096 for general widths within 32 bits (k_ifft4096_w32's native-order instantiation, round 4);
    ragged batches, edge frames; against the oracle and the generic kernel."""
    log2n, dw, tw, fmt, rnd = case
    monkeypatch.setenv("INTFFT_NO_NARROW16", "1")
    n = 1 << log2n
    kw = dict(direction="INV", in_order=in_order, out_order=out_order)
    for batch in (1, 3, 70):
        x = uniform_frames(batch, n, dw, 6100 + log2n + batch + dw)
        got, info = run_gpu(x, log2n, dw, tw, fmt, rnd, True, **kw)
        assert info["kernel_name"] == "k_ifft4096_w32" and info["n_passes"] == 1, info
        assert np.array_equal(got, run_ref(x, log2n, dw, tw, fmt, rnd, True, **kw))
    x = np.concatenate([edge_frames(n, dw), uniform_frames(3, n, dw, 6200 + log2n)])
    check(x, log2n, dw, tw, fmt, rnd, False, **kw)
    with monkeypatch.context() as m:
        m.setenv("INTFFT_GENERIC_ONLY", "1")
        got_g, info_g = run_gpu(x, log2n, dw, tw, fmt, rnd, True, **kw)
        assert info_g["kernel_name"].startswith("k_pass"), info_g
    got, _ = run_gpu(x, log2n, dw, tw, fmt, rnd, True, **kw)
    assert np.array_equal(got, got_g)


def test_config1_chirp_frame():
    x = (chirp_frame(1024) * 64)[None]
    check(x, 10, 16, 16, 0, 0, True)


@pytest.mark.parametrize("fmt", [0, 1])
def test_use_fly_zero(fmt):
    x = uniform_frames(3, 512, 16, 5)
    for direction in ("FWD", "INV", "PAIR"):
        check(x, 9, 16, 16, fmt, 0, True, direction=direction, use_fly=0, out_order="BITREV")


@pytest.mark.parametrize("tw,new", [(16, True), (16, False), (24, True), (24, False), (18, True), (10, True)])
def test_twiddle_tables_all_stages(tw, new):
    """k_twiddle_stage (ROM + Taylor) against the oracle for every STAGE 0..19."""
    from intfftk_amd import IntFFTCore

    core = IntFFTCore(20, 16, tw, 0, 0, "NEW" if new else "OLD", "FWD")
    for s in range(20):
        got = core.twiddles(s)
        re, im = C.twiddles(s, tw, new)
        assert np.array_equal(got[:, 0], re) and np.array_equal(got[:, 1], im), (s, tw, new)
    core.close()


def test_batch_index_independence_and_ragged_batches():
    """Frames are independent units: any batch split gives the same rows (incl. batch = 1 and
    batches that do not fill the last block)."""
    import torch

    from intfftk_amd import int_fft_single_path

    core = int_fft_single_path(7, 16, 16, 0, 0)
    x = torch.from_numpy(uniform_frames(37, 128, 16, 3).astype(np.int16)).cuda()
    full = core(x).cpu().numpy()
    for cut in (1, 5, 16, 36):
        a = core(x[:cut].contiguous()).cpu().numpy()
        b = core(x[cut:].contiguous()).cpu().numpy()
        assert np.array_equal(np.concatenate([a, b]), full)
    empty = core(x[:0].contiguous())
    assert empty.shape[0] == 0
    core.close()


def test_errors_mirror_elaboration_failures():
    from intfftk_amd import ERR_INVALID, ERR_UNSUPPORTED, IntFFTCore, IntFFTError

    for kw, code in [(dict(NFFT=10, FORMAT=1, RNDMODE=1), ERR_UNSUPPORTED),
                     (dict(NFFT=10, TWDL_WIDTH=28), ERR_UNSUPPORTED),
                     (dict(NFFT=10, DATA_WIDTH=60, TWDL_WIDTH=24, FORMAT=0), ERR_UNSUPPORTED),
                     (dict(NFFT=2), ERR_INVALID), (dict(NFFT=21), ERR_INVALID)]:
        with pytest.raises(IntFFTError) as ei:
            IntFFTCore(**kw)
        assert ei.value.status == code


def test_in_place_exec():
    import torch

    from intfftk_amd import int_fft_single_path

    core = int_fft_single_path(10, 16, 16, 0, 0)
    x = torch.from_numpy(uniform_frames(64, 1024, 15, 9).astype(np.int16)).cuda()
    want = core(x).clone()
    got = core(x, out=x)
    assert torch.equal(got, want)
    core.close()


@pytest.mark.parametrize("log2n,direction,batch", [(4, "FWD", 1000), (5, "PAIR", 77), (7, "FWD", 1001), (7, "INV", 1001), (9, "PAIR", 33),
                                                   (11, "FWD", 35), (12, "INV", 17), (13, "FWD", 515), (16, "PAIR", 5), (17, "INV", 3)])
def test_in_place_exec_all_kernel_families(log2n, direction, batch):
    """d_in == d_out (equal containers) through every packed kernel family, with ragged batches: each kernel reads the
    whole chunk / tile it is about to overwrite before its first store, multi-pass plans go through the plan scratch."""
    import torch

    from intfftk_amd import IntFFTCore

    core = IntFFTCore(log2n, 16, 16, 0, 0, "NEW", direction)
    x = torch.from_numpy(uniform_frames(batch, 1 << log2n, 15, 77 + log2n).astype(np.int16)).cuda()
    want = core(x).clone()
    got = core(x, out=x)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    core.close()


def test_concurrent_plans_on_two_streams():
    """Re-entrancy across plans (include/intfft.h): two plans -- one of them a three-pass plan with its own scratch --
    driven back to back on two streams without host synchronisation give the same bits as run alone."""
    import torch

    from intfftk_amd import IntFFTCore

    a = IntFFTCore(14, 16, 16, 0, 0, "NEW", "FWD")
    b = IntFFTCore(10, 16, 16, 1, 0, "NEW", "FWD")
    xa = torch.from_numpy(uniform_frames(300, 1 << 14, 15, 1).astype(np.int16)).cuda()
    xb = torch.from_numpy(uniform_frames(5000, 1024, 15, 2).astype(np.int16)).cuda()
    want_a, want_b = a(xa).clone(), b(xb).clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs_a, outs_b = [], []
    for _ in range(10):
        with torch.cuda.stream(sa):
            outs_a.append(a(xa))
        with torch.cuda.stream(sb):
            outs_b.append(b(xb))
    torch.cuda.synchronize()
    assert all(torch.equal(o, want_a) for o in outs_a) and all(torch.equal(o, want_b) for o in outs_b)
    a.close()
    b.close()


@pytest.mark.parametrize("chunk", [0, 1, 7, 64])
def test_exec_host_streaming(chunk):
    """intfft_exec_host: chunked double-buffered H2D/transform/D2H gives the same rows as one
    resident call, for any chunking (incl. a ragged last chunk and a multi-pass plan)."""
    from intfftk_amd import IntFFTCore

    for cfg in [(10, 16, 16, 0, 0, "FWD"), (8, 24, 16, 1, 0, "PAIR"), (15, 16, 16, 0, 0, "FWD")]:
        log2n, dw, tw, fmt, rnd, d = cfg
        core = IntFFTCore(log2n, dw, tw, fmt, rnd, "NEW", d)
        x = uniform_frames(37 if log2n < 15 else 5, 1 << log2n, dw, 21)
        got = core.exec_host(x.astype({2: np.int16, 4: np.int32, 8: np.int64}[core.in_container]), chunk).astype(np.int64)
        want = run_ref(x, log2n, dw, tw, fmt, rnd, True, direction=d)
        assert np.array_equal(got, want), (cfg, chunk)
        core.close()


@pytest.mark.parametrize("rnd", [0, 1])
@pytest.mark.parametrize("in_order", ["NATURAL", "HALVES"])
@pytest.mark.parametrize("out_order", ["NATURAL", "BITREV"])
def test_fast1024_native_orders(rnd, in_order, out_order):
    """The wave kernel's four order combinations (int_fftNk's native HALVES -> BITREV included),
    on guard-bit frames (fast extraction) and full-scale frames (exact extraction)."""
    x = np.concatenate([edge_frames(1024, 16), uniform_frames(600, 1024, 15, 31), uniform_frames(40, 1024, 16, 32)])
    info = check(x, 10, 16, 16, 0, rnd, True, in_order=in_order, out_order=out_order)
    assert info["fast_path"] == 1


@pytest.mark.parametrize("in_order", ["NATURAL", "BITREV"])
@pytest.mark.parametrize("out_order", ["NATURAL", "HALVES"])
def test_fast1024x_native_orders(in_order, out_order):
    """The inverse wave kernel's order combinations (int_ifftNk's native BITREV -> HALVES included)."""
    x = np.concatenate([edge_frames(1024, 16), uniform_frames(300, 1024, 15, 41), uniform_frames(20, 1024, 16, 42)])
    info = check(x, 10, 16, 16, 0, 0, True, direction="INV", in_order=in_order, out_order=out_order)
    assert info["fast_path"] == 1


@pytest.mark.parametrize("log2n", [7, 9, 10, 11, 12])
@pytest.mark.parametrize("direction,time_order,freq_order", [("FWD", "HALVES", "BITREV"), ("FWD", "NATURAL", "BITREV"),
                                                              ("FWD", "HALVES", "NATURAL"), ("INV", "HALVES", "BITREV"),
                                                              ("INV", "NATURAL", "BITREV"), ("INV", "HALVES", "NATURAL")])
def test_round_mode_native_orders(log2n, direction, time_order, freq_order):
    """RNDMODE = 1 with the cores' own beat orders (int_fftNk: HALVES in / BITREV out, int_ifftNk: BITREV in / HALVES out) on the
    packed single-pass kernels."""
    n = 1 << log2n
    fp = 1 << max(0, 10 - log2n)
    x = np.concatenate([edge_frames(n, 16), uniform_frames(fp + 3, n, 16, 600 + log2n), uniform_frames(2 * fp + 1, n, 15, 601 + log2n)])
    kw = dict(in_order=time_order, out_order=freq_order) if direction == "FWD" else dict(in_order=freq_order, out_order=time_order)
    info = check(x, log2n, 16, 16, 0, 1, True, direction=direction, **kw)
    assert info["fast_path"] == 1 and "_i16" in info["kernel_name"], info


@pytest.mark.parametrize("tw,new", [(16, True), (16, False), (12, True), (8, True)])
def test_fast1024u_unscaled_wave_kernel(tw, new):
    """The unscaled (bit-growth) wave kernel -- the testbench's "UNSCALED" mode at N = 1024: guard-bit frames
    (wrap-free path), full-scale frames and the edge patterns (exact path with the per-stage width wrap)."""
    x = np.concatenate([edge_frames(1024, 16), uniform_frames(300, 1024, 15, 51), uniform_frames(40, 1024, 16, 52),
                        chirp_frame(1024)[None] * 64])
    info = check(x, 10, 16, tw, 1, 0, new)
    assert info["fast_path"] == 1 and info["kernel_name"].startswith("k_fft1024_u32")


@pytest.mark.parametrize("log2n", [6, 7, 8, 9])
def test_unscaled_wave_kernel_short_frames(log2n):
    """The testbench's "UNSCALED" UUT at 64 <= N < 1024 (NFFT = 7 is what fft_signle_test.vhd ships with)."""
    n = 1 << log2n
    for batch, seed in [(1, 1), (3, 2), ((1 << (10 - log2n)) + 1, 3), (1000, 4), (4099, 5)]:
        x = np.concatenate([uniform_frames(batch, n, 15, 500 + seed), edge_frames(n, 16),
                            uniform_frames(5, n, 16, 600 + seed)])
        info = check(x, log2n, 16, 16, 1, 0, True)
        assert info["fast_path"] == 1 and info["kernel_name"].startswith("k_fft1024_u32")
    check(uniform_frames(77, n, 16, 9), log2n, 16, 12, 1, 0, True)
    check(uniform_frames(77, n, 16, 9), log2n, 16, 16, 1, 0, False)


@pytest.mark.parametrize("log2n,direction", [(6, "INV"), (7, "INV"), (8, "INV"), (9, "INV"), (10, "INV"),
                                             (6, "PAIR"), (7, "PAIR"), (8, "PAIR")])
def test_unscaled_wave_kernel_inverse_and_pair(log2n, direction):
    """int_ifftNk / int_fft_ifft_pair with FORMAT = 1 (fft_double_test.vhd:83-88 ships NFFT = 7, FORMAT = 1): bit
    growth through both cores; the pair's later IFFT stages run the two-DSP multiplier regime."""
    n = 1 << log2n
    for batch, seed in [(1, 1), (3, 2), ((1 << (10 - log2n)) + 1, 3), (1000, 4)]:
        x = np.concatenate([uniform_frames(batch, n, 15, 900 + seed), edge_frames(n, 16),
                            uniform_frames(5, n, 16, 950 + seed)])
        info = check(x, log2n, 16, 16, 1, 0, True, direction=direction)
        assert info["fast_path"] == 1 and info["kernel_name"].startswith("k_fft1024ux_u32")
    check(uniform_frames(77, n, 16, 9), log2n, 16, 12, 1, 0, True, direction=direction)
    check(uniform_frames(77, n, 16, 9), log2n, 16, 16, 1, 0, False, direction=direction)  # XSER "OLD": earlier dbl18


def narrow_packed(dw, tw, fmt, rnd):
    """DATA_WIDTH 9 .. 15, scaled, twiddles of at most 16 bits: served by the packed int16 kernels (intfft_pk16.hpp)"""
    return 9 <= dw <= 15 and fmt == 0 and 8 <= tw <= 16  # both sum / difference modes (single-pass kernels)


W32_CASES = [(12, 16, 0, 0), (12, 16, 0, 1), (14, 18, 0, 0), (18, 18, 0, 0), (24, 24, 0, 1), (32, 24, 0, 0), (32, 16, 0, 0),
             (32, 16, 0, 1), (8, 8, 0, 0), (20, 16, 1, 0), (22, 24, 1, 0), (10, 12, 1, 0), (16, 18, 0, 0), (16, 24, 1, 0),
             (26, 26, 0, 0), (5, 10, 1, 0)]


@pytest.mark.parametrize("log2n", [6, 7, 8, 9, 10])
@pytest.mark.parametrize("case", W32_CASES)
def test_general_width_wave_kernel(log2n, case):
    """Any DATA_WIDTH / TWDL_WIDTH / FORMAT / RNDMODE within 32 bits at 64 <= N <= 1024: every multiplier regime
    reachable below 33 bits (sngl, dbl18, sngl25, dbl35), all three sum/difference variants, both containers."""
    dw, tw, fmt, rnd = case
    if dw + fmt * log2n > 32:
        pytest.skip("results exceed 32 bits: served by the 64-bit generic kernel")
    n = 1 << log2n
    for new in (True, False):
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, new), C.FWD) != 0:
            continue
        x = np.concatenate([edge_frames(n, dw), uniform_frames((1 << (10 - log2n)) + 3, n, dw, 40 + dw),
                            uniform_frames(70, n, max(2, dw - 1), 41 + dw)])
        info = check(x, log2n, dw, tw, fmt, rnd, new)
        assert info["fast_path"] == 1 and info["kernel_name"].startswith("k_fft1024_i16" if narrow_packed(*case) else "k_fft1024_w32"), info


@pytest.mark.parametrize("log2n", [11, 12])
@pytest.mark.parametrize("case", [(16, 16, 1, 0), (16, 16, 0, 1), (12, 16, 0, 0), (20, 24, 1, 0), (32, 24, 0, 0), (32, 16, 0, 1),
                                  (18, 18, 0, 0), (14, 12, 1, 0)])
def test_general_width_block_kernel(log2n, case):
    """N = 2048 / 4096 with widths within 32 bits: the unscaled 16-bit transform, round mode, other data widths."""
    dw, tw, fmt, rnd = case
    n = 1 << log2n
    for new in (True, False):
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, new), C.FWD) != 0:
            continue
        x = np.concatenate([edge_frames(n, dw), uniform_frames(5, n, dw, 60 + dw), uniform_frames(40, n, max(2, dw - 1), 61 + dw)])
        info = check(x, log2n, dw, tw, fmt, rnd, new)
        packed_round = ((dw, fmt, rnd) == (16, 0, 1) and tw <= 16) or narrow_packed(*case)  # the "ROUNDING" UUT runs on the packed block kernel
        assert info["fast_path"] == 1 and info["kernel_name"].startswith("k_fft4096_i16" if packed_round else "k_fft4096_w32"), info


@pytest.mark.parametrize("log2n", [6, 7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("case", [(16, 16, 0, 1), (12, 16, 0, 0), (12, 16, 0, 1), (18, 18, 0, 0), (24, 24, 0, 1), (32, 24, 0, 0),
                                  (32, 16, 0, 0), (8, 8, 0, 0), (20, 16, 1, 0), (16, 16, 1, 0), (10, 12, 1, 0), (16, 24, 1, 0),
                                  (26, 26, 0, 0)])
def test_general_width_inverse_kernels(log2n, case):
    """int_ifftNk with any widths within 32 bits at 64 <= N <= 4096 (wave kernel up to 1024, block kernel above)."""
    dw, tw, fmt, rnd = case
    if dw + fmt * log2n > 32:
        pytest.skip("results exceed 32 bits")
    if (dw, tw, fmt, rnd) == (16, 16, 1, 0) and log2n <= 10:
        pytest.skip("served by the tuned unscaled kernel")
    n = 1 << log2n
    for new in (True, False):
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, new), C.INV) != 0:
            continue
        fp = 1 << max(0, 10 - log2n)
        x = np.concatenate([edge_frames(n, dw), uniform_frames(fp + 3, n, dw, 70 + dw), uniform_frames(20, n, max(2, dw - 1), 71 + dw)])
        info = check(x, log2n, dw, tw, fmt, rnd, new, direction="INV")
        packed_round = ((dw, fmt, rnd) == (16, 0, 1) and tw <= 16) or narrow_packed(*case)  # packed wave / block kernels
        assert info["fast_path"] == 1 and info["kernel_name"].startswith(("k_fft1024x_i16", "k_fft4096_i16") if packed_round else "k_ifft"), info


@pytest.mark.parametrize("log2n", [3, 5, 6, 7, 10, 11, 12])
@pytest.mark.parametrize("dw", [9, 12, 15])
@pytest.mark.parametrize("tw", [16, 12])
@pytest.mark.parametrize("direction", ["FWD", "INV", "PAIR"])
def test_narrow_data_on_packed_kernels(log2n, dw, tw, direction, monkeypatch):
    """DATA_WIDTH 9 .. 15 (12 / 14-bit converters) in truncate mode run on the packed int16 kernels: guard-safe frames
    (|re|, |im| < 2^(w-2): fast extraction with 16-bit twiddles), full-scale w-bit frames (w-bit exact extraction) and
    containers that hold more than w bits (wrapped to DATA_WIDTH on load), mixed in one batch."""
    n = 1 << log2n
    fp = 1 << max(0, 10 - log2n)
    for new in (True, False):
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, 0, 0, new), DIR[direction]) != 0:
            continue
        x = np.concatenate([edge_frames(n, dw), uniform_frames(fp + 5, n, dw - 1, 300 + dw), uniform_frames(7, n, dw, 301 + dw),
                            uniform_frames(3, n, 16, 302 + dw), uniform_frames(2 * fp + 1, n, dw - 1, 303 + dw)])
        info = check(x, log2n, dw, tw, 0, 0, new, direction=direction)
        want = ("k_fftsmall_i16" if log2n <= 5 else "k_fft1024_i16" if direction == "FWD" and log2n <= 10 else
                "k_fft1024x_i16" if log2n <= 10 else "k_fft4096_i16")
        assert info["fast_path"] == 1 and info["kernel_name"].startswith(want), info
        if new and dw == 12:  # the 32-bit kernels serve the same plan with INTFFT_NO_NARROW16 (A/B against the same oracle)
            with monkeypatch.context() as m:
                m.setenv("INTFFT_NO_NARROW16", "1")
                info = check(x, log2n, dw, tw, 0, 0, new, direction=direction)
                assert "_i16" not in info["kernel_name"], info
    for in_order, out_order in ([("HALVES", "BITREV")] if direction == "FWD" else [("BITREV", "HALVES")] if direction == "INV" else []):
        if log2n >= 7:
            check(uniform_frames(fp + 2, n, dw, 310 + dw), log2n, dw, tw, 0, 0, True, direction=direction, in_order=in_order, out_order=out_order)


@pytest.mark.parametrize("log2n,batch", [(13, 37), (13, 515), (14, 9), (15, 5), (16, 5), (16, 33), (17, 3), (18, 2), (19, 1), (20, 1)])
@pytest.mark.parametrize("direction", ["FWD", "INV"])
@pytest.mark.parametrize("tw", [16, 12])
def test_round_mode_multi_pass(log2n, batch, direction, tw, monkeypatch):
    """RNDMODE = 1 (the testbench's "ROUNDING" UUT) at N >= 8192 on the packed multi-pass kernels: plain values between the
    passes, rhu2 sums, exact extraction; two-pass and three-pass splits of the same plan against the oracle."""
    if tw != 16 and log2n > 16:
        pytest.skip("long frames: one twiddle width is enough")
    n = 1 << log2n
    x = np.concatenate([uniform_frames(batch, n, 16, 500 + log2n), edge_frames(n, 16)[3:6], uniform_frames(2, n, 15, 501 + log2n)])
    if log2n <= 14:  # N = 8192 / 16384 run ONE pass since round 4 (test_single_pass_n8192_n16384_round_mode): this test keeps the multi-pass plans covered
        monkeypatch.setenv("INTFFT_NO_FAST16K", "1")
    info = check(x, log2n, 16, tw, 0, 1, True, direction=direction)
    assert info["compute_word"] == 2 and info["kernel_name"].startswith(("k_big2", "k_mid")), info
    # every length in two passes since round 4: the 32-register passes of N = 2^17, 2^18 and the half-line tiles of N = 2^19, 2^20 in their
    # ROUND instantiations (quarter turns through the negated twiddle: D = rhu2(A - B) can be -2^15, -D is not exact)
    assert info["n_passes"] == 2, info
    if log2n >= 17:  # ... against the three-pass plans they replace
        with monkeypatch.context() as m:
            m.setenv("INTFFT_NO_BIG2P" if log2n <= 18 else "INTFFT_NO_BIG2X", "1")
            got3, info3 = run_gpu(x, log2n, 16, tw, 0, 1, True, direction=direction)
            assert info3["n_passes"] == 3, info3
        got2, _ = run_gpu(x, log2n, 16, tw, 0, 1, True, direction=direction)
        assert np.array_equal(got2, got3)
    if log2n <= 16:
        with monkeypatch.context() as m:
            m.setenv("INTFFT_NO_TWOPASS", "1")
            info = check(x[:batch + 2], log2n, 16, tw, 0, 1, True, direction=direction)
            assert info["n_passes"] == 3, info
    if batch <= 37:  # the cores' native beat orders (HALVES on the time side, BITREV on the frequency side) and the mixed ones
        t_o, f_o = ("in_order", "out_order") if direction == "FWD" else ("out_order", "in_order")
        for time_order, freq_order in (("HALVES", "BITREV"), ("NATURAL", "BITREV"), ("HALVES", "NATURAL")):
            info = check(x[:batch + 3], log2n, 16, tw, 0, 1, True, direction=direction, **{t_o: time_order, f_o: freq_order})
            assert info["kernel_name"].startswith(("k_big2", "k_mid")), info


@pytest.mark.parametrize("log2n", [3, 5, 7, 10, 11, 12, 13, 16, 17])
@pytest.mark.parametrize("dw", [9, 14, 15])
@pytest.mark.parametrize("direction", ["FWD", "INV", "PAIR"])
def test_narrow_data_round_mode(log2n, dw, direction):
    """RNDMODE = 1 on narrow data (e.g. a 14-bit converter with rounding): the packed kernels with the w-bit wrap of the rhu2
    differences (+2^(w-1) wraps to -2^(w-1), int_dif2_fly.vhd:173-218).  The edge frames hold the operands that reach it
    (alternating +full / -full scale, most negative value everywhere); containers with more than w bits are wrapped on load."""
    n = 1 << log2n
    fp = 1 << max(0, 10 - log2n)
    nb = 3 if log2n >= 13 else fp + 5
    for tw in (16, 11):
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, 0, 1, True), DIR[direction]) != 0:
            continue
        x = np.concatenate([edge_frames(n, dw), uniform_frames(nb, n, dw, 800 + dw), uniform_frames(2, n, 16, 801 + dw),
                            uniform_frames(nb, n, dw - 1, 802 + dw)])
        # +max next to -min in both orders and both components: the rhu2 difference that needs the wrap, at every stage distance
        hi, lo = (1 << (dw - 1)) - 1, -(1 << (dw - 1))
        x[-1, 0::2, :] = hi
        x[-1, 1::2, :] = lo
        x[-2, : n // 2, :] = hi
        x[-2, n // 2 :, :] = lo
        info = check(x, log2n, dw, tw, 0, 1, True, direction=direction)
        if log2n <= 12:
            assert "_i16" in info["kernel_name"], info
        elif log2n <= 14:  # one pass since round 4: k_fft16k_i16<., ., ., ROUND = 2>
            assert info["kernel_name"] == "k_fft16k_i16", info
        elif not (direction == "PAIR" and log2n > 16):  # the multi-pass kernels in their ROUND = 2 forms
            assert info["kernel_name"].startswith(("k_big2", "k_mid")), info


@pytest.mark.parametrize("log2n,batch", [(13, 37), (13, 259), (14, 9), (15, 5), (16, 5), (16, 19), (17, 3), (18, 2), (19, 1), (20, 1)])
@pytest.mark.parametrize("tw", [16, 12])
def test_round_mode_pair_multi_pass(log2n, batch, tw, monkeypatch):
    """int_fft_ifft_pair with RNDMODE = 1 at N = 8192 .. 65536: DIF pass, the pair of STAGE 7..0 / 0..7 per 256-point group
    (k_mid_pair in the DIF packing), DIT pass."""
    n = 1 << log2n
    x = np.concatenate([uniform_frames(batch, n, 16, 700 + log2n), edge_frames(n, 16)[3:6], uniform_frames(2, n, 15, 701 + log2n)])
    if tw != 16 and log2n > 16:
        pytest.skip("long frames: one twiddle width is enough")
    if log2n <= 14:  # one pass since round 4 (test_single_pass_n8192_n16384_round_mode): keep the three-pass pair of these lengths covered
        monkeypatch.setenv("INTFFT_NO_FAST16K", "1")
    info = check(x, log2n, 16, tw, 0, 1, True, direction="PAIR")
    assert info["kernel_name"] == ("k_big20_p1/k_mid_pair/q1" if log2n <= 16 else "k_big20_p1/k_fft4096_i16<MID>/q1"), info
    if log2n in (13, 16):  # the same plan through the 4096-point middle pass
        with monkeypatch.context() as m:
            m.setenv("INTFFT_NO_TWOPASS", "1")
            info = check(x[:batch + 2], log2n, 16, tw, 0, 1, True, direction="PAIR")
            assert info["kernel_name"] == "k_big20_p1/k_fft4096_i16<MID>/q1", info


@pytest.mark.parametrize("log2n,batch", [(13, 37), (14, 9), (15, 5), (16, 5), (17, 3), (18, 2), (19, 1), (20, 1)])
@pytest.mark.parametrize("dw,tw", [(12, 16), (9, 16), (14, 12)])
@pytest.mark.parametrize("direction", ["FWD", "INV", "PAIR"])
def test_narrow_data_multi_pass(log2n, batch, dw, tw, direction):
    """DATA_WIDTH 9 .. 15 at N >= 8192: the packed multi-pass kernels (every pass votes its guard condition at w bits; exact
    paths extract w bits; first passes wrap containers that hold more than w bits)."""
    if (dw, tw) != (12, 16) and log2n in (15, 19, 20):
        pytest.skip("long frames: one width is enough")
    n = 1 << log2n
    x = np.concatenate([uniform_frames(batch, n, dw - 1, 400 + dw), uniform_frames(2, n, dw, 401 + dw), uniform_frames(1, n, 16, 402 + dw),
                        uniform_frames(1, n, dw - 1, 403 + dw)])
    info = check(x, log2n, dw, tw, 0, 0, True, direction=direction)
    assert info["compute_word"] == 2 and info["kernel_name"].startswith(("k_big", "k_mid", "k_fft16k")), info  # (N = 8192 / 16384 FWD / INV: one pass)
    if log2n in (13, 16):
        for in_order, out_order in ([("HALVES", "BITREV")] if direction == "FWD" else [("BITREV", "HALVES")] if direction == "INV" else []):
            check(x[:batch + 2], log2n, dw, tw, 0, 0, True, direction=direction, in_order=in_order, out_order=out_order)


@pytest.mark.parametrize("case", [(10, 24, 24), (10, 24, 16), (10, 23, 18), (11, 23, 24), (11, 22, 16), (12, 22, 24), (12, 21, 16)])
def test_unscaled_results_of_33_and_34_bits(case):
    """Unscaled plans whose results need 33 / 34 bits (24-bit data at N = 1024, 22-bit at N = 4096): every multiplier stage
    still fits 32 bits, the two multiplier-free stages run in 64 bits, results in int64 containers."""
    log2n, dw, tw = case
    n = 1 << log2n
    for new in (True, False):
        x = np.concatenate([edge_frames(n, dw), uniform_frames(9, n, dw, 170 + dw), uniform_frames(30, n, dw - 1, 171 + dw)])
        info = check(x, log2n, dw, tw, 1, 0, new)
        assert info["fast_path"] == 1 and info["out_container"] == 8, info


@pytest.mark.parametrize("case", [(12, 24, 24), (12, 24, 16), (12, 23, 18), (12, 23, 25), (11, 24, 24), (11, 25, 16), (11, 25, 20),
                                  (12, 24, 10)])
def test_unscaled_results_of_35_and_36_bits(case, monkeypatch):
    """24-bit unscaled data at N = 2048 / 4096 (the lengths below BASELINE config 3's): STAGE 4 still fits 32 bits, the whole
    last register round of k_fft4096_w32 (STAGE 3, 2 with their multipliers, then 1, 0) runs in 64 bits (gfly64); every
    multiplier regime the widths reach, both XSER, odd batches incl. the two-frames-per-block form at N = 2048; and equal to
    the all-64-bit block kernel (INTFFT_NO_FASTW32) and the generic kernel (+ INTFFT_NO_FASTW64) on the same plan."""
    log2n, dw, tw = case
    n = 1 << log2n
    for new in (True, False):
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, 1, 0, new), C.FWD) != 0:
            continue
        x = np.concatenate([edge_frames(n, dw), uniform_frames(7, n, dw, 270 + dw), uniform_frames(12, n, dw - 1, 271 + dw)])
        info = check(x, log2n, dw, tw, 1, 0, new)
        assert info["fast_path"] == 1 and info["out_container"] == 8 and info["kernel_name"] == "k_fft4096_w32", info
        assert info["out_bits"] == dw + log2n and 35 <= info["out_bits"] <= 36
    x = uniform_frames(5, n, dw, 99)
    a, _ = run_gpu(x, log2n, dw, tw, 1, 0, True)
    monkeypatch.setenv("INTFFT_NO_FASTW32", "1")  # ... then the all-64-bit block kernel serves the plan,
    b, ib = run_gpu(x, log2n, dw, tw, 1, 0, True)
    assert ib["kernel_name"] == "k_fft4096_w64" and np.array_equal(a, b), ib
    monkeypatch.setenv("INTFFT_NO_FASTW64", "1")  # ... then the generic one
    b, ib = run_gpu(x, log2n, dw, tw, 1, 0, True)
    assert ib["kernel_name"] == "k_pass<long>" and np.array_equal(a, b), ib


@pytest.mark.parametrize("log2n,dw,tw,batch", [(13, 24, 24, 9), (13, 24, 16, 8), (14, 24, 24, 5), (14, 24, 18, 4), (15, 24, 24, 3), (15, 24, 16, 1),
                                               (16, 24, 24, 2), (13, 27, 24, 3), (14, 20, 16, 6), (15, 18, 24, 2), (16, 18, 16, 1), (15, 25, 20, 2),
                                               (13, 20, 24, 17)])
def test_wide_two_pass_kernels_n8192_to_n65536(log2n, dw, tw, batch, monkeypatch):
    """BASELINE config 3's kernel pair (int32 first pass, 64-bit second pass) at N = 2^13 .. 2^16 and DATA_WIDTH 17 .. 27:
    shorter lengths run as virtual 2^16-point frames (batches that do not fill the last group included), second-pass stages
    of at most 32 bits take the general slice form.  Bit-exact to the oracle incl. the edge frames, and equal to the generic
    kernels it replaces."""
    n = 1 << log2n
    x = np.concatenate([uniform_frames(batch, n, dw, 300 + log2n + dw), edge_frames(n, dw)[[0, 1, 4]]])[:max(batch, 1) + (2 if log2n < 15 else 0)]
    for new in (True, False):
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, 1, 0, new), C.FWD) != 0:
            continue
        info = check(x, log2n, dw, tw, 1, 0, new)
        assert info["kernel_name"] == "k_wide16_p1+p2" and info["n_passes"] == 2 and info["out_container"] == 8, info
    a, _ = run_gpu(x, log2n, dw, tw, 1, 0, True)
    monkeypatch.setenv("INTFFT_NO_WIDE16", "1")
    b, ib = run_gpu(x, log2n, dw, tw, 1, 0, True)
    assert ib["kernel_name"] != "k_wide16_p1+p2" and np.array_equal(a, b)


@pytest.mark.parametrize("log2n,dw,tw,batch", [(13, 24, 24, 9), (13, 24, 16, 8), (14, 24, 24, 5), (14, 22, 18, 4), (15, 24, 24, 3), (15, 24, 16, 1),
                                               (16, 24, 24, 2), (16, 24, 16, 1), (14, 20, 16, 6), (15, 18, 24, 2), (16, 17, 16, 1), (13, 21, 24, 17)])
def test_wide_two_pass_inverse_n8192_to_n65536(log2n, dw, tw, batch, monkeypatch):
    """int_ifftNk on the class of BASELINE config 3 (round 4): k_wide16_q1 (bit-reversed gather, DIT STAGE 0..7 on int32) + k_wide16_q2
    (STAGE 8..L-1 on 64-bit words), N = 2^13 .. 2^16, DATA_WIDTH 17 .. 24 (int_ifftNk.vhd:183-341; the multiplier of STAGE s works
    at 24 + s bits, int_dit2_fly.vhd:290-325): bit-exact to the oracle incl. edge frames and partial last groups, both XSER, and equal to
    the generic kernels it replaces (INTFFT_NO_WIDE16)."""
    n = 1 << log2n
    x = np.concatenate([uniform_frames(batch, n, dw, 350 + log2n + dw), edge_frames(n, dw)[[0, 1, 4]]])[:max(batch, 1) + (2 if log2n < 15 else 0)]
    for new in (True, False):
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, 1, 0, new), C.INV) != 0:
            continue
        info = check(x, log2n, dw, tw, 1, 0, new, direction="INV")
        assert info["kernel_name"] == "k_wide16_q1+q2" and info["n_passes"] == 2 and info["out_container"] == 8, info
    a, _ = run_gpu(x, log2n, dw, tw, 1, 0, True, direction="INV")
    monkeypatch.setenv("INTFFT_NO_WIDE16", "1")
    b, ib = run_gpu(x, log2n, dw, tw, 1, 0, True, direction="INV")
    assert ib["kernel_name"] != "k_wide16_q1+q2" and np.array_equal(a, b)


WIDE64_CASES = [
    # (log2n, dw, tw, batch): DATA_WIDTH 25 .. 32 with bit growth at N = 2^13 .. 2^16 (41 .. 48-bit results), and what class 1 leaves out
    (13, 32, 16, 9), (13, 28, 16, 3), (14, 32, 16, 5), (14, 30, 14, 4), (15, 32, 16, 3), (15, 26, 16, 1), (16, 32, 16, 2), (16, 32, 12, 1),
    (16, 25, 16, 1), (16, 28, 18, 1), (14, 32, 10, 2), (13, 27, 24, 2), (16, 24, 12, 1), (13, 32, 8, 17),
]


@pytest.mark.parametrize("direction", ["FWD", "INV"])
@pytest.mark.parametrize("log2n,dw,tw,batch", WIDE64_CASES)
def test_wide64_first_pass_data_width_up_to_32(log2n, dw, tw, batch, direction, monkeypatch):
    """DATA_WIDTH 25 .. 32 (the wrapper's documented range ends at 32: int_fft_single_path.vhd:15) with FORMAT = 1 at N = 2^13 .. 2^16: results of
    41 .. 48 bits, the widths leave int32 inside the first pass, so BOTH passes run on 64-bit words -- k_wide64_p1 + k_wide16_p2<IN64> forward,
    k_wide64_q1 + k_wide16_q2<IN64> inverse (round 5; k_pass<long> before).  Multiplier regimes sngl / dbl18 / trpl18 by stage width
    (int_cmult_dsp48.vhd:184-303, the pre-truncation of int_cmult_dbl18_dsp48.vhd:163-175 as a mask): bit-exact to the oracle for both XSER, on
    full-scale and edge frames and partial last groups (N < 2^16: virtual frames), and equal to the generic kernels (INTFFT_NO_WIDE16)."""
    d = {"FWD": C.FWD, "INV": C.INV}[direction]
    n = 1 << log2n
    x = np.concatenate([uniform_frames(batch, n, dw, 900 + log2n + dw + tw), edge_frames(n, dw)[[0, 1, 3, 4, 5]]])[:batch + (3 if log2n < 15 else 1)]
    name = "k_wide64_q1+k_wide16_q2" if direction == "INV" else "k_wide64_p1+k_wide16_p2"
    ran = 0
    for new in (True, False):
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, 1, 0, new), d) != 0:
            continue
        info = check(x, log2n, dw, tw, 1, 0, new, direction=direction)
        if info["kernel_name"] != name:  # class 1 (the int32 first pass still fits), or a stage outside the 64-bit product / slice conditions
            assert info["kernel_name"].startswith("k_pass") or info["kernel_name"].startswith("k_wide16_"), info
            continue
        assert info["n_passes"] == 2 and info["out_container"] == 8 and info["in_container"] == 4, info
        ran += 1
    if (dw, tw) in ((32, 16), (32, 12), (28, 16), (25, 16), (32, 10)):
        assert ran >= 1, "the class itself must be served by the dedicated kernels"
    a, ia = run_gpu(x, log2n, dw, tw, 1, 0, True, direction=direction)
    monkeypatch.setenv("INTFFT_NO_WIDE16", "1")
    b, ib = run_gpu(x, log2n, dw, tw, 1, 0, True, direction=direction)
    assert ib["kernel_name"].startswith("k_pass") and np.array_equal(a, b)


@pytest.mark.parametrize("log2n,dw,tw,batch", [(16, 24, 24, 2), (13, 24, 16, 9), (14, 22, 18, 5), (15, 24, 24, 3), (16, 32, 16, 1), (13, 32, 16, 11), (14, 28, 16, 3), (15, 32, 12, 2)])
def test_wide_two_pass_native_orders(log2n, dw, tw, batch, monkeypatch):
    """The cores' own beat orders on the two-pass wide classes (round 5): int_fftNk HALVES in -> BITREV out and int_ifftNk BITREV in -> HALVES out
    (int_fftNk.vhd:15-21, int_ifftNk.vhd:15-21) and the mixed forms with natural order, for BASELINE config 3's class (k_wide16_p1/p2, q1/q2) and the
    DATA_WIDTH 25 .. 32 class (k_wide64_*): a HALVES beat is one 16- / 32-byte access of a register pair, BITREV order goes through a 16 x 16 exchange in
    the transpose planes.  Bit-exact to the oracle incl. partial virtual frames, and equal to the generic kernels (INTFFT_NO_WIDE16)."""
    n = 1 << log2n
    x = np.concatenate([uniform_frames(batch, n, dw, 4400 + log2n + dw), edge_frames(n, dw)[[1, 4]]])[:batch + (2 if log2n < 15 else 0)]
    fwd = "k_wide64_p1+k_wide16_p2" if dw + log2n - 8 > 32 or tw < 16 else "k_wide16_p1+p2"
    inv = "k_wide64_q1+k_wide16_q2" if dw + 8 > 32 or tw < 16 else "k_wide16_q1+q2"
    for direction, orders, name in (("FWD", [("HALVES", "BITREV"), ("HALVES", "NATURAL"), ("NATURAL", "BITREV")], fwd),
                                    ("INV", [("BITREV", "HALVES"), ("BITREV", "NATURAL"), ("NATURAL", "HALVES")], inv)):
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, 1, 0, True), DIR[direction]) != 0:
            continue
        for in_o, out_o in orders:
            info = check(x, log2n, dw, tw, 1, 0, True, direction=direction, in_order=in_o, out_order=out_o)
            assert info["kernel_name"] == name and info["n_passes"] == 2, (info, in_o, out_o)
        a, _ = run_gpu(x, log2n, dw, tw, 1, 0, True, direction=direction, in_order=orders[0][0], out_order=orders[0][1])
        with monkeypatch.context() as m:
            m.setenv("INTFFT_NO_WIDE16", "1")
            b, ib = run_gpu(x, log2n, dw, tw, 1, 0, True, direction=direction, in_order=orders[0][0], out_order=orders[0][1])
        assert ib["kernel_name"].startswith("k_pass") and np.array_equal(a, b)


def test_wide_family_random_configurations():
    """Seeded fuzz over the unscaled plans with int64 results (N = 2^10 .. 2^16, DATA_WIDTH 17 .. 30, TWDL_WIDTH 10 .. 25, both
    XSER): whichever kernel the planner picks (k_fft1024_w32 / k_fft4096_w32 with 64-bit tails, k_wide16_p1+p2<L>, k_pass<long>),
    the result is the oracle's, bit for bit."""
    rng = np.random.default_rng(20260929)
    seen, done = set(), 0
    for _ in range(400):
        log2n = int(rng.integers(10, 17))
        dw = int(rng.integers(17, 31))
        tw = int(rng.integers(10, 26))
        new = bool(rng.integers(0, 2))
        if dw + log2n <= 32 or dw + log2n > 44 or C.lib().orc_validate(C.make_params(log2n, dw, tw, 1, 0, new), C.FWD) != 0:
            continue
        n = 1 << log2n
        batch = int(rng.integers(1, 6)) if log2n >= 13 else int(rng.integers(1, 20))
        x = np.concatenate([uniform_frames(batch, n, dw, int(rng.integers(1, 1 << 30))), edge_frames(n, dw)[[1, 3]]])
        info = check(x, log2n, dw, tw, 1, 0, new)
        seen.add(info["kernel_name"])
        done += 1
        if done >= 48:
            break
    # (round 5: what was k_pass<long> here -- DATA_WIDTH beyond the int32 first pass at N >= 8192 -- now runs k_wide64_p1 + k_wide16_p2)
    assert done >= 40 and {"k_wide16_p1+p2", "k_fft4096_w32", "k_wide64_p1+k_wide16_p2"} <= seen, (done, seen)


@pytest.mark.parametrize("batch", [1, 2, 5, 1027])
def test_fast1024u_ragged_batches(batch):
    x = uniform_frames(batch, 1024, 16 if batch % 2 else 15, 200 + batch)
    info = check(x, 10, 16, 16, 1, 0, True)
    assert info["fast_path"] == 1


def test_fast1024u_matches_the_generic_pass_kernel(monkeypatch):
    """Same plan through three device paths -- the tuned unscaled wave kernel, the general-width wave kernel
    (INTFFT_NO_FAST1024U=1) and the generic k_pass<int32> kernels (also INTFFT_NO_FASTW32=1): all agree bit for bit
    (and with the oracle, test_fast1024u_unscaled_wave_kernel)."""
    x = np.concatenate([uniform_frames(64, 1024, 16, 61), uniform_frames(64, 1024, 15, 62)])
    fast, info_f = run_gpu(x, 10, 16, 16, 1, 0, True)
    monkeypatch.setenv("INTFFT_NO_FAST1024U", "1")
    mid, info_m = run_gpu(x, 10, 16, 16, 1, 0, True)
    monkeypatch.setenv("INTFFT_NO_FASTW32", "1")
    slow, info_s = run_gpu(x, 10, 16, 16, 1, 0, True)
    assert info_f["kernel_name"].startswith("k_fft1024_u32") and info_m["kernel_name"].startswith("k_fft1024_w32")
    assert info_s["fast_path"] == 0
    assert np.array_equal(fast, slow) and np.array_equal(mid, slow)


AB_CASES = [(4, 16, 16, 0, 0, "PAIR"), (5, 16, 16, 0, 1, "FWD"), (7, 16, 16, 0, 0, "FWD"), (7, 16, 16, 0, 1, "FWD"), (7, 16, 16, 0, 0, "PAIR"), (7, 16, 16, 1, 0, "PAIR"),
            (9, 16, 16, 1, 0, "INV"), (10, 16, 16, 0, 0, "INV"), (11, 16, 16, 0, 0, "PAIR"), (12, 16, 16, 1, 0, "FWD"),
            (12, 16, 16, 0, 1, "FWD"), (10, 14, 18, 0, 0, "FWD"), (13, 16, 16, 0, 0, "FWD"), (15, 16, 16, 0, 0, "FWD"),
            (16, 24, 24, 1, 0, "FWD"), (18, 16, 16, 0, 0, "FWD"), (14, 16, 16, 0, 0, "PAIR"), (17, 16, 16, 0, 0, "PAIR"), (13, 16, 16, 0, 0, "INV"), (18, 16, 16, 0, 0, "INV"), (14, 16, 16, 1, 0, "FWD"), (16, 12, 16, 0, 1, "FWD"), (10, 24, 24, 1, 0, "FWD"), (12, 22, 16, 1, 0, "FWD"), (14, 16, 16, 1, 0, "INV"), (15, 12, 16, 0, 0, "INV")]


@pytest.mark.parametrize("case", AB_CASES)
def test_dedicated_and_generic_kernels_agree(case, monkeypatch):
    """A/B on the device: the dedicated kernel of a configuration and the generic LDS pass kernels
    (INTFFT_GENERIC_ONLY=1) produce identical bits on batches far larger than the oracle-checked ones -- two
    independent implementations of the same arithmetic, and the generic kernels stay under test."""
    log2n, dw, tw, fmt, rnd, d = case
    batch = max(3, (1 << 22) >> log2n) + 1
    x = uniform_frames(batch, 1 << log2n, dw, 5150 + log2n)
    a, ia = run_gpu(x, log2n, dw, tw, fmt, rnd, True, direction=d)
    monkeypatch.setenv("INTFFT_GENERIC_ONLY", "1")
    b, ib = run_gpu(x, log2n, dw, tw, fmt, rnd, True, direction=d)
    assert ia["kernel_name"] != ib["kernel_name"] and ib["kernel_name"].startswith("k_pass"), (ia, ib)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("log2n", [7, 8, 9])
@pytest.mark.parametrize("rnd", [0, 1])
@pytest.mark.parametrize("in_order,out_order", [("HALVES", "BITREV"), ("NATURAL", "BITREV"), ("HALVES", "NATURAL")])
def test_wave_kernel_short_frames_native_orders(log2n, rnd, in_order, out_order):
    """int_fftNk's own beat orders (HALVES in, BITREV out) at 128 <= N < 1024, ragged batches."""
    n = 1 << log2n
    for batch in (1, (1 << (10 - log2n)) + 1, 517):
        x = np.concatenate([uniform_frames(batch, n, 15, 81 + batch), edge_frames(n, 16)])
        info = check(x, log2n, 16, 16, 0, rnd, True, in_order=in_order, out_order=out_order)
        assert info["fast_path"] == 1 and info["kernel_name"].startswith("k_fft1024_i16")


@pytest.mark.parametrize("log2n", [7, 8, 9])
@pytest.mark.parametrize("in_order,out_order", [("BITREV", "HALVES"), ("NATURAL", "HALVES"), ("BITREV", "NATURAL")])
def test_inverse_wave_kernel_short_frames_native_orders(log2n, in_order, out_order):
    """int_ifftNk's own beat orders (BITREV in, HALVES out) at 128 <= N < 1024."""
    n = 1 << log2n
    for batch in (1, (1 << (10 - log2n)) + 1, 517):
        x = np.concatenate([uniform_frames(batch, n, 15, 91 + batch), edge_frames(n, 16)])
        info = check(x, log2n, 16, 16, 0, 0, True, direction="INV", in_order=in_order, out_order=out_order)
        assert info["fast_path"] == 1 and info["kernel_name"].startswith("k_fft1024x_i16")


@pytest.mark.parametrize("log2n", [11, 12])
@pytest.mark.parametrize("direction,in_order,out_order", [("FWD", "HALVES", "BITREV"), ("FWD", "NATURAL", "BITREV"),
                                                          ("FWD", "HALVES", "NATURAL"), ("INV", "BITREV", "HALVES"),
                                                          ("INV", "NATURAL", "HALVES"), ("INV", "BITREV", "NATURAL")])
def test_block_kernel_native_orders(log2n, direction, in_order, out_order):
    """The cores' own beat orders at N = 2048 / 4096 (odd batch: partial chunk at N = 2048)."""
    n = 1 << log2n
    x = np.concatenate([uniform_frames(37, n, 15, 131 + log2n), edge_frames(n, 16)])
    info = check(x, log2n, 16, 16, 0, 0, True, direction=direction, in_order=in_order, out_order=out_order)
    assert info["fast_path"] == 1 and info["kernel_name"].startswith("k_fft4096_i16")


@pytest.mark.parametrize("log2n", [3, 4, 5])
@pytest.mark.parametrize("direction,rnd", [("FWD", 0), ("FWD", 1), ("INV", 0), ("PAIR", 0)])
def test_lane_per_frame_kernel_n8_to_n32(log2n, direction, rnd):
    """N = 8, 16, 32: a whole frame in the registers of one lane (k_fftsmall_i16); ragged batches, edge frames."""
    n = 1 << log2n
    for batch, seed in [(1, 1), (63, 2), (257, 3), (100003, 4)]:
        x = np.concatenate([uniform_frames(batch, n, 16 if seed % 2 else 15, 1300 + seed), edge_frames(n, 16)])
        info = check(x, log2n, 16, 16, 0, rnd, True, direction=direction)
        assert info["fast_path"] == 1 and info["kernel_name"].startswith("k_fftsmall_i16")
    check(uniform_frames(77, n, 16, 9), log2n, 16, 11, 0, rnd, False, direction=direction)


def test_native_cores_chain_like_the_pair():
    """int_fftNk (HALVES -> BITREV) feeding int_ifftNk (BITREV -> HALVES) equals int_fft_ifft_pair on the
    same frames re-ordered (int_fft_ifft_pair.vhd:209-280 wires exactly this chain)."""
    import torch

    from intfftk_amd import int_fft_ifft_pair, int_fftNk, int_ifftNk

    x = torch.from_numpy(uniform_frames(64, 1024, 15, 77).astype(np.int16)).cuda()
    fwd, inv, pair = int_fftNk(10, 16, 16, 0, 0), int_ifftNk(10, 16, 16, 0, 0), int_fft_ifft_pair(10, 16, 16, 0, 0)
    # natural -> HALVES beats: mem[2i + l] = x[i + 512 l]
    halves = torch.stack([x[:, :512], x[:, 512:]], dim=2).reshape(64, 1024, 2).contiguous()
    y = inv(fwd(halves))
    nat = torch.cat([y[:, 0::2], y[:, 1::2]], dim=1)
    assert torch.equal(nat, pair(x))
    for c in (fwd, inv, pair):
        c.close()


@pytest.mark.parametrize("log2n", [6, 7, 8, 9])
@pytest.mark.parametrize("rnd", [0, 1])
def test_wave_kernel_short_frames(log2n, rnd):
    """64 <= N < 1024: 2^(10 - log2n) frames share a wave of the N = 1024 kernel (the reference testbench's own
    NFFT = 7 is one of them: fft_signle_test.vhd:93).  Guard-bit frames, full-scale frames and edge patterns mixed
    inside one chunk; batch sizes that leave the last chunk partial."""
    n = 1 << log2n
    for batch, seed in [(1, 1), (3, 2), ((1 << (10 - log2n)) + 1, 3), (1000, 4), (4099, 5)]:
        x = np.concatenate([uniform_frames(batch, n, 15, 300 + seed), edge_frames(n, 16),
                            uniform_frames(5, n, 16, 400 + seed)])
        info = check(x, log2n, 16, 16, 0, rnd, True)
        assert info["fast_path"] == 1 and info["kernel_name"].startswith("k_fft1024_i16")
    check(uniform_frames(77, n, 15, 9), log2n, 16, 12, 0, rnd, True)  # narrower twiddles: exact extraction
    check(uniform_frames(77, n, 15, 9), log2n, 16, 16, 0, rnd, False)  # XSER = "OLD"


@pytest.mark.parametrize("log2n", [6, 7, 8, 9])
@pytest.mark.parametrize("direction", ["INV", "PAIR"])
def test_wave_kernel_short_frames_inverse_and_pair(log2n, direction):
    """int_ifftNk / int_fft_ifft_pair at 64 <= N < 1024 (fft_double_test.vhd ships with NFFT = 7)."""
    n = 1 << log2n
    for batch, seed in [(1, 1), (3, 2), ((1 << (10 - log2n)) + 1, 3), (1000, 4), (4099, 5)]:
        x = np.concatenate([uniform_frames(batch, n, 15, 700 + seed), edge_frames(n, 16),
                            uniform_frames(5, n, 16, 800 + seed)])
        info = check(x, log2n, 16, 16, 0, 0, True, direction=direction)
        assert info["fast_path"] == 1 and info["kernel_name"].startswith("k_fft1024x_i16")
    check(uniform_frames(77, n, 15, 9), log2n, 16, 13, 0, 0, True, direction=direction)


@pytest.mark.parametrize("cfg", [(10, "FWD"), (10, "INV"), (10, "PAIR"), (12, "FWD"), (12, "INV"), (12, "PAIR"),
                                 (11, "FWD"), (11, "INV"), (11, "PAIR")])
@pytest.mark.parametrize("batch", [1, 3, 5, 1027])
def test_fast_kernels_ragged_batches(cfg, batch):
    """Persistent-grid kernels: batches smaller than the grid, not a multiple of the waves per block, odd."""
    log2n, d = cfg
    x = uniform_frames(batch, 1 << log2n, 15, 100 + batch)
    info = check(x, log2n, 16, 16, 0, 0, True, direction=d)
    assert info["fast_path"] == 1


@pytest.mark.parametrize("nplans,root,batch", [(1, 0, 5), (2, 0, 7), (3, 1, 10), (4, 3, 2), (3, 2, 4099)])
def test_exec_sharded_c_abi(nplans, root, batch):
    """intfft_exec_sharded: contiguous shards (remainder to the last plans), peer copies, no collective.  On a
    one-GPU box every plan sits on device 0 (peer copies degenerate to device copies); the control flow, shard
    bounds, staging buffers and error paths are the same as across devices."""
    import torch

    from intfftk_amd import IntFFTCore, exec_sharded
    from intfftk_amd import _capi as capi

    ndev = torch.cuda.device_count()
    cores = [IntFFTCore(10, 16, 16, 0, 0, "NEW", "FWD", device=i % ndev) for i in range(nplans)]
    x = uniform_frames(batch, 1024, 15, 31337 + batch)
    xd = torch.from_numpy(x.astype(np.int16)).to("cuda:%d" % (root % ndev))
    y = exec_sharded(cores, xd, root)
    assert np.array_equal(y.cpu().numpy().astype(np.int64), run_ref(x, 10, 16, 16, 0, 0, True))
    # error behaviour: mismatching generics, a repeated plan, bad root
    other = IntFFTCore(10, 16, 16, 0, 1, "NEW", "FWD")
    import ctypes
    arr = (ctypes.c_void_p * 2)(cores[0]._plan, other._plan)
    assert capi.lib().intfft_exec_sharded(arr, 2, 0, xd.data_ptr(), y.data_ptr(), batch) == capi.ERR_INVALID
    arr = (ctypes.c_void_p * 2)(cores[0]._plan, cores[0]._plan)
    assert capi.lib().intfft_exec_sharded(arr, 2, 0, xd.data_ptr(), y.data_ptr(), batch) == capi.ERR_INVALID
    assert capi.lib().intfft_exec_sharded(arr, 2, 5, xd.data_ptr(), y.data_ptr(), batch) == capi.ERR_INVALID
    for c in cores + [other]:
        c.close()


def test_hip_graph_capture_and_replay():
    """intfft_exec is capturable (no allocation / sync inside): replaying the graph on new input data gives
    the same rows as eager execution -- the launch-bound small-batch regime is where this matters."""
    import torch

    from intfftk_amd import int_fft_ifft_pair, int_fft_single_path

    for core in (int_fft_single_path(10, 16, 16, 0, 0), int_fft_ifft_pair(12, 16, 16, 0, 0), int_fft_single_path(7, 16, 16, 1, 0),
                 int_fft_single_path(14, 16, 16, 0, 0), int_fft_single_path(13, 16, 16, 1, 0)):  # the last two: two-pass plans
        x = torch.from_numpy(uniform_frames(16, core.n, 15, 3).astype(np.int16)).cuda()
        y = torch.empty((16, core.n, 2), dtype=core.out_dtype, device="cuda")
        core(x, out=y)  # warm: one-time occupancy queries happen outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(4):
                core(x, out=y)
        x.copy_(torch.from_numpy(uniform_frames(16, core.n, 15, 4).astype(np.int16)).cuda())
        g.replay()
        torch.cuda.synchronize()
        want = core(x).clone()
        assert torch.equal(y, want)
        core.close()


def _fuzz_cases(count, seed):
    """Deterministic pseudo-random generics that the reference can elaborate (orc_validate)."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        log2n = int(rng.integers(3, 15))
        fmt = int(rng.integers(0, 2))
        rnd = 0 if fmt else int(rng.integers(0, 2))
        dw = int(rng.integers(4, 33))
        tw = int(rng.integers(8, 27))
        new = bool(rng.integers(0, 2))
        d = ["FWD", "INV", "PAIR"][int(rng.integers(0, 3))]
        if C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, new), DIR[d]) != 0:
            continue
        out.append((log2n, dw, tw, fmt, rnd, new, d))
    return out


@pytest.mark.parametrize("case", _fuzz_cases(240, 20260928))
def test_fuzz_generics_three_way(case, monkeypatch):
    """Random elaboratable generics: whatever kernel the planner picks, the generic LDS pass kernels and the oracle
    agree bit for bit (ragged batch sizes, full-range data)."""
    log2n, dw, tw, fmt, rnd, new, d = case
    n = 1 << log2n
    batch = int(3 + (log2n * 7 + dw) % 11)
    x = uniform_frames(batch, n, dw, 9000 + log2n * 100 + dw)
    got, info = run_gpu(x, log2n, dw, tw, fmt, rnd, new, direction=d)
    want = run_ref(x, log2n, dw, tw, fmt, rnd, new, direction=d)
    assert np.array_equal(got, want), (case, info["kernel_name"])
    if not info["kernel_name"].startswith("k_pass"):
        monkeypatch.setenv("INTFFT_GENERIC_ONLY", "1")
        gen, _ = run_gpu(x, log2n, dw, tw, fmt, rnd, new, direction=d)
        assert np.array_equal(gen, want), (case, "generic")


def _fuzz_cases_64(count, seed):
    """Random elaboratable generics whose results need 33 .. 64 bits at N = 64 .. 4096 (the 64-bit wave / block kernels' domain and
    the sub-plan pairs), every direction."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        log2n = int(rng.integers(6, 13))
        fmt = int(rng.integers(0, 2))
        rnd = 0 if fmt else int(rng.integers(0, 2))
        dw = int(rng.integers(12, 65))
        tw = int(rng.integers(8, 27))
        new = bool(rng.integers(0, 2))
        d = ["FWD", "INV", "PAIR"][int(rng.integers(0, 3))]
        ob = dw + (fmt * log2n) * (2 if d == "PAIR" else 1)
        if ob <= 32 or ob > 64 or C.lib().orc_validate(C.make_params(log2n, dw, tw, fmt, rnd, new), DIR[d]) != 0:
            continue
        out.append((log2n, dw, tw, fmt, rnd, new, d))
    return out


@pytest.mark.parametrize("case", _fuzz_cases_64(120, 20260929))
def test_fuzz_64_bit_word_plans(case, monkeypatch):
    """Seeded fuzz over the plans on 64-bit words at N = 64 .. 4096: whichever kernel serves them (k_fft1024_w64 / k_ifft1024_w64,
    the block kernels, k_fft1024_w32 / k_fft4096_w32 with their 64-bit tails, forward + inverse sub-plans for a pair, k_pass<long>),
    the oracle and the generic kernels agree bit for bit; ragged batches, full-range data, edge frames."""
    log2n, dw, tw, fmt, rnd, new, d = case
    n = 1 << log2n
    batch = int(2 + (log2n * 5 + dw) % 9)
    x = np.concatenate([uniform_frames(batch, n, dw, 15000 + log2n * 100 + dw), edge_frames(n, dw)[[1, 4]]])
    got, info = run_gpu(x, log2n, dw, tw, fmt, rnd, new, direction=d)
    want = run_ref(x, log2n, dw, tw, fmt, rnd, new, direction=d)
    assert np.array_equal(got, want), (case, info["kernel_name"])
    if not info["kernel_name"].startswith("k_pass"):
        monkeypatch.setenv("INTFFT_GENERIC_ONLY", "1")
        gen, ig = run_gpu(x, log2n, dw, tw, fmt, rnd, new, direction=d)
        assert ig["kernel_name"].startswith("k_pass") and np.array_equal(gen, want), (case, "generic")


def _fuzz_order_cases(count, seed):
    rng = np.random.default_rng(seed)
    names = list(ORD)
    out = []
    for c in _fuzz_cases(count, seed + 1):
        out.append(c + (names[int(rng.integers(0, 4))], names[int(rng.integers(0, 4))]))
    return out


@pytest.mark.parametrize("case", _fuzz_order_cases(100, 777))
def test_fuzz_generics_with_orders(case):
    """Random generics x random I/O orders (HALVES / BITREV / BITREV_LANES / NATURAL on either side)."""
    log2n, dw, tw, fmt, rnd, new, d, in_o, out_o = case
    x = uniform_frames(4, 1 << log2n, dw, 12000 + log2n * 10 + dw)
    check(x, log2n, dw, tw, fmt, rnd, new, direction=d, in_order=in_o, out_order=out_o)


@pytest.mark.parametrize("log2n", [3, 4, 5])
@pytest.mark.parametrize("direction", ["FWD", "INV", "PAIR"])
@pytest.mark.parametrize("dw", [16, 12])
def test_rounding_mode_short_frames(log2n, direction, dw):
    """RNDMODE = 1 at N = 8 .. 32, every direction, on k_fftsmall_i16 (one frame per lane, exact extraction)."""
    n = 1 << log2n
    for tw in (16, 10):
        x = np.concatenate([edge_frames(n, dw), uniform_frames(1000, n, dw, 900 + log2n + dw), uniform_frames(65, n, 16, 901 + log2n)])
        info = check(x, log2n, dw, tw, 0, 1, True, direction=direction)
        assert info["fast_path"] == 1 and info["kernel_name"] == "k_fftsmall_i16", info


@pytest.mark.parametrize("log2n", [6, 7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("direction", ["FWD", "INV", "PAIR"])
def test_rounding_mode_on_the_packed_kernels(log2n, direction, monkeypatch):
    """RNDMODE = 1 (the testbench's "ROUNDING" UUT, fft_signle_test.vhd:93-112) on the packed wave / block kernels:
    bit-exact to the oracle incl. the full-scale edge frames (rhu2 of the maximum wraps, int_dif2_fly.vhd:173-218),
    odd batches, and equal to the general-width kernels they replace (INTFFT_NO_PACKED_ROUND)."""
    n = 1 << log2n
    x = np.concatenate([edge_frames(n, 16), uniform_frames(45, n, 16, 600 + log2n)])
    info = check(x, log2n, 16, 16, 0, 1, True, direction=direction)
    assert info["fast_path"] == 1 and info["compute_word"] == 2, info
    monkeypatch.setenv("INTFFT_NO_PACKED_ROUND", "1")
    got_b, info_b = run_gpu(x, log2n, 16, 16, 0, 1, True, direction=direction)
    monkeypatch.delenv("INTFFT_NO_PACKED_ROUND")
    got_a, _ = run_gpu(x, log2n, 16, 16, 0, 1, True, direction=direction)
    assert np.array_equal(got_a, got_b)
    if direction != "FWD" or log2n > 10:
        assert info_b["compute_word"] != 2 or info_b["fast_path"] == 0  # the fallback really is a different kernel
