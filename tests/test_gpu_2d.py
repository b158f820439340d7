"""GPU parity of the N > 512K "2D-FFT scheme" plans (intfft_plan_create_2d) against the oracle's flat form, bit-exact:
small lengths over every direction / order / width class, and the lengths the scheme exists for (2^20, 2^21)."""
import numpy as np
import pytest

from oracle import oracle_c as C
from tests.helpers import edge_frames, to_complex, uniform_frames

pytestmark = pytest.mark.gpu
DIRS = {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}
ORD = {"NATURAL": C.NATURAL, "BITREV": C.BITREV, "HALVES": C.HALVES, "BITREV_LANES": C.BITREV_LANES}
DIR = {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}
NP = {2: np.int16, 4: np.int32, 8: np.int64}


def run_gpu(x, log2n, l1, dw, tw, fmt, rnd, new, direction="FWD", in_order="NATURAL", out_order="NATURAL"):
    import torch

    from intfftk_amd import IntFFTCore

    core = IntFFTCore(log2n, dw, tw, fmt, rnd, "NEW" if new else "OLD", direction, in_order, out_order, NFFT1=l1)
    y = core(torch.from_numpy(np.ascontiguousarray(x.astype(NP[core.in_container]))).cuda())
    torch.cuda.synchronize()
    info = core.info
    core.close()
    return y.cpu().numpy().astype(np.int64), info


def check(x, log2n, l1, dw, tw, fmt, rnd, new, direction="FWD", in_order="NATURAL", out_order="NATURAL"):
    got, info = run_gpu(x, log2n, l1, dw, tw, fmt, rnd, new, direction, in_order, out_order)
    want = C.execute_2d(x, C.make_params(log2n, dw, tw, fmt, rnd, new), l1, DIR[direction], ORD[in_order], ORD[out_order], form=1)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError("2-D GPU != oracle (%r): %d mismatches, first at %r: got %r want %r"
                             % ((log2n, l1, dw, tw, fmt, rnd, new, direction, in_order, out_order), len(bad), bad[0],
                                got[tuple(bad[0])], want[tuple(bad[0])]))
    return info


CASES = [(6, 3, 16, 16, 0, 0, True), (7, 3, 16, 16, 0, 1, True), (7, 4, 16, 16, 1, 0, True), (8, 4, 24, 24, 1, 0, True),
         (8, 5, 16, 16, 0, 0, False), (9, 3, 30, 16, 1, 0, True), (9, 6, 12, 10, 0, 0, True), (10, 5, 16, 18, 0, 0, True),
         (11, 4, 32, 24, 1, 0, True), (12, 6, 44, 16, 0, 0, True), (13, 6, 16, 16, 0, 0, True), (14, 9, 16, 16, 1, 0, True),
         (15, 5, 24, 16, 0, 1, True), (16, 8, 16, 16, 0, 0, True)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("direction", list(DIR))
def test_2d_every_direction_and_width(case, direction):
    log2n, l1, dw, tw, fmt, rnd, new = case
    if C.lib().orc_validate_2d(C.make_params(log2n, dw, tw, fmt, rnd, new), l1, DIR[direction]):
        pytest.skip("not elaboratable")
    n = 1 << log2n
    x = np.concatenate([uniform_frames(5, n, dw, 31 + log2n), edge_frames(n, dw)])
    check(x, *case, direction=direction)


@pytest.mark.parametrize("in_order", list(ORD))
@pytest.mark.parametrize("out_order", list(ORD))
@pytest.mark.parametrize("direction", list(DIR))
def test_2d_io_orders(in_order, out_order, direction):
    x = uniform_frames(3, 1 << 13, 16, 9)
    check(x, 13, 6, 16, 16, 0, 0, True, direction=direction, in_order=in_order, out_order=out_order)


@pytest.mark.parametrize("log2n,l1,frames", [(20, 10, 3), (20, 8, 2), (21, 10, 2), (21, 11, 1), (22, 11, 1), (22, 10, 2)])
@pytest.mark.parametrize("direction", ["FWD", "INV"])
def test_2d_lengths_beyond_the_native_cores(log2n, l1, frames, direction):
    """N = 2^20 .. 2^22 (the reference's cores stop at 2^19): 16-bit scaled, bit-exact to the oracle; the forward result is
    within log2N + 2 LSB of fft(x)/N."""
    n = 1 << log2n
    x = uniform_frames(frames, n, 15, 1234 + log2n)
    check(x, log2n, l1, 16, 16, 0, 0, True, direction=direction)
    if direction == "FWD":
        got, info = run_gpu(x[:1], log2n, l1, 16, 16, 0, 0, True)
        # N1 = 1024: the column cores + multiplier run on tiles (k_big2x_c): two launches at N = 2^20, else column pass + row sub-plan + one
        # layout change; other splits: the five-launch composite
        want = 2 if (log2n, l1) in ((20, 10), (21, 10), (22, 11)) else 3 if l1 == 10 else None  # (2^21, 2^22 = 2048 x 2048: the row cores write X themselves since round 5)
        assert (info["n_passes"] == want if want else info["n_passes"] >= 4) and info["kernel_name"].startswith("2d["), info
        assert np.abs(to_complex(got) - np.fft.fft(to_complex(x[:1]), axis=1) / n).max() <= log2n + 2


@pytest.mark.parametrize("frames", [1, 5, 70])
def test_2d_n2pow20_two_launches(frames, monkeypatch):
    """N = 2^20 = 1024 x 1024, 16-bit scaled-truncate forward: k_big2x_c (column cores on XCD-paired half-line tiles + the
    inter-core multiplier) + k_big2x_b (row cores + store) against the oracle and against the five-launch composite plan
    (INTFFT_2D_NO_FUSED_CORES); full-scale frames (exact extraction in their tiles), HALVES order in, 13-bit twiddles, a batch
    beyond one scratch chunk (two streams against one)."""
    n = 1 << 20
    x = uniform_frames(frames, n, 15, 777 + frames)
    x[0] = uniform_frames(1, n, 16, 5)[0]
    if frames <= 5:
        info = check(x, 20, 10, 16, 16, 0, 0, True)
        assert info["kernel_name"] == "2d[k_big2x_c|k_big2x_b]" and info["n_passes"] == 2, info
        check(x[:2], 20, 10, 16, 16, 0, 0, True, in_order="HALVES")
        check(x[:1], 20, 10, 16, 13, 0, 0, False)
        with monkeypatch.context() as m:
            m.setenv("INTFFT_2D_NO_FUSED_CORES", "1")
            got5, info5 = run_gpu(x, 20, 10, 16, 16, 0, 0, True)
            assert info5["n_passes"] >= 4, info5
        got2, _ = run_gpu(x, 20, 10, 16, 16, 0, 0, True)
        assert np.array_equal(got2, got5)
    else:  # chunk loop (64 frames per 256 MiB layout buffer): frames 0, 63, 64, 69 against the oracle
        got, info = run_gpu(x, 20, 10, 16, 16, 0, 0, True)
        assert info["n_passes"] == 2
        sel = [0, 31, 32, 63, 64, 69]
        want = C.execute_2d(x[sel], C.make_params(20, 16, 16, 0, 0, True), 10, C.FWD, C.NATURAL, C.NATURAL, form=1)
        assert np.array_equal(got[sel], want)
        with monkeypatch.context() as m:  # the chunks alternate between two streams (32-frame halves); one stream: 64-frame chunks
            m.setenv("INTFFT_ONE_STREAM", "1")
            got1, _ = run_gpu(x, 20, 10, 16, 16, 0, 0, True)
        assert np.array_equal(got, got1)


def test_2d_n2pow20_pair_four_launches(monkeypatch):
    """N = 2^20 = 1024 x 1024 FFT -> IFFT pair, 16-bit scaled-truncate: the forward two launches into the second layout buffer (X in
    natural order), the inverse two launches from there -- against the oracle's pair and the eight-launch composite; HALVES orders."""
    n = 1 << 20
    x = uniform_frames(3, n, 15, 2777)
    x[0] = uniform_frames(1, n, 16, 16)[0]
    info = check(x, 20, 10, 16, 16, 0, 0, True, "PAIR")
    assert info["kernel_name"] == "2d[k_big2x_c|k_big2x_b|k_big2x_qb|k_big2x_ci]" and info["n_passes"] == 4, info
    check(x[:1], 20, 10, 16, 16, 0, 0, True, "PAIR", "HALVES", "HALVES")
    with monkeypatch.context() as m:
        m.setenv("INTFFT_2D_NO_FUSED_CORES", "1")
        got8, info8 = run_gpu(x, 20, 10, 16, 16, 0, 0, True, "PAIR")
        assert info8["n_passes"] >= 7, info8
    got4, _ = run_gpu(x, 20, 10, 16, 16, 0, 0, True, "PAIR")
    assert np.array_equal(got4, got8)


def test_2d_n2pow21_pair_four_launches(monkeypatch):
    """N = 2^21 = 1024 x 2048 FFT -> IFFT pair (round 5): k_big2x_c<11> + k_rows2k_tr into the second layout buffer (X in natural order), k_rows2k_qtr +
    k_big2x_ci<., 11> from there -- against the oracle's pair and the composite (INTFFT_2D_NO_ROWS2K); HALVES orders; a batch beyond one scratch chunk."""
    n = 1 << 21
    x = uniform_frames(3, n, 15, 3777)
    x[0] = uniform_frames(1, n, 16, 18)[0]
    info = check(x, 21, 10, 16, 16, 0, 0, True, "PAIR")
    assert info["kernel_name"] == "2d[k_big2x_c|k_rows2k_tr|k_rows2k_qtr|k_big2x_ci]" and info["n_passes"] == 4, info
    check(x[:1], 21, 10, 16, 16, 0, 0, True, "PAIR", "HALVES", "HALVES")
    with monkeypatch.context() as m:
        m.setenv("INTFFT_2D_NO_ROWS2K", "1")
        got8, info8 = run_gpu(x, 21, 10, 16, 16, 0, 0, True, "PAIR")
        assert info8["n_passes"] >= 6, info8
    got4, _ = run_gpu(x, 21, 10, 16, 16, 0, 0, True, "PAIR")
    assert np.array_equal(got4, got8)
    xb = np.concatenate([x] * 7)  # 21 frames: more than one chunk of the layout buffers
    gb, _ = run_gpu(xb, 21, 10, 16, 16, 0, 0, True, "PAIR")
    assert all(np.array_equal(gb[3 * i:3 * i + 3], got4) for i in range(7))


@pytest.mark.parametrize("frames", [3, 70])
def test_2d_n2pow20_inverse_two_launches(frames, monkeypatch):
    """N = 2^20 = 1024 x 1024, 16-bit scaled-truncate INVERSE: k_big2x_qb (row cores: pass QB of the 1-D two-pass inverse as it is) +
    k_big2x_ci (conj multiplier + column cores on pass QA's tiles) against the oracle and the five-launch composite plan; full-scale
    frames, HALVES order out, 13-bit twiddles / XSER OLD, a batch beyond one scratch chunk (two streams against one)."""
    n = 1 << 20
    x = uniform_frames(frames, n, 15, 1777 + frames)
    x[0] = uniform_frames(1, n, 16, 15)[0]
    if frames <= 5:
        info = check(x, 20, 10, 16, 16, 0, 0, True, "INV")
        assert info["kernel_name"] == "2d[k_big2x_qb|k_big2x_ci]" and info["n_passes"] == 2, info
        check(x[:2], 20, 10, 16, 16, 0, 0, True, "INV", "NATURAL", "HALVES")
        check(x[:1], 20, 10, 16, 13, 0, 0, False, "INV")
        with monkeypatch.context() as m:
            m.setenv("INTFFT_2D_NO_FUSED_CORES", "1")
            got5, info5 = run_gpu(x, 20, 10, 16, 16, 0, 0, True, "INV")
            assert info5["n_passes"] >= 4, info5
        got2, _ = run_gpu(x, 20, 10, 16, 16, 0, 0, True, "INV")
        assert np.array_equal(got2, got5)
    else:
        got, info = run_gpu(x, 20, 10, 16, 16, 0, 0, True, "INV")
        assert info["n_passes"] == 2
        sel = [0, 31, 32, 63, 64, 69]
        want = C.execute_2d(x[sel], C.make_params(20, 16, 16, 0, 0, True), 10, C.INV, C.NATURAL, C.NATURAL, form=1)
        assert np.array_equal(got[sel], want)
        with monkeypatch.context() as m:
            m.setenv("INTFFT_ONE_STREAM", "1")
            got1, _ = run_gpu(x, 20, 10, 16, 16, 0, 0, True, "INV")
        assert np.array_equal(got, got1)


@pytest.mark.parametrize("frames", [3, 37])
def test_2d_n2pow21_inverse_two_launches(frames, monkeypatch):
    """N = 2^21 = 1024 x 2048, 16-bit scaled-truncate INVERSE (round 5): k_rows2k_qtr (the 2048-point row cores, sixteen per workgroup, reading 64-byte
    pieces of X[k1 + 1024 k2]) + k_big2x_ci<., 11> (conj multiplier + column cores) against the oracle, the five-launch composite plan
    (INTFFT_2D_NO_FUSED_CORES) and the three-launch form (INTFFT_2D_NO_ROWS2K); a full-scale frame, HALVES order out, 13-bit twiddles / XSER OLD, a batch beyond one scratch chunk."""
    n = 1 << 21
    x = uniform_frames(frames, n, 15, 2777 + frames)
    x[0] = uniform_frames(1, n, 16, 17)[0]
    if frames <= 5:
        info = check(x, 21, 10, 16, 16, 0, 0, True, "INV")
        assert info["kernel_name"] == "2d[k_rows2k_qtr|k_big2x_ci]" and info["n_passes"] == 2, info
        check(x[:2], 21, 10, 16, 16, 0, 0, True, "INV", "NATURAL", "HALVES")
        check(x[:1], 21, 10, 16, 13, 0, 0, False, "INV")
        with monkeypatch.context() as m:
            m.setenv("INTFFT_2D_NO_FUSED_CORES", "1")
            got5, info5 = run_gpu(x, 21, 10, 16, 16, 0, 0, True, "INV")
            assert info5["n_passes"] >= 4, info5
        with monkeypatch.context() as m:
            m.setenv("INTFFT_2D_NO_ROWS2K", "1")  # the three-launch form (test_2d_1024_by_n2_inverse_three_launches)
            got3, info3 = run_gpu(x, 21, 10, 16, 16, 0, 0, True, "INV")
            assert info3["n_passes"] == 3, info3
        got2, _ = run_gpu(x, 21, 10, 16, 16, 0, 0, True, "INV")
        assert np.array_equal(got2, got5) and np.array_equal(got2, got3)
    else:
        got, info = run_gpu(x, 21, 10, 16, 16, 0, 0, True, "INV")
        assert info["n_passes"] == 2, info
        sel = [0, 15, 16, 31, 32, 36]
        want = C.execute_2d(x[sel], C.make_params(21, 16, 16, 0, 0, True), 10, C.INV, C.NATURAL, C.NATURAL, form=1)
        assert np.array_equal(got[sel], want)


@pytest.mark.parametrize("case", [(13, 6, 16, 16, 0, 0, True), (14, 9, 16, 16, 1, 0, True), (12, 6, 44, 16, 0, 0, True), (16, 8, 16, 16, 0, 1, True),
                                  (15, 3, 24, 24, 1, 0, False), (17, 4, 16, 16, 0, 0, True), (17, 13, 16, 16, 0, 0, True)])
@pytest.mark.parametrize("direction", list(DIR))
def test_2d_composite_equals_flat_form(case, direction, monkeypatch):
    """The shipped 2-D plans are composites (layout changes + 1-D sub-plans on the dedicated kernels + the multiplier between
    the cores); INTFFT_2D_GENERIC=1 evaluates the flat form on the generic pass kernels instead.  Same bits, both = oracle."""
    log2n, l1, dw, tw, fmt, rnd, new = case
    if C.lib().orc_validate_2d(C.make_params(log2n, dw, tw, fmt, rnd, new), l1, DIR[direction]):
        pytest.skip("not elaboratable")
    x = uniform_frames(3, 1 << log2n, dw, 500 + log2n + l1)
    a, ia = run_gpu(x, log2n, l1, dw, tw, fmt, rnd, new, direction, "HALVES", "BITREV_LANES")
    monkeypatch.setenv("INTFFT_2D_GENERIC", "1")
    b, ib = run_gpu(x, log2n, l1, dw, tw, fmt, rnd, new, direction, "HALVES", "BITREV_LANES")
    monkeypatch.delenv("INTFFT_2D_GENERIC")
    assert ia["kernel_name"].startswith("2d[") and ib["kernel_name"].startswith("k_pass"), (ia, ib)
    monkeypatch.setenv("INTFFT_2D_NO_FUSE", "1")  # the multiplier between the cores as its own launch instead of fused
    c, _ = run_gpu(x, log2n, l1, dw, tw, fmt, rnd, new, direction, "HALVES", "BITREV_LANES")
    monkeypatch.delenv("INTFFT_2D_NO_FUSE")
    want = C.execute_2d(x, C.make_params(log2n, dw, tw, fmt, rnd, new), l1, DIR[direction], C.HALVES, C.BITREV_LANES, form=1)
    assert np.array_equal(a, want) and np.array_equal(b, want) and np.array_equal(c, want)


def test_2d_in_place_and_ragged_chunks(monkeypatch):
    """d_in == d_out with equal containers, and a batch that is not a multiple of the plan's chunk of frames."""
    import torch

    from intfftk_amd import int_fft_2d

    core = int_fft_2d(NFFT=20, NFFT1=9, direction="PAIR")
    x = uniform_frames(3, 1 << 20, 15, 4)
    buf = torch.from_numpy(x.astype(np.int16)).cuda()
    core(buf, out=buf)
    torch.cuda.synchronize()
    want = C.execute_2d(x, C.make_params(20, 16, 16, 0, 0, True), 9, C.PAIR)
    assert np.array_equal(buf.cpu().numpy().astype(np.int64), want)
    monkeypatch.setenv("INTFFT_2D_CHUNK_FRAMES", "4")  # 11 frames = two full chunks of scratch + a ragged one
    small = int_fft_2d(NFFT=13, NFFT1=6)
    monkeypatch.delenv("INTFFT_2D_CHUNK_FRAMES")
    y = uniform_frames(11, 1 << 13, 15, 5)
    got = small(torch.from_numpy(y.astype(np.int16)).cuda()).cpu().numpy().astype(np.int64)
    assert np.array_equal(got, C.execute_2d(y, C.make_params(13, 16, 16, 0, 0, True), 6, C.FWD))


def test_2d_unscaled_and_pair_at_2pow20():
    x = uniform_frames(1, 1 << 20, 15, 77)
    check(x, 20, 10, 16, 16, 1, 0, True, direction="FWD")   # 36-bit results, int64 containers
    check(x, 20, 10, 16, 16, 0, 0, True, direction="PAIR")


def test_2d_twiddle_introspection():
    from intfftk_amd import int_fft_2d

    core = int_fft_2d(NFFT=12, NFFT1=5, TWDL_WIDTH=16)
    tab = core.twiddles(-1)
    assert tab.shape == (4096, 2)
    want = np.array([C.twiddle_2d(12, 16, m) for m in range(4096)])
    assert np.array_equal(tab, want)
    re, im = C.twiddles(6, 16)
    assert np.array_equal(core.twiddles(6), np.stack([re, im], axis=-1))
    with pytest.raises(Exception):
        core.twiddles(7)  # stages exist up to max(log2 N1, log2 N2) - 1 = 6
    # the library's host-side evaluation == the oracle's on a whole 2^16 circle, 16- and 24-bit twiddles
    for t in (16, 24):
        big = int_fft_2d(NFFT=16, NFFT1=8, DATA_WIDTH=16, TWDL_WIDTH=t, FORMAT=1)
        assert np.array_equal(big.twiddles(-1), np.array([C.twiddle_2d(16, t, m) for m in range(1 << 16)]))


def test_2d_random_configurations():
    """Seeded fuzz over (N1, N2, widths, mode, XSER, direction, orders): every elaboratable draw is bit-exact to the oracle."""
    rng = np.random.default_rng(20260928)
    orders = list(ORD)
    done = 0
    for _ in range(120):
        log2n = int(rng.integers(6, 15))
        l1 = int(rng.integers(3, log2n - 2))
        dw = int(rng.choice([8, 12, 16, 18, 24, 30]))
        tw = int(rng.choice([10, 16, 18, 24]))
        fmt = int(rng.integers(0, 2))
        rnd = 0 if fmt else int(rng.integers(0, 2))
        new = bool(rng.integers(0, 2))
        direction = str(rng.choice(list(DIR)))
        if C.lib().orc_validate_2d(C.make_params(log2n, dw, tw, fmt, rnd, new), l1, DIR[direction]):
            continue
        x = uniform_frames(int(rng.integers(1, 4)), 1 << log2n, dw, int(rng.integers(1, 1 << 30)))
        check(x, log2n, l1, dw, tw, fmt, rnd, new, direction, str(rng.choice(orders)), str(rng.choice(orders)))
        done += 1
    assert done >= 60


@pytest.mark.parametrize("log2n,frames,out_order", [(21, 3, "NATURAL"), (22, 2, "BITREV"), (23, 1, "NATURAL"), (22, 18, "NATURAL")])
def test_2d_1024_by_n2_three_launches(log2n, frames, out_order, monkeypatch):
    """N = 2^21 .. 2^23 as 1024 x N2, 16-bit scaled-truncate forward: k_big2x_c (column cores + multiplier on tiles), the N2-point row
    sub-plan, one layout change -- against the oracle and the five-launch composite (INTFFT_2D_NO_FUSED_CORES); every output order
    goes through the same last launch; HALVES order in; a batch beyond one scratch chunk (16 frames at N = 2^22)."""
    n = 1 << log2n
    x = uniform_frames(frames, n, 15, 888 + log2n)
    x[0] = uniform_frames(1, n, 16, 6)[0]
    got, info = run_gpu(x, log2n, 10, 16, 16, 0, 0, True, out_order=out_order)
    # (N2 = 8192 / 16384: a one-pass row core since round 4; N2 = 2048 in natural order out: the row cores write X themselves, two launches, round 5)
    two = log2n == 21 and out_order == "NATURAL"
    assert info["kernel_name"] == ("2d[k_big2x_c|k_rows2k_tr]" if two else info["kernel_name"]) and info["kernel_name"].startswith("2d[k_big2x_c|"), info
    assert info["n_passes"] == (2 if two else 3), info
    if two:
        with monkeypatch.context() as m:
            m.setenv("INTFFT_2D_NO_ROWS2K", "1")
            got3, info3 = run_gpu(x, log2n, 10, 16, 16, 0, 0, True, out_order=out_order)
            assert info3["n_passes"] == 3 and np.array_equal(got, got3), info3
    with monkeypatch.context() as m:
        m.setenv("INTFFT_2D_NO_FUSED_CORES", "1")
        got5, info5 = run_gpu(x, log2n, 10, 16, 16, 0, 0, True, out_order=out_order)
        assert info5["n_passes"] >= 5, info5
    assert np.array_equal(got, got5)
    sel = [0, frames - 1] if frames > 1 else [0]
    want = C.execute_2d(x[sel], C.make_params(log2n, 16, 16, 0, 0, True), 10, C.FWD, C.NATURAL, ORD[out_order], form=1)
    assert np.array_equal(got[sel], want)
    if frames <= 3 and log2n <= 22:
        check(x[:1], log2n, 10, 16, 16, 0, 0, True, in_order="HALVES", out_order=out_order)


@pytest.mark.parametrize("log2n,frames,out_order", [(22, 2, "NATURAL"), (22, 18, "HALVES"), (23, 1, "NATURAL"), (21, 3, "NATURAL")])
def test_2d_1024_by_n2_inverse_three_launches(log2n, frames, out_order, monkeypatch):
    """N = 2^22 .. 2^24 as 1024 x N2, 16-bit scaled-truncate INVERSE (round 5): one layout change (X[k1 + 1024 k2] -> rows [r][k2], k1 = brev10(r)), the N2-point
    inverse row sub-plan, k_big2x_ci<., L2, ROWS> (conj multiplier + column cores reading plain rows) -- against the oracle and the five-launch composite
    (INTFFT_2D_NO_FUSED_CORES); N = 2^21 takes this form when its two-launch kernels are switched off; a batch beyond one scratch chunk."""
    n = 1 << log2n
    x = uniform_frames(frames, n, 15, 988 + log2n)
    x[0] = uniform_frames(1, n, 16, 7)[0]
    if log2n == 21:
        monkeypatch.setenv("INTFFT_2D_NO_ROWS2K", "1")
    got, info = run_gpu(x, log2n, 10, 16, 16, 0, 0, True, "INV", out_order=out_order)
    assert info["kernel_name"].startswith("2d[") and info["kernel_name"].endswith("|k_big2x_ci]") and "rows2k" not in info["kernel_name"], info
    assert info["n_passes"] == 3, info  # (the 2048- / 4096- / 8192-point inverse row cores are one launch each)
    with monkeypatch.context() as m:
        m.setenv("INTFFT_2D_NO_FUSED_CORES", "1")
        got5, info5 = run_gpu(x, log2n, 10, 16, 16, 0, 0, True, "INV", out_order=out_order)
        assert info5["n_passes"] >= 5, info5
    assert np.array_equal(got, got5)
    sel = [0, frames - 1] if frames > 1 else [0]
    want = C.execute_2d(x[sel], C.make_params(log2n, 16, 16, 0, 0, True), 10, C.INV, C.NATURAL, ORD[out_order], form=1)
    assert np.array_equal(got[sel], want)


@pytest.mark.parametrize("frames", [2, 19])
def test_2d_n2pow22_two_launches(frames, monkeypatch):
    """N = 2^22 = 2048 x 2048, 16-bit scaled-truncate forward (round 5): k_cols2k_c (the 2048-point column cores + the multiplier on tiles of 2048 rows x 16
    columns, one workgroup per CU) + k_rows2k_tr<., 11> (the row cores + the store of X[k1 + 2048 k2]) against the oracle and the five-launch composite
    (INTFFT_2D_NO_ROWS2K); a full-scale frame, 13-bit twiddles / XSER OLD, a batch beyond one scratch chunk."""
    n = 1 << 22
    x = uniform_frames(frames, n, 15, 4777 + frames)
    x[0] = uniform_frames(1, n, 16, 19)[0]
    got, info = run_gpu(x, 22, 11, 16, 16, 0, 0, True)
    assert info["kernel_name"] == "2d[k_cols2k_c|k_rows2k_tr]" and info["n_passes"] == 2, info
    sel = [0, frames - 1]
    want = C.execute_2d(x[sel], C.make_params(22, 16, 16, 0, 0, True), 11, C.FWD, C.NATURAL, C.NATURAL, form=1)
    assert np.array_equal(got[sel], want)
    with monkeypatch.context() as m:
        m.setenv("INTFFT_2D_NO_ROWS2K", "1")
        got5, info5 = run_gpu(x, 22, 11, 16, 16, 0, 0, True)
        assert info5["n_passes"] >= 4, info5
    assert np.array_equal(got, got5)
    if frames <= 2:
        check(x[:1], 22, 11, 16, 13, 0, 0, False)


@pytest.mark.parametrize("direction,frames", [("INV", 2), ("INV", 19), ("PAIR", 2), ("PAIR", 11)])
def test_2d_n2pow22_inverse_and_pair(direction, frames, monkeypatch):
    """N = 2^22 = 2048 x 2048, 16-bit scaled-truncate (round 5): the inverse in two launches -- k_rows2k_qtr<., 11, true> (the 2048-point row cores, plain rows out) +
    k_cols2k_ci (conj multiplier + the 2048-point column cores on tiles of 2048 rows x 16 columns) -- and the pair in four (the forward two into the second layout
    buffer); against the oracle and the composite (INTFFT_2D_NO_ROWS2K); a full-scale frame, 13-bit twiddles / XSER OLD, batches beyond one scratch chunk."""
    n = 1 << 22
    x = uniform_frames(frames, n, 15, 5777 + frames)
    x[0] = uniform_frames(1, n, 16, 21)[0]
    got, info = run_gpu(x, 22, 11, 16, 16, 0, 0, True, direction)
    name = "2d[k_rows2k_qtr|k_cols2k_ci]" if direction == "INV" else "2d[k_cols2k_c|k_rows2k_tr|k_rows2k_qtr|k_cols2k_ci]"
    assert info["kernel_name"] == name and info["n_passes"] == (2 if direction == "INV" else 4), info
    sel = [0, frames - 1]
    want = C.execute_2d(x[sel], C.make_params(22, 16, 16, 0, 0, True), 11, DIRS[direction], C.NATURAL, C.NATURAL, form=1)
    assert np.array_equal(got[sel], want)
    if frames <= 2:
        with monkeypatch.context() as m:
            m.setenv("INTFFT_2D_NO_ROWS2K", "1")
            got5, info5 = run_gpu(x, 22, 11, 16, 16, 0, 0, True, direction)
            assert info5["n_passes"] >= 4 and info5["kernel_name"] != name, info5
        assert np.array_equal(got, got5)
        check(x[:1], 22, 11, 16, 13, 0, 0, False, direction)

