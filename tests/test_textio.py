"""Text-file compatibility helpers: formats of the reference testbenches."""
import numpy as np
import pytest

from intfftk_amd import textio
from tests.helpers import uniform_frames


def test_di_single_round_trip(tmp_path):
    x = uniform_frames(3, 128, 16, 1)
    p = str(tmp_path / "di_single.dat")
    textio.write_di_single(p, x)
    assert open(p).readline().split() == [str(x[0, 0, 0]), str(x[0, 0, 1])]
    assert np.array_equal(textio.read_di_single(p, 128), x)
    with pytest.raises(ValueError):
        textio.read_di_single(p, 100)


def test_di_double_is_the_interleave2_stream(tmp_path):
    x = uniform_frames(2, 16, 12, 2)
    p = str(tmp_path / "di_double.dat")
    textio.write_di_double(p, x)
    first = [int(v) for v in open(p).readline().split()]
    assert first == [x[0, 0, 0], x[0, 1, 0], x[0, 0, 1], x[0, 1, 1]]  # D0_RE D1_RE D0_IM D1_IM of beat 0
    assert np.array_equal(textio.read_di_double(p, 16), x)


def test_dout_pair_top17_and_reference_wiring(tmp_path):
    width = 16 + 2 * 7  # fft_double_test: NFFT = 7, unscaled pair output width
    y = uniform_frames(1, 128, width, 3)
    rows = textio.dout_pair_lines(y, width)
    assert rows.shape == (64, 4)
    assert rows[0, 0] == y[0, 0, 0] >> (width - 17) and rows[0, 3] == y[0, 1, 1] >> (width - 17)
    assert np.abs(rows).max() < 1 << 16
    bug = textio.dout_pair_lines(y, width, reference_wiring=True)
    assert np.array_equal(bug[:, 2], bug[:, 0]) and np.array_equal(bug[:, 1], bug[:, 3])
    p = str(tmp_path / "dout_pair.dat")
    textio.write_dout_pair(p, y, width)
    assert np.array_equal(textio.read_dout_pair(p), rows)


@pytest.mark.gpu
def test_pair_through_text_files_matches_oracle(tmp_path):
    """fft_double_test's flow end to end: di_double.dat -> int_fft_ifft_pair -> dout_pair.dat."""
    import torch

    from intfftk_amd import int_fft_ifft_pair
    from oracle import oracle_c as C

    x = uniform_frames(4, 128, 16, 9)
    pin, pout = str(tmp_path / "di_double.dat"), str(tmp_path / "dout_pair.dat")
    textio.write_di_double(pin, x)
    core = int_fft_ifft_pair(NFFT=7, DATA_WIDTH=16, TWDL_WIDTH=16, FORMAT=1)
    y = core(torch.from_numpy(textio.read_di_double(pin, 128).astype(np.int16)).cuda()).cpu().numpy()
    textio.write_dout_pair(pout, y, core.out_bits)
    want = C.execute(x, C.make_params(7, 16, 16, 1, 0, True), C.PAIR)
    assert np.array_equal(textio.read_dout_pair(pout), textio.dout_pair_lines(want, 16 + 14))
    core.close()


@pytest.mark.gpu
def test_cli_single_flow(tmp_path):
    """python -m intfftk_amd.cli single: di_single.dat -> natural-order spectrum, three tb modes."""
    from intfftk_amd import cli
    from oracle import oracle_c as C

    x = uniform_frames(2, 128, 16, 4)
    pin = str(tmp_path / "di_single.dat")
    textio.write_di_single(pin, x)
    for mode, (fmt, rnd) in {"UNSCALED": (1, 0), "TRUNCATE": (0, 0), "ROUNDING": (0, 1)}.items():
        pout = str(tmp_path / ("out_%s.dat" % mode))
        assert cli.main(["single", pin, pout, "--nfft", "7", "--mode", mode]) == 0
        want = C.execute(x, C.make_params(7, 16, 16, fmt, rnd, True), C.FWD)
        assert np.array_equal(textio.read_di_single(pout, 128), want)
