"""Results beyond 64 bits (the trpl18 tail of int_cmult_dsp48.vhd:267-303 with bit growth, e.g. the 32-bit unscaled pair at
NFFT >= 17): k_pass<__int128>, 16-byte containers.  The checker is the pure-Python twin of the oracle (Python ints have no
width limit; the C oracle holds values in int64 and stops at 64 bits), so the sizes here are small."""
import numpy as np
import pytest

from oracle import oracle_py as P
from tests.helpers import uniform_frames

pytestmark = pytest.mark.gpu
NP = {2: np.int16, 4: np.int32, 8: np.int64}
DIR = {"FWD": P.FWD, "INV": P.INV, "PAIR": P.PAIR}
ORD = {"NATURAL": P.NATURAL, "BITREV": P.BITREV, "HALVES": P.HALVES, "BITREV_LANES": P.BITREV_LANES}


def run_gpu(x, log2n, dw, tw, new, direction, in_order="NATURAL", out_order="NATURAL"):
    import torch

    from intfftk_amd import IntFFTCore
    from intfftk_amd.engine import wide_to_int

    core = IntFFTCore(log2n, dw, tw, 1, 0, "NEW" if new else "OLD", direction, in_order, out_order)
    y = core(torch.from_numpy(np.ascontiguousarray(x.astype(NP[core.in_container]))).cuda())
    torch.cuda.synchronize()
    info = dict(core.info)
    core.close()
    y = y.cpu().numpy()
    return (wide_to_int(y) if info["out_container"] == 16 else y.astype(object)), info


def check(x, log2n, dw, tw, new, direction, in_order="NATURAL", out_order="NATURAL"):
    got, info = run_gpu(x, log2n, dw, tw, new, direction, in_order, out_order)
    assert info["out_container"] == 16 and info["compute_word"] == 16 and info["kernel_name"] == "k_pass<__int128>", info
    for f in range(x.shape[0]):
        frame = [(int(a), int(b)) for a, b in x[f]]
        want = P.execute(frame, log2n, dw, tw, 1, 0, new, DIR[direction], ORD[in_order], ORD[out_order])
        for m, (wr, wi) in enumerate(want):
            assert (got[f, m, 0], got[f, m, 1]) == (wr, wi), (f, m, got[f, m], (wr, wi), info)
    return info


def frames(batch, n, dw, seed):
    x = uniform_frames(batch, n, min(dw, 63), seed)
    if dw == 64:  # full-range int64
        rng = np.random.default_rng(seed)
        x = rng.integers(-(1 << 63), (1 << 63) - 1, size=(batch, n, 2), dtype=np.int64, endpoint=True)
    lim = (1 << (dw - 1)) - 1
    x[0, 0] = (lim, -lim - 1)  # the extremes of the width
    x[0, 1] = (-lim - 1, lim)
    return x


# The trpl18 multiplier elaborates for MAW + MBW <= 80 (NEW) / 78 (OLD) (its product slice must lie inside the 79 / 77 bits of
# P, int_cmult_trpl18_dsp48.vhd:151-152), so the widest multiplier of a configuration bounds what can be built: DIF multiplies at
# DTW + 1 down to STAGE 2, DIT at DTW from STAGE 2 up.  Beyond 61 / 59 bits the A port cuts the operand (:161-162).
@pytest.mark.parametrize("log2n,dw,tw,new,direction", [
    (6, 60, 16, True, "FWD"), (10, 56, 16, True, "FWD"), (10, 60, 12, True, "FWD"), (10, 64, 8, True, "FWD"),
    (3, 63, 16, True, "FWD"), (7, 58, 16, True, "INV"), (10, 58, 10, False, "INV"), (9, 56, 14, True, "INV"),
    (8, 50, 14, True, "PAIR"), (5, 56, 12, False, "PAIR")])
def test_results_beyond_64_bits(log2n, dw, tw, new, direction):
    x = frames(3, 1 << log2n, dw, 77 + log2n + dw)
    info = check(x, log2n, dw, tw, new, direction)
    assert info["out_bits"] > 64 and info["n_passes"] == 1


def test_wide_elaboration_bound():
    """One bit past the product slice: the reference does not elaborate, neither do we (both oracles agree)."""
    from intfftk_amd import ERR_UNSUPPORTED, IntFFTCore, IntFFTError

    for kw in (dict(NFFT=10, DATA_WIDTH=57, TWDL_WIDTH=16, FORMAT=1), dict(NFFT=7, DATA_WIDTH=59, TWDL_WIDTH=16, FORMAT=1, direction="INV"),
               dict(NFFT=17, DATA_WIDTH=32, TWDL_WIDTH=16, FORMAT=1, direction="PAIR")):
        with pytest.raises(IntFFTError) as e:
            IntFFTCore(**kw)
        assert e.value.status == ERR_UNSUPPORTED
    assert P.cmult_regime(65, 16, True) is None and P.cmult_regime(64, 16, True) == "trpl18" and P.cmult_regime(63, 16, False) is None


@pytest.mark.parametrize("in_order,out_order", [("HALVES", "BITREV"), ("BITREV", "HALVES"), ("BITREV_LANES", "NATURAL")])
@pytest.mark.parametrize("direction", ["FWD", "INV", "PAIR"])
def test_wide_io_orders(direction, in_order, out_order):
    dw = 52 if direction == "PAIR" else 60
    x = frames(2, 1 << 7, dw, 5)
    check(x, 7, dw, 12, True, direction, in_order, out_order)


@pytest.mark.parametrize("log2n,dw,tw,direction", [(12, 62, 8, "FWD"), (13, 60, 8, "INV"), (12, 50, 6, "PAIR")])
def test_wide_multi_pass(log2n, dw, tw, direction):
    """Beyond one LDS tile (2^11 points of 32 bytes): scratch of 128-bit words between the passes."""
    x = frames(2, 1 << log2n, dw, 31 + log2n)
    info = check(x, log2n, dw, tw, True, direction)
    assert info["n_passes"] >= 2


def test_wide_pair_of_the_verdict():
    """The configuration VERDICT round 1 names: int_fft_ifft_pair, 32-bit unscaled, NFFT = 17 (66-bit results) -- with 15-bit
    twiddles, the widest for which its last DIT multiplier (65 bits) elaborates: one frame, spot-checked against the Python
    twin on a 1-in-4099 sample of the outputs (the twin computes the whole frame)."""
    log2n, dw, tw = 17, 32, 15
    x = uniform_frames(1, 1 << log2n, dw, 4242)
    got, info = run_gpu(x, log2n, dw, tw, True, "PAIR")
    assert info["out_bits"] == 66 and info["out_container"] == 16
    want = P.execute([(int(a), int(b)) for a, b in x[0]], log2n, dw, tw, 1, 0, True, P.PAIR, P.NATURAL, P.NATURAL)
    for m in range(0, 1 << log2n, 4099):
        assert (got[0, m, 0], got[0, m, 1]) == want[m], m
    # the pair returns the input scaled by N, up to the error of the 15-bit twiddles (about N log2 N 2^-15 of the full scale)
    m = 12345
    assert abs(int(got[0, m, 0]) - (int(x[0, m, 0]) << log2n)) < (1 << (dw + log2n - 8))


def test_wide_exec_host_and_plan_info():
    from intfftk_amd import IntFFTCore
    from intfftk_amd.engine import wide_to_int

    core = IntFFTCore(8, 60, 12, 1, 0)
    assert (core.in_container, core.out_container, core.out_bits) == (8, 16, 68)
    x = frames(5, 256, 60, 9)
    y = wide_to_int(core.exec_host(x, chunk_frames=2))
    core.close()
    for f in (0, 4):
        want = P.execute([(int(a), int(b)) for a, b in x[f]], 8, 60, 12, 1, 0, True, P.FWD, P.NATURAL, P.NATURAL)
        assert all((y[f, m, 0], y[f, m, 1]) == want[m] for m in range(256))
