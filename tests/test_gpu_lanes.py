"""BITREV_LANES -- the serial stream between outbuf_half_path and int_bitrev_order
(src/vhdl/buffers/outbuf_half_path.vhd:160-172, int_bitrev_order.vhd:82-104) -- off the generic kernels (round 6):

* the packed 16-bit kernels of N = 128 .. 16384 carry it as a store map (forward: `k_fft1024_i16`, `k_fft4096_i16`,
  `k_fft16k_i16`) / load map (inverse: `k_fft1024x_i16`, `k_fft4096_i16`, `k_fft16k_i16`) of their BITREV instantiations: one launch;
* every other plan whose BITREV twin has dedicated kernels runs that twin and one bit permutation (`lanes[...]`): the order
  is a rotation of the BITREV memory index by one bit.

Bit-exact against the C oracle through the C-ABI, like every parity test.
"""
import numpy as np
import pytest

from tests.helpers import edge_frames, uniform_frames
from tests.test_gpu_parity import check

pytestmark = pytest.mark.gpu


def frames(n, dw, batch, seed):
    x = uniform_frames(batch, n, dw, seed)
    e = edge_frames(n, dw)
    x[: min(batch, len(e))] = e[: min(batch, len(e))]
    return x


def fwd_kernel(log2n):
    return "k_fft1024_i16" if log2n <= 10 else "k_fft4096_i16" if log2n <= 12 else "k_fft16k_i16"


def inv_kernel(log2n):
    return "k_fft1024x_i16" if log2n <= 10 else "k_fft4096_i16" if log2n <= 12 else "k_fft16k_i16"


@pytest.mark.parametrize("log2n", [7, 8, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("rnd", [0, 1])
@pytest.mark.parametrize("in_order", ["NATURAL", "HALVES"])
def test_packed_forward_lanes_store_map(log2n, rnd, in_order):
    n = 1 << log2n
    for dw, tw, batch in ((16, 16, 37), (16, 13, 9), (12, 16, 21)):  # fast + exact extraction, narrow data, partial last chunk
        info = check(frames(n, dw, batch, 100 + log2n), log2n, dw, tw, 0, rnd, True, direction="FWD", in_order=in_order,
                     out_order="BITREV_LANES")
        assert info["kernel_name"] == fwd_kernel(log2n) and info["n_passes"] == 1, info
    # quiet frames (the fast extraction path) and a full multiple of the chunk
    x = uniform_frames(64, n, 14, 7)
    info = check(x, log2n, 16, 16, 0, rnd, True, direction="FWD", in_order=in_order, out_order="BITREV_LANES")
    assert info["kernel_name"] == fwd_kernel(log2n), info


@pytest.mark.parametrize("log2n", [7, 8, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("rnd", [0, 1])
@pytest.mark.parametrize("out_order", ["NATURAL", "HALVES"])
def test_packed_inverse_lanes_load_map(log2n, rnd, out_order):
    n = 1 << log2n
    for dw, tw, batch in ((16, 16, 37), (16, 13, 9), (12, 16, 21)):
        info = check(frames(n, dw, batch, 200 + log2n), log2n, dw, tw, 0, rnd, True, direction="INV", in_order="BITREV_LANES",
                     out_order=out_order)
        assert info["kernel_name"] == inv_kernel(log2n) and info["n_passes"] == 1, info
    x = uniform_frames(64, n, 14, 8)
    info = check(x, log2n, 16, 16, 0, rnd, True, direction="INV", in_order="BITREV_LANES", out_order=out_order)
    assert info["kernel_name"] == inv_kernel(log2n), info


def test_forward_lanes_then_inverse_lanes_is_the_pair():
    """int_fft_single_path's serial stream fed straight into an inverse core = int_fft_ifft_pair on the same frames."""
    from tests.test_gpu_parity import run_gpu
    x = frames(1024, 16, 33, 5)
    mid, _ = run_gpu(x, 10, 16, 16, 0, 0, True, direction="FWD", out_order="BITREV_LANES")
    back, _ = run_gpu(mid, 10, 16, 16, 0, 0, True, direction="INV", in_order="BITREV_LANES")
    pair, _ = run_gpu(x, 10, 16, 16, 0, 0, True, direction="PAIR")
    assert np.array_equal(back, pair)


# (log2n, dw, tw, fmt, rnd, direction, other-end order, batch): one plan per dedicated family behind the composite
COMPOSITE = [
    (15, 16, 16, 0, 1, "INV", "HALVES", 3),      # two-pass packed, round mode
    (17, 16, 16, 0, 0, "FWD", "HALVES", 2),      # k_big2p_a
    (16, 16, 16, 0, 0, "INV", "NATURAL", 2),     # two-pass packed
    (10, 16, 16, 1, 0, "FWD", "NATURAL", 9),     # unscaled 16-bit: 32-bit results
    (10, 24, 24, 1, 0, "FWD", "HALVES", 6),      # 64-bit words
    (10, 24, 16, 0, 1, "INV", "NATURAL", 6),     # 32-bit words, scaled
    (12, 18, 18, 0, 0, "FWD", "NATURAL", 4),
    (16, 24, 24, 1, 0, "FWD", "NATURAL", 2),     # C3's plan
]


def test_short_frames_stay_on_the_generic_kernels():
    """N = 8 .. 64: the register kernels take natural orders only, so the BITREV twin is generic and no composite is built."""
    for log2n, direction, kw in ((5, "FWD", dict(out_order="BITREV_LANES")), (6, "INV", dict(in_order="BITREV_LANES"))):
        info = check(frames(1 << log2n, 16, 40, 3), log2n, 16, 16, 0, 0, True, direction=direction, **kw)
        assert not info["kernel_name"].startswith("lanes["), info


@pytest.mark.parametrize("cfg", COMPOSITE, ids=lambda c: "n%d_w%d_t%d_f%d_r%d_%s_%s" % c[:7])
def test_lanes_composite_equals_oracle(cfg, monkeypatch):
    log2n, dw, tw, fmt, rnd, direction, other, batch = cfg
    x = frames(1 << log2n, dw, batch, 300 + log2n)
    kw = dict(direction=direction)
    if direction == "FWD":
        kw.update(in_order=other, out_order="BITREV_LANES")
    else:
        kw.update(in_order="BITREV_LANES", out_order=other)
    info = check(x, log2n, dw, tw, fmt, rnd, True, **kw)
    assert info["kernel_name"].startswith("lanes["), info
    # the generic kernels give the same bytes (A/B switch), under their own name
    monkeypatch.setenv("INTFFT_DIAG", "1")
    monkeypatch.setenv("INTFFT_NO_LANES_COMPOSITE", "1")
    info_g = check(x, log2n, dw, tw, fmt, rnd, True, **kw)
    assert not info_g["kernel_name"].startswith("lanes["), info_g


@pytest.mark.parametrize("cb", [2, 4, 8])
@pytest.mark.parametrize("log2n", [3, 4, 9, 12, 17])
def test_reorder_one_bit_rotations_equal_the_tiled_mover(log2n, cb, monkeypatch):
    """BITREV <-> BITREV_LANES and NATURAL <-> HALVES run on the streaming kernel k_rotate1; with it switched off the tiled mover
    gives the same bytes (tests/test_gpu_cabi.py::test_reorder_all_order_pairs checks both against the index maps)."""
    import torch

    from intfftk_amd import _capi as capi
    n = 1 << log2n
    batch = 5 if log2n < 17 else 2
    dt = {2: torch.int16, 4: torch.int32, 8: torch.int64}[cb]
    x = torch.randint(-30000, 30000, (batch, n, 2), dtype=dt, device="cuda")
    monkeypatch.setenv("INTFFT_DIAG", "1")
    for a, b in ((capi.ORDER_BITREV, capi.ORDER_BITREV_LANES), (capi.ORDER_BITREV_LANES, capi.ORDER_BITREV),
                 (capi.ORDER_NATURAL, capi.ORDER_HALVES), (capi.ORDER_HALVES, capi.ORDER_NATURAL)):
        y, z = torch.zeros_like(x), torch.zeros_like(x)
        assert capi.lib().intfft_reorder(log2n, cb, a, b, x.data_ptr(), y.data_ptr(), batch, 0, None) == 0
        monkeypatch.setenv("INTFFT_NO_ROTATE1", "1")
        assert capi.lib().intfft_reorder(log2n, cb, a, b, x.data_ptr(), z.data_ptr(), batch, 0, None) == 0
        monkeypatch.delenv("INTFFT_NO_ROTATE1")
        torch.cuda.synchronize()
        assert torch.equal(y, z), (a, b)
        # and the inverse direction undoes it
        w = torch.zeros_like(x)
        assert capi.lib().intfft_reorder(log2n, cb, b, a, y.data_ptr(), w.data_ptr(), batch, 0, None) == 0
        torch.cuda.synchronize()
        assert torch.equal(w, x), (a, b)


def test_lanes_composite_workspace_and_chunks(monkeypatch):
    """A middle buffer smaller than the batch (several chunks) and a caller-supplied workspace."""
    import torch

    from intfftk_amd import IntFFTCore
    from oracle import oracle_c as C
    monkeypatch.setenv("INTFFT_DIAG", "1")
    monkeypatch.setenv("INTFFT_SCRATCH_MB", "1")  # 1 MiB middle buffer: 16 frames of N = 4096 in int32 pairs ... several chunks
    x = frames(4096, 16, 50, 9)
    check(x, 12, 16, 16, 1, 0, True, direction="FWD", out_order="BITREV_LANES")
    monkeypatch.delenv("INTFFT_SCRATCH_MB")
    core = IntFFTCore(12, 16, 16, 1, 0, "NEW", "FWD", "NATURAL", "BITREV_LANES", 1)
    assert core.info["kernel_name"].startswith("lanes["), core.info
    xin = torch.from_numpy(x.astype(np.int16)).cuda()
    need = core.workspace_bytes(50)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    core.release_scratch()
    y = core.exec_ws(xin, ws)
    torch.cuda.synchronize()
    want = C.execute(x, C.make_params(12, 16, 16, 1, 0, True, 1), C.FWD, C.NATURAL, C.BITREV_LANES, form=1)
    assert np.array_equal(y.cpu().numpy().astype(np.int64), want)
    core.close()
