"""USE_FLY = 0 off the generic kernels (round 6): the bypass mux of int_fftNk.vhd:260-277 / int_ifftNk.vhd:259-276 takes the butterflies
out of the path.  In place terms nothing moves: position p holds input index p and is read out as output index bitrev(p), so a single
core is the bit reversal of the logical index (a pair: the identity) of the input wrapped to DATA_WIDTH bits -- sign-extended (scaled)
or zero-extended (unscaled) -- in the output container.  A plan = that conversion + one bit permutation of the memory index, or no
permutation at all in the cores' own beat orders.  `bypass[k_convert]` / `bypass[k_convert|k_reorder]`.
Bit-exact against the C oracle (whose stream form really walks the delay lines) through the C-ABI."""
import numpy as np
import pytest

from tests.helpers import edge_frames, uniform_frames
from tests.test_gpu_parity import check

pytestmark = pytest.mark.gpu

ORDERS = ["NATURAL", "BITREV", "HALVES", "BITREV_LANES"]


def mem_bit(order, L, j):
    """Memory-index bit that carries logical-index bit j (include/intfft.h: the four INTFFT_ORDER_* maps)."""
    return {"NATURAL": j, "BITREV": L - 1 - j, "HALVES": 0 if j == L - 1 else j + 1, "BITREV_LANES": L - 1 if j == L - 1 else L - 2 - j}[order]


def frames(n, bits, batch, seed):
    x = uniform_frames(batch, n, bits, seed)
    e = edge_frames(n, bits)
    x[: min(batch, len(e))] = e[: min(batch, len(e))]
    return x


@pytest.mark.parametrize("direction", ["FWD", "INV", "PAIR"])
@pytest.mark.parametrize("fmt", [0, 1])
def test_bypass_every_order_pair(direction, fmt):
    for log2n, dw, batch in ((9, 16, 11), (5, 12, 40), (12, 24, 3)):
        # data wider than DATA_WIDTH in the container: the wrap of the first stage is visible
        x = frames(1 << log2n, min(dw + 3, {16: 16, 12: 16, 24: 32}[dw]), batch, 40 + log2n)
        for in_o in ORDERS:
            for out_o in ORDERS:
                info = check(x, log2n, dw, 16, fmt, 0, True, direction=direction, in_order=in_o, out_order=out_o, use_fly=0)
                assert info["kernel_name"] in ("bypass[k_convert]", "bypass[k_convert|k_reorder]"), (info, in_o, out_o)
                # no data movement exactly where the memory-index permutation is the identity: e.g. NATURAL -> BITREV for a single
                # core (the in-place picture), equal orders for a pair
                ident = all(mem_bit(out_o, log2n, j) == mem_bit(in_o, log2n, j if direction == "PAIR" else log2n - 1 - j)
                            for j in range(log2n))
                assert (info["kernel_name"] == "bypass[k_convert]") == ident, (info, direction, in_o, out_o)
                if direction != "PAIR" and (in_o, out_o) in (("NATURAL", "BITREV"), ("BITREV", "NATURAL")):
                    assert ident


@pytest.mark.parametrize("cfg", [(10, 16, 1, "FWD"), (10, 20, 1, "PAIR"), (8, 32, 1, "INV"), (11, 40, 0, "FWD"), (16, 16, 0, "FWD"), (16, 24, 1, "INV")],
                         ids=lambda c: "n%d_w%d_f%d_%s" % c)
def test_bypass_equals_the_generic_kernels(cfg, monkeypatch):
    log2n, dw, fmt, direction = cfg
    x = frames(1 << log2n, min(dw, 62), 3 if log2n >= 16 else 9, 7)
    kw = dict(direction=direction, in_order="HALVES" if direction != "INV" else "BITREV", out_order="NATURAL", use_fly=0)
    info = check(x, log2n, dw, 16, fmt, 0, True, **kw)
    assert info["kernel_name"].startswith("bypass["), info
    monkeypatch.setenv("INTFFT_DIAG", "1")
    monkeypatch.setenv("INTFFT_NO_BYPASS_COPY", "1")
    info_g = check(x, log2n, dw, 16, fmt, 0, True, **kw)
    assert not info_g["kernel_name"].startswith("bypass["), info_g


def test_bypass_chunks_workspace_and_in_place(monkeypatch):
    import torch

    from intfftk_amd import IntFFTCore
    from oracle import oracle_c as C
    monkeypatch.setenv("INTFFT_DIAG", "1")
    monkeypatch.setenv("INTFFT_SCRATCH_MB", "1")  # several chunks of the middle buffer
    x = frames(4096, 16, 70, 9)
    check(x, 12, 16, 16, 0, 0, True, direction="FWD", out_order="BITREV", use_fly=0)
    monkeypatch.delenv("INTFFT_SCRATCH_MB")
    p = C.make_params(12, 16, 16, 0, 0, True, 0)
    want = C.execute(x, p, C.FWD, C.NATURAL, C.BITREV, form=1)
    core = IntFFTCore(12, 16, 16, 0, 0, "NEW", "FWD", "NATURAL", "BITREV", 0)
    xin = torch.from_numpy(x.astype(np.int16)).cuda()
    ws = torch.empty(core.workspace_bytes(70), dtype=torch.uint8, device="cuda")
    core.release_scratch()
    assert np.array_equal(core.exec_ws(xin, ws).cpu().numpy().astype(np.int64), want)
    y = xin.clone()  # in place: equal containers
    core.exec_ws(y, ws, out=y)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy().astype(np.int64), want)
    core.close()


def test_results_beyond_64_bits_stay_generic():
    """16-byte containers: the bypass copy is not built (the plan keeps the generic kernels)."""
    from intfftk_amd import IntFFTCore
    core = IntFFTCore(5, 60, 12, 1, 0, "NEW", "FWD", "NATURAL", "NATURAL", 0)  # 65-bit results
    assert core.out_container == 16 and not core.info["kernel_name"].startswith("bypass["), core.info
    core.close()
