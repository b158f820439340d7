"""Whole frames that came out of the reference's own VHDL text, clocked (tests/golden/from_rtl_sim.npz, written by
tests/golden/make_golden_from_rtl_sim.py through tools/rtl_sim.py where the reference tree exists) -- against the C oracle, the Python
twin (CPU) and the HIP path through the C-ABI (-m gpu).

The cases: int_fftNk and int_ifftNk alone (the cores' own beat orders), int_fft_single_path (natural in, natural out through the
reference's input / output / bit-reverse buffers) and int_fft_ifft_pair (BASELINE's C5 shape: FFT -> IFFT through iobuf_flow_int2).
The expected frames do not come from any oracle of this repository: the chain is reference text -> fixture -> engine."""
import os

import numpy as np
import pytest

from oracle import oracle_c as C
from oracle import oracle_py as P

PATH = os.path.join(os.path.dirname(__file__), "golden", "from_rtl_sim.npz")
Z = dict(np.load(PATH))
BIG = os.path.join(os.path.dirname(__file__), "golden", "from_rtl_sim_big.npz")  # one full-size frame of C3's shape (make_golden_from_rtl_sim.py --big)
if os.path.exists(BIG):
    _b = dict(np.load(BIG))
    Z.update({k: v for k, v in _b.items() if k != "cases"})
    Z["cases"] = np.concatenate([Z["cases"], _b["cases"]])
CASES = [ln.split() for ln in Z["cases"]]
IDS = [c[0] for c in CASES]
NP = {2: np.int16, 4: np.int32, 8: np.int64}


def unpack(c):
    name, kind, d = c[:3]
    nfft, dw, tw, fmt, rnd = (int(v) for v in c[3:8])
    return name, kind, d, nfft, dw, tw, fmt, rnd, c[8], c[9], c[10]


def test_fixture_covers_what_it_says():
    kinds = {c[1] for c in CASES}
    assert kinds == {"core_fwd", "core_inv", "single", "pair"} and len(CASES) >= 60
    assert {int(c[3]) for c in CASES} >= {3, 4, 5, 6, 7, 10, 12, 13}
    assert {(int(c[6]), int(c[7])) for c in CASES} == {(0, 0), (0, 1), (1, 0)} and {c[8] for c in CASES} == {"NEW", "OLD"}
    names = {c[0] for c in CASES}
    # BASELINE's C2 and C5 shapes, N = 4096 / 8192 (Taylor twiddles from STAGE 11 on), C3's 24-bit unscaled regime walk at N = 4096
    assert {"single_n10_w16_t16_f0_r0_NEW", "pair_n12_w16_t16_f0_r0_NEW", "core_fwd_n13_w16_t16_f0_r0_NEW", "core_fwd_n12_w24_t24_f1_r0_NEW"} <= names
    for c in CASES:
        x, y = Z[c[0] + "_x"], Z[c[0] + "_y"]
        assert x.shape[1:] == y.shape[1:] == (1 << int(c[3]), 2) and 1 <= y.shape[0] <= x.shape[0]
        if c[1].startswith("core"):
            assert y.shape[0] == x.shape[0]  # RAMB_TYPE = "CONT": every frame comes out


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_c_oracle_equals_the_text(c):
    name, kind, d, nfft, dw, tw, fmt, rnd, xser, in_o, out_o = unpack(c)
    x, y = Z[name + "_x"], Z[name + "_y"]
    p = C.make_params(nfft, dw, tw, fmt, rnd, xser == "NEW", 1)
    assert C.lib().orc_validate(p, getattr(C, d)) == 0
    for form in (0, 1):  # in place and the stream form that walks the delay lines
        got = C.execute(x[: y.shape[0]], p, getattr(C, d), getattr(C, in_o), getattr(C, out_o), form=form)
        assert np.array_equal(got, y), (name, form)


@pytest.mark.parametrize("c", [c for c in CASES if int(c[3]) <= 4], ids=[c[0] for c in CASES if int(c[3]) <= 4])
def test_python_twin_equals_the_text(c):
    name, kind, d, nfft, dw, tw, fmt, rnd, xser, in_o, out_o = unpack(c)
    x, y = Z[name + "_x"], Z[name + "_y"]
    n = 1 << nfft
    rev = [int(format(m, "0%db" % nfft)[::-1], 2) for m in range(n)]
    ow = dw + fmt * nfft * (2 if d == "PAIR" else 1)
    for f in range(y.shape[0]):
        fr = [(int(a), int(b)) for a, b in x[f]]
        if kind == "core_fwd":
            got = P.fft_dif(fr, nfft, dw, tw, fmt, rnd, xser == "NEW", 1)            # the bit-reversed sequence = BITREV memory
        elif kind == "core_inv":
            got = P.ifft_dit(fr, nfft, dw, tw, fmt, rnd, xser == "NEW", 1)           # takes the bit-reversed sequence
        else:
            got = P.execute(fr, nfft, dw, tw, fmt, rnd, xser == "NEW", getattr(P, d))
        got = [(P.sgn(a, ow), P.sgn(b, ow)) for a, b in got]
        assert got == [(int(a), int(b)) for a, b in y[f]], (name, f)
    assert rev[1] == n >> 1


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_hip_engine_equals_the_text(c):
    """The product path: plan + exec through the C-ABI, whatever kernel the planner picks for the shape."""
    import torch

    from intfftk_amd import IntFFTCore
    name, kind, d, nfft, dw, tw, fmt, rnd, xser, in_o, out_o = unpack(c)
    x, y = Z[name + "_x"], Z[name + "_y"]
    core = IntFFTCore(nfft, dw, tw, fmt, rnd, xser, d, in_o, out_o, 1)
    xin = torch.from_numpy(np.ascontiguousarray(x.astype(NP[core.in_container]))).cuda()
    got = core(xin)
    torch.cuda.synchronize()
    got = got.cpu().numpy().astype(np.int64)
    info = dict(core.info)
    core.close()
    assert np.array_equal(got[: y.shape[0]], y), (name, info["kernel_name"])


@pytest.mark.gpu
def test_hip_engine_equals_the_text_in_one_ragged_batch():
    """The frames of all 16-bit scaled N = 32 forward cases in one call, repeated to a batch that does not divide the kernels' tiles."""
    import torch

    from intfftk_amd import IntFFTCore
    name = "core_fwd_n5_w16_t16_f0_r0_NEW"
    x, y = Z[name + "_x"], Z[name + "_y"]
    reps = 37
    core = IntFFTCore(5, 16, 16, 0, 0, "NEW", "FWD", "NATURAL", "BITREV", 1)
    xin = torch.from_numpy(np.ascontiguousarray(np.tile(x, (reps, 1, 1)).astype(np.int16))).cuda()
    got = core(xin).cpu().numpy().astype(np.int64)
    core.close()
    assert np.array_equal(got, np.tile(y, (reps, 1, 1)))


@pytest.mark.gpu
def test_the_headline_kernels_themselves_meet_the_text():
    """C2's and C5's exact shapes: the frames that the reference's wrappers handed out, against the kernels bench.py times."""
    import torch

    from intfftk_amd import IntFFTCore
    for name, d, nfft, kernel in (("single_n10_w16_t16_f0_r0_NEW", "FWD", 10, "k_fft1024_i16"), ("pair_n12_w16_t16_f0_r0_NEW", "PAIR", 12, "k_fft4096_i16")):
        x, y = Z[name + "_x"], Z[name + "_y"]
        core = IntFFTCore(nfft, 16, 16, 0, 0, "NEW", d, "NATURAL", "NATURAL", 1)
        assert core.info["kernel_name"] == kernel and core.info["fast_path"] == 1, core.info
        # a batch that fills the chip, the fixture's frames scattered through it
        reps = 257
        xin = torch.from_numpy(np.ascontiguousarray(np.tile(x, (reps, 1, 1)).astype(np.int16))).cuda()
        got = core(xin).cpu().numpy().astype(np.int64)
        core.close()
        want = np.tile(np.concatenate([y, np.zeros((x.shape[0] - y.shape[0],) + y.shape[1:], np.int64)]), (reps, 1, 1))
        mask = np.tile(np.arange(x.shape[0]) < y.shape[0], reps)
        assert np.array_equal(got[mask], want[mask]), name
