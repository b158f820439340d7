"""The DSP48-primitive-level structural twin (oracle/dsp48_twin.py) against the two "slice of P" oracles.

CPU only.  The twin wires every mlt* / int_cmult* / int_addsub_dsp48 / row_twiddle_tay entity the way its port map reads
(src/vhdl/math/mults/*.vhd, src/vhdl/math/cmult/*.vhd, src/vhdl/math/int_addsub_dsp48.vhd, src/vhdl/twiddle/row_twiddle_tay.vhd)
around one generic model of a DSP48E1 / DSP48E2 slice; oracle_py.py and intfft_oracle.c state the same arithmetic as slices
of exact products.  A bounded sample here; tools/dsp48_fuzz.py is the long soak (profiles/r06_dsp48_twin_fuzz.txt).
"""
import random
import zlib

import pytest

from oracle import dsp48_twin as tw
from oracle import oracle_c
from oracle import oracle_py as op

import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import dsp48_fuzz as fz  # noqa: E402


# ---------------------------------------------------------------------------------------------- the slice itself

def test_slice_multiplier_sees_only_the_low_a_bits():
    # DSP48E1 multiplies A[24:0], DSP48E2 A[26:0]: a 27-bit operand survives on E2 only
    a = tw.vec(-(1 << 26) + 12345, 30)
    b = tw.vec(-777, 18)
    p2, _, _ = tw.dsp48("E2", opmode="000000101", a=a, b=b)
    assert tw.signed(p2, 48) == (-(1 << 26) + 12345) * -777
    p1, _, _ = tw.dsp48("E1", opmode="0000101", a=a, b=b)
    assert tw.signed(p1, 48) == tw.signed(tw.sl(a, 24, 0), 25) * -777 != tw.signed(p2, 48)


def test_slice_cascade_shift_is_arithmetic():
    pc = tw.vec(-5 << 17 | 0x1ABCD, 48)
    p, _, _ = tw.dsp48("E1", opmode="1010101", a=3, b=4, pcin=pc)
    assert tw.signed(p, 48) == -5 + 12


def test_slice_subtract_borrow_chain_is_a_96_bit_subtract():
    rng = random.Random(7)
    for _ in range(2000):
        x, y = rng.getrandbits(96), rng.getrandbits(96)
        lo, _, cy = tw.dsp48("E2", use_mult="NONE", opmode="000110011", alumode="0011", a=tw.sl(y, 47, 18), b=tw.sl(y, 17, 0),
                             c=tw.sl(x, 47, 0))
        hi, _, _ = tw.dsp48("E2", use_mult="NONE", opmode="000110011", alumode="0011", a=tw.sl(y, 95, 66), b=tw.sl(y, 65, 48),
                            c=tw.sl(x, 95, 48), carryinsel="010", carrycascin=cy)
        assert lo | (hi << 48) == (x - y) & ((1 << 96) - 1)


def test_slice_simd_two24_has_no_carry_between_lanes():
    ab = 0xFFFFFF | (1 << 24)
    p, _, _ = tw.dsp48("E1", use_mult="NONE", opmode="0110011", a=tw.sl(ab, 47, 18), b=tw.sl(ab, 17, 0), c=1, use_simd="TWO24")
    assert p == (1 << 24)  # lane 0 wraps to 0 and does not touch lane 1


# ---------------------------------------------------------------------------------------------- exact multipliers

@pytest.mark.parametrize("name", ["mlt42x18_dsp48e1", "mlt44x18_dsp48e2", "mlt35x25_dsp48e1", "mlt35x27_dsp48e2", "mlt59x18_dsp48e1",
                                  "mlt61x18_dsp48e2", "mlt52x25_dsp48e1", "mlt52x27_dsp48e2"])
def test_multipliers_are_exact(name):
    assert fz.run_mlt(name, 3000, 11) == 0


# ---------------------------------------------------------------------------------------------- complex multiplier

SURVEY_KATS = [  # (d.re, d.im, wr, wi, w, t, NEW?) -> (re, im)
    ((-12345, 23456, 23170, -23170, 16, 16, True), (7856, 25314)),
    ((32767, -32768, -30273, -12539, 16, 16, True), (22724, 17734)),
    ((-123456789, 98765432, 30273, -12539, 30, 16, True), (-76263050, 138487261)),
    ((-123456789, 98765432, 30273, -12539, 30, 16, False), (-76263050, 138487261)),
    ((291770562, 216703618, 21856, -24413, 30, 16, True), (356058436, -72836929)),
    ((291770562, 216703618, 21856, -24413, 30, 16, False), (356058435, -72836928)),
    ((-123456789012, 98765432101, 12539, -30273, 46, 16, True), (44003334002, 151850193080)),
    ((-30000, 29999, 2965820, -2965820, 17, 24, True), (-1, 42425)),
    ((-123456789, 98765432, 2965820, -2965820, 30, 24, True), (-17459422, 157134796)),
    ((-123456789012, 98765432101, -2965820, -2965820, 40, 24, True), (157134797054, 17459421194)),
]


@pytest.mark.parametrize("args,want", SURVEY_KATS)
def test_cmult_survey_kats_through_the_slices(args, want):
    dr, di, wr, wi, w, t, new = args
    g = tw.int_cmult_dsp48(tw.vec(dr, w), tw.vec(di, w), tw.vec(wr, t), tw.vec(wi, t), w, t, "NEW" if new else "OLD")
    assert (tw.signed(g[0], w), tw.signed(g[1], w)) == want


def _cases(new):
    by = {}
    for r, w, t in fz.cmult_cases(new):
        by.setdefault(r, []).append((w, t))
    return by


@pytest.mark.parametrize("new", [True, False])
@pytest.mark.parametrize("regime", ["sngl", "sngl25", "dbl18", "dbl35", "trpl18", "trpl52"])
def test_cmult_regime_equals_the_python_oracle(regime, new):
    cases = _cases(new)[regime]
    rng = random.Random(zlib.crc32(regime.encode()) + int(new))
    pick = rng.sample(cases, min(24, len(cases)))
    # always the extremes of the regime: the narrowest / widest data and twiddle it elaborates
    pick += [min(cases), max(cases), min(cases, key=lambda c: (c[1], c[0])), max(cases, key=lambda c: (c[1], c[0]))]
    for i, (w, t) in enumerate(pick):
        assert fz.run_cmult(new, w, t, 150, 1000 + i) == 0, (regime, w, t)


def test_cmult_generate_tree_elaborates_exactly_where_the_oracle_says():
    for new in (True, False):
        for t in range(8, 30):
            for w in range(8, 82):
                r = op.cmult_regime(w, t, new)
                try:
                    g = tw.int_cmult_dsp48(0, 0, 0, 0, w, t, "NEW" if new else "OLD")
                except AssertionError:  # a slice outside P: the RTL does not elaborate
                    g = None
                assert (g is None) == (r is None), (w, t, new, r)


def test_cmult_equals_the_c_oracle_where_results_fit_64_bits():
    rng = random.Random(5)
    for new in (True, False):
        for regime, cases in _cases(new).items():
            for w, t in rng.sample([c for c in cases if c[0] <= 62], 6):
                for _ in range(100):
                    dr, di, wr, wi = fz.operand(rng, w), fz.operand(rng, w), fz.operand(rng, t), fz.operand(rng, t)
                    g = tw.int_cmult_dsp48(tw.vec(dr, w), tw.vec(di, w), tw.vec(wr, t), tw.vec(wi, t), w, t, "NEW" if new else "OLD")
                    assert (tw.signed(g[0], w), tw.signed(g[1], w)) == tuple(oracle_c.cmult(dr, di, wr, wi, w, t, new)), (regime, w, t)


# ---------------------------------------------------------------------------------------------- adder / Taylor

@pytest.mark.parametrize("new", [True, False])
def test_addsub_all_three_generate_branches(new):
    for dspw in list(range(2, 30)) + [46, 47, 48, 49, 63, 64, 65, 80, 95]:
        assert fz.run_addsub(new, dspw, 200, dspw) == 0, dspw


@pytest.mark.parametrize("new", [True, False])
@pytest.mark.parametrize("stage", [11, 12, 15, 19])
def test_taylor_equals_the_python_oracle(stage, new):
    for t in (16, 24) + ((27,) if new else (25,)):
        assert fz.run_taylor(new, stage, t, 1500, stage * 31 + t) == 0, (stage, t)


@pytest.mark.parametrize("new", [True, False])
@pytest.mark.parametrize("kind", ["dif", "dit"])
def test_butterflies_wired_through_the_slices(kind, new):
    assert fz.run_fly(new, kind, 4000, 77) == 0


# ---------------------------------------------------------------------------------------------- the comparison has teeth

def test_seeded_defects_are_caught(monkeypatch):
    """Three misreadings of the wiring, each of which the comparison above must flag."""
    real = tw.dsp48

    def shift16(series, **kw):  # a 16-bit instead of a 17-bit cascade shift
        if kw.get("opmode", "")[-7:-4] == "101":
            kw = dict(kw, opmode=kw["opmode"][:-7] + "001" + kw["opmode"][-4:], pcin=tw.vec(tw.signed(kw["pcin"], 48) >> 16, 48))
        return real(series, **kw)

    monkeypatch.setattr(tw, "dsp48", shift16)
    assert fz.run_cmult(True, 30, 16, 50, 1) > 0
    assert fz.run_mlt("mlt52x27_dsp48e2", 50, 1) > 0

    def no_carry(series, **kw):  # the 96-bit adders without their carry chain
        return real(series, **dict(kw, carrycascin=0))

    monkeypatch.setattr(tw, "dsp48", no_carry)
    assert fz.run_cmult(True, 60, 16, 300, 2) > 0
    assert fz.run_addsub(True, 70, 300, 2) > 0

    def e1_as_e2(series, **kw):  # a DSP48E1 whose multiplier took 27 bits of A: invisible, the OLD wiring never needs them ...
        if series == "E1":
            return real("E2", **dict(kw, opmode="00" + kw["opmode"]))
        return real(series, **kw)

    monkeypatch.setattr(tw, "dsp48", e1_as_e2)
    assert fz.run_cmult(False, 25, 16, 200, 3) == 0

    def e2_as_e1(series, **kw):  # ... and a DSP48E2 that took only 25: the NEW wiring relies on all 27
        if series == "E2":
            return real("E1", **dict(kw, opmode=kw["opmode"][2:]))
        return real(series, **kw)

    monkeypatch.setattr(tw, "dsp48", e2_as_e1)
    assert fz.run_cmult(True, 27, 16, 200, 4) > 0
