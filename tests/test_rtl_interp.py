"""The reference's own structural VHDL, parsed and evaluated (tools/rtl_interp.py), against the hand-wired DSP48 twin and the oracle.

CPU only, and only where the reference tree is present (this container): elsewhere every test here skips.  The interpreter reads
src/vhdl/math/mults/*.vhd, src/vhdl/math/cmult/*.vhd, src/vhdl/math/int_addsub_dsp48.vhd and src/vhdl/fft/int_di[ft]2_fly.vhd at run time -- generics, the XSER -> constant
functions, if / for generate, signals, slices, SXT, entity and DSP48 instantiations with their port maps -- and runs them as a dataflow
network on the DSP48 slice model of oracle/dsp48_twin.py.  So the WIRING in these comparisons is the reference's text, not anybody's
reading of it; what remains assumed is the slice model (UG479 / UG579).  Parity stays unpinned (no RTL simulation), but a misread port
map can no longer hide in the oracles."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import rtl_interp as R  # noqa: E402

from oracle import dsp48_twin as tw  # noqa: E402
from oracle import oracle_py as op  # noqa: E402

pytestmark = pytest.mark.skipif(not R.available(), reason="the reference tree is not present on this host")


@pytest.mark.parametrize("name", sorted(R.MULTS))
def test_multipliers_from_the_text_are_exact(name):
    assert R.check_mult(name, 100, random.Random(3)) == 0


def _cases():
    by = {}
    for new in (True, False):
        for t in range(8, 28):
            for w in range(8, 79):
                r = op.cmult_regime(w, t, new)
                if r:
                    by.setdefault((r, new), []).append((w, t))
    return by


@pytest.mark.parametrize("new", [True, False])
@pytest.mark.parametrize("regime", ["sngl", "sngl25", "dbl18", "dbl35", "trpl18", "trpl52"])
def test_complex_multiplier_tree_from_the_text(regime, new):
    lst = _cases()[(regime, new)]
    rng = random.Random(len(regime) * 2 + int(new))
    for w, t in rng.sample(lst, 4) + [min(lst), max(lst)]:
        assert R.check_cmult(w, t, "NEW" if new else "OLD", 15, rng) == 0, (regime, w, t)


@pytest.mark.parametrize("xser", ["NEW", "OLD"])
def test_adder_from_the_text(xser):
    rng = random.Random(9)
    for dspw in (8, 15, 16, 23, 24, 25, 31, 47, 48, 49, 64, 80, 95):
        assert R.check_addsub(dspw, xser, 20, rng) == 0, dspw


FLY_CASES = [  # (dtw, tfw, scale, rndmode, stage, odd, xser)
    (16, 16, 1, 0, 0, 0, "NEW"), (16, 16, 1, 0, 1, 1, "NEW"), (16, 16, 1, 0, 1, 0, "OLD"), (16, 16, 1, 0, 9, 0, "NEW"), (16, 16, 1, 1, 3, 0, "NEW"),
    (16, 16, 1, 1, 1, 1, "OLD"), (16, 16, 1, 1, 0, 0, "NEW"), (24, 24, 0, 0, 5, 0, "NEW"), (24, 24, 0, 0, 12, 1, "NEW"), (30, 16, 0, 0, 1, 1, "NEW"),
    (40, 24, 1, 1, 7, 0, "OLD"), (50, 16, 0, 0, 2, 0, "NEW"), (24, 16, 1, 0, 0, 0, "OLD"), (12, 10, 1, 0, 4, 0, "NEW"), (47, 16, 0, 0, 0, 0, "OLD"),
]


@pytest.mark.parametrize("kind", ["dif", "dit"])
def test_butterflies_from_the_text(kind):
    """int_dif2_fly / int_dit2_fly as the files read: the adder instance and its generics, pr_rnd, pr_inv (the not(x) quirk), the
    multiplier instance with the DIT's exchanged feed -- against the twin and oracle_py."""
    rng = random.Random(12)
    for c in FLY_CASES:
        assert R.check_fly(kind, *c, 25, rng) == 0, c


@pytest.mark.parametrize("xser", ["NEW", "OLD"])
@pytest.mark.parametrize("use_mlt", [False, True])
def test_taylor_correction_from_the_text(xser, use_mlt):
    """row_twiddle_tay: XSHIFT and MATHPI from its functions (INTEGER(MATH_PI * 2.0**(13 - ii - del))), the MATHPI * cnt ROM that read_rom
    fills (or the multiplier process), the operand placement loops, the two DSP48 slices with their ALUMODE aggregates, pr_rnd and the
    crossed outputs -- against the twin (which tests/test_dsp48_twin.py holds against oracle_py.twiddles)."""
    rng = random.Random(21)
    for awd, ii in ((16, 0), (16, 4), (24, 4), (24, 7), (12, 7), (25 if xser == "OLD" else 27, 2), (19, 5)):
        assert R.check_taylor(awd, ii, xser, use_mlt, 12, rng) == 0, (awd, ii)


def test_taylor_ii_8_does_not_elaborate_in_the_reference():
    """N = 2^20 needs STAGE 19 = ii 8, where rom_cnt (ii + 1 = 9 bits) no longer fits cnt_exp(7 downto 0): the reference's text does not
    elaborate there, and what the engine computes for C4's top stage is the extension SURVEY.md section 9.6 / DESIGN.md section 2 define."""
    with pytest.raises(AssertionError, match="does not elaborate"):
        R.evaluate("row_twiddle_tay", {"awd": 16, "xser": "new", "use_mlt": False, "ii": 8}, {"rom_ww": 1, "rom_cnt": 300, "rstp": 0})


def test_generate_tree_elaborates_where_the_oracle_says():
    """Width pairs outside every generate condition leave DO_RE / DO_IM undriven: the oracle calls them unsupported."""
    for w, t, new in ((28, 17, True), (80, 16, True), (30, 28, True), (26, 16, False), (53, 24, True), (18, 19, True), (78, 8, False)):
        try:
            R.evaluate("int_cmult_dsp48", {"dtw": w, "twd": t, "xser": "new" if new else "old"},
                       {"di_re": 1, "di_im": 1, "ww_re": 1, "ww_im": 1})
            driven = True
        except (R.NotReady, AssertionError, KeyError):
            driven = False
        assert driven == (op.cmult_regime(w, t, new) is not None), (w, t, new)


def test_the_comparison_reads_the_text(monkeypatch):
    """Three edits of the TEXT (not of the twin) must each show up as mismatches: the interpreter really follows the files."""
    real = R._load

    def edited(old, new):
        def load(entity):
            t = real(entity)
            return t.replace(old, new)
        return load

    rng = random.Random(5)
    # 1. the A-operand split of mlt44x18 one bit lower
    assert "mlt_a(43 downto 17)" in real("mlt44x18_dsp48e2")
    monkeypatch.setattr(R, "_load", edited("mlt_a(43 downto 17)", "mlt_a(42 downto 16)"))
    R.forget()
    assert R.check_mult("mlt44x18_dsp48e2", 20, rng) > 0
    # 2. the product window of the dbl18 block one bit higher
    assert "dsppm1(pwd-1-(18-mbw)" not in real("int_cmult_dbl18_dsp48") and "dspp_m1(pwd-1-(18-mbw) downto pwd-48-(18-mbw))" in real("int_cmult_dbl18_dsp48")
    monkeypatch.setattr(R, "_load", edited("dspp_m1(pwd-1-(18-mbw) downto pwd-48-(18-mbw))", "dspp_m1(pwd-(18-mbw) downto pwd-47-(18-mbw))"))
    R.forget()
    assert R.check_cmult(30, 16, "NEW", 20, rng) > 0
    # 3. the carry chain of the 96-bit closing adder of trpl18 cut
    assert 'carryinsel => "010"' in real("int_cmult_trpl18_dsp48")
    monkeypatch.setattr(R, "_load", edited('carryinsel => "010"', 'carryinsel => "000"'))
    R.forget()
    assert R.check_cmult(60, 16, "NEW", 200, rng) > 0
    # 4. the DIT butterfly's multiplier fed straight instead of exchanged (int_dit2_fly.vhd:304-322)
    assert "di_re => ib_im" in real("int_dit2_fly")
    monkeypatch.setattr(R, "_load", edited("di_re => ib_im, di_im => ib_re", "di_re => ib_re, di_im => ib_im"))
    R.forget()
    assert R.check_fly("dit", 16, 16, 1, 0, 5, 0, "NEW", 20, rng) > 0
    # 5. the truncating DIF butterfly adding full-width operands instead of ia(dtw-1 downto 1)
    assert "ia_re => ia_re(dtw-1 downto 1)" in real("int_dif2_fly")
    monkeypatch.setattr(R, "_load", edited("(dtw-1 downto 1)", "(dtw-2 downto 0)"))
    R.forget()
    assert R.check_fly("dif", 16, 16, 1, 0, 0, 0, "NEW", 20, rng) > 0
    # 6. the Taylor correction rounded one bit lower
    assert "cos_prod(47 downto xshift-1)" in real("row_twiddle_tay")
    monkeypatch.setattr(R, "_load", edited("_prod(47 downto xshift-1)", "_prod(46 downto xshift-2)"))
    R.forget()
    assert R.check_taylor(16, 3, "NEW", False, 20, rng) > 0
    monkeypatch.setattr(R, "_load", real)
    R.forget()
    assert R.check_taylor(16, 3, "NEW", False, 10, rng) == 0 and R.check_cmult(60, 16, "NEW", 20, rng) == 0 and R.check_fly("dit", 16, 16, 1, 0, 5, 0, "NEW", 20, rng) == 0
