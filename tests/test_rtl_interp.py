"""The reference's own structural VHDL, parsed and evaluated (tools/rtl_interp.py), against the hand-wired DSP48 twin and the oracle.

CPU only, and only where the reference tree is present (this container): elsewhere every test here skips.  The interpreter reads
src/vhdl/math/mults/*.vhd, src/vhdl/math/cmult/*.vhd, src/vhdl/math/int_addsub_dsp48.vhd, src/vhdl/fft/int_di[ft]2_fly.vhd,
src/vhdl/twiddle/*.vhd and the generate loops of src/vhdl/fft/int_fftNk.vhd / int_ifftNk.vhd at run time -- generics, the XSER -> constant
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


@pytest.mark.parametrize("xser", ["NEW", "OLD"])
def test_twiddle_generator_from_the_text(xser):
    """rom_twiddle_int as its file reads -- the quarter-wave ROM its function fills from MATH_PI, COS and SIN (magnitude 2^(AWD-1) - 1 below
    18 bits, 2^(AWD-2) - 1 from there), the quadrant rotation of pr_ww, the address slicing, row_twiddle_tay instantiated for STAGE >= 11 --
    read at counter values cnt, against oracle_py.twiddles: the tables every kernel of the engine is checked against."""
    rng = random.Random(31)
    for stage in (2, 3, 4, 7, 10, 11, 12, 15, 18):
        for awd in (16, 24) if stage % 3 else (12, 18, 19, 25 if xser == "OLD" else 27):
            assert R.check_twiddles(stage, awd, xser, bool(stage & 1) and stage >= 11, 10, rng) == 0, (stage, awd)


def _oracle_schedule(direction, log2n, dw, fmt, rnd, monkeypatch):
    """what oracle_py really does per stage: (STAGE, DTW, SCALE, RNDMODE) of every butterfly call and the block length of every commutation"""
    flies, blocks = [], []
    name = "dif_fly" if direction == "FWD" else "dit_fly"
    real_fly, real_rev = getattr(op, name), op._rev2rdx

    def fly(a, b, ww, stage, dtw, t, scale, r, odd, new=True):
        if not flies or flies[-1] != (stage, dtw, scale, r):
            flies.append((stage, dtw, scale, r))
        return real_fly(a, b, ww, stage, dtw, t, scale, r, odd, new)

    def rev(ia, ib, cnti):
        blocks.append(cnti)
        return real_rev(ia, ib, cnti)

    monkeypatch.setattr(op, name, fly)
    monkeypatch.setattr(op, "_rev2rdx", rev)
    x = [(i + 1, -i) for i in range(1 << log2n)]
    (op.fft_dif if direction == "FWD" else op.ifft_dit)(x, log2n, dw, 16, fmt, rnd)
    monkeypatch.setattr(op, name, real_fly)
    monkeypatch.setattr(op, "_rev2rdx", real_rev)
    return flies, blocks


@pytest.mark.parametrize("direction", ["FWD", "INV"])
@pytest.mark.parametrize("cfg", [(5, 16, 0, 0), (5, 16, 0, 1), (6, 12, 1, 0), (4, 24, 1, 0)])
def test_stage_schedule_from_the_text(direction, cfg, monkeypatch):
    """The generate loops of int_fftNk / int_ifftNk: which STAGE, DTW, SCALE, RNDMODE each butterfly instance gets, which STAGE and AWD its
    twiddle generator, which STAGE and NWIDTH each delay line (whose N_INV = NFFT - STAGE - 2 is read from its own file) -- against what
    oracle_py's stream form actually does, recorded while it runs."""
    log2n, dw, fmt, rnd = cfg
    top = "int_fftnk" if direction == "FWD" else "int_ifftnk"
    sch = R.stage_schedule(top, {"nfft": log2n, "ramb_type": "wrap", "format": fmt, "rndmode": rnd, "data_width": dw, "twdl_width": 16,
                                 "xser": "new", "use_mlt": False})
    flies, blocks = _oracle_schedule(direction, log2n, dw, fmt, rnd, monkeypatch)
    text_flies = [(g["stage"], g["dtw"], g["scale"], g["rndmode"]) for _, unit, g, _ in sch if unit in ("int_dif2_fly", "int_dit2_fly")]
    assert text_flies == flies
    assert [(g["stage"], g["awd"]) for _, unit, g, _ in sch if unit == "rom_twiddle_int"] == [(f[0], 16) for f in flies]
    delays = [(unit, g) for _, unit, g, _ in sch if unit.startswith("int_delay")]
    assert [1 << R.delay_block_log2(unit, g["nfft"], g["stage"]) for unit, g in delays] == blocks
    assert [g["nwidth"] for _, g in delays] == [2 * (dw + (ii + 1) * fmt) for ii in range(log2n - 1)]
    # the other RAMB_TYPE instantiates the other delay line with the same generics
    sch2 = R.stage_schedule(top, {"nfft": log2n, "ramb_type": "cont", "format": fmt, "rndmode": rnd, "data_width": dw, "twdl_width": 16,
                                  "xser": "new", "use_mlt": False})
    assert [(u.replace("wrap", "line"), g) for _, u, g, _ in sch] == [(u, g) for _, u, g, _ in sch2]


@pytest.mark.parametrize("unit,gap", [("int_delay_line", 0), ("int_delay_wrap", 0), ("int_delay_wrap", 1), ("int_delay_wrap", 7)])
def test_cross_commutation_clocked_from_the_text(unit, gap):
    """The delay lines are counters, two memories and a crossbar: time is their function, so they are CLOCKED from the text -- every
    register, counter, memory and process, cycle by cycle, four frames back to back (int_delay_wrap also with idle clocks between the
    frames, which is what RAMB_TYPE = "WRAP" tolerates) -- and the stream of valid output beats must be the cross-commutation
    oracle_py._rev2rdx (= fn_rev2rdx, math/fn_radix2.m:51-69) performs with block length 2^N_INV."""
    rng = random.Random(41)
    for nfft in (3, 4, 5, 6, 7):
        for stage in range(nfft - 1):  # N_INV = NFFT - STAGE - 2 from NFFT - 2 down to 0 (the register-only form / null address ranges)
            assert R.check_delay(unit, nfft, stage, gap, rng) == 0, (nfft, stage)


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
    # 7. the sine of the twiddle ROM with the other sign; 8. the ROM magnitude of wide twiddles one bit larger
    assert "sin(-pi_std)" in real("rom_twiddle_int") and "(2.0 ** (xmag-2)) - 1.0" in real("rom_twiddle_int")
    monkeypatch.setattr(R, "_load", edited("sin(-pi_std)", "sin(pi_std)"))
    R.forget()
    assert R.check_twiddles(5, 16, "NEW", False, 20, rng) > 0
    monkeypatch.setattr(R, "_load", edited("(2.0 ** (xmag-2)) - 1.0", "(2.0 ** (xmag-1)) - 1.0"))
    R.forget()
    assert R.check_twiddles(6, 24, "NEW", False, 20, rng) > 0 and R.check_twiddles(6, 16, "NEW", False, 20, rng) == 0
    # 9. the crossbar of the delay line switched by the wrong address bit
    assert "cross <= cnt_adr(n_inv)" in real("int_delay_line")
    monkeypatch.setattr(R, "_load", edited("cross <= cnt_adr(n_inv)", "cross <= cnt_adr(n_inv-1)"))
    R.forget()
    assert R.check_delay("int_delay_line", 5, 0, 0, rng) > 0
    monkeypatch.setattr(R, "_load", real)
    R.forget()
    assert R.check_delay("int_delay_line", 5, 0, 0, rng) == 0
    assert R.check_twiddles(5, 16, "NEW", False, 10, rng) == 0 and R.check_taylor(16, 3, "NEW", False, 10, rng) == 0 and R.check_cmult(60, 16, "NEW", 20, rng) == 0 and R.check_fly("dit", 16, 16, 1, 0, 5, 0, "NEW", 20, rng) == 0
