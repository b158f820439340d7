"""Structural twin of the reference's DSP48 arithmetic at PRIMITIVE level (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: like everything under oracle/, this file is the builder's reading of the RTL text; it is
not an RTL simulation.  What it adds over oracle_py.py / intfft_oracle.c (which work at the "slice of P"
level that SURVEY.md section 9.4 summarises) is the level below: one generic model of a Xilinx DSP48E1 /
DSP48E2 slice (`dsp48`: the X / Y / Z / W multiplexers decoded from OPMODE, the ALUMODE decode, the 25 x 18 /
27 x 18 multiplier that only sees A[24:0] / A[26:0], the 17-bit-shifted PCIN cascade, SIMD TWO24, the
CARRYCASCOUT -> CARRYCASCIN chain, everything wrapping at 48 bits), and every multiplier / complex
multiplier / adder entity of the reference WIRED THE WAY ITS PORT MAP READS: each function below follows one
entity, takes and returns raw std_logic_vector contents (unsigned Python ints of a stated width), slices them
with the VHDL's own `hi downto lo` indices and calls `dsp48` with the OPMODE / ALUMODE strings of that
instance.  A misreading shared by the two "slice of P" oracles (a cascade that does not add up to the exact
product, an operand that does not fit the multiplier's A port, a carry chain that does not make a 96-bit
subtract) would show here as a mismatch in tests/test_dsp48_twin.py / tools/dsp48_fuzz.py.

The pipeline registers (AREG / BREG / MREG / PREG, the `when rising_edge(clk)` delays that align the
partial products) are modelled as wires: the data path is feed-forward, and latency alignment is the RTL's
concern, not the arithmetic's.

DSP48 semantics used (Xilinx UG479 "7 Series DSP48E1 Slice", UG579 "UltraScale Architecture DSP Slice"):
  * multiplier: A[24:0] x B[17:0] (E1), A[26:0] x B[17:0] (E2), both two's complement, M sign-extended to 48;
    with X = Y = "01" the two partial products add to M;
  * X: 00 -> 0, 01 -> M, 10 -> P, 11 -> A:B (A[29:0] & B[17:0]);  Y: 00 -> 0, 01 -> M, 10 -> all ones, 11 -> C;
    Z: 000 -> 0, 001 -> PCIN, 010 -> P, 011 -> C, 101 -> PCIN >> 17 (arithmetic), 110 -> P >> 17;
    W (E2 only, OPMODE[8:7]): 00 -> 0, 01 -> P, 11 -> C;
  * ALUMODE 0000: Z + (W + X + Y + CIN); 0011: Z - (W + X + Y + CIN), built as not(not(Z) + W + X + Y + CIN);
    0001: not(Z) + (W + X + Y + CIN); 0010: not(Z + W + X + Y + CIN);
  * CARRYINSEL 000 -> CARRYIN, 010 -> CARRYCASCIN; CARRYCASCOUT = the carry out of bit 47 of that internal
    addition (for ALUMODE 0011 this is the borrow of Z - (X + Y + CIN): the documented 96-bit subtracter);
  * USE_SIMD TWO24: the 48-bit adder splits into two 24-bit lanes with no carry between them.
All citations are relative to the reference repository (hukenovs/intfftk).
"""
from __future__ import annotations

import math

M48 = (1 << 48) - 1


# ------------------------------------------------------------------------------------------------
# std_logic_vector helpers: a vector is (unsigned int, width) with the width kept by the caller
# ------------------------------------------------------------------------------------------------

def vec(v: int, w: int) -> int:
    """conv_std_logic_vector(v, w): the low w bits of a two's-complement integer."""
    return v & ((1 << w) - 1)


def sl(v: int, hi: int, lo: int) -> int:
    """v(hi downto lo) as a vector of hi-lo+1 bits."""
    assert hi >= lo >= 0
    return (v >> lo) & ((1 << (hi - lo + 1)) - 1)


def sxt(v: int, w_from: int, w_to: int) -> int:
    """ieee.std_logic_arith.SXT(v, w_to) of a w_from-bit vector: sign-extends, or drops the top bits if w_to < w_from."""
    if w_to <= w_from:
        return v & ((1 << w_to) - 1)
    if (v >> (w_from - 1)) & 1:
        v |= ((1 << (w_to - w_from)) - 1) << w_from
    return v


def signed(v: int, w: int) -> int:
    return v - (1 << w) if (v >> (w - 1)) & 1 else v


def rep(bit: int, n: int) -> int:
    """(others => bit) over n positions."""
    return ((1 << n) - 1) if bit else 0


# ------------------------------------------------------------------------------------------------
# one DSP48E1 / DSP48E2 slice
# ------------------------------------------------------------------------------------------------

def dsp48(series: str, *, opmode: str, alumode: str = "0000", use_mult: str = "MULTIPLY", a: int = 0, b: int = 0,
          c: int = 0, pcin: int = 0, carryin: int = 0, carryinsel: str = "000", carrycascin: int = 0,
          use_simd: str = "ONE48"):
    """-> (P, PCOUT, CARRYCASCOUT).  a: 30-bit, b: 18-bit, c / pcin: 48-bit vectors.  P has no feedback use in the
    reference (no accumulator), so X = 10 / Z = 010 / W = 01 are rejected rather than modelled."""
    assert series in ("E1", "E2")
    assert len(opmode) == (7 if series == "E1" else 9), (series, opmode)
    assert 0 <= a < (1 << 30) and 0 <= b < (1 << 18) and 0 <= c <= M48 and 0 <= pcin <= M48
    wsel = opmode[:-7] if series == "E2" else "00"
    zsel, ysel, xsel = opmode[-7:-4], opmode[-4:-2], opmode[-2:]
    aw = 25 if series == "E1" else 27
    m = 0
    if use_mult == "MULTIPLY":
        m = (signed(sl(a, aw - 1, 0), aw) * signed(b, 18)) & M48
    else:
        assert use_mult == "NONE" and xsel != "01" and ysel != "01"
    if xsel == "01" or ysel == "01":
        assert xsel == "01" and ysel == "01", "X and Y must both select the multiplier"
        xy = m  # the two partial products add to M
    else:
        x = {"00": 0, "11": (a << 18) | b}[xsel]
        y = {"00": 0, "10": M48, "11": c}[ysel]
        xy = x + y
    z = {"000": 0, "001": pcin, "011": c, "101": vec(signed(pcin, 48) >> 17, 48)}[zsel]
    wv = {"00": 0, "11": c}[wsel]
    cin = {"000": carryin, "010": carrycascin}[carryinsel]
    if use_simd == "TWO24":
        assert use_mult == "NONE" and cin == 0 and alumode in ("0000", "0011")
        out = 0
        for lane in (0, 24):
            zl, sl_ = sl(z, lane + 23, lane), sl(xy, lane + 23, lane) + sl(wv, lane + 23, lane)
            r = (zl + sl_) if alumode == "0000" else (zl - sl_)
            out |= (r & 0xFFFFFF) << lane
        return out, out, 0
    assert use_simd == "ONE48"
    s = wv + xy + cin
    if alumode == "0000":
        full = z + s
        p, cout = full & M48, (full >> 48) & 1
    elif alumode == "0011":  # not(not(Z) + W + X + Y + CIN)
        full = ((~z) & M48) + s
        p, cout = (~full) & M48, (full >> 48) & 1
    elif alumode == "0001":
        full = ((~z) & M48) + s
        p, cout = full & M48, (full >> 48) & 1
    elif alumode == "0010":
        full = z + s
        p, cout = (~full) & M48, (full >> 48) & 1
    else:
        raise ValueError(alumode)
    return p, p, cout


def _op(series: str, e1: str) -> str:
    """The reference writes each OPMODE twice: 7 bits on DSP48E1, the same with W = "00" in front on DSP48E2."""
    return e1 if series == "E1" else "00" + e1


# ------------------------------------------------------------------------------------------------
# exact multipliers: src/vhdl/math/mults/
# ------------------------------------------------------------------------------------------------

def mlt25x18(mlt_a: int, a_width: int, mlt_b: int, b_width: int, xseries: str) -> int:
    """mlt25x18_dsp48.vhd:84-230: dspA <= SXT(MLT_A, 30), dspB <= SXT(MLT_B, 18), OPMODE 0000101, MLT_P <= P (48 bits)."""
    series = "E1" if xseries == "OLD" else "E2"
    p, _, _ = dsp48(series, opmode=_op(series, "0000101"), a=sxt(mlt_a, a_width, 30), b=sxt(mlt_b, b_width, 18))
    return p


def _mlt_a_split2(series: str, mlt_a: int, aw: int, mlt_b: int) -> int:
    """mlt42x18_dsp48e1.vhd:82-89 / mlt44x18_dsp48e2.vhd:82-89: the A operand in two pieces on the A ports.
    aw = 42 (E1) / 44 (E2); result PWD = aw + 18 bits."""
    mw = 25 if series == "E1" else 27            # multiplier's A width
    dsp_a_m2 = sl(mlt_a, 16, 0)                  # dspA_M2(16 downto 0) <= MLT_A(16 downto 0); (29 downto 17) <= '0'
    dsp_a_m1 = sl(mlt_a, aw - 1, 17) | (rep(sl(mlt_a, aw - 1, aw - 1), 30 - mw) << mw)  # (mw-1..0) <= MLT_A(aw-1..17); rest sign
    assert aw - 17 == mw
    dsp_b_12 = mlt_b
    p_m2, pc_12, _ = dsp48(series, opmode=_op(series, "0000101"), a=dsp_a_m2, b=dsp_b_12)          # xDSP_M2
    p_m1, _, _ = dsp48(series, opmode=_op(series, "1010101"), a=dsp_a_m1, b=dsp_b_12, pcin=pc_12)  # xDSP_M1: (PCIN >> 17) + A*B
    # MLT_P(16 downto 0) <= dspP_M2(16 downto 0); MLT_P(PWD-1 downto 17) <= dspP_M1(mw+17 downto 0)
    return sl(p_m2, 16, 0) | (sl(p_m1, mw + 17, 0) << 17)


def mlt42x18_dsp48e1(mlt_a: int, mlt_b: int) -> int:
    return _mlt_a_split2("E1", mlt_a, 42, mlt_b)   # 60-bit product


def mlt44x18_dsp48e2(mlt_a: int, mlt_b: int) -> int:
    return _mlt_a_split2("E2", mlt_a, 44, mlt_b)   # 62-bit product


def _mlt_b_split2(series: str, mlt_a: int, mlt_b: int) -> int:
    """mlt35x25_dsp48e1.vhd:82-90 / mlt35x27_dsp48e2.vhd:83-91: the 35-bit MLT_A in two pieces on the B ports,
    the 25 / 27-bit MLT_B sign-extended on the A port of both slices."""
    mw = 25 if series == "E1" else 27
    dsp_a_12 = sl(mlt_b, mw - 1, 0) | (rep(sl(mlt_b, mw - 1, mw - 1), 30 - mw) << mw)
    dsp_b_m2 = sl(mlt_a, 16, 0)                  # dspB_M2(17) <= '0'
    dsp_b_m1 = sl(mlt_a, 34, 17)
    p_m2, pc_12, _ = dsp48(series, opmode=_op(series, "0000101"), a=dsp_a_12, b=dsp_b_m2)
    p_m1, _, _ = dsp48(series, opmode=_op(series, "1010101"), a=dsp_a_12, b=dsp_b_m1, pcin=pc_12)
    return sl(p_m2, 16, 0) | (sl(p_m1, mw + 17, 0) << 17)   # (59 downto 17) <= P_M1(42 downto 0) / (61 downto 17) <= P_M1(44 downto 0)


def mlt35x25_dsp48e1(mlt_a: int, mlt_b: int) -> int:
    return _mlt_b_split2("E1", mlt_a, mlt_b)       # 60 bits


def mlt35x27_dsp48e2(mlt_a: int, mlt_b: int) -> int:
    return _mlt_b_split2("E2", mlt_a, mlt_b)       # 62 bits


def _mlt_a_split3(series: str, mlt_a: int, aw: int, mlt_b: int) -> int:
    """mlt59x18_dsp48e1.vhd:88-102 / mlt61x18_dsp48e2.vhd:88-102: the A operand in three pieces (17 + 17 + 25 / 27 bits)."""
    mw = 25 if series == "E1" else 27
    assert aw - 34 == mw
    dsp_a_m3 = sl(mlt_a, 16, 0)
    dsp_a_m2 = sl(mlt_a, 33, 17)
    dsp_a_m1 = sl(mlt_a, aw - 1, 34) | (rep(sl(mlt_a, aw - 1, aw - 1), 30 - mw) << mw)
    p_m3, pc_23, _ = dsp48(series, opmode=_op(series, "0000101"), a=dsp_a_m3, b=mlt_b)
    p_m2, pc_12, _ = dsp48(series, opmode=_op(series, "1010101"), a=dsp_a_m2, b=mlt_b, pcin=pc_23)
    p_m1, _, _ = dsp48(series, opmode=_op(series, "1010101"), a=dsp_a_m1, b=mlt_b, pcin=pc_12)
    # MLT_P(16..0) <= P_MZ(16..0) (= P_M3 delayed); (33..17) <= P_M2(16..0); (PWD-1..34) <= P_M1(mw+17..0)
    return sl(p_m3, 16, 0) | (sl(p_m2, 16, 0) << 17) | (sl(p_m1, mw + 17, 0) << 34)


def mlt59x18_dsp48e1(mlt_a: int, mlt_b: int) -> int:
    return _mlt_a_split3("E1", mlt_a, 59, mlt_b)   # 77 bits


def mlt61x18_dsp48e2(mlt_a: int, mlt_b: int) -> int:
    return _mlt_a_split3("E2", mlt_a, 61, mlt_b)   # 79 bits


def _mlt_b_split3(series: str, mlt_a: int, mlt_b: int) -> int:
    """mlt52x25_dsp48e1.vhd:88-102 / mlt52x27_dsp48e2.vhd:88-102: the 52-bit MLT_A in three pieces on the B ports."""
    mw = 25 if series == "E1" else 27
    dsp_a_12 = sl(mlt_b, mw - 1, 0) | (rep(sl(mlt_b, mw - 1, mw - 1), 30 - mw) << mw)
    dsp_b_m3 = sl(mlt_a, 16, 0)
    dsp_b_m2 = sl(mlt_a, 33, 17)
    dsp_b_m1 = sl(mlt_a, 51, 34)
    p_m3, pc_23, _ = dsp48(series, opmode=_op(series, "0000101"), a=dsp_a_12, b=dsp_b_m3)
    p_m2, pc_12, _ = dsp48(series, opmode=_op(series, "1010101"), a=dsp_a_12, b=dsp_b_m2, pcin=pc_23)
    p_m1, _, _ = dsp48(series, opmode=_op(series, "1010101"), a=dsp_a_12, b=dsp_b_m1, pcin=pc_12)
    return sl(p_m3, 16, 0) | (sl(p_m2, 16, 0) << 17) | (sl(p_m1, mw + 17, 0) << 34)


def mlt52x25_dsp48e1(mlt_a: int, mlt_b: int) -> int:
    return _mlt_b_split3("E1", mlt_a, mlt_b)       # 77 bits


def mlt52x27_dsp48e2(mlt_a: int, mlt_b: int) -> int:
    return _mlt_b_split3("E2", mlt_a, mlt_b)       # 79 bits


# ------------------------------------------------------------------------------------------------
# complex-multiplier halves: src/vhdl/math/cmult/
# ------------------------------------------------------------------------------------------------

def _alumode(xalu: str) -> str:
    return {"ADD": "0000", "SUB": "0011"}[xalu]


def int_cmult18x25_dsp48(m1_aa, m1_bb, m2_aa, m2_bb, maw, mbw, xalu, xser) -> int:
    """int_cmult18x25_dsp48.vhd:108-406: xDSP_M2 (OPMODE 0000101) makes PCOUT = M2_AA * M2_BB,
    xDSP_M1 (OPMODE 0010101, ALUMODE by XALU) makes P = PCIN +/- M1_AA * M1_BB; MP_12 <= dspP_M1 (48 bits)."""
    series = "E1" if xser == "OLD" else "E2"
    _, pc_m2, _ = dsp48(series, opmode=_op(series, "0000101"), a=sxt(m2_aa, maw, 30), b=sxt(m2_bb, mbw, 18))
    p_m1, _, _ = dsp48(series, opmode=_op(series, "0010101"), alumode=_alumode(xalu), a=sxt(m1_aa, maw, 30),
                       b=sxt(m1_bb, mbw, 18), pcin=pc_m2)
    return p_m1


def _add48(series, xalu, d1_48, d2_48, **kw):
    """The closing adder of the wide regimes: A:B <= dsp1 (the M1 product), C <= dsp2, OPMODE 0110011: P = C +/- A:B."""
    return dsp48(series, use_mult="NONE", opmode=_op(series, "0110011"), alumode=_alumode(xalu), a=sl(d1_48, 47, 18),
                 b=sl(d1_48, 17, 0), c=d2_48, **kw)


def int_cmult_dbl18_dsp48(m1_aa, m1_bb, m2_aa, m2_bb, maw, mbw, xalu, xser) -> int:
    """int_cmult_dbl18_dsp48.vhd:146-360."""
    series, awd, pwd = ("E1", 42, 60) if xser == "OLD" else ("E2", 44, 62)
    mlt = mlt42x18_dsp48e1 if xser == "OLD" else mlt44x18_dsp48e2
    p_m1 = mlt(sxt(m1_aa, maw, awd), sxt(m1_bb, mbw, 18))
    p_m2 = mlt(sxt(m2_aa, maw, awd), sxt(m2_bb, mbw, 18))
    d1 = sl(p_m1, pwd - 1 - (18 - mbw), pwd - 48 - (18 - mbw))   # :174
    d2 = sl(p_m2, pwd - 1 - (18 - mbw), pwd - 48 - (18 - mbw))   # :175
    p12, _, _ = _add48(series, xalu, d1, d2)
    return sl(p12, 47 - 1 - (awd - maw), 47 - awd)               # :163


def int_cmult_dbl35_dsp48(m1_aa, m1_bb, m2_aa, m2_bb, maw, mbw, xalu, xser) -> int:
    """int_cmult_dbl35_dsp48.vhd:155-361."""
    series, pwd, bwd = ("E1", 60, 25) if xser == "OLD" else ("E2", 62, 27)
    mlt = mlt35x25_dsp48e1 if xser == "OLD" else mlt35x27_dsp48e2
    p_m1 = mlt(sxt(m1_aa, maw, 35), sxt(m1_bb, mbw, bwd))
    p_m2 = mlt(sxt(m2_aa, maw, 35), sxt(m2_bb, mbw, bwd))
    d1 = sl(p_m1, pwd - 1 - (bwd - mbw) - 1, pwd - 48 - (bwd - mbw) - 1)  # :163
    d2 = sl(p_m2, pwd - 1 - (bwd - mbw) - 1, pwd - 48 - (bwd - mbw) - 1)  # :164
    p12, _, _ = _add48(series, xalu, d1, d2)
    return sl(p12, 47 - 1 - (35 - maw), 47 - 35)                           # :168


def _close_wide(series, xalu, d1, d2, maw) -> int:
    """xDT48 (MAW < 49) / xDT96 (MAW > 48) of int_cmult_trpl18_dsp48.vhd:195-474 and int_cmult_trpl52_dsp48.vhd:193-472:
    one 48-bit adder on the sign-extended operands, or two with the carry cascading from the low to the high one
    (CARRYCASCOUT -> CARRYCASCIN, CARRYINSEL "010")."""
    if maw < 49:
        d1_dt, d2_dt = sxt(d1, maw, 48), sxt(d2, maw, 48)       # xG48: bits >= MAW take bit MAW-1
        p48, _, _ = _add48(series, xalu, d1_dt, d2_dt)
        return sl(p48, maw - 1, 0)
    assert maw > 48
    d1_lo, d2_lo = sl(d1, 47, 0), sl(d2, 47, 0)
    d1_hi, d2_hi = sxt(sl(d1, maw - 1, 48), maw - 48, 48), sxt(sl(d2, maw - 1, 48), maw - 48, 48)
    p_lo, _, cy = _add48(series, xalu, d1_lo, d2_lo)                                       # xDSP_ADD1 (CARRYCASCOUT => dspP_CY)
    p_hi, _, _ = _add48(series, xalu, d1_hi, d2_hi, carryinsel="010", carrycascin=cy)      # xDSP_ADD2
    return p_lo | (sl(p_hi, maw - 1 - 48, 0) << 48)


def int_cmult_trpl18_dsp48(m1_aa, m1_bb, m2_aa, m2_bb, maw, mbw, xalu, xser) -> int:
    """int_cmult_trpl18_dsp48.vhd:151-162 + the closing adders."""
    series, awd, pwd = ("E1", 59, 77) if xser == "OLD" else ("E2", 61, 79)
    mlt = mlt59x18_dsp48e1 if xser == "OLD" else mlt61x18_dsp48e2
    p_m1 = mlt(sxt(m1_aa, maw, awd), sxt(m1_bb, mbw, 18))      # dspA <= SXT(M_AA, AWD): cuts an operand above AWD bits
    p_m2 = mlt(sxt(m2_aa, maw, awd), sxt(m2_bb, mbw, 18))
    assert maw + mbw - 2 <= pwd - 1, "dspP(MAW+MBW-2 downto MBW-1) outside P: does not elaborate"
    d1 = sl(p_m1, maw + mbw - 2, mbw - 1)
    d2 = sl(p_m2, maw + mbw - 2, mbw - 1)
    return _close_wide(series, xalu, d1, d2, maw)


def int_cmult_trpl52_dsp48(m1_aa, m1_bb, m2_aa, m2_bb, maw, mbw, xalu, xser) -> int:
    """int_cmult_trpl52_dsp48.vhd:150-170 + the closing adders."""
    series, pwd, bwd = ("E1", 77, 25) if xser == "OLD" else ("E2", 79, 27)
    mlt = mlt52x25_dsp48e1 if xser == "OLD" else mlt52x27_dsp48e2
    p_m1 = mlt(sxt(m1_aa, maw, 52), sxt(m1_bb, mbw, bwd))
    p_m2 = mlt(sxt(m2_aa, maw, 52), sxt(m2_bb, mbw, bwd))
    assert maw + mbw - 3 <= pwd - 1
    d1 = sl(p_m1, maw + mbw - 2 - 1, mbw - 1 - 1)
    d2 = sl(p_m2, maw + mbw - 2 - 1, mbw - 1 - 1)
    return _close_wide(series, xalu, d1, d2, maw)


def int_cmult_dsp48(di_re: int, di_im: int, ww_re: int, ww_im: int, dtw: int, twd: int, xser: str):
    """int_cmult_dsp48.vhd:176-436: the generate tree and the operand routing of the RE (XALU = SUB) and IM (ADD) instances.
    Vectors in, vectors out (DTW bits).  Returns None where no branch generates (D_RE / D_IM undriven)."""
    sngl18, dbl18, trpl18, twd_dsp = (28, 45, 79, 28) if xser == "NEW" else (26, 43, 77, 26)
    if twd < 19:
        if dtw < sngl18:                                            # xGEN_SNGL :184-225
            p_re = int_cmult18x25_dsp48(di_im, ww_im, di_re, ww_re, dtw, twd, "SUB", xser)
            p_im = int_cmult18x25_dsp48(di_im, ww_re, di_re, ww_im, dtw, twd, "ADD", xser)
            return sl(p_re, dtw + twd - 2, twd - 1), sl(p_im, dtw + twd - 2, twd - 1)
        ent = None
        if sngl18 - 1 < dtw < dbl18:                                # xGEN_DBL :228-264
            ent = int_cmult_dbl18_dsp48
        elif dbl18 - 1 < dtw < trpl18:                              # xGEN_TRPL :267-303
            ent = int_cmult_trpl18_dsp48
        if ent is None:
            return None
        return (ent(di_im, ww_im, di_re, ww_re, dtw, twd, "SUB", xser),
                ent(di_im, ww_re, di_re, ww_im, dtw, twd, "ADD", xser))
    if 18 < twd < twd_dsp:
        if dtw < 19:                                                # xGEN_SNGL :309-354: the TWIDDLE goes to the A port
            p_re = int_cmult18x25_dsp48(ww_im, di_im, ww_re, di_re, twd, dtw, "SUB", xser)
            p_im = int_cmult18x25_dsp48(ww_re, di_im, ww_im, di_re, twd, dtw, "ADD", xser)
            return sl(p_re, dtw + twd - 3, twd - 2), sl(p_im, dtw + twd - 3, twd - 2)
        ent = None
        if 18 < dtw < 36:                                           # xGEN_DBL :357-393
            ent = int_cmult_dbl35_dsp48
        elif 35 < dtw < 53:                                         # xGEN_TRPL :396-433
            ent = int_cmult_trpl52_dsp48
        if ent is None:
            return None
        return (ent(di_im, ww_im, di_re, ww_re, dtw, twd, "SUB", xser),
                ent(di_im, ww_re, di_re, ww_im, dtw, twd, "ADD", xser))
    return None


# ------------------------------------------------------------------------------------------------
# adder / subtracter: src/vhdl/math/int_addsub_dsp48.vhd
# ------------------------------------------------------------------------------------------------

def int_addsub_dsp48(ia_re: int, ia_im: int, ib_re: int, ib_im: int, dspw: int, xser: str):
    """-> (OX_RE, OX_IM, OY_RE, OY_IM), each DSPW+1 bits.  C port <= IA, A:B <= IB, P = C +/- A:B (OPMODE 0110011)."""
    series = "E1" if xser == "OLD" else "E2"
    op = _op(series, "0110011")
    if dspw < 24:                                                   # xGEN_LOW :713-1018, USE_SIMD TWO24: re in lane 0, im in lane 1
        dsp_c = sxt(ia_re, dspw, 24) | (sxt(ia_im, dspw, 24) << 24)
        dsp_ab = sxt(ib_re, dspw, 24) | (sxt(ib_im, dspw, 24) << 24)
        kw = dict(use_mult="NONE", opmode=op, a=sl(dsp_ab, 47, 18), b=sl(dsp_ab, 17, 0), c=dsp_c, use_simd="TWO24")
        p_xx, _, _ = dsp48(series, alumode="0000", **kw)
        p_yy, _, _ = dsp48(series, alumode="0011", **kw)
        return sl(p_xx, dspw, 0), sl(p_xx, dspw + 24, 24), sl(p_yy, dspw, 0), sl(p_yy, dspw + 24, 24)
    if dspw < 48:                                                   # xGEN_HIGH :112-710: one slice per output
        out = []
        for alumode in ("0000", "0011"):
            for ia, ib in ((ia_re, ib_re), (ia_im, ib_im)):
                dsp_b = sl(ib, 17, 0)
                dsp_a = sxt(sl(ib, dspw - 1, 18), dspw - 18, 30)    # xFOR_A: bits >= DSPW-18 take IB(DSPW-1)
                dsp_c = sxt(ia, dspw, 48)                           # xFOR_C
                p, _, _ = dsp48(series, use_mult="NONE", opmode=op, alumode=alumode, a=dsp_a, b=dsp_b, c=dsp_c)
                out.append(sl(p, dspw, 0))
        return tuple(out)
    out = []                                                        # xGEN_DBL :1021-2190: 96-bit operands on two slices
    for alumode in ("0000", "0011"):
        for ia, ib in ((ia_re, ib_re), (ia_im, ib_im)):
            a96, b96 = sxt(ia, dspw, 96), sxt(ib, dspw, 96)
            p1, _, cy = dsp48(series, use_mult="NONE", opmode=op, alumode=alumode, a=sl(b96, 47, 18), b=sl(b96, 17, 0),
                              c=sl(a96, 47, 0))
            p2, _, _ = dsp48(series, use_mult="NONE", opmode=op, alumode=alumode, a=sl(b96, 95, 66), b=sl(b96, 65, 48),
                             c=sl(a96, 95, 48), carryinsel="010", carrycascin=cy)
            out.append(p1 | (sl(p2, dspw - 48, 0) << 48))           # OX(47..0) <= P1; OX(DSPW..48) <= P2(DSPW-48..0)
    return tuple(out)


# ------------------------------------------------------------------------------------------------
# Taylor twiddle correction: src/vhdl/twiddle/row_twiddle_tay.vhd
# ------------------------------------------------------------------------------------------------

def row_twiddle_tay(rom_ww: int, rom_cnt: int, awd: int, xser: str, ii: int, use_mlt: bool = False):
    """-> (rom_re, rom_im), AWD bits each.  rom_ww: 2*AWD bits (low half = re, high half = im of rom_twiddle_int's
    ww_rom, rom_twiddle_int.vhd:171-184,229-244); rom_cnt: ii+1 bits."""
    series = "E1" if xser == "OLD" else "E2"
    xshift = 23 if xser == "OLD" else 21                                     # find_widthA :123-132
    mathpi = _vhdl_integer(math.pi * 2.0 ** (13 - ii - (0 if xser == "OLD" else 2)))   # const_pi :134-148
    cnt_exp = sl(rom_cnt, ii, 0)                                             # cnt_exp(7 downto ii+1) <= '0' :199-202
    if not use_mlt:
        mpi = vec(mathpi * cnt_exp, 16)                                      # rom_pi(jj) = conv_std_logic_vector(MATHPI*jj, 16); mpi(23..16) = 0
    else:
        mpi = vec(vec(mathpi, 16) * cnt_exp, 24)                             # unsigned(std_pi) * unsigned(cnt_exp): 16 x 8 -> 24 bits
    mpx = sl(mpi, 17, 1)                                                     # mpx <= '0' & mpi(17 downto 1) :247
    sin_aa = sxt(sl(rom_ww, awd - 1, 0), awd, 30)                            # :250-257
    cos_aa = sxt(sl(rom_ww, 2 * awd - 1, awd), awd, 30)
    cos_cc = sxt(sl(cos_aa, awd - 1, 0), awd, 48 - xshift) << xshift         # :260-268: value << XSHIFT, zeros below, sign above
    sin_cc = sxt(sl(sin_aa, awd - 1, 0), awd, 48 - xshift) << xshift
    op = _op(series, "0110101")                                              # Z = C, X = Y = M
    cos_prod, _, _ = dsp48(series, opmode=op, alumode="0011", a=sin_aa, b=mpx, c=cos_cc)   # MULT_ADD: ALUMODE(1..0) = "11" :304,448
    sin_prod, _, _ = dsp48(series, opmode=op, alumode="0000", a=cos_aa, b=mpx, c=sin_cc)   # MULT_SUB :374,520
    cos_pdt = sl(cos_prod, 47, xshift - 1)                                   # 49-XSHIFT bits :195-196
    sin_pdt = sl(sin_prod, 47, xshift - 1)
    wr = 48 - xshift
    cos_rnd = vec(sl(cos_pdt, 48 - xshift, 1) + (cos_pdt & 1), wr)           # pr_rnd :178-193
    sin_rnd = vec(sl(sin_pdt, 48 - xshift, 1) + (sin_pdt & 1), wr)
    return sl(sin_rnd, awd - 1, 0), sl(cos_rnd, awd - 1, 0)                  # rom_re <= sin_rnd, rom_im <= cos_rnd :174-175


def _vhdl_integer(v: float) -> int:
    """VHDL INTEGER(real): round to nearest."""
    return int(math.floor(v + 0.5))


# ------------------------------------------------------------------------------------------------
# butterflies: src/vhdl/fft/int_dif2_fly.vhd, int_dit2_fly.vhd -- the adder, the rounding process, the STAGE 1
# swap / negate process and the multiplier wired as the architectures read, on vectors
# ------------------------------------------------------------------------------------------------

def _rnd_proc(v: int, dtw: int) -> int:
    """pr_rnd: out <= v(DTW downto 1) [+ '1' when v(0) = '1'], DTW bits (int_dif2_fly.vhd:194-218, int_dit2_fly.vhd:192-216)."""
    return vec(sl(v, dtw, 1) + (v & 1), dtw)


def _inv_proc(x: int, w: int) -> int:
    """pr_inv: not(x) + '1' when the sign bit is '0', not(x) otherwise (int_dif2_fly.vhd:299-303, int_dit2_fly.vhd:271-275)."""
    nx = (~x) & ((1 << w) - 1)
    return vec(nx + 1, w) if not (x >> (w - 1)) & 1 else nx


def int_dif2_fly(ia_re, ia_im, ib_re, ib_im, ww_re, ww_im, *, stage, scale, dtw, tfw, rndmode, xser, dt_sw=0):
    """-> (OA_RE, OA_IM, OB_RE, OB_IM), DTW-SCALE+1 bits each.  dt_sw: the STAGE 1 toggle (odd butterflies)."""
    wo = dtw - scale + 1
    if rndmode == 0 and scale == 1:      # xTRUNC :144-164: DSPW = DTW-1 on IA(DTW-1 downto 1)
        ad_re, ad_im, su_re, su_im = int_addsub_dsp48(sl(ia_re, dtw - 1, 1), sl(ia_im, dtw - 1, 1), sl(ib_re, dtw - 1, 1),
                                                      sl(ib_im, dtw - 1, 1), dtw - 1, xser)
    elif scale == 1:                     # xROUND :167-219
        r = int_addsub_dsp48(ia_re, ia_im, ib_re, ib_im, dtw, xser)
        ad_re, ad_im, su_re, su_im = (_rnd_proc(v, dtw) for v in r)
    else:                                # xUNSCALED :221-241
        ad_re, ad_im, su_re, su_im = int_addsub_dsp48(ia_re, ia_im, ib_re, ib_im, dtw, xser)
    if stage == 0:                       # xST0 :245-255
        return ad_re, ad_im, su_re, su_im
    if stage == 1:                       # xST1 :259-318
        if dt_sw == 0:
            return ad_re, ad_im, su_re, su_im
        return ad_re, ad_im, su_im, _inv_proc(su_re, wo)
    o = int_cmult_dsp48(su_re, su_im, ww_re, ww_im, wo, tfw, xser)   # xSTn :322-373, DTW => DTW+1-SCALE
    return None if o is None else (ad_re, ad_im, o[0], o[1])


def int_dit2_fly(ia_re, ia_im, ib_re, ib_im, ww_re, ww_im, *, stage, scale, dtw, tfw, rndmode, xser, dt_sw=0):
    """-> (OA_RE, OA_IM, OB_RE, OB_IM), DTW-SCALE+1 bits each."""
    if stage == 0:                       # xST0 :221-230
        bw_re, bw_im = ib_re, ib_im
    elif stage == 1:                     # xST1 :234-286
        if dt_sw == 0:
            bw_re, bw_im = ib_re, ib_im
        else:
            bw_re, bw_im = _inv_proc(ib_im, dtw), ib_re
    else:                                # xSTn :290-325: DI_RE => IB_IM, DI_IM => IB_RE, DO_RE => bw_im, DO_IM => bw_re
        o = int_cmult_dsp48(ib_im, ib_re, ww_re, ww_im, dtw, tfw, xser)
        if o is None:
            return None
        bw_im, bw_re = o
    az_re, az_im = ia_re, ia_im
    if scale == 0 or rndmode == 0:       # xUNSCALED :142-162: DSPW = DTW-SCALE on (DTW-1 downto SCALE)
        return int_addsub_dsp48(sl(az_re, dtw - 1, scale), sl(az_im, dtw - 1, scale), sl(bw_re, dtw - 1, scale),
                                sl(bw_im, dtw - 1, scale), dtw - scale, xser)
    r = int_addsub_dsp48(az_re, az_im, bw_re, bw_im, dtw, xser)      # xROUND :164-217
    return tuple(_rnd_proc(v, dtw) for v in r)
