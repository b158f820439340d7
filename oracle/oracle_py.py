"""Independent pure-Python twin of oracle/intfft_oracle.c (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: the reference ships no golden vectors for this path; this twin
exists so that two restatements written in different styles -- C with arithmetic shifts on
__int128 (intfft_oracle.c) and Python big-ints with literal std_logic_vector bit slicing (here) --
can be compared bit for bit, and so that tests/golden/ can be regenerated (tests/golden/make_golden.py).

Style rule for this file: follow the RTL *slices* literally (P(hi downto lo)), never the shift
shorthand used by the C file, so that a misreading in one of them shows up as a mismatch.
All citations are relative to the reference repository (hukenovs/intfftk).
"""
from __future__ import annotations

import math
from functools import lru_cache

FWD, INV, PAIR = 0, 1, 2
NATURAL, BITREV, HALVES, BITREV_LANES = 0, 1, 2, 3


def sgn(v: int, w: int) -> int:
    """Signed value of the low w bits (a std_logic_vector(w-1 downto 0) read as signed)."""
    v &= (1 << w) - 1
    return v - (1 << w) if v >> (w - 1) else v


def bits(v: int, hi: int, lo: int) -> int:
    """v(hi downto lo) of a two's-complement vector, returned as a signed (hi-lo+1)-bit number."""
    return sgn(v >> lo, hi - lo + 1)


def bitrev(v: int, n: int) -> int:
    return int(format(v, "0%db" % n)[::-1], 2) if n else 0


# ------------------------------------------------------------------------------------------------
# complex multiplier
# ------------------------------------------------------------------------------------------------

def cmult_regime(w: int, t: int, new: bool) -> str | None:
    """int_cmult_dsp48.vhd:182-434 generate conditions."""
    l18, h18, t18, twd = (28, 45, 79, 28) if new else (26, 43, 77, 26)
    if t < 19:
        if w < l18:
            return "sngl"
        if l18 - 1 < w < h18:
            # product slice P(pwd-1-(18-t) downto pwd-48-(18-t)): its low index t-4 (NEW) / t-6 (OLD) must exist
            return "dbl18" if (t - 4 if new else t - 6) >= 0 else None
        if h18 - 1 < w < t18:
            # the product slice P(MAW+MBW-2 downto MBW-1) (int_cmult_trpl18_dsp48.vhd:151-152) must lie inside the PWD = 79 / 77
            # bits of P, or it does not elaborate
            return "trpl18" if w + t <= t18 + 1 else None
        return None
    if 18 < t < twd:
        if w < 19:
            return "sngl25"
        if 18 < w < 36:
            return "dbl35"
        if 35 < w < 53:
            return "trpl52"
    return None


def _half(m2: int, m1: int, sub: bool, regime: str, w: int, t: int, new: bool) -> int:
    """One int_cmult*_dsp48 instance: MP_12 = f(M2_AA*M2_BB, M1_AA*M1_BB)."""
    if regime == "sngl":  # int_cmult18x25_dsp48: 48-bit P = M2 -/+ M1; D = P(w+t-2 downto t-1)
        p = sgn(m2 - m1 if sub else m2 + m1, 48)
        return bits(p, w + t - 2, t - 1)
    if regime == "sngl25":  # int_cmult_dsp48.vhd:316-317  P(w+t-3 downto t-2)
        p = sgn(m2 - m1 if sub else m2 + m1, 48)
        return bits(p, w + t - 3, t - 2)
    if regime == "dbl18":  # int_cmult_dbl18_dsp48.vhd:104-128,163,174-175
        awd, pwd = (44, 62) if new else (42, 60)
        p1, p2 = sgn(m1, pwd), sgn(m2, pwd)
        d1 = bits(p1, pwd - 1 - (18 - t), pwd - 48 - (18 - t))
        d2 = bits(p2, pwd - 1 - (18 - t), pwd - 48 - (18 - t))
        p12 = sgn(d2 - d1 if sub else d2 + d1, 48)  # P = C -/+ A:B
        return bits(p12, 47 - 1 - (awd - w), 47 - awd)
    if regime == "dbl35":  # int_cmult_dbl35_dsp48.vhd:102-127,163-168
        pwd, bwd = (62, 27) if new else (60, 25)
        p1, p2 = sgn(m1, pwd), sgn(m2, pwd)
        d1 = bits(p1, pwd - 1 - (bwd - t) - 1, pwd - 48 - (bwd - t) - 1)
        d2 = bits(p2, pwd - 1 - (bwd - t) - 1, pwd - 48 - (bwd - t) - 1)
        p12 = sgn(d2 - d1 if sub else d2 + d1, 48)
        return bits(p12, 47 - 1 - (35 - w), 47 - 35)
    if regime == "trpl18":  # int_cmult_trpl18_dsp48.vhd:151-155
        pwd = 79 if new else 77
        d1 = bits(sgn(m1, pwd), w + t - 2, t - 1)
        d2 = bits(sgn(m2, pwd), w + t - 2, t - 1)
        return sgn(d2 - d1 if sub else d2 + d1, w)
    if regime == "trpl52":  # int_cmult_trpl52_dsp48.vhd:166-170
        pwd = 79 if new else 77
        d1 = bits(sgn(m1, pwd), w + t - 2 - 1, t - 1 - 1)
        d2 = bits(sgn(m2, pwd), w + t - 2 - 1, t - 1 - 1)
        return sgn(d2 - d1 if sub else d2 + d1, w)
    raise ValueError(regime)


def cmult(d_re: int, d_im: int, wr: int, wi: int, w: int, t: int, new: bool = True):
    """int_cmult_dsp48: RE instance M1 = (DI_IM, WW_IM), M2 = (DI_RE, WW_RE), SUB;
    IM instance M1 = (DI_IM, WW_RE), M2 = (DI_RE, WW_IM), ADD (int_cmult_dsp48.vhd:192-224)."""
    regime = cmult_regime(w, t, new)
    if regime is None:
        raise ValueError("unsupported widths w=%d t=%d" % (w, t))
    if regime == "trpl18":  # dspA <= SXT(M_AA, AWD), AWD = 61 / 59 (int_cmult_trpl18_dsp48.vhd:161-162): a longer operand is cut
        awd = 61 if new else 59
        if w > awd:
            d_re, d_im = sgn(d_re, awd), sgn(d_im, awd)
    re = _half(d_re * wr, d_im * wi, True, regime, w, t, new)
    im = _half(d_re * wi, d_im * wr, False, regime, w, t, new)
    return re, im


# ------------------------------------------------------------------------------------------------
# twiddles
# ------------------------------------------------------------------------------------------------

def _integer(v: float) -> int:
    """VHDL INTEGER(real) = round to nearest; flag anything close to a tie."""
    r = math.floor(v + 0.5)
    frac = v - math.floor(v)
    if abs(frac - 0.5) < 1e-7:
        raise ArithmeticError("near-tie in twiddle rounding: %r" % v)
    return int(r)


def _rom(depth: int, t: int):
    """rom_twiddle_int.vhd:135-159: (re, im) of the 2^depth quarter-wave entries."""
    mg = 2.0 ** (t - 1) - 1.0 if t < 18 else 2.0 ** (t - 2) - 1.0
    out = []
    for ii in range(2 ** depth):
        pi_std = (float(ii) * math.pi) / (2.0 ** (depth + 1))
        out.append((_integer(mg * math.cos(pi_std)), _integer(mg * math.sin(-pi_std))))
    return out


@lru_cache(maxsize=None)
def twiddles(stage: int, t: int, new: bool = True):
    """Twiddle stream of one butterfly stage as a list of 2^stage (re, im)."""
    if stage == 0:
        return [_rom(0, t)[0]]
    depth = stage - 1 if stage < 11 else 9  # find_depth rom_twiddle_int.vhd:118-129
    rom = _rom(depth, t)
    out = []
    if stage >= 11:
        ii = stage - 11
        xshift = 21 if new else 23
        mathpi = _integer(math.pi * 2.0 ** (13 - ii - (2 if new else 0)))
    for cnt in range(2 ** stage):
        div = (cnt >> (stage - 1)) & 1
        addr = cnt & (2 ** (stage - 1) - 1)
        ram = rom[addr] if stage < 11 else rom[addr >> (stage - 10)]
        if div == 0:  # pr_ww rom_twiddle_int.vhd:171-184
            re, im = ram
        else:
            re, im = ram[1], sgn(~ram[0] + 1, t)
        if stage < 11:
            out.append((re, im))
            continue
        count = addr & (2 ** (stage - 10) - 1)
        mpi = (mathpi * count) & 0xFFFF            # row_twiddle_tay.vhd:208-221
        mpx = (mpi >> 1) & 0x1FFFF                 # '0' & mpi(17 downto 1)
        sin_aa, cos_aa = re, im                    # :250-251 (names are swapped in the RTL)
        cos_cc, sin_cc = cos_aa << xshift, sin_aa << xshift
        cos_prod = sgn(cos_cc - sin_aa * mpx, 48)  # MULT_ADD, ALUMODE 0011: C - A*B
        sin_prod = sgn(sin_cc + cos_aa * mpx, 48)  # MULT_SUB, ALUMODE 0000: C + A*B
        cos_pdt = bits(cos_prod, 47, xshift - 1)
        sin_pdt = bits(sin_prod, 47, xshift - 1)
        cos_rnd = bits(cos_pdt, 48 - xshift, 1) + (cos_pdt & 1)
        sin_rnd = bits(sin_pdt, 48 - xshift, 1) + (sin_pdt & 1)
        out.append((sgn(sin_rnd, t), sgn(cos_rnd, t)))  # rom_re <= sin_rnd, rom_im <= cos_rnd
    return out


# ------------------------------------------------------------------------------------------------
# butterflies
# ------------------------------------------------------------------------------------------------

def _rnd(v: int, w_in: int) -> int:
    """pr_rnd int_dif2_fly.vhd:196-217: v is (w_in+1)-bit; out = v(w_in downto 1) + v(0), w_in bits."""
    return sgn(bits(v, w_in, 1) + (v & 1), w_in)


def _negq(x: int, w: int) -> int:
    """int_dif2_fly.vhd:297-303: not(x)+1 when the sign bit is 0, not(x) otherwise."""
    return sgn(~x + 1, w) if x >= 0 else sgn(~x, w)


def dif_fly(a, b, ww, stage, dtw, t, scale, rnd, odd, new=True):
    (are, aim), (bre, bim) = a, b
    wo = dtw - scale + 1
    if scale and not rnd:  # int_addsub on IA(DTW-1 downto 1)
        are, aim, bre, bim = (bits(v, dtw - 1, 1) for v in (are, aim, bre, bim))
        s = (are + bre, aim + bim)
        d = (are - bre, aim - bim)
    elif scale:
        s = (_rnd(are + bre, dtw), _rnd(aim + bim, dtw))
        d = (_rnd(are - bre, dtw), _rnd(aim - bim, dtw))
    else:
        s = (are + bre, aim + bim)
        d = (are - bre, aim - bim)
    if stage == 0:
        y = d
    elif stage == 1:
        y = d if not odd else (d[1], _negq(d[0], wo))
    else:
        y = cmult(d[0], d[1], ww[0], ww[1], wo, t, new)
    return s, y


def dit_fly(a, b, ww, stage, dtw, t, scale, rnd, odd, new=True):
    (are, aim), (bre, bim) = a, b
    if stage == 0:
        bw = (bre, bim)
    elif stage == 1:
        bw = (bre, bim) if not odd else (_negq(bim, dtw), bre)
    else:  # DI_RE <= IB_IM, DI_IM <= IB_RE; DO_RE => bw_im, DO_IM => bw_re (int_dit2_fly.vhd:304-322)
        o_re, o_im = cmult(bim, bre, ww[0], ww[1], dtw, t, new)
        bw = (o_im, o_re)
    if scale and not rnd:
        are, aim = bits(are, dtw - 1, 1), bits(aim, dtw - 1, 1)
        bre2, bim2 = bits(bw[0], dtw - 1, 1), bits(bw[1], dtw - 1, 1)
        return (are + bre2, aim + bim2), (are - bre2, aim - bim2)
    if scale:
        return ((_rnd(are + bw[0], dtw), _rnd(aim + bw[1], dtw)),
                (_rnd(are - bw[0], dtw), _rnd(aim - bw[1], dtw)))
    return (are + bw[0], aim + bw[1]), (are - bw[0], aim - bw[1])


# ------------------------------------------------------------------------------------------------
# stream-form transforms (fn_radix2.m dataflow)
# ------------------------------------------------------------------------------------------------

def _rev2rdx(ia, ib, cnti):
    """fn_rev2rdx / fn_rdx2rev body (fn_radix2.m:51-89) with block length CNTi, 0-based."""
    half = len(ia)
    cntj = half // cnti
    oa, ob = [None] * half, [None] * half
    for i in range(cnti):
        for j in range(1, cntj + 1):
            stp = 2 * (math.ceil(j / 2) - 1) * cnti
            src = ia if j % 2 == 1 else ib
            oa[i + cnti * (j - 1)] = src[i + stp]
            ob[i + cnti * (j - 1)] = src[i + stp + cnti]
    return oa, ob


def _zext(v, w):
    return (v[0] & ((1 << w) - 1), v[1] & ((1 << w) - 1))


def fft_dif(x, log2n, dw, t, fmt=0, rnd=0, new=True, use_fly=1):
    """int_fftNk: natural x[0..N) -> v (bit-reversed sequence, v[2i] = lane0[i], v[2i+1] = lane1[i])."""
    n = 1 << log2n
    scale = 1 - fmt
    ta = [(sgn(r, dw), sgn(i, dw)) for r, i in x[: n // 2]]
    tb = [(sgn(r, dw), sgn(i, dw)) for r, i in x[n // 2:]]
    for ii in range(log2n):
        stage = log2n - ii - 1
        dtw = dw + ii * fmt
        tw = twiddles(stage, t, new) if stage >= 2 else None
        oa, ob = [], []
        for q in range(n // 2):
            if not use_fly:
                xa, xb = (ta[q], tb[q]) if not fmt else (_zext(ta[q], dtw), _zext(tb[q], dtw))
            else:
                ww = tw[q % (1 << stage)] if tw else (0, 0)
                xa, xb = dif_fly(ta[q], tb[q], ww, stage, dtw, t, scale, rnd, q & 1, new)
            oa.append(xa)
            ob.append(xb)
        if ii < log2n - 1:
            ta, tb = _rev2rdx(oa, ob, (n // 2) >> (ii + 1))
    v = []
    for q in range(n // 2):
        v += [oa[q], ob[q]]
    return v


def ifft_dit(v, log2n, dw, t, fmt=0, rnd=0, new=True, use_fly=1):
    """int_ifftNk: bit-reversed pair stream v -> natural x."""
    n = 1 << log2n
    scale = 1 - fmt
    ta = [(sgn(v[2 * q][0], dw), sgn(v[2 * q][1], dw)) for q in range(n // 2)]
    tb = [(sgn(v[2 * q + 1][0], dw), sgn(v[2 * q + 1][1], dw)) for q in range(n // 2)]
    for ii in range(log2n):
        stage = ii
        dtw = dw + ii * fmt
        tw = twiddles(stage, t, new) if stage >= 2 else None
        oa, ob = [], []
        for q in range(n // 2):
            if not use_fly:
                xa, xb = (ta[q], tb[q]) if not fmt else (_zext(ta[q], dtw), _zext(tb[q], dtw))
            else:
                ww = tw[q % (1 << stage)] if tw else (0, 0)
                xa, xb = dit_fly(ta[q], tb[q], ww, stage, dtw, t, scale, rnd, q & 1, new)
            oa.append(xa)
            ob.append(xb)
        if ii < log2n - 1:
            ta, tb = _rev2rdx(oa, ob, 1 << ii)
    return oa + ob


def order_index(order, log2n, m):
    half = 1 << (log2n - 1)
    if order == NATURAL:
        return m
    if order == BITREV:
        return bitrev(m, log2n)
    if order == HALVES:
        return (m >> 1) + (m & 1) * half
    if order == BITREV_LANES:
        return bitrev(2 * (m % half) + m // half, log2n)
    raise ValueError(order)


def execute(frame, log2n, dw, t, fmt=0, rnd=0, new=True, direction=FWD, in_order=NATURAL,
            out_order=NATURAL, use_fly=1):
    """One frame with the ABI semantics of include/intfft.h; frame = list of (re, im)."""
    n = 1 << log2n
    if direction in (FWD, PAIR):
        x = [None] * n
        for m in range(n):
            x[order_index(in_order, log2n, m)] = frame[m]
        v = fft_dif(x, log2n, dw, t, fmt, rnd, new, use_fly)
        if direction == FWD:
            return [v[bitrev(order_index(out_order, log2n, m), log2n)] for m in range(n)]
        y = ifft_dit(v, log2n, dw + fmt * log2n, t, fmt, rnd, new, use_fly)
        return [y[order_index(out_order, log2n, m)] for m in range(n)]
    v = [None] * n
    for m in range(n):
        v[bitrev(order_index(in_order, log2n, m), log2n)] = frame[m]
    y = ifft_dit(v, log2n, dw, t, fmt, rnd, new, use_fly)
    return [y[order_index(out_order, log2n, m)] for m in range(n)]


# ------------------------------------------------------------------------------------------------
# N > 512K: the "2D-FFT scheme" (int_fftNk.vhd:11-13) -- this project's extension, structural definition
# (see the comment block in intfft_oracle.c): columns -> inter-pass twiddle -> rows with the 1-D cores above
# ------------------------------------------------------------------------------------------------

_COS_C = [float.fromhex(h) for h in ("-0x1.0000000000000p-1", "0x1.5555555555555p-5", "-0x1.6c16c16c16c17p-10", "0x1.a01a01a01a01ap-16",
                                     "-0x1.27e4fb7789f5cp-22", "0x1.1eed8eff8d898p-29", "-0x1.93974a8c07c9dp-37", "0x1.ae7f3e733b81fp-45")]
_SIN_C = [float.fromhex(h) for h in ("-0x1.5555555555555p-3", "0x1.1111111111111p-7", "-0x1.a01a01a01a01ap-13", "0x1.71de3a556c734p-19",
                                     "-0x1.ae64567f544e4p-26", "0x1.6124613a86d09p-33", "-0x1.ae7f3e733b81fp-41", "0x1.952c77030ad4ap-49")]


def twiddle_2d(log2n: int, t: int, m: int):
    """W_N^m, m in [0, N): quarter-wave ROM formula of rom_twiddle_int.vhd:143-152 at full depth, quadrants by
    (re, im) <- (im, -re) (:177-183).  cos / sin are the specified sequence of separately rounded double operations
    (octant reduction, Taylor polynomials to z^8 in Horner form; CPython floats are IEEE doubles and never fuse)."""
    quarter = 1 << (log2n - 2)
    a, q = m % quarter, (m // quarter) & 3
    swap = a > quarter // 2
    b = quarter - a if swap else a
    scale = 3.14159265358979323846
    for _ in range(log2n - 1):
        scale = scale * 0.5
    x = float(b) * scale
    z = x * x
    pc, ps = _COS_C[7], _SIN_C[7]
    for k in range(6, -1, -1):
        pc = _COS_C[k] + z * pc
        ps = _SIN_C[k] + z * ps
    cosx = 1.0 + z * pc
    sinx = x + (x * z) * ps
    mg = 2.0 ** (t - 1) - 1.0 if t < 18 else 2.0 ** (t - 2) - 1.0
    vc = mg * (sinx if swap else cosx) + 0.5
    vs = mg * (cosx if swap else sinx) + 0.5
    re, im = int(vc), -int(vs)
    for _ in range(q):
        re, im = im, sgn(~re + 1, t)
    return re, im


def fft2d(x, log2n, l1, dw, t, fmt=0, rnd=0, new=True):
    """natural x -> v with v[j] = X[bitrev_L(j)] (the layout fft_dif returns)."""
    l2 = log2n - l1
    n1, n2, n = 1 << l1, 1 << l2, 1 << log2n
    a = [[None] * n2 for _ in range(n1)]  # a[k1][n2]
    for i2 in range(n2):
        col = fft_dif([x[i1 * n2 + i2] for i1 in range(n1)], l1, dw, t, fmt, rnd, new)
        for j1 in range(n1):
            a[bitrev(j1, l1)][i2] = col[j1]
    w1 = dw + fmt * l1
    v = [None] * n
    for k1 in range(n1):
        row = [cmult(a[k1][i2][0], a[k1][i2][1], *twiddle_2d(log2n, t, (k1 * i2) % n), w1, t, new) for i2 in range(n2)]
        res = fft_dif(row, l2, w1, t, fmt, rnd, new)
        for j2 in range(n2):
            v[bitrev(k1, l1) * n2 + j2] = res[j2]
    return v


def ifft2d(v, log2n, l1, dw, t, fmt=0, rnd=0, new=True):
    """v[j] = X[bitrev_L(j)] -> natural x."""
    l2 = log2n - l1
    n1, n2, n = 1 << l1, 1 << l2, 1 << log2n
    w = dw + fmt * l2
    d = []
    for j1 in range(n1):
        row = ifft_dit(v[j1 * n2:(j1 + 1) * n2], l2, dw, t, fmt, rnd, new)
        k1 = bitrev(j1, l1)
        out = []
        for i2 in range(n2):  # swapped feed: DI_RE <- B.im, DI_IM <- B.re; DO_RE -> T.im, DO_IM -> T.re
            o_re, o_im = cmult(row[i2][1], row[i2][0], *twiddle_2d(log2n, t, (k1 * i2) % n), w, t, new)
            out.append((o_im, o_re))
        d.append(out)
    x = [None] * n
    for i2 in range(n2):
        res = ifft_dit([d[j1][i2] for j1 in range(n1)], l1, w, t, fmt, rnd, new)
        for i1 in range(n1):
            x[i1 * n2 + i2] = res[i1]
    return x


def execute_2d(frame, log2n, l1, dw, t, fmt=0, rnd=0, new=True, direction=FWD, in_order=NATURAL, out_order=NATURAL):
    n = 1 << log2n
    if direction in (FWD, PAIR):
        x = [None] * n
        for m in range(n):
            x[order_index(in_order, log2n, m)] = (sgn(frame[m][0], dw), sgn(frame[m][1], dw))
        v = fft2d(x, log2n, l1, dw, t, fmt, rnd, new)
        if direction == FWD:
            return [v[bitrev(order_index(out_order, log2n, m), log2n)] for m in range(n)]
        y = ifft2d(v, log2n, l1, dw + fmt * log2n, t, fmt, rnd, new)
        return [y[order_index(out_order, log2n, m)] for m in range(n)]
    v = [None] * n
    for m in range(n):
        v[bitrev(order_index(in_order, log2n, m), log2n)] = (sgn(frame[m][0], dw), sgn(frame[m][1], dw))
    y = ifft2d(v, log2n, l1, dw, t, fmt, rnd, new)
    return [y[order_index(out_order, log2n, m)] for m in range(n)]


# ------------------------------------------------------------------------------------------------
# double-precision restatement of math/fn_radix2.m (the reference's own model), for the
# ordering check against numpy.fft
# ------------------------------------------------------------------------------------------------

def fn_radix2_float(din, n, mode):
    """fn_fft_dif (fn_radix2.m:152-190) / fn_fft_dit (:193-232) in double precision."""
    import numpy as np

    nl = int(round(math.log2(n)))
    half = n // 2
    idx = [bitrev(i, nl) for i in range(n)]
    if mode == "FWD":
        ww = [complex(math.cos(i * 2 * math.pi / n), -math.sin(i * 2 * math.pi / n)) for i in range(half)]
        ta, tb = list(din[:half]), list(din[half:])
        for i in range(1, nl + 1):
            cnt = 2 ** (i - 1)
            stp = half // cnt
            wx = [ww[(p % stp) * cnt] for p in range(half)]  # fn_twiddleN_dif :109-117
            oa = [a + b for a, b in zip(ta, tb)]
            ob = [(a - b) * w for a, b, w in zip(ta, tb, wx)]
            if i < nl:
                ta, tb = _rev2rdx(oa, ob, half // (2 ** i))
        oo = []
        for q in range(half):
            oo += [oa[q], ob[q]]
        return np.array([oo[idx[k]] for k in range(n)])  # bitrevorder
    ww = [complex(math.cos(i * 2 * math.pi / n), math.sin(i * 2 * math.pi / n)) for i in range(half)]
    dx = [din[idx[k]] for k in range(n)]
    ta, tb = dx[0::2], dx[1::2]
    for i in range(1, nl + 1):
        cnt = 2 ** (nl - i)
        stp = half // cnt
        wx = [ww[(p % stp) * cnt] for p in range(half)]  # fn_twiddleN_dit :119-128
        oa = [a + b * w for a, b, w in zip(ta, tb, wx)]
        ob = [a - b * w for a, b, w in zip(ta, tb, wx)]
        if i < nl:
            ta, tb = _rev2rdx(oa, ob, half // (2 ** (nl - i)))
    return np.array(oa + ob)
