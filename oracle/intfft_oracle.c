/*
 * intfft_oracle.c -- CPU restatement of the intfftk fixed-point radix-2 FFT/IFFT hot path.
 *
 * TEST INFRASTRUCTURE ONLY (checker for tests/, smoke() and bench.py's cpu_baseline leg).
 * PARITY UNPINNED by the reference's own tests -- see intfft_oracle.h and oracle/README.md.
 *
 * All paths cited below are relative to the reference repository (hukenovs/intfftk).
 * Values are two's-complement integers held in int64_t; every RTL signal is a fixed-width
 * std_logic_vector, so results are wrapped to the signal width with orc_wrap().
 */
#include "intfft_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef __int128 i128;

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* interpret the low w bits of v as a signed number (std_logic_vector(w-1 downto 0), signed) */
int64_t orc_wrap(int64_t v, int w)
{
    if (w >= 64) return v;
    return (int64_t)((uint64_t)v << (64 - w)) >> (64 - w);
}

static inline int64_t wrap128(i128 v, int w)
{
    return orc_wrap((int64_t)(uint64_t)(unsigned __int128)v, w);
}

/* round-half-up of v/2: int_dif2_fly.vhd:196-217, row_twiddle_tay.vhd:176-196 */
static inline int64_t rhu2(int64_t v) { return (v >> 1) + (v & 1); }
/* ... of the exact (w + 1)-bit sum / difference of two w-bit values, w up to 64 (the sum of two 64-bit operands does not fit int64: found by
 * tools/fuzz_soak.py against the Python twin and the GPU, DATA_WIDTH = 64 with RNDMODE = 1); the low 64 bits are what orc_wrap keeps */
static inline int64_t rhu2_sum(int64_t a, int64_t b) { return (int64_t)(((__int128)a + (__int128)b + 1) >> 1); }
static inline int64_t rhu2_diff(int64_t a, int64_t b) { return (int64_t)(((__int128)a - (__int128)b + 1) >> 1); }

static inline size_t bitrev(size_t v, int bits)
{
    size_t r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* complex multiplier: int_cmult_dsp48.vhd:182-434                                            */
/* ------------------------------------------------------------------------------------------ */

int orc_cmult_regime(int w, int t, int xser)
{
    const int L = xser ? 28 : 26;  /* find_sngl_18  int_cmult_dsp48.vhd:115-127 */
    const int H = xser ? 45 : 43;  /* find_dbl_18   :129-141 */
    const int T = xser ? 79 : 77;  /* find_trpl_18  :143-155 */
    const int TD = xser ? 28 : 26; /* find_twd_25   :161-173 */
    if (t < 19) {                  /* xGEN_TWD18 :182 */
        if (w < L) return ORC_SNGL;               /* :184 */
        if (w < H)                                /* :228; the product slice starts at bit t-4 (NEW) / t-6 (OLD): */
            return (xser ? t - 4 : t - 6) >= 0 ? ORC_DBL18 : ORC_UNSUPPORTED; /* a negative index does not elaborate */
        if (w < T)                                /* :267; the product slice P(MAW+MBW-2 downto MBW-1) of               */
            return w + t <= T + 1 ? ORC_TRPL18 : ORC_UNSUPPORTED; /* int_cmult_trpl18_dsp48.vhd:151-152 must lie inside the */
        return ORC_UNSUPPORTED;                   /* PWD = 79 / 77 bits of P, or the slice does not elaborate            */
    }
    if (t < TD) {                  /* xGEN_TWD25 :307 */
        if (w < 19) return ORC_SNGL25;            /* :309 */
        if (w < 36) return ORC_DBL35;             /* :357 */
        if (w < 53) return ORC_TRPL52;            /* :396 */
        return ORC_UNSUPPORTED;
    }
    return ORC_UNSUPPORTED;        /* find_delay -> 0, int_dif2_fly.vhd:112-114 */
}

/* (M2, M1, op) -> result at the regime's truncation points.  op = +1 add, -1 subtract.
 * M2 - M1 / M2 + M1: int_cmult18x25_dsp48.vhd:19-20,111-116 (PCIN +/- A*B). */
static inline int64_t combine(i128 m2, i128 m1, int op, int regime, int w, int t, int xser)
{
    i128 v;
    switch (regime) {
    case ORC_SNGL: /* P(DTW+TWD-2 downto TWD-1)  int_cmult_dsp48.vhd:189-190 */
        v = (op > 0 ? m2 + m1 : m2 - m1) >> (t - 1);
        break;
    case ORC_DBL18: { /* product slice from bit t-4 (NEW) / t-6 (OLD), then result slice from
                         bit 3 / 5: int_cmult_dbl18_dsp48.vhd:163,174-175 (AWD 44/42, PWD 62/60) */
        const int a = xser ? t - 4 : t - 6, b = xser ? 3 : 5;
        i128 s2 = m2 >> a, s1 = m1 >> a;
        v = (op > 0 ? s2 + s1 : s2 - s1) >> b;
        break;
    }
    case ORC_TRPL18: { /* P(MAW+MBW-2 downto MBW-1) of each product, then add:
                          int_cmult_trpl18_dsp48.vhd:151-155 */
        i128 s2 = m2 >> (t - 1), s1 = m1 >> (t - 1);
        v = (op > 0 ? s2 + s1 : s2 - s1);
        break;
    }
    case ORC_SNGL25: /* P(DTW+TWD-3 downto TWD-2)  int_cmult_dsp48.vhd:316-317 */
        v = (op > 0 ? m2 + m1 : m2 - m1) >> (t - 2);
        break;
    case ORC_DBL35: { /* slice from bit t-14 (both XSER), result from bit 12:
                         int_cmult_dbl35_dsp48.vhd:163-168 */
        i128 s2 = m2 >> (t - 14), s1 = m1 >> (t - 14);
        v = (op > 0 ? s2 + s1 : s2 - s1) >> 12;
        break;
    }
    case ORC_TRPL52: { /* int_cmult_trpl52_dsp48.vhd:166-170 */
        i128 s2 = m2 >> (t - 2), s1 = m1 >> (t - 2);
        v = (op > 0 ? s2 + s1 : s2 - s1);
        break;
    }
    default:
        v = 0;
    }
    return wrap128(v, w);
}

/* DO_RE = DI_RE*WW_RE - DI_IM*WW_IM, DO_IM = DI_RE*WW_IM + DI_IM*WW_RE, truncated and wrapped
 * to w bits (int_cmult_dsp48.vhd:192-224: RE: M2 = re*wr, M1 = im*wi, SUB; IM: M2 = re*wi,
 * M1 = im*wr, ADD). */
int orc_cmult(int64_t d_re, int64_t d_im, int64_t wr, int64_t wi, int w, int t, int xser,
              int64_t *o_re, int64_t *o_im)
{
    const int regime = orc_cmult_regime(w, t, xser);
    if (regime < 0) return -1;
    if ((regime == ORC_SNGL || regime == ORC_SNGL25) && w + t < 62) { /* fast path, same maths */
        const int sh = regime == ORC_SNGL ? t - 1 : t - 2;
        *o_re = orc_wrap((d_re * wr - d_im * wi) >> sh, w);
        *o_im = orc_wrap((d_re * wi + d_im * wr) >> sh, w);
        return 0;
    }
    if (regime == ORC_TRPL18 && w > (xser ? 61 : 59)) { /* the A port is SXT(M_AA, AWD), AWD = 61 / 59: a longer operand is CUT */
        d_re = orc_wrap(d_re, xser ? 61 : 59);           /* to its low AWD bits (int_cmult_trpl18_dsp48.vhd:161-162; the block is  */
        d_im = orc_wrap(d_im, xser ? 61 : 59);           /* written for "data width from 42/44 to 59/61", int_cmult_dsp48.vhd:266)  */
    }
    *o_re = combine((i128)d_re * wr, (i128)d_im * wi, -1, regime, w, t, xser);
    *o_im = combine((i128)d_re * wi, (i128)d_im * wr, +1, regime, w, t, xser);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* twiddles: rom_twiddle_int.vhd:118-246, row_twiddle_tay.vhd:123-268                         */
/* ------------------------------------------------------------------------------------------ */

/* VHDL INTEGER(real): round to nearest (rom_twiddle_int.vhd:151-152) */
static inline int64_t rn(double v) { return (int64_t)llround(v); }

/* quarter-wave ROM entry ii of a 2^depth table: rom_twiddle_int.vhd:135-159 */
static void rom_entry(int depth, int t, int64_t ii, int64_t *re, int64_t *im)
{
    const double mg = (t < 18) ? ldexp(1.0, t - 1) - 1.0 : ldexp(1.0, t - 2) - 1.0; /* :143-147 */
    const double phi = ((double)ii * M_PI) / ldexp(1.0, depth + 1);                  /* :149 */
    *re = rn(mg * cos(phi));
    *im = rn(mg * sin(-phi));
}

int orc_twiddles(int stage, int twd, int xser, int64_t *re, int64_t *im)
{
    if (stage < 0 || stage > 19 || twd < 2 || twd > 32) return -1;
    const size_t cnt_n = (size_t)1 << stage;
    if (stage == 0) { /* DEPTH = 0, never read by the butterflies (int_dif2_fly.vhd:245) */
        rom_entry(0, twd, 0, &re[0], &im[0]);
        return 0;
    }
    if (stage < 11) { /* xSTD rom_twiddle_int.vhd:205-212 */
        const int depth = stage - 1; /* find_depth :118-129 */
        for (size_t k = 0; k < cnt_n; ++k) {
            const int div = (int)(k >> (stage - 1));                   /* cnt(STAGE-1) :189 */
            const int64_t addr = (int64_t)(k & ((cnt_n >> 1) - 1));     /* cnt(STAGE-2..0) :188 */
            int64_t r, i;
            rom_entry(depth, twd, addr, &r, &i);
            if (div) { /* second quadrant: (re, im) <- (im, -re)  :177-183 */
                re[k] = i;
                im[k] = orc_wrap(-r, twd);
            } else {
                re[k] = r;
                im[k] = i;
            }
        }
        return 0;
    }
    /* xLNG :215-246 + row_twiddle_tay */
    const int ii = stage - 11;
    const int del = xser ? 2 : 0;                 /* const_pi row_twiddle_tay.vhd:135-149 */
    const int xs = xser ? 21 : 23;                /* find_widthA :123-133 */
    const int64_t mathpi = rn(M_PI * ldexp(1.0, 13 - ii - del));
    for (size_t k = 0; k < cnt_n; ++k) {
        const int div = (int)(k >> (stage - 1));
        const int64_t addr = (int64_t)(k & ((cnt_n >> 1) - 1));
        const int64_t ax = addr >> (stage - 10);                       /* addrx :221 */
        const int64_t cnt = addr & (((int64_t)1 << (stage - 10)) - 1); /* count :225 */
        int64_t r, i;
        rom_entry(9, twd, ax, &r, &i);
        if (div) { /* pr_ww applies the quadrant rotation before the Taylor block :174-184,237 */
            int64_t tr = i, ti = orc_wrap(-r, twd);
            r = tr;
            i = ti;
        }
        const int64_t mpi = (mathpi * cnt) & 0xFFFF;   /* 16-bit ROM word :208-221 */
        const int64_t mpx = mpi >> 1;                  /* '0' & mpi(17 downto 1) :247 */
        /* sin_aa carries re, cos_aa carries im (:250-251).  MULT_SUB: P = C + A*B with
         * A = cos_aa (im), C = sin_cc (re << XS); MULT_ADD: P = C - A*B with A = sin_aa (re),
         * C = cos_cc (im << XS)  (:304-312,374-382 ALUMODE 0011 / 0000). */
        const int64_t p_re = orc_wrap(r * ((int64_t)1 << xs) + i * mpx, 48); /* 48-bit P */
        const int64_t p_im = orc_wrap(i * ((int64_t)1 << xs) - r * mpx, 48);
        re[k] = orc_wrap(rhu2(p_re >> (xs - 1)), twd); /* pdt = P(47 downto XS-1); rnd :176-199 */
        im[k] = orc_wrap(rhu2(p_im >> (xs - 1)), twd);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* parameter checks / widths                                                                  */
/* ------------------------------------------------------------------------------------------ */

int orc_out_width(const orc_params *p, int direction)
{
    const int g = p->format ? p->log2n : 0;
    return p->data_width + (direction == ORC_PAIR ? 2 * g : g);
}

static int check_core(const orc_params *p, int dw_in, int inverse)
{
    for (int ii = 0; ii < p->log2n; ++ii) {
        const int stage = inverse ? ii : p->log2n - ii - 1;
        const int dtw = dw_in + ii * p->format;
        if (stage < 2) continue;
        /* DIF multiplies at DTW+1-SCALE (int_dif2_fly.vhd:351), DIT at DTW (int_dit2_fly.vhd:307) */
        const int w = inverse ? dtw : dtw + p->format;
        if (orc_cmult_regime(w, p->twdl_width, p->xser) < 0) return -1;
    }
    return 0;
}

int orc_validate(const orc_params *p, int direction)
{
    if (p->log2n < 3 || p->log2n > 20) return -1;
    if (p->data_width < 2 || p->twdl_width < 4) return -1;
    if (p->format != 0 && p->format != 1) return -1;
    if (p->rndmode != 0 && p->rndmode != 1) return -1;
    if (p->format == 1 && p->rndmode == 1) return -1; /* not elaboratable: int_dif2_fly.vhd:339-346 */
    if (orc_out_width(p, direction) > 64) return -1;
    if (direction == ORC_FWD || direction == ORC_PAIR)
        if (check_core(p, p->data_width, 0)) return -1;
    if (direction == ORC_INV)
        if (check_core(p, p->data_width, 1)) return -1;
    if (direction == ORC_PAIR)
        if (check_core(p, p->data_width + p->format * p->log2n, 1)) return -1;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* butterflies                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* "for positive values use Y = not(X) + 1, for negative values use Y = not(X)"
 * int_dif2_fly.vhd:280-304, int_dit2_fly.vhd:251-276 */
static inline int64_t neg_quirk(int64_t x, int w) { return x >= 0 ? orc_wrap(-x, w) : ~x; }

/* int_dif2_fly.vhd:144-373.  a, b: DTW-bit inputs; out width DTW-SCALE+1. */
void orc_dif_fly(const orc_params *p, int stage, int dtw, int odd, orc_cplx a, orc_cplx b,
                 int64_t wr, int64_t wi, orc_cplx *x, orc_cplx *y)
{
    const int scale = 1 - p->format;
    const int wo = dtw - scale + 1;
    orc_cplx s, d;
    if (scale && p->rndmode == 0) { /* xTRUNC :144-164, inputs sliced (DTW-1 downto 1) */
        s.re = (a.re >> 1) + (b.re >> 1);
        s.im = (a.im >> 1) + (b.im >> 1);
        d.re = (a.re >> 1) - (b.re >> 1);
        d.im = (a.im >> 1) - (b.im >> 1);
    } else if (scale) { /* xROUND :167-219 */
        s.re = orc_wrap(rhu2_sum(a.re, b.re), wo);
        s.im = orc_wrap(rhu2_sum(a.im, b.im), wo);
        d.re = orc_wrap(rhu2_diff(a.re, b.re), wo);
        d.im = orc_wrap(rhu2_diff(a.im, b.im), wo);
    } else { /* xUNSCALED :221-241 */
        s.re = a.re + b.re;
        s.im = a.im + b.im;
        d.re = a.re - b.re;
        d.im = a.im - b.im;
    }
    *x = s;
    if (stage == 0) { /* xST0 :245-255 */
        *y = d;
    } else if (stage == 1) { /* xST1 :259-318 */
        if (!odd) {
            *y = d;
        } else {
            y->re = d.im;
            y->im = neg_quirk(d.re, wo);
        }
    } else { /* xSTn :322-373 */
        orc_cmult(d.re, d.im, wr, wi, wo, p->twdl_width, p->xser, &y->re, &y->im);
    }
}

/* int_dit2_fly.vhd:142-325.  a, b: DTW-bit inputs; out width DTW-SCALE+1. */
void orc_dit_fly(const orc_params *p, int stage, int dtw, int odd, orc_cplx a, orc_cplx b,
                 int64_t wr, int64_t wi, orc_cplx *x, orc_cplx *y)
{
    const int scale = 1 - p->format;
    const int wo = dtw - scale + 1;
    orc_cplx t;
    if (stage == 0) { /* xST0 :221-230 */
        t = b;
    } else if (stage == 1) { /* xST1 :234-286 */
        if (!odd) {
            t = b;
        } else {
            t.im = b.re;
            t.re = neg_quirk(b.im, dtw);
        }
    } else { /* xSTn :290-325: DI_RE <- IB_IM, DI_IM <- IB_RE, DO_RE -> bw_im, DO_IM -> bw_re */
        int64_t o_re, o_im;
        orc_cmult(b.im, b.re, wr, wi, dtw, p->twdl_width, p->xser, &o_re, &o_im);
        t.im = o_re;
        t.re = o_im;
    }
    if (scale && p->rndmode == 0) { /* xUNSCALED with SCALE=1: slices (DTW-1 downto 1) :142-162 */
        x->re = (a.re >> 1) + (t.re >> 1);
        x->im = (a.im >> 1) + (t.im >> 1);
        y->re = (a.re >> 1) - (t.re >> 1);
        y->im = (a.im >> 1) - (t.im >> 1);
    } else if (scale) { /* xROUND :164-217 */
        x->re = orc_wrap(rhu2_sum(a.re, t.re), wo);
        x->im = orc_wrap(rhu2_sum(a.im, t.im), wo);
        y->re = orc_wrap(rhu2_diff(a.re, t.re), wo);
        y->im = orc_wrap(rhu2_diff(a.im, t.im), wo);
    } else {
        x->re = a.re + t.re;
        x->im = a.im + t.im;
        y->re = a.re - t.re;
        y->im = a.im - t.im;
    }
}

/* USE_FLY = '0' (int_fftNk.vhd:260-277): the stage forwards its input vector.  The per-stage
 * vectors are zero-initialised at full width and only the low DTW bits are driven
 * (int_fftNk.vhd:131-134,179-182,326-329), so in unscaled mode the pattern is zero-extended. */
static inline orc_cplx bypass(const orc_params *p, int dtw, orc_cplx v)
{
    if (p->format && dtw < 64) {
        const uint64_t mask = ((uint64_t)1 << dtw) - 1;
        v.re = (int64_t)((uint64_t)v.re & mask);
        v.im = (int64_t)((uint64_t)v.im & mask);
    }
    return v;
}

/* ------------------------------------------------------------------------------------------ */
/* twiddle cache for one transform                                                            */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    int64_t *re[20], *im[20];
} tw_set;

static int tw_build(tw_set *tw, int log2n, int twd, int xser)
{
    memset(tw, 0, sizeof(*tw));
    for (int s = 2; s < log2n; ++s) {
        tw->re[s] = (int64_t *)malloc(sizeof(int64_t) << s);
        tw->im[s] = (int64_t *)malloc(sizeof(int64_t) << s);
        if (!tw->re[s] || !tw->im[s] || orc_twiddles(s, twd, xser, tw->re[s], tw->im[s])) return -1;
    }
    return 0;
}

static void tw_free(tw_set *tw)
{
    for (int s = 0; s < 20; ++s) {
        free(tw->re[s]);
        free(tw->im[s]);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* stream form: math/fn_radix2.m dataflow with the RTL butterflies                            */
/* ------------------------------------------------------------------------------------------ */

/* cross-commutation int_delay_line.vhd:60-104 == fn_rev2rdx / fn_rdx2rev (fn_radix2.m:51-89):
 * blocks of b words: A' = [A0 B0 A2 B2 ...], B' = [A1 B1 A3 B3 ...] */
static void commutate(const orc_cplx *ia, const orc_cplx *ib, orc_cplx *oa, orc_cplx *ob,
                      size_t half, size_t b)
{
    const size_t nblk = half / b;
    for (size_t j = 0; j < nblk; ++j) {
        const size_t m = (j >> 1) << 1;
        const orc_cplx *src = (j & 1) ? ib : ia;
        memcpy(oa + j * b, src + m * b, b * sizeof(orc_cplx));
        memcpy(ob + j * b, src + (m + 1) * b, b * sizeof(orc_cplx));
    }
}

static int fft_stream_tw(const orc_params *p, const tw_set *tw, const orc_cplx *x, orc_cplx *v)
{
    const int L = p->log2n;
    const size_t n = (size_t)1 << L, half = n >> 1;
    orc_cplx *buf = (orc_cplx *)malloc(4 * half * sizeof(orc_cplx));
    if (!buf) return -1;
    orc_cplx *ta = buf, *tb = buf + half, *oa = buf + 2 * half, *ob = buf + 3 * half;
    for (size_t i = 0; i < half; ++i) { /* input buffer fn_radix2.m:154-155, int_fftNk.vhd:15-17 */
        ta[i].re = orc_wrap(x[i].re, p->data_width);
        ta[i].im = orc_wrap(x[i].im, p->data_width);
        tb[i].re = orc_wrap(x[i + half].re, p->data_width);
        tb[i].im = orc_wrap(x[i + half].im, p->data_width);
    }
    for (int ii = 0; ii < L; ++ii) { /* xCALC int_fftNk.vhd:184; fn_radix2.m:161 */
        const int stage = L - ii - 1;                 /* :192 */
        const int dtw = p->data_width + ii * p->format; /* :193 */
        const size_t msk = ((size_t)1 << stage) - 1;
        for (size_t q = 0; q < half; ++q) {
            if (!p->use_fly) {
                oa[q] = bypass(p, dtw, ta[q]);
                ob[q] = bypass(p, dtw, tb[q]);
                continue;
            }
            const size_t k = q & msk; /* twiddle counter mod 2^STAGE rom_twiddle_int.vhd:187-202 */
            const int64_t wr = stage >= 2 ? tw->re[stage][k] : 0;
            const int64_t wi = stage >= 2 ? tw->im[stage][k] : 0;
            orc_dif_fly(p, stage, dtw, (int)(q & 1), ta[q], tb[q], wr, wi, &oa[q], &ob[q]);
        }
        if (ii < L - 1) { /* xDELAYS int_fftNk.vhd:281-331: STAGE = ii, N_INV = NFFT-ii-2 */
            commutate(oa, ob, ta, tb, half, (size_t)1 << (L - ii - 2));
        }
    }
    for (size_t i = 0; i < half; ++i) { /* fn_radix2.m:182-185: Oo(2i-1) = Oa(i), Oo(2i) = Ob(i) */
        v[2 * i] = oa[i];
        v[2 * i + 1] = ob[i];
    }
    free(buf);
    return 0;
}

static int ifft_stream_tw(const orc_params *p, const tw_set *tw, const orc_cplx *v, orc_cplx *x)
{
    const int L = p->log2n;
    const size_t n = (size_t)1 << L, half = n >> 1;
    orc_cplx *buf = (orc_cplx *)malloc(4 * half * sizeof(orc_cplx));
    if (!buf) return -1;
    orc_cplx *ta = buf, *tb = buf + half, *oa = buf + 2 * half, *ob = buf + 3 * half;
    for (size_t i = 0; i < half; ++i) { /* fn_radix2.m:198-201 */
        ta[i].re = orc_wrap(v[2 * i].re, p->data_width);
        ta[i].im = orc_wrap(v[2 * i].im, p->data_width);
        tb[i].re = orc_wrap(v[2 * i + 1].re, p->data_width);
        tb[i].im = orc_wrap(v[2 * i + 1].im, p->data_width);
    }
    for (int ii = 0; ii < L; ++ii) { /* xCALC int_ifftNk.vhd:183 */
        const int stage = ii;                          /* :189 */
        const int dtw = p->data_width + ii * p->format;
        const size_t msk = ((size_t)1 << stage) - 1;
        for (size_t q = 0; q < half; ++q) {
            if (!p->use_fly) {
                oa[q] = bypass(p, dtw, ta[q]);
                ob[q] = bypass(p, dtw, tb[q]);
                continue;
            }
            const size_t k = q & msk;
            const int64_t wr = stage >= 2 ? tw->re[stage][k] : 0;
            const int64_t wi = stage >= 2 ? tw->im[stage][k] : 0;
            orc_dit_fly(p, stage, dtw, (int)(q & 1), ta[q], tb[q], wr, wi, &oa[q], &ob[q]);
        }
        if (ii < L - 1) { /* int_ifftNk.vhd:289-311: commutator STAGE = NFFT-ii-2 -> block 2^ii */
            commutate(oa, ob, ta, tb, half, (size_t)1 << ii);
        }
    }
    memcpy(x, oa, half * sizeof(orc_cplx)); /* fn_radix2.m:229-230 */
    memcpy(x + half, ob, half * sizeof(orc_cplx));
    free(buf);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* in-place form: flat array, Gentleman-Sande DIF / Cooley-Tukey DIT (independent of the      */
/* commutator network; must agree bit-for-bit with the stream form)                           */
/* ------------------------------------------------------------------------------------------ */

static int fft_inplace_tw(const orc_params *p, const tw_set *tw, const orc_cplx *x, orc_cplx *v)
{
    const int L = p->log2n;
    const size_t n = (size_t)1 << L;
    for (size_t i = 0; i < n; ++i) {
        v[i].re = orc_wrap(x[i].re, p->data_width);
        v[i].im = orc_wrap(x[i].im, p->data_width);
    }
    for (int ii = 0; ii < L; ++ii) {
        const int stage = L - ii - 1;
        const int dtw = p->data_width + ii * p->format;
        const size_t h = (size_t)1 << stage;
        for (size_t g = 0; g < n; g += 2 * h) {
            for (size_t k = 0; k < h; ++k) {
                orc_cplx *a = &v[g + k], *b = &v[g + k + h];
                if (!p->use_fly) {
                    *a = bypass(p, dtw, *a);
                    *b = bypass(p, dtw, *b);
                    continue;
                }
                const int64_t wr = stage >= 2 ? tw->re[stage][k] : 0;
                const int64_t wi = stage >= 2 ? tw->im[stage][k] : 0;
                orc_cplx ox, oy;
                orc_dif_fly(p, stage, dtw, (int)(k & 1), *a, *b, wr, wi, &ox, &oy);
                *a = ox;
                *b = oy;
            }
        }
    }
    return 0;
}

static int ifft_inplace_tw(const orc_params *p, const tw_set *tw, const orc_cplx *v, orc_cplx *x)
{
    const int L = p->log2n;
    const size_t n = (size_t)1 << L;
    for (size_t i = 0; i < n; ++i) {
        x[i].re = orc_wrap(v[i].re, p->data_width);
        x[i].im = orc_wrap(v[i].im, p->data_width);
    }
    for (int ii = 0; ii < L; ++ii) {
        const int stage = ii;
        const int dtw = p->data_width + ii * p->format;
        const size_t h = (size_t)1 << stage;
        for (size_t g = 0; g < n; g += 2 * h) {
            for (size_t k = 0; k < h; ++k) {
                orc_cplx *a = &x[g + k], *b = &x[g + k + h];
                if (!p->use_fly) {
                    *a = bypass(p, dtw, *a);
                    *b = bypass(p, dtw, *b);
                    continue;
                }
                const int64_t wr = stage >= 2 ? tw->re[stage][k] : 0;
                const int64_t wi = stage >= 2 ? tw->im[stage][k] : 0;
                orc_cplx ox, oy;
                orc_dit_fly(p, stage, dtw, (int)(k & 1), *a, *b, wr, wi, &ox, &oy);
                *a = ox;
                *b = oy;
            }
        }
    }
    return 0;
}

#define ORC_WRAPPER(name, impl)                                                        \
    int name(const orc_params *p, const orc_cplx *in, orc_cplx *out)                   \
    {                                                                                  \
        tw_set tw;                                                                     \
        if (tw_build(&tw, p->log2n, p->twdl_width, p->xser)) { tw_free(&tw); return -1; } \
        const int rc = impl(p, &tw, in, out);                                          \
        tw_free(&tw);                                                                  \
        return rc;                                                                     \
    }
ORC_WRAPPER(orc_fft_stream, fft_stream_tw)
ORC_WRAPPER(orc_fft_inplace, fft_inplace_tw)
ORC_WRAPPER(orc_ifft_stream, ifft_stream_tw)
ORC_WRAPPER(orc_ifft_inplace, ifft_inplace_tw)

/* ------------------------------------------------------------------------------------------ */
/* I/O orders and the batched driver                                                          */
/* ------------------------------------------------------------------------------------------ */

size_t orc_order_index(int order, int log2n, size_t m)
{
    const size_t half = (size_t)1 << (log2n - 1);
    switch (order) {
    case ORC_NATURAL: return m;
    case ORC_BITREV: return bitrev(m, log2n); /* fn_radix2.m:182-188 */
    case ORC_HALVES: return (m >> 1) + (m & 1) * half; /* int_fftNk.vhd:15-17, beat-major */
    case ORC_BITREV_LANES: /* outbuf_half_path.vhd:160-172 serial [lane0 ; lane1], then
                              int_bitrev_order.vhd:82-104 undoes exactly this map */
        return bitrev(2 * (m & (half - 1)) + (m >> (log2n - 1)), log2n);
    default: return m;
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int exec_frame(const orc_params *p, const tw_set *tw, int direction, int in_order,
                      int out_order, const orc_cplx *in, orc_cplx *out, orc_cplx *t0,
                      orc_cplx *t1, int form)
{
    const int L = p->log2n;
    const size_t n = (size_t)1 << L;
    int rc = 0;
    if (direction == ORC_FWD || direction == ORC_PAIR) {
        for (size_t m = 0; m < n; ++m) t0[orc_order_index(in_order, L, m)] = in[m];
        rc = form ? fft_inplace_tw(p, tw, t0, t1) : fft_stream_tw(p, tw, t0, t1);
        if (rc) return rc;
        if (direction == ORC_FWD) { /* t1[j] = X[rev j] */
            for (size_t m = 0; m < n; ++m)
                out[m] = t1[bitrev(orc_order_index(out_order, L, m), L)];
            return 0;
        }
        /* PAIR: FFT lane outputs feed the IFFT lane inputs directly
         * (int_fft_ifft_pair.vhd:242-280), IFFT DATA_WIDTH = DATA_WIDTH + FORMAT*NFFT (:261) */
        orc_params q = *p;
        q.data_width = p->data_width + p->format * L;
        rc = form ? ifft_inplace_tw(&q, tw, t1, t0) : ifft_stream_tw(&q, tw, t1, t0);
        if (rc) return rc;
        for (size_t m = 0; m < n; ++m) out[m] = t0[orc_order_index(out_order, L, m)];
        return 0;
    }
    /* INV: in[m] = X[order(m)]; the core wants v[j] = X[rev j] */
    for (size_t m = 0; m < n; ++m) t0[bitrev(orc_order_index(in_order, L, m), L)] = in[m];
    rc = form ? ifft_inplace_tw(p, tw, t0, t1) : ifft_stream_tw(p, tw, t0, t1);
    if (rc) return rc;
    for (size_t m = 0; m < n; ++m) out[m] = t1[orc_order_index(out_order, L, m)];
    return 0;
}

/* threads <= 0: sized to the work (about 2^21 stage-samples per thread, never more threads than frames) -- a team of every host
 * thread for a handful of short frames costs ~0.15 s per call on a 256-thread box (spin-up / spin-down beside torch's own pool),
 * which was 85 % of the GPU parity suite's run time.  An explicit count (bench.py's cpu_baseline passes one) is used as given except
 * that it, too, is capped at the number of frames: frames are the only unit of parallelism here, extra threads would idle.
 * Callers that relied on the earlier "threads <= 0 means every core" now get the work-sized team; pass omp_get_max_threads() for that. */
static int pick_threads(int threads, size_t batch, size_t n, int log2n)
{
#ifdef _OPENMP
    const int mx = omp_get_max_threads();
    if (threads <= 0) {
        const size_t want = (batch * n * (size_t)log2n) >> 21;
        threads = want < 1 ? 1 : want > (size_t)mx ? mx : (int)want;
    }
    if ((size_t)threads > batch) threads = (int)(batch ? batch : 1);
    return threads;
#else
    (void)batch; (void)n; (void)log2n;
    return threads > 0 ? threads : 1;
#endif
}

int orc_exec(const orc_params *p, int direction, int in_order, int out_order, const int64_t *in,
             int64_t *out, size_t batch, int form, int threads)
{
    if (orc_validate(p, direction)) return -1;
    const size_t n = (size_t)1 << p->log2n;
    tw_set tw;
    if (tw_build(&tw, p->log2n, p->twdl_width, p->xser)) { tw_free(&tw); return -2; }
    int err = 0;
#ifdef _OPENMP
    threads = pick_threads(threads, batch, n, p->log2n);
#pragma omp parallel num_threads(threads)
#endif
    {
        orc_cplx *t0 = (orc_cplx *)malloc(2 * n * sizeof(orc_cplx));
        orc_cplx *t1 = t0 ? t0 + n : NULL;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long long f = 0; f < (long long)batch; ++f) {
            if (!t0) { err = -2; continue; }
            const orc_cplx *fi = (const orc_cplx *)(in + 2 * n * (size_t)f);
            orc_cplx *fo = (orc_cplx *)(out + 2 * n * (size_t)f);
            if (exec_frame(p, &tw, direction, in_order, out_order, fi, fo, t0, t1, form)) err = -3;
        }
        free(t0);
    }
    tw_free(&tw);
    (void)threads;
    return err;
}

int orc_exec_i16(const orc_params *p, int direction, int in_order, int out_order,
                 const int16_t *in, int16_t *out, size_t batch, int form, int threads)
{
    if (orc_validate(p, direction)) return -1;
    if (orc_out_width(p, direction) > 16 || p->data_width > 16) return -1;
    const size_t n = (size_t)1 << p->log2n;
    tw_set tw;
    if (tw_build(&tw, p->log2n, p->twdl_width, p->xser)) { tw_free(&tw); return -2; }
    int err = 0;
#ifdef _OPENMP
    threads = pick_threads(threads, batch, n, p->log2n);
#pragma omp parallel num_threads(threads)
#endif
    {
        orc_cplx *t0 = (orc_cplx *)malloc(4 * n * sizeof(orc_cplx));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long long f = 0; f < (long long)batch; ++f) {
            if (!t0) { err = -2; continue; }
            orc_cplx *fi = t0 + 2 * n, *fo = t0 + 3 * n;
            const int16_t *src = in + 2 * n * (size_t)f;
            int16_t *dst = out + 2 * n * (size_t)f;
            for (size_t m = 0; m < n; ++m) {
                fi[m].re = src[2 * m];
                fi[m].im = src[2 * m + 1];
            }
            if (exec_frame(p, &tw, direction, in_order, out_order, fi, fo, t0, t0 + n, form)) err = -3;
            for (size_t m = 0; m < n; ++m) {
                dst[2 * m] = (int16_t)fo[m].re;
                dst[2 * m + 1] = (int16_t)fo[m].im;
            }
        }
        free(t0);
    }
    tw_free(&tw);
    (void)threads;
    return err;
}

/* ------------------------------------------------------------------------------------------ */
/* N > 512K: the "2D-FFT scheme" (int_fftNk.vhd:11-13, row_twiddle_tay.vhd:31)                 */
/* ------------------------------------------------------------------------------------------ */
/* The reference only NAMES the scheme ("For N > 512K you should use 2D-FFT scheme"); its exact   *
 * arithmetic is undefined there.  This is the extension this project defines (DESIGN.md          *
 * section 4.5), built from the reference's own blocks:                                           *
 *   N = N1 * N2, n = n1*N2 + n2, k = k1 + N1*k2, both cores native (log2 N1, log2 N2 in 3..19,   *
 *   Taylor range ii <= 7):                                                                       *
 *   FWD  columns: int_fftNk(N1) over n1 for every n2      -> A[k1][n2]                          *
 *        twiddle: B = int_cmult_dsp48(A, W_N^(k1*n2)) at the column core's output width          *
 *        rows:    int_fftNk(N2) over n2 for every k1 with DATA_WIDTH = that width -> X[k1+N1*k2] *
 *   INV  the mirror: rows int_ifftNk(N2), T = D * conj(W) through the re/im-swapped multiplier   *
 *        feed of int_dit2_fly.vhd:304-322, columns int_ifftNk(N1).                               *
 * Inter-pass twiddle W_N^m, m in [0, N): the quarter-wave ROM formula of                          *
 * rom_twiddle_int.vhd:143-152 at full depth (no Taylor step): for a = m mod N/4,                  *
 * (c, s) = (RN(mg cos(2 pi a / N)), RN(mg sin(-2 pi a / N))), mg as in :143-147; quadrant         *
 * q = m div N/4 applies (re, im) <- (im, -re) q times (:177-183 is the q = 1 case).  cos / sin are *
 * a specified sequence of IEEE double operations (orc_twiddle_2d), not a libm call, so that the   *
 * GPU can evaluate them on the fly and still produce the same integers.                           */

void orc_twiddle_2d(int log2n, int twd, size_t m, int64_t *re, int64_t *im)
{
    /* cos / sin as a FIXED sequence of separately rounded IEEE double operations (this file is compiled with
     * -ffp-contract=off), so that every implementation yields the same integers: octant reduction to x in [0, pi/4],
     * x = b * (pi * 2^-(L-1)), Taylor polynomials in z = x*x up to z^8 (Horner), RN(v) = floor(v + 0.5). */
    const size_t quarter = (size_t)1 << (log2n - 2);
    const size_t a = m & (quarter - 1);
    const int q = (int)((m >> (log2n - 2)) & 3);
    const int swap = a > (quarter >> 1);
    const size_t b = swap ? quarter - a : a;
    double scale = 3.14159265358979323846;
    for (int i = 0; i < log2n - 1; ++i) scale = scale * 0.5;
    const double x = (double)b * scale;
    const double z = x * x;
    static const double C[8] = {-0x1.0000000000000p-1, 0x1.5555555555555p-5, -0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-16,
                                -0x1.27e4fb7789f5cp-22, 0x1.1eed8eff8d898p-29, -0x1.93974a8c07c9dp-37, 0x1.ae7f3e733b81fp-45};
    static const double S[8] = {-0x1.5555555555555p-3, 0x1.1111111111111p-7, -0x1.a01a01a01a01ap-13, 0x1.71de3a556c734p-19,
                                -0x1.ae64567f544e4p-26, 0x1.6124613a86d09p-33, -0x1.ae7f3e733b81fp-41, 0x1.952c77030ad4ap-49};
    double pc = C[7], ps = S[7];
    for (int k = 6; k >= 0; --k) {
        double t = z * pc;
        pc = C[k] + t;
        t = z * ps;
        ps = S[k] + t;
    }
    double t1 = z * pc;
    const double cosx = 1.0 + t1;
    const double xz = x * z;
    t1 = xz * ps;
    const double sinx = x + t1;
    double mg = 1.0; /* rom_twiddle_int.vhd:143-147 */
    for (int i = 0; i < (twd < 18 ? twd - 1 : twd - 2); ++i) mg = mg * 2.0;
    mg = mg - 1.0;
    double vc = mg * (swap ? sinx : cosx), vs = mg * (swap ? cosx : sinx);
    vc = vc + 0.5;
    vs = vs + 0.5;
    int64_t c = (int64_t)vc, s = -(int64_t)vs;
    for (int i = 0; i < q; ++i) { /* (re, im) <- (im, -re)  rom_twiddle_int.vhd:177-183 */
        const int64_t t = c;
        c = s;
        s = orc_wrap(-t, twd);
    }
    *re = c;
    *im = s;
}

static int check_2d(const orc_params *p, int l1, int direction)
{
    const int L = p->log2n, l2 = L - l1;
    if (l1 < 3 || l1 > 19 || l2 < 3 || l2 > 19 || L > 24) return -1;
    if (!p->use_fly) return -1;
    if (p->data_width < 2 || p->twdl_width < 4) return -1;
    if ((p->format | 1) != 1 || (p->rndmode | 1) != 1 || (p->format && p->rndmode)) return -1;
    if (orc_out_width(p, direction) > 64) return -1;
    orc_params c = *p;
    int dw = p->data_width;
    if (direction == ORC_FWD || direction == ORC_PAIR) {
        c.log2n = l1;
        if (check_core(&c, dw, 0)) return -1;
        dw += p->format * l1;
        if (orc_cmult_regime(dw, p->twdl_width, p->xser) < 0) return -1;
        c.log2n = l2;
        if (check_core(&c, dw, 0)) return -1;
        dw += p->format * l2;
    }
    if (direction == ORC_INV || direction == ORC_PAIR) {
        c.log2n = l2;
        if (check_core(&c, dw, 1)) return -1;
        dw += p->format * l2;
        if (orc_cmult_regime(dw, p->twdl_width, p->xser) < 0) return -1;
        c.log2n = l1;
        if (check_core(&c, dw, 1)) return -1;
    }
    return 0;
}

int orc_validate_2d(const orc_params *p, int log2_n1, int direction) { return check_2d(p, log2_n1, direction); }

typedef struct {
    tw_set tw;          /* per-stage tables, shared by the two cores (a table depends on STAGE only) */
    int64_t *wre, *wim; /* inter-pass table, N entries */
} tw2d_set;

static int tw2d_build(tw2d_set *t, const orc_params *p, int l1)
{
    const int L = p->log2n, l2 = L - l1;
    memset(t, 0, sizeof(*t));
    if (tw_build(&t->tw, l1 > l2 ? l1 : l2, p->twdl_width, p->xser)) return -1;
    const size_t n = (size_t)1 << L;
    t->wre = (int64_t *)malloc(n * sizeof(int64_t));
    t->wim = (int64_t *)malloc(n * sizeof(int64_t));
    if (!t->wre || !t->wim) return -1;
    for (size_t m = 0; m < n; ++m) orc_twiddle_2d(L, p->twdl_width, m, &t->wre[m], &t->wim[m]);
    return 0;
}

static void tw2d_free(tw2d_set *t)
{
    tw_free(&t->tw);
    free(t->wre);
    free(t->wim);
}

/* forward, structural form: literally columns -> twiddle -> rows with the 1-D core functions.
 * x natural (n = n1*N2 + n2); v[j1*N2 + j2] = X[rev(j1) + N1*rev(j2)] = X[rev_L(j)]. */
static int fft2d_struct(const orc_params *p, int l1, const tw2d_set *t, const orc_cplx *x, orc_cplx *v, int stream)
{
    const int L = p->log2n, l2 = L - l1;
    const size_t n1 = (size_t)1 << l1, n2 = (size_t)1 << l2, n = n1 * n2;
    orc_cplx *col = (orc_cplx *)malloc(2 * (n1 > n2 ? n1 : n2) * sizeof(orc_cplx));
    orc_cplx *a = (orc_cplx *)malloc(n * sizeof(orc_cplx)); /* a[k1*N2 + n2] */
    if (!col || !a) { free(col); free(a); return -1; }
    orc_cplx *res = col + (n1 > n2 ? n1 : n2);
    orc_params c = *p;
    c.log2n = l1;
    int rc = 0;
    for (size_t i2 = 0; i2 < n2 && !rc; ++i2) {
        for (size_t i1 = 0; i1 < n1; ++i1) col[i1] = x[i1 * n2 + i2];
        rc = stream ? fft_stream_tw(&c, &t->tw, col, res) : fft_inplace_tw(&c, &t->tw, col, res);
        for (size_t j1 = 0; j1 < n1; ++j1) a[bitrev(j1, l1) * n2 + i2] = res[j1]; /* res[j1] = A[rev j1] */
    }
    const int w1 = p->data_width + p->format * l1;
    for (size_t k1 = 0; k1 < n1 && !rc; ++k1) {
        for (size_t i2 = 0; i2 < n2; ++i2) {
            const size_t m = (k1 * i2) & (n - 1);
            orc_cplx *e = &a[k1 * n2 + i2];
            int64_t yr, yi;
            if (orc_cmult(e->re, e->im, t->wre[m], t->wim[m], w1, p->twdl_width, p->xser, &yr, &yi)) rc = -1;
            e->re = yr;
            e->im = yi;
        }
        c.log2n = l2;
        c.data_width = w1;
        if (!rc) rc = stream ? fft_stream_tw(&c, &t->tw, a + k1 * n2, res) : fft_inplace_tw(&c, &t->tw, a + k1 * n2, res);
        for (size_t j2 = 0; j2 < n2; ++j2) v[bitrev(k1, l1) * n2 + j2] = res[j2]; /* res[j2] = C[k1][rev j2] */
    }
    free(col);
    free(a);
    return rc;
}

/* inverse, structural form: v[j] = X[rev_L j] in, x natural out */
static int ifft2d_struct(const orc_params *p, int l1, const tw2d_set *t, const orc_cplx *v, orc_cplx *x, int stream)
{
    const int L = p->log2n, l2 = L - l1;
    const size_t n1 = (size_t)1 << l1, n2 = (size_t)1 << l2, n = n1 * n2;
    orc_cplx *col = (orc_cplx *)malloc(2 * (n1 > n2 ? n1 : n2) * sizeof(orc_cplx));
    orc_cplx *d = (orc_cplx *)malloc(n * sizeof(orc_cplx)); /* d[j1*N2 + n2], row j1 <-> k1 = rev(j1) */
    if (!col || !d) { free(col); free(d); return -1; }
    orc_cplx *res = col + (n1 > n2 ? n1 : n2);
    orc_params c = *p;
    int rc = 0;
    const int w = p->data_width + p->format * l2; /* DIT multiplies at its input width (int_dit2_fly.vhd:307) */
    for (size_t j1 = 0; j1 < n1 && !rc; ++j1) {
        c.log2n = l2;
        c.data_width = p->data_width;
        rc = stream ? ifft_stream_tw(&c, &t->tw, v + j1 * n2, d + j1 * n2) : ifft_inplace_tw(&c, &t->tw, v + j1 * n2, d + j1 * n2);
        const size_t k1 = bitrev(j1, l1);
        for (size_t i2 = 0; i2 < n2 && !rc; ++i2) {
            const size_t m = (k1 * i2) & (n - 1);
            orc_cplx *e = &d[j1 * n2 + i2];
            int64_t ore, oim; /* swapped feed: DI_RE <- B.im, DI_IM <- B.re; DO_RE -> T.im, DO_IM -> T.re */
            if (orc_cmult(e->im, e->re, t->wre[m], t->wim[m], w, p->twdl_width, p->xser, &ore, &oim)) rc = -1;
            e->im = ore;
            e->re = oim;
        }
    }
    c.log2n = l1;
    c.data_width = w;
    for (size_t i2 = 0; i2 < n2 && !rc; ++i2) {
        for (size_t j1 = 0; j1 < n1; ++j1) col[j1] = d[j1 * n2 + i2]; /* bit-reversed in k1: the core's native input */
        rc = stream ? ifft_stream_tw(&c, &t->tw, col, res) : ifft_inplace_tw(&c, &t->tw, col, res);
        for (size_t i1 = 0; i1 < n1; ++i1) x[i1 * n2 + i2] = res[i1];
    }
    free(col);
    free(d);
    return rc;
}

/* flat form: one array, the column stages on index bits L-1..L2 with the column core's tables (index = position
 * div N2), one multiply sweep, the row stages on bits L2-1..0 -- what the GPU kernels evaluate */
static int fft2d_flat(const orc_params *p, int l1, const tw2d_set *t, const orc_cplx *x, orc_cplx *v)
{
    const int L = p->log2n, l2 = L - l1;
    const size_t n = (size_t)1 << L, n2 = (size_t)1 << l2;
    for (size_t i = 0; i < n; ++i) {
        v[i].re = orc_wrap(x[i].re, p->data_width);
        v[i].im = orc_wrap(x[i].im, p->data_width);
    }
    for (int ii = 0; ii < L; ++ii) {
        const int col = ii < l1;
        const int stage = col ? l1 - ii - 1 : L - ii - 1; /* STAGE generic of the core the stage belongs to */
        const int sh = col ? l2 : 0;
        const int dtw = p->data_width + ii * p->format;
        const size_t h = (size_t)1 << (stage + sh);
        if (ii == l1) { /* between the cores */
            const int w1 = p->data_width + p->format * l1;
            for (size_t j = 0; j < n; ++j) {
                const size_t m = (bitrev(j >> l2, l1) * (j & (n2 - 1))) & (n - 1);
                int64_t yr, yi;
                if (orc_cmult(v[j].re, v[j].im, t->wre[m], t->wim[m], w1, p->twdl_width, p->xser, &yr, &yi)) return -1;
                v[j].re = yr;
                v[j].im = yi;
            }
        }
        for (size_t g = 0; g < n; g += 2 * h)
            for (size_t k = 0; k < h; ++k) {
                const size_t kt = k >> sh;
                const int64_t wr = stage >= 2 ? t->tw.re[stage][kt] : 0, wi = stage >= 2 ? t->tw.im[stage][kt] : 0;
                orc_cplx ox, oy;
                orc_dif_fly(p, stage, dtw, (int)(kt & 1), v[g + k], v[g + k + h], wr, wi, &ox, &oy);
                v[g + k] = ox;
                v[g + k + h] = oy;
            }
    }
    return 0;
}

static int ifft2d_flat(const orc_params *p, int l1, const tw2d_set *t, const orc_cplx *v, orc_cplx *x)
{
    const int L = p->log2n, l2 = L - l1;
    const size_t n = (size_t)1 << L, n2 = (size_t)1 << l2;
    for (size_t i = 0; i < n; ++i) {
        x[i].re = orc_wrap(v[i].re, p->data_width);
        x[i].im = orc_wrap(v[i].im, p->data_width);
    }
    for (int ii = 0; ii < L; ++ii) {
        const int col = ii >= l2;
        const int stage = col ? ii - l2 : ii;
        const int sh = col ? l2 : 0;
        const int dtw = p->data_width + ii * p->format;
        const size_t h = (size_t)1 << (stage + sh);
        if (ii == l2) {
            for (size_t j = 0; j < n; ++j) {
                const size_t m = (bitrev(j >> l2, l1) * (j & (n2 - 1))) & (n - 1);
                int64_t ore, oim;
                if (orc_cmult(x[j].im, x[j].re, t->wre[m], t->wim[m], dtw, p->twdl_width, p->xser, &ore, &oim)) return -1;
                x[j].im = ore;
                x[j].re = oim;
            }
        }
        for (size_t g = 0; g < n; g += 2 * h)
            for (size_t k = 0; k < h; ++k) {
                const size_t kt = k >> sh;
                const int64_t wr = stage >= 2 ? t->tw.re[stage][kt] : 0, wi = stage >= 2 ? t->tw.im[stage][kt] : 0;
                orc_cplx ox, oy;
                orc_dit_fly(p, stage, dtw, (int)(kt & 1), x[g + k], x[g + k + h], wr, wi, &ox, &oy);
                x[g + k] = ox;
                x[g + k + h] = oy;
            }
    }
    return 0;
}

/* form: 0 = structural with the stream-form cores, 1 = flat, 2 = structural with the in-place cores */
static int exec_frame_2d(const orc_params *p, int l1, const tw2d_set *t, int direction, int in_order, int out_order,
                         const orc_cplx *in, orc_cplx *out, orc_cplx *t0, orc_cplx *t1, int form)
{
    const int L = p->log2n;
    const size_t n = (size_t)1 << L;
    int rc = 0;
    if (direction == ORC_FWD || direction == ORC_PAIR) {
        for (size_t m = 0; m < n; ++m) {
            t0[orc_order_index(in_order, L, m)].re = orc_wrap(in[m].re, p->data_width);
            t0[orc_order_index(in_order, L, m)].im = orc_wrap(in[m].im, p->data_width);
        }
        rc = form == 1 ? fft2d_flat(p, l1, t, t0, t1) : fft2d_struct(p, l1, t, t0, t1, form == 0);
        if (rc) return rc;
        if (direction == ORC_FWD) {
            for (size_t m = 0; m < n; ++m) out[m] = t1[bitrev(orc_order_index(out_order, L, m), L)];
            return 0;
        }
        orc_params q = *p;
        q.data_width = p->data_width + p->format * L;
        rc = form == 1 ? ifft2d_flat(&q, l1, t, t1, t0) : ifft2d_struct(&q, l1, t, t1, t0, form == 0);
        if (rc) return rc;
        for (size_t m = 0; m < n; ++m) out[m] = t0[orc_order_index(out_order, L, m)];
        return 0;
    }
    for (size_t m = 0; m < n; ++m) {
        t0[bitrev(orc_order_index(in_order, L, m), L)].re = orc_wrap(in[m].re, p->data_width);
        t0[bitrev(orc_order_index(in_order, L, m), L)].im = orc_wrap(in[m].im, p->data_width);
    }
    rc = form == 1 ? ifft2d_flat(p, l1, t, t0, t1) : ifft2d_struct(p, l1, t, t0, t1, form == 0);
    if (rc) return rc;
    for (size_t m = 0; m < n; ++m) out[m] = t1[orc_order_index(out_order, L, m)];
    return 0;
}

int orc_exec_2d(const orc_params *p, int log2_n1, int direction, int in_order, int out_order, const int64_t *in,
                int64_t *out, size_t batch, int form, int threads)
{
    if (check_2d(p, log2_n1, direction)) return -1;
    const size_t n = (size_t)1 << p->log2n;
    tw2d_set t;
    if (tw2d_build(&t, p, log2_n1)) { tw2d_free(&t); return -2; }
    int err = 0;
#ifdef _OPENMP
    threads = pick_threads(threads, batch, n, p->log2n);
#pragma omp parallel num_threads(threads)
#endif
    {
        orc_cplx *t0 = (orc_cplx *)malloc(2 * n * sizeof(orc_cplx));
        orc_cplx *t1 = t0 ? t0 + n : NULL;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long long f = 0; f < (long long)batch; ++f) {
            if (!t0) { err = -2; continue; }
            const orc_cplx *fi = (const orc_cplx *)(in + 2 * n * (size_t)f);
            orc_cplx *fo = (orc_cplx *)(out + 2 * n * (size_t)f);
            if (exec_frame_2d(p, log2_n1, &t, direction, in_order, out_order, fi, fo, t0, t1, form)) err = -3;
        }
        free(t0);
    }
    tw2d_free(&t);
    (void)threads;
    return err;
}
