// intfft_pk16.hpp -- packed-int16 butterfly arithmetic shared by the wave / block kernels
// (intfft_fast1024.hip, intfft_fast4096.hip).  See intfft_fast1024.hip for the derivations:
// v_dot2 multiplies from hazard-safe asm blocks, pre-shifted outputs, fast extraction under the
// per-frame guard-bit condition, quarter-turn twiddle sharing.
#pragma once

#include "intfft_internal.hpp"

namespace intfft {

using u32 = uint32_t;
typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s as_v2s(u32 x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ u32 as_u32(v2s x) { return __builtin_bit_cast(u32, x); }

// result slicing of the exact 32-bit sums (t = TWDL_WIDTH)
struct Slice {
    int off_y;  // t - 1: Y      = sum[t+w-2 : t-1]
    int off_y1; // t    : Y >> 1 = sext(sum[t+w-2 : t])
    u32 sel;    // v_perm_b32 selector {S0.b1, S0.b0, S1.b1, S1.b0} (exact extraction)
    u32 sel_hi; // v_perm_b32 selector {S0.b3, S0.b2, S1.b3, S1.b2} (fast extraction)
    // DATA_WIDTH w = 9 .. 16 in int16 containers (set_width()).  Narrow data runs in the same 16-bit lanes: sums and differences of
    // sign-extended w-bit values are exact there, the exact extraction takes w (or w - 1) bits of the dot products -- the RTL's
    // w-bit wrap, then sign-extended -- and the guard-bit test of the fast path scales to |re|, |im| < 2^(w-2).
    int wd = 16;             // width of the exact extraction (v_bfe_i32)
    u32 gbias = 0x40004000u; // guard test: (x + gbias) & gmask == 0 for both halves <=> -2^(w-2) <= re, im < 2^(w-2)
    u32 gmask = 0x80008000u;
    u32 gbias1 = 0x20002000u, gmask1 = 0xC000C000u; // the same test for values of the Y >> 1 kind (one bit less)
    int round = 0; // multi-pass kernels (intfft_big20.hip): RNDMODE = 1 on the exact-path instantiations (a runtime switch there)
    __host__ __device__ void set_width(int w)
    {
        wd = w;
        const u32 b = 1u << (w - 2), m = (0xFFFFu << (w - 1)) & 0xFFFFu;
        gbias = b | (b << 16);
        gmask = m | (m << 16);
        gbias1 = gbias >> 1;
        gmask1 = gmask | ((gmask >> 1) & 0x7FFF7FFFu);
    }
};

__device__ __forceinline__ u32 pack_wa(int2 w) { return ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16); }
__device__ __forceinline__ u32 pack_wb(int2 w) { return ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16); }

// ---- sum / difference ---------------------------------------------------------------------------
// PRE: the inputs already hold A >> 1, B >> 1 (truncate mode only).
// SW: the difference comes out with its halves exchanged, (D.im, D.re) -- op_sel of the packed subtract, no extra instruction
// (group4's DPK form multiplies the swapped D by DIT-packed twiddles)
template <bool SW> __device__ __forceinline__ u32 pk_sub(v2s x, v2s y)
{
    if (!SW) return as_u32(x - y);
    u32 r;
    asm("v_pk_sub_i16 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(as_u32(x)), "v"(as_u32(y)));
    return r;
}
template <bool ROUND, bool PRE, bool SW = false> __device__ __forceinline__ void sumdiff(u32 a, u32 b, u32 &s, u32 &d)
{
    const v2s A = as_v2s(a), B = as_v2s(b);
    if (!ROUND) { // int_dif2_fly.vhd:144-164
        const v2s A1 = PRE ? A : A >> (short)1, B1 = PRE ? B : B >> (short)1;
        s = as_u32(A1 + B1);
        d = pk_sub<SW>(A1, B1);
    } else { // :167-219  rhu2(x) = floor((x + 1) / 2).  With T = (A ^ B) >> 1 (arithmetic):
        //   A + B = 2 (A | B) - (A ^ B)               ->  rhu2(A + B) = (A | B) - T
        //   A - B + 1 = A + ~B + 2, floor((A + ~B) / 2) = (A & ~B) + ((A ^ ~B) >> 1) = (A & ~B) - T - 1
        //                                              ->  rhu2(A - B) = (A & ~B) - T
        // and, cheaper still, (A - B + 1) = (A + B + 1) - 2 B  ->  rhu2(A - B) = rhu2(A + B) - B.  Exact over the integers, hence also
        // modulo 2^16 (the RTL's 16-bit wrap; rhu2(A + B) itself always fits 16 bits): FIVE operations for both results, two of them
        // (xor, or) in the VOP2 encoding that issues at ~2.6 clk against ~4.5 for the packed ones (profiles/r03_valubench.txt)
        const v2s T = (A ^ B) >> (short)1;
        const v2s S = (A | B) - T;
        s = as_u32(S);
        d = pk_sub<SW>(S, B);
    }
}
// Round mode on narrow data (DATA_WIDTH w < 16 in 16-bit lanes): rhu2(A - B) reaches +2^(w-1) when A = 2^(w-1) - 1, B = -2^(w-1),
// where the RTL's w-bit result wraps to -2^(w-1) (int_dif2_fly.vhd:173-218); nothing else leaves the w-bit range.  The callers
// wrap the differences in their ROUND == 2 instantiations (tested per group behind a wave-uniform branch the 16-bit round-mode
// kernels lost 2..7 %); the kernels select that body once per frame (Slice::wd != 16).
__device__ __forceinline__ u32 wrap_w(u32 x, int w)
{
    const short sh = (short)(16 - w);
    const v2s shv = {sh, sh};
    return as_u32((as_v2s(x) << shv) >> shv);
}
// truncate mode with a per-lane shift amount (0 where the lane's registers already hold X >> 1)
template <bool SW = false> __device__ __forceinline__ void sumdiff_var(u32 a, u32 b, v2s sh, u32 &s, u32 &d)
{
    const v2s A1 = as_v2s(a) >> sh, B1 = as_v2s(b) >> sh;
    s = as_u32(A1 + B1);
    d = pk_sub<SW>(A1, B1);
}

// ---- complex multiplies: cmult_{16,t}(D, W), single-DSP regime (int_cmult_dsp48.vhd:184-225) ----
// dr/di are the data operands of the re / im dot products (equal for a table twiddle; (D, -D) with
// the base twiddle's (Wb, Wa) for a quarter-turn twiddle).  SG: twiddles in SGPRs.
#define INTFFT_MUL2X_BODY                                                                              \
    "v_dot2_i32_i16 %[r0], %[dr0], %[wa0], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[i0], %[di0], %[wb0], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[r1], %[dr1], %[wa1], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[i1], %[di1], %[wb1], 0\n\t"                                                      \
    "v_bfe_i32 %[y0], %[r0], %[off], %[wd]\n\t"                                                        \
    "v_bfe_i32 %[r0], %[i0], %[off], %[wd]\n\t"                                                        \
    "v_bfe_i32 %[y1], %[r1], %[off], %[wd]\n\t"                                                        \
    "v_bfe_i32 %[i0], %[i1], %[off], %[wd]\n\t"                                                        \
    "v_perm_b32 %[y0], %[r0], %[y0], %[sel]\n\t"                                                       \
    "v_perm_b32 %[y1], %[i0], %[y1], %[sel]"

// exact extraction, 2 butterflies: (off, WIDTH) = (t-1, 16) -> Y, (t, 15) -> Y >> 1
// The width comes in a VGPR (two SGPR operands exceed gfx9's constant-bus limit): w for Y, w - 1 for Y >> 1, w = Slice::wd
template <int WIDTH, bool SG>
__device__ __forceinline__ void mul2x(u32 dr0, u32 di0, u32 wa0, u32 wb0, u32 dr1, u32 di1, u32 wa1, u32 wb1,
                                      int off, u32 sel, u32 &y0, u32 &y1, int wd_full = 16)
{
    static_assert(WIDTH == 16 || WIDTH == 15, "Y or Y >> 1");
    const int wdv = wd_full - (16 - WIDTH);
    u32 r0, i0, r1, i1;
    if (SG)
        asm(INTFFT_MUL2X_BODY
            : [y0] "=&v"(y0), [y1] "=&v"(y1), [r0] "=&v"(r0), [i0] "=&v"(i0), [r1] "=&v"(r1), [i1] "=&v"(i1)
            : [dr0] "v"(dr0), [di0] "v"(di0), [wa0] "s"(wa0), [wb0] "s"(wb0), [dr1] "v"(dr1), [di1] "v"(di1),
              [wa1] "s"(wa1), [wb1] "s"(wb1), [off] "s"(off), [wd] "v"(wdv), [sel] "s"(sel));
    else
        asm(INTFFT_MUL2X_BODY
            : [y0] "=&v"(y0), [y1] "=&v"(y1), [r0] "=&v"(r0), [i0] "=&v"(i0), [r1] "=&v"(r1), [i1] "=&v"(i1)
            : [dr0] "v"(dr0), [di0] "v"(di0), [wa0] "v"(wa0), [wb0] "v"(wb0), [dr1] "v"(dr1), [di1] "v"(di1),
              [wa1] "v"(wa1), [wb1] "v"(wb1), [off] "s"(off), [wd] "v"(wdv), [sel] "s"(sel));
}

// exact extraction of the FULL 16-bit Y = sum[off+15 : off] (round mode on 16-bit data; any t): a VOP2 shift puts re's slice into the
// low half (the high half is overwritten next), an SDWA shift writes im's slice into the high half -- 2 dot + 2 shifts per butterfly
// instead of 2 dot + 2 bfe + 1 perm, and the plain shift is in the fast-issue class.  gfx940-class hazard "VALU with dst_sel != DWORD ->
// VALU read of that VGPR needs one wait state": inside the block the second SDWA write covers the first; the s_nop covers the second.
#define INTFFT_MUL2X_W16_BODY                                                                          \
    "v_dot2_i32_i16 %[r0], %[dr0], %[wa0], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[i0], %[di0], %[wb0], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[r1], %[dr1], %[wa1], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[i1], %[di1], %[wb1], 0\n\t"                                                      \
    "v_lshrrev_b32 %[y0], %[off], %[r0]\n\t"                                                           \
    "v_lshrrev_b32 %[y1], %[off], %[r1]\n\t"                                                           \
    "v_lshrrev_b32_sdwa %[y0], %[off], %[i0] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t" \
    "v_lshrrev_b32_sdwa %[y1], %[off], %[i1] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t" \
    "s_nop 0"

template <bool SG>
__device__ __forceinline__ void mul2x_w16(u32 dr0, u32 di0, u32 wa0, u32 wb0, u32 dr1, u32 di1, u32 wa1, u32 wb1, int off, u32 &y0,
                                          u32 &y1)
{
    u32 r0, i0, r1, i1;
    if (SG)
        asm(INTFFT_MUL2X_W16_BODY
            : [y0] "=&v"(y0), [y1] "=&v"(y1), [r0] "=&v"(r0), [i0] "=&v"(i0), [r1] "=&v"(r1), [i1] "=&v"(i1)
            : [dr0] "v"(dr0), [di0] "v"(di0), [wa0] "s"(wa0), [wb0] "s"(wb0), [dr1] "v"(dr1), [di1] "v"(di1), [wa1] "s"(wa1),
              [wb1] "s"(wb1), [off] "s"(off));
    else
        asm(INTFFT_MUL2X_W16_BODY
            : [y0] "=&v"(y0), [y1] "=&v"(y1), [r0] "=&v"(r0), [i0] "=&v"(i0), [r1] "=&v"(r1), [i1] "=&v"(i1)
            : [dr0] "v"(dr0), [di0] "v"(di0), [wa0] "v"(wa0), [wb0] "v"(wb0), [dr1] "v"(dr1), [di1] "v"(di1), [wa1] "v"(wa1),
              [wb1] "v"(wb1), [off] "s"(off));
}

// exact extraction of Y >> 1 for t = 16 without v_bfe: Y >> 1 = sext(sum[30:16]) is the high half of the sum with its bit 15
// (= sum[31], which differs from sum[30] exactly when the fast path's precondition fails) replaced by bit 14.  One v_perm_b32
// takes both high halves, then P' = (P & 0x7FFF7FFF) | ((P << 1) & 0x80008000): a VOP2 shift and one v_bfi_b32 for the two
// components together -- 2 dot + 1 perm + 1 bfi + 1 shift per butterfly instead of 2 dot + 2 bfe + 1 perm.
#define INTFFT_MUL2X_T16_BODY                                                                          \
    "v_dot2_i32_i16 %[r0], %[dr0], %[wa0], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[i0], %[di0], %[wb0], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[r1], %[dr1], %[wa1], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[i1], %[di1], %[wb1], 0\n\t"                                                      \
    "v_perm_b32 %[y0], %[i0], %[r0], %[selh]\n\t"                                                      \
    "v_lshlrev_b32 %[r0], 1, %[y0]\n\t"                                                                \
    "v_perm_b32 %[y1], %[i1], %[r1], %[selh]\n\t"                                                      \
    "v_lshlrev_b32 %[r1], 1, %[y1]\n\t"                                                                \
    "v_bfi_b32 %[y0], %[msk], %[r0], %[y0]\n\t"                                                        \
    "v_bfi_b32 %[y1], %[msk], %[r1], %[y1]"

template <bool SG>
__device__ __forceinline__ void mul2x_t16(u32 dr0, u32 di0, u32 wa0, u32 wb0, u32 dr1, u32 di1, u32 wa1, u32 wb1, u32 selh, u32 &y0,
                                          u32 &y1)
{
    u32 r0, i0, r1, i1;
    const u32 msk = 0x80008000u;
    if (SG)
        asm(INTFFT_MUL2X_T16_BODY
            : [y0] "=&v"(y0), [y1] "=&v"(y1), [r0] "=&v"(r0), [i0] "=&v"(i0), [r1] "=&v"(r1), [i1] "=&v"(i1)
            : [dr0] "v"(dr0), [di0] "v"(di0), [wa0] "s"(wa0), [wb0] "s"(wb0), [dr1] "v"(dr1), [di1] "v"(di1), [wa1] "s"(wa1),
              [wb1] "s"(wb1), [selh] "s"(selh), [msk] "s"(msk));
    else
        asm(INTFFT_MUL2X_T16_BODY
            : [y0] "=&v"(y0), [y1] "=&v"(y1), [r0] "=&v"(r0), [i0] "=&v"(i0), [r1] "=&v"(r1), [i1] "=&v"(i1)
            : [dr0] "v"(dr0), [di0] "v"(di0), [wa0] "v"(wa0), [wb0] "v"(wb0), [dr1] "v"(dr1), [di1] "v"(di1), [wa1] "v"(wa1),
              [wb1] "v"(wb1), [selh] "s"(selh), [msk] "s"(msk));
}

#define INTFFT_MUL4F_BODY                                                                              \
    "v_dot2_i32_i16 %[y0], %[dr0], %[wa0], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[i0], %[di0], %[wb0], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[y1], %[dr1], %[wa1], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[i1], %[di1], %[wb1], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[y2], %[dr2], %[wa2], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[i2], %[di2], %[wb2], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[y3], %[dr3], %[wa3], 0\n\t"                                                      \
    "v_dot2_i32_i16 %[i3], %[di3], %[wb3], 0\n\t"                                                      \
    "v_perm_b32 %[y0], %[i0], %[y0], %[sel]\n\t"                                                       \
    "v_perm_b32 %[y1], %[i1], %[y1], %[sel]\n\t"                                                       \
    "v_perm_b32 %[y2], %[i2], %[y2], %[sel]\n\t"                                                       \
    "v_perm_b32 %[y3], %[i3], %[y3], %[sel]"

// fast extraction, 4 butterflies, t = 16, no 31-bit overflow: Y >> 1 = { im[31:16], re[31:16] }
template <bool SG>
__device__ __forceinline__ void mul4f(const u32 (&dr)[4], const u32 (&di)[4], const u32 (&wa)[4], const u32 (&wb)[4],
                                      u32 sel, u32 (&y)[4])
{
    u32 i0, i1, i2, i3;
    if (SG)
        asm(INTFFT_MUL4F_BODY
            : [y0] "=&v"(y[0]), [y1] "=&v"(y[1]), [y2] "=&v"(y[2]), [y3] "=&v"(y[3]), [i0] "=&v"(i0), [i1] "=&v"(i1),
              [i2] "=&v"(i2), [i3] "=&v"(i3)
            : [dr0] "v"(dr[0]), [di0] "v"(di[0]), [dr1] "v"(dr[1]), [di1] "v"(di[1]), [dr2] "v"(dr[2]), [di2] "v"(di[2]),
              [dr3] "v"(dr[3]), [di3] "v"(di[3]), [wa0] "s"(wa[0]), [wb0] "s"(wb[0]), [wa1] "s"(wa[1]), [wb1] "s"(wb[1]),
              [wa2] "s"(wa[2]), [wb2] "s"(wb[2]), [wa3] "s"(wa[3]), [wb3] "s"(wb[3]), [sel] "s"(sel));
    else
        asm(INTFFT_MUL4F_BODY
            : [y0] "=&v"(y[0]), [y1] "=&v"(y[1]), [y2] "=&v"(y[2]), [y3] "=&v"(y[3]), [i0] "=&v"(i0), [i1] "=&v"(i1),
              [i2] "=&v"(i2), [i3] "=&v"(i3)
            : [dr0] "v"(dr[0]), [di0] "v"(di[0]), [dr1] "v"(dr[1]), [di1] "v"(di[1]), [dr2] "v"(dr[2]), [di2] "v"(di[2]),
              [dr3] "v"(dr[3]), [di3] "v"(di[3]), [wa0] "v"(wa[0]), [wb0] "v"(wb[0]), [wa1] "v"(wa[1]), [wb1] "v"(wb[1]),
              [wa2] "v"(wa[2]), [wb2] "v"(wb[2]), [wa3] "v"(wa[3]), [wb3] "v"(wb[3]), [sel] "s"(sel));
}

// ---- a group of four general butterflies (a_i, b_i): a_i <- S, b_i <- cmult(D, W_i) ---------------
//   FASTX   1: fast extraction (implies truncate mode and pre-shifted outputs); 2: exact extraction specialised for t = 16
//           (mul2x_t16; kernels launched only for 16-bit twiddles use it on the frames that fail the guard test); 0: exact
//   QTURN   the twiddles are the quarter turns of the given base twiddles
//   OUT_PRE emit Y >> 1 (truncate mode)
//   SG      twiddles in SGPRs
//   PREMASK bit i: inputs of butterfly i already hold X >> 1;  VARSH: per-lane shift amount instead
//   DPK     the twiddles come in the DIT packing (wa, wb) = (Wc, Wd) = ((wr, wi), (-wi, wr)) -- what a kernel that runs BOTH
//           cores shares with its DIT half (group4_dit<.., DITPACK>).  D is then produced with its halves exchanged (free:
//           op_sel of the packed subtract) and Y.re = dot(Dsw, Wd), Y.im = dot(Dsw, Wc); a quarter turn is
//           Y.re = dot(Dsw, Wc), Y.im = dot(-Dsw, Wd): the code below with the roles of (wa, wb) exchanged
template <int ROUND, int FASTX, bool QTURN, bool OUT_PRE, bool SG, int PREMASK, bool VARSH = false, bool DPK = false>
__device__ __forceinline__ void group4(u32 &a0, u32 &b0, u32 &a1, u32 &b1, u32 &a2, u32 &b2, u32 &a3, u32 &b3,
                                       const u32 (&wa_in)[4], const u32 (&wb_in)[4], const Slice &sl, v2s shv = v2s{0, 0})
{
    static_assert(FASTX == 0 || (!ROUND && OUT_PRE), "fast extraction yields Y >> 1 only");
    static_assert(!QTURN || !SG || !ROUND, "a round-mode quarter turn negates its twiddle operand: a VGPR");
    const u32 (&wa)[4] = DPK ? wb_in : wa_in;
    const u32 (&wb)[4] = DPK ? wa_in : wb_in;
    u32 d[4];
    if (VARSH) {
        sumdiff_var<DPK>(a0, b0, shv, a0, d[0]);
        sumdiff_var<DPK>(a1, b1, shv, a1, d[1]);
        sumdiff_var<DPK>(a2, b2, shv, a2, d[2]);
        sumdiff_var<DPK>(a3, b3, shv, a3, d[3]);
    } else {
        sumdiff<(ROUND != 0), (PREMASK & 1) != 0, DPK>(a0, b0, a0, d[0]);
        sumdiff<(ROUND != 0), (PREMASK & 2) != 0, DPK>(a1, b1, a1, d[1]);
        sumdiff<(ROUND != 0), (PREMASK & 4) != 0, DPK>(a2, b2, a2, d[2]);
        sumdiff<(ROUND != 0), (PREMASK & 8) != 0, DPK>(a3, b3, a3, d[3]);
    }
    if constexpr (ROUND) {
        if constexpr (ROUND == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = wrap_w(d[i], sl.wd);
        }
    }
    u32 y[4];
    if constexpr (QTURN && ROUND != 0) {
        // RNDMODE = 1: D = rhu2(A - B) can be -2^15, so -D is not exact; the negated TWIDDLE operand is (no table entry is -2^15: the
        // planners' *_tables_ok check it): Y.re = dot(D, Wb), Y.im = dot(D, -Wa), full-width results, exact extraction
        const v2s z = {0, 0};
        const u32 nwa[4] = {as_u32(z - as_v2s(wa[0])), as_u32(z - as_v2s(wa[1])), as_u32(z - as_v2s(wa[2])), as_u32(z - as_v2s(wa[3]))};
        if constexpr (ROUND == 1) { // 16-bit data: the two-shift extraction
            mul2x_w16<SG>(d[0], d[0], wb[0], nwa[0], d[1], d[1], wb[1], nwa[1], sl.off_y, y[0], y[1]);
            mul2x_w16<SG>(d[2], d[2], wb[2], nwa[2], d[3], d[3], wb[3], nwa[3], sl.off_y, y[2], y[3]);
        } else {
            mul2x<16, SG>(d[0], d[0], wb[0], nwa[0], d[1], d[1], wb[1], nwa[1], sl.off_y, sl.sel, y[0], y[1], sl.wd);
            mul2x<16, SG>(d[2], d[2], wb[2], nwa[2], d[3], d[3], wb[3], nwa[3], sl.off_y, sl.sel, y[2], y[3], sl.wd);
        }
    } else if (QTURN) {
        // W' = (W.im, -W.re): Y.re = dot(D, Wb), Y.im = dot(-D, Wa).  (Round 4 tried the negated TWIDDLE operand instead -- one packed
        // subtract per distinct twiddle, shared by the compiler between the butterflies of a round: 49 fewer operations per thread and tile
        // in k_big2x_a, but the longer live ranges spill 4-8 VGPRs there: C4 270 against 308 Gsample/s, N = 16384 449 against 541.)
        const v2s z = {0, 0};
        const u32 n[4] = {as_u32(z - as_v2s(d[0])), as_u32(z - as_v2s(d[1])), as_u32(z - as_v2s(d[2])),
                          as_u32(z - as_v2s(d[3]))};
        if (FASTX == 1) {
            mul4f<SG>(d, n, wb, wa, sl.sel_hi, y);
        } else if (FASTX == 2) {
            mul2x_t16<SG>(d[0], n[0], wb[0], wa[0], d[1], n[1], wb[1], wa[1], sl.sel_hi, y[0], y[1]);
            mul2x_t16<SG>(d[2], n[2], wb[2], wa[2], d[3], n[3], wb[3], wa[3], sl.sel_hi, y[2], y[3]);
        } else {
            mul2x<OUT_PRE ? 15 : 16, SG>(d[0], n[0], wb[0], wa[0], d[1], n[1], wb[1], wa[1],
                                         OUT_PRE ? sl.off_y1 : sl.off_y, sl.sel, y[0], y[1], sl.wd);
            mul2x<OUT_PRE ? 15 : 16, SG>(d[2], n[2], wb[2], wa[2], d[3], n[3], wb[3], wa[3],
                                         OUT_PRE ? sl.off_y1 : sl.off_y, sl.sel, y[2], y[3], sl.wd);
        }
    } else {
        if constexpr (ROUND == 1) { // round mode on 16-bit data: full-width Y by the two-shift extraction
            mul2x_w16<SG>(d[0], d[0], wa[0], wb[0], d[1], d[1], wa[1], wb[1], sl.off_y, y[0], y[1]);
            mul2x_w16<SG>(d[2], d[2], wa[2], wb[2], d[3], d[3], wa[3], wb[3], sl.off_y, y[2], y[3]);
        } else if (FASTX == 1) {
            mul4f<SG>(d, d, wa, wb, sl.sel_hi, y);
        } else if (FASTX == 2) {
            mul2x_t16<SG>(d[0], d[0], wa[0], wb[0], d[1], d[1], wa[1], wb[1], sl.sel_hi, y[0], y[1]);
            mul2x_t16<SG>(d[2], d[2], wa[2], wb[2], d[3], d[3], wa[3], wb[3], sl.sel_hi, y[2], y[3]);
        } else {
            mul2x<OUT_PRE ? 15 : 16, SG>(d[0], d[0], wa[0], wb[0], d[1], d[1], wa[1], wb[1],
                                         OUT_PRE ? sl.off_y1 : sl.off_y, sl.sel, y[0], y[1], sl.wd);
            mul2x<OUT_PRE ? 15 : 16, SG>(d[2], d[2], wa[2], wb[2], d[3], d[3], wa[3], wb[3],
                                         OUT_PRE ? sl.off_y1 : sl.off_y, sl.sel, y[2], y[3], sl.wd);
        }
    }
    b0 = y[0];
    b1 = y[1];
    b2 = y[2];
    b3 = y[3];
}

// STAGE 0 and even positions of STAGE 1: Y = D (int_dif2_fly.vhd:245-255, :293-296)
template <bool ROUND, bool IN_PRE> __device__ __forceinline__ void bfly_triv(u32 &a, u32 &b)
{
    u32 s, d;
    sumdiff<ROUND, IN_PRE>(a, b, s, d);
    a = s;
    b = d;
}

// odd positions of STAGE 1: Y.re = D.im, Y.im = D.re >= 0 ? -D.re : ~D.re (int_dif2_fly.vhd:297-304)
template <bool ROUND, bool IN_PRE> __device__ __forceinline__ void bfly_mj(u32 &a, u32 &b)
{
    u32 s, d;
    sumdiff<ROUND, IN_PRE>(a, b, s, d);
    a = s;
    const u32 rot = __builtin_amdgcn_alignbit(d, d, 16); // lo = D.im, hi = D.re
    const u32 nx = rot ^ 0xFFFF0000u;                     // hi = ~D.re
    b = nx + ((nx >> 31) << 16);                          // + 1 in the high half iff D.re >= 0
}

// RNDMODE = 1, DIF STAGE 1 and 0 on NV registers (pairs (g, g + 2), (g + 1, g + 3), then (g, g + 1)): bfly_triv / bfly_mj with the
// w-bit wrap of the differences (wrap_w) between the rhu2 sums and the -j rotation
template <int NV, bool NARROW> __device__ __forceinline__ void round_stages10(u32 (&v)[NV], const Slice &sl)
{
#pragma unroll
    for (int g = 0; g < NV; g += 4) {
        sumdiff<true, false>(v[g], v[g + 2], v[g], v[g + 2]);
        sumdiff<true, false>(v[g + 1], v[g + 3], v[g + 1], v[g + 3]);
    }
    if constexpr (NARROW) {
#pragma unroll
        for (int g = 0; g < NV; g += 4) v[g + 2] = wrap_w(v[g + 2], sl.wd), v[g + 3] = wrap_w(v[g + 3], sl.wd);
    }
#pragma unroll
    for (int g = 0; g < NV; g += 4) { // odd positions: Y = -j D with the negation quirk (bfly_mj)
        const u32 d = v[g + 3];
        const u32 rot = __builtin_amdgcn_alignbit(d, d, 16);
        const u32 nx = rot ^ 0xFFFF0000u;
        v[g + 3] = nx + ((nx >> 31) << 16);
    }
#pragma unroll
    for (int g = 0; g < NV; g += 2) sumdiff<true, false>(v[g], v[g + 1], v[g], v[g + 1]);
    if constexpr (NARROW) {
#pragma unroll
        for (int g = 1; g < NV; g += 2) v[g] = wrap_w(v[g], sl.wd);
    }
}

// lane-half / row exchanges (gfx950 v_permlane32_swap / v_permlane16_swap).  hipcc pads the hazards it can see;
// swap_guard() covers "VALU write -> v_permlane read" for producers it cannot see (the asm multiplies above):
// one s_nop tied to the registers about to be exchanged, once per group of swaps.
__device__ __forceinline__ void swap32(u32 &a, u32 &b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ void swap16(u32 &a, u32 &b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ void swap_guard(u32 (&v)[16])
{
    asm("s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
          "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
}

// Guard-bit test of one frame (wave-uniform result).  t = x + 0x40004000 has bit 15 / bit 31 clear
// for every sample iff re in [-2^14, 2^14) and im in [-2^14 - 1, 2^14) (the low half may carry
// into the high half).  Then |z| <= 23172 for every input sample z.  Through a scaled-truncate
// stage the complex magnitude M grows by at most 1.42 (floors: <= 0.71 on S and D, twiddle
// magnitude <= 32767.71, final floor 0.71), so M <= 23187 at every butterfly input, |D| <= M + 0.71,
// and |re|, |im| of D*W are <= 23188 * 32767.71 < 2^30: bit 31 equals bit 30 in every sum, which is
// what fast extraction needs.  (The bound that would actually be needed is M <= 32752.)
// guard_acc(): per-lane accumulator; a frame is safe iff no lane of any of its waves has a flagged bit
// Narrow data (Slice::set_width): the same argument scaled to w bits -- |z| <= 2^(w-1.5), M <= that + 17 through twelve stages,
// needed M < 2^(w-1) - 16: holds for w >= 9.
__device__ __forceinline__ u32 guard_acc(const u32 (&v)[16], u32 gbias = 0x40004000u, u32 gmask = 0x80008000u)
{
    u32 acc = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc |= v[j] + gbias;
    return acc & gmask;
}
__device__ __forceinline__ bool frame_has_guard_bit(const u32 (&v)[16], u32 gbias = 0x40004000u, u32 gmask = 0x80008000u)
{
    return __builtin_amdgcn_ballot_w64(guard_acc(v, gbias, gmask) != 0) == 0;
}
// Block-wide "does any thread object?" with ONE barrier per call (HIP's __syncthreads_or is three: reset, or, read).  Three flag words
// rotate: call k uses flag k mod 3; behind its barrier thread 0 clears flag (k + 2) mod 3 -- last read behind barrier k - 1, which
// every thread has left, and first set behind barrier k + 1, which thread 0 has yet to reach -- so no reset can race a set or a read.
// flags: three dwords of LDS, zeroed (+ one __syncthreads) before the first call; `phase` is the caller's wave-uniform call counter
// mod 3 (block_any advances it).  Like __syncthreads_or the barrier also orders the caller's earlier LDS reads against its later writes.
__device__ __forceinline__ bool block_any(u32 *flags, unsigned &phase, bool mine) // (flags: derived from a __shared__ array, NOT volatile: through a volatile generic pointer the accesses become flat_* with vmcnt waits)
{
#ifdef INTFFT_VOTE_OCKL // A/B: HIP's own three-barrier reduction
    (void)flags, (void)phase;
    return __syncthreads_or(mine) != 0;
#endif
    if (mine) flags[phase] = 1u;
    // LDS-only barrier: __syncthreads() is a full workgroup fence and would also wait for the global loads the callers keep in flight
    // across the vote (the per-tile twiddle re-reads issued right behind the data loads): measured C4 288 against 305 Gsample/s
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const bool any = flags[phase] != 0u;
    const unsigned clr = phase == 0 ? 2u : phase - 1u; // (phase + 2) mod 3
    if (threadIdx.x == 0) flags[clr] = 0u;
    phase = phase == 2 ? 0u : phase + 1u;
    return any;
}
__device__ __forceinline__ void block_any_init(u32 *flags)
{
    if (threadIdx.x < 3) flags[threadIdx.x] = 0u;
    __syncthreads();
}

// input wrap to DATA_WIDTH (conv_std_logic_vector) of int16 containers that hold more than w bits: exact path of narrow plans
template <int NV> __device__ __forceinline__ void wrap_inputs(u32 (&v)[NV], int w)
{
    const short sh = (short)(16 - w);
    const v2s shv = {sh, sh};
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = as_u32((as_v2s(v[j]) << shv) >> shv);
}


// ---- reusable rounds (four stages in registers) for the wave / block kernels ----------------------
// wave-uniform twiddles of stages 3 and 2, both packings (kernel argument -> SGPRs)
struct RoundCConsts {
    u32 wa3[8], wb3[8]; // STAGE 3: table index r & 7
    u32 wa2[4], wb2[4]; // STAGE 2: table index r & 3
};

// frame-invariant per-thread twiddles of one round: stage with register offset 8 / 4 / 2 / 1
struct RoundTw {
    u32 wa8[8], wb8[8], wa4[4], wb4[4], wa2[2], wb2[2], wa1[1], wb1[1];
};

__device__ __forceinline__ constexpr int rev4c(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

// ---- DIT group of four general butterflies: a_i <- X, b_i <- Y (int_dit2_fly.vhd:142-162, 290-325) ----
// DITPACK: the twiddle operands are held in the DIT packing {Wc = (wr, wi), Wd = (-wi, wr)} (to_dit_packing below)
// instead of the DIF packing {Wa, Wb}: T.re = dot(B, Wc), T.im = dot(B, Wd) on the UNSWAPPED B -- one v_alignbit less
// per butterfly.  Kernels that run only the inverse core use it; the pair shares one DIF-packed set between its cores.
// QTURN: the twiddles are the quarter turns W' = (W.im, -W.re) of the given base twiddles: T'.re = -(T.im of the base) and
// T'.im = T.re of the base, so the re operand is the NEGATED im operand of the base (exact: no table entry is -2^15, the
// planner checks) and the im operand is the base's re operand.
template <bool FASTX, bool SG, int ROUND = 0, bool DITPACK = false, bool QTURN = false>
__device__ __forceinline__ void group4_dit(u32 &a0, u32 &b0, u32 &a1, u32 &b1, u32 &a2, u32 &b2, u32 &a3, u32 &b3,
                                           const u32 (&wa_in)[4], const u32 (&wb_in)[4], const Slice &sl)
{
    static_assert(!QTURN || !SG, "the negated operand lives in a VGPR");
    const u32 bs[4] = {DITPACK ? b0 : __builtin_amdgcn_alignbit(b0, b0, 16), DITPACK ? b1 : __builtin_amdgcn_alignbit(b1, b1, 16),
                       DITPACK ? b2 : __builtin_amdgcn_alignbit(b2, b2, 16), DITPACK ? b3 : __builtin_amdgcn_alignbit(b3, b3, 16)};
    // operand of the re / im dot product: (Wb, Wa) on the swapped B, (Wc, Wd) = (wa_in, wb_in) on the unswapped B
    const u32 (&wb0)[4] = DITPACK ? wa_in : wb_in;
    const u32 (&wa0)[4] = DITPACK ? wb_in : wa_in;
    const v2s z2 = {0, 0};
    const u32 wb[4] = {QTURN ? as_u32(z2 - as_v2s(wa0[0])) : wb0[0], QTURN ? as_u32(z2 - as_v2s(wa0[1])) : wb0[1],
                       QTURN ? as_u32(z2 - as_v2s(wa0[2])) : wb0[2], QTURN ? as_u32(z2 - as_v2s(wa0[3])) : wb0[3]};
    const u32 wa[4] = {QTURN ? wb0[0] : wa0[0], QTURN ? wb0[1] : wa0[1], QTURN ? wb0[2] : wa0[2], QTURN ? wb0[3] : wa0[3]};
    if constexpr (ROUND) { // RNDMODE = 1 (int_dit2_fly.vhd:164-217): T at full width, then rhu2(A +/- T)
        static_assert(!FASTX, "fast extraction yields T >> 1 only");
        u32 tf[4];
        if constexpr (ROUND == 1) {
            mul2x_w16<SG>(bs[0], bs[0], wb[0], wa[0], bs[1], bs[1], wb[1], wa[1], sl.off_y, tf[0], tf[1]);
            mul2x_w16<SG>(bs[2], bs[2], wb[2], wa[2], bs[3], bs[3], wb[3], wa[3], sl.off_y, tf[2], tf[3]);
        } else {
            mul2x<16, SG>(bs[0], bs[0], wb[0], wa[0], bs[1], bs[1], wb[1], wa[1], sl.off_y, sl.sel, tf[0], tf[1], sl.wd);
            mul2x<16, SG>(bs[2], bs[2], wb[2], wa[2], bs[3], bs[3], wb[3], wa[3], sl.off_y, sl.sel, tf[2], tf[3], sl.wd);
        }
        sumdiff<true, false>(a0, tf[0], a0, b0);
        sumdiff<true, false>(a1, tf[1], a1, b1);
        sumdiff<true, false>(a2, tf[2], a2, b2);
        sumdiff<true, false>(a3, tf[3], a3, b3);
        if constexpr (ROUND == 2) b0 = wrap_w(b0, sl.wd), b1 = wrap_w(b1, sl.wd), b2 = wrap_w(b2, sl.wd), b3 = wrap_w(b3, sl.wd);
        return;
    }
    u32 t[4]; // T >> 1
    if (FASTX) {
        mul4f<SG>(bs, bs, wb, wa, sl.sel_hi, t);
    } else {
        mul2x<15, SG>(bs[0], bs[0], wb[0], wa[0], bs[1], bs[1], wb[1], wa[1], sl.off_y1, sl.sel, t[0], t[1], sl.wd);
        mul2x<15, SG>(bs[2], bs[2], wb[2], wa[2], bs[3], bs[3], wb[3], wa[3], sl.off_y1, sl.sel, t[2], t[3], sl.wd);
    }
    const v2s A0 = as_v2s(a0) >> (short)1, A1 = as_v2s(a1) >> (short)1, A2 = as_v2s(a2) >> (short)1,
              A3 = as_v2s(a3) >> (short)1;
    a0 = as_u32(A0 + as_v2s(t[0]));
    b0 = as_u32(A0 - as_v2s(t[0]));
    a1 = as_u32(A1 + as_v2s(t[1]));
    b1 = as_u32(A1 - as_v2s(t[1]));
    a2 = as_u32(A2 + as_v2s(t[2]));
    b2 = as_u32(A2 - as_v2s(t[2]));
    a3 = as_u32(A3 + as_v2s(t[3]));
    b3 = as_u32(A3 - as_v2s(t[3]));
}

// DIF packing -> DIT packing of one twiddle: Wc = (wr, wi) = Wb with its halves swapped, Wd = (-wi, wr) = Wa swapped
__device__ __forceinline__ void to_dit_packing(u32 &wa, u32 &wb)
{
    const u32 c = __builtin_amdgcn_alignbit(wb, wb, 16), d = __builtin_amdgcn_alignbit(wa, wa, 16);
    wa = c;
    wb = d;
}
__device__ __forceinline__ void to_dit_packing(RoundTw &t)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) to_dit_packing(t.wa8[j], t.wb8[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) to_dit_packing(t.wa4[j], t.wb4[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) to_dit_packing(t.wa2[j], t.wb2[j]);
    to_dit_packing(t.wa1[0], t.wb1[0]);
}
// host side: the wave-uniform twiddles of stages 3 and 2 in the DIT packing
inline void to_dit_packing_host(RoundCConsts &c)
{
    auto rot = [](u32 x) { return (x >> 16) | (x << 16); };
    for (int k = 0; k < 8; ++k) {
        const u32 a = c.wa3[k], b = c.wb3[k];
        c.wa3[k] = rot(b), c.wb3[k] = rot(a);
    }
    for (int k = 0; k < 4; ++k) {
        const u32 a = c.wa2[k], b = c.wb2[k];
        c.wa2[k] = rot(b), c.wb2[k] = rot(a);
    }
}

// DIT STAGE 1, odd positions: T.im = B.re, T.re = B.im >= 0 ? -B.im : ~B.im (int_dit2_fly.vhd:264-276)
template <bool ROUND = false> __device__ __forceinline__ void bfly_pj_dit(u32 &a, u32 &b)
{
    const u32 rot = __builtin_amdgcn_alignbit(b, b, 16); // lo = B.im, hi = B.re
    const u32 nx = rot ^ 0x0000FFFFu;                     // lo = ~B.im
    const v2s add = {(short)((nx >> 15) & 1u), 0};        // + 1 in the low half iff B.im >= 0
    const u32 t = as_u32(as_v2s(nx) + add);
    sumdiff<ROUND, false>(a, t, a, b);
}

// ---- four DIF stages on register offsets 8, 4, 2, 1 (stage numbers s0+3 .. s0) -----------------------
// kinds: inputs of the first stage are S-type (unshifted) unless VARSH0 gives a per-thread shift amount.
// NS < 4 runs only the last NS stages (short frames: the leading stages belong to frame-number bits).
template <bool FASTX, bool VARSH0, int NS = 4, int ROUND = 0, bool DPK = false>
__device__ __forceinline__ void dif_round(u32 (&v)[16], const RoundTw &tw, const Slice &sl, v2s shv)
{
    if constexpr (ROUND) { // RNDMODE = 1: full-width values everywhere (no pre-shifted outputs), exact extraction
        static_assert(!FASTX, "round mode uses the exact extraction");
        if constexpr (NS >= 4) {
            const u32 wa0[4] = {tw.wa8[0], tw.wa8[1], tw.wa8[2], tw.wa8[3]}, wb0[4] = {tw.wb8[0], tw.wb8[1], tw.wb8[2], tw.wb8[3]};
            const u32 wa1[4] = {tw.wa8[4], tw.wa8[5], tw.wa8[6], tw.wa8[7]}, wb1[4] = {tw.wb8[4], tw.wb8[5], tw.wb8[6], tw.wb8[7]};
            group4<ROUND, false, false, false, false, 0, false, DPK>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa0, wb0, sl);
            group4<ROUND, false, false, false, false, 0, false, DPK>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa1, wb1, sl);
        }
        if constexpr (NS >= 3) {
            group4<ROUND, false, false, false, false, 0, false, DPK>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], tw.wa4, tw.wb4, sl);
            group4<ROUND, false, false, false, false, 0, false, DPK>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], tw.wa4, tw.wb4, sl);
        }
        if constexpr (NS >= 2) {
            const u32 wa[4] = {tw.wa2[0], tw.wa2[1], tw.wa2[0], tw.wa2[1]}, wb[4] = {tw.wb2[0], tw.wb2[1], tw.wb2[0], tw.wb2[1]};
            group4<ROUND, false, false, false, false, 0, false, DPK>(v[0], v[2], v[1], v[3], v[8], v[10], v[9], v[11], wa, wb, sl);
            group4<ROUND, false, false, false, false, 0, false, DPK>(v[4], v[6], v[5], v[7], v[12], v[14], v[13], v[15], wa, wb, sl);
        }
        if constexpr (NS >= 1) {
            const u32 wa[4] = {tw.wa1[0], tw.wa1[0], tw.wa1[0], tw.wa1[0]}, wb[4] = {tw.wb1[0], tw.wb1[0], tw.wb1[0], tw.wb1[0]};
            group4<ROUND, false, false, false, false, 0, false, DPK>(v[0], v[1], v[4], v[5], v[8], v[9], v[12], v[13], wa, wb, sl);
            group4<ROUND, false, false, false, false, 0, false, DPK>(v[2], v[3], v[6], v[7], v[10], v[11], v[14], v[15], wa, wb, sl);
        }
        return;
    }
    constexpr int M0 = 0;
    constexpr int MA4 = NS >= 4 ? 0xF : 0, MA2 = NS >= 3 ? 0xF : 0, MA1 = NS >= 2 ? 0xF : 0;
    if constexpr (NS >= 4) {
        const u32 wa0[4] = {tw.wa8[0], tw.wa8[1], tw.wa8[2], tw.wa8[3]}, wb0[4] = {tw.wb8[0], tw.wb8[1], tw.wb8[2], tw.wb8[3]};
        const u32 wa1[4] = {tw.wa8[4], tw.wa8[5], tw.wa8[6], tw.wa8[7]}, wb1[4] = {tw.wb8[4], tw.wb8[5], tw.wb8[6], tw.wb8[7]};
        group4<false, FASTX, false, true, false, M0, VARSH0, DPK>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa0, wb0, sl, shv);
        group4<false, FASTX, false, true, false, M0, VARSH0, DPK>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa1, wb1, sl, shv);
    }
    // offset 4: pairs (j, j+4); kind = j & 8
    if constexpr (NS >= 3) {
        group4<false, FASTX, false, true, false, M0, false, DPK>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], tw.wa4, tw.wb4, sl);
        group4<false, FASTX, false, true, false, MA4, false, DPK>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], tw.wa4, tw.wb4, sl);
    }
    // offset 2: pairs (j, j+2); twiddle j & 1; kind = j & 4
    if constexpr (NS >= 2) {
        const u32 wa[4] = {tw.wa2[0], tw.wa2[1], tw.wa2[0], tw.wa2[1]}, wb[4] = {tw.wb2[0], tw.wb2[1], tw.wb2[0], tw.wb2[1]};
        group4<false, FASTX, false, true, false, M0, false, DPK>(v[0], v[2], v[1], v[3], v[8], v[10], v[9], v[11], wa, wb, sl);
        group4<false, FASTX, false, true, false, MA2, false, DPK>(v[4], v[6], v[5], v[7], v[12], v[14], v[13], v[15], wa, wb, sl);
    }
    // offset 1: pairs (j, j+1); kind = j & 2
    if constexpr (NS >= 1) {
        const u32 wa[4] = {tw.wa1[0], tw.wa1[0], tw.wa1[0], tw.wa1[0]}, wb[4] = {tw.wb1[0], tw.wb1[0], tw.wb1[0], tw.wb1[0]};
        group4<false, FASTX, false, true, false, M0, false, DPK>(v[0], v[1], v[4], v[5], v[8], v[9], v[12], v[13], wa, wb, sl);
        group4<false, FASTX, false, true, false, MA1, false, DPK>(v[2], v[3], v[6], v[7], v[10], v[11], v[14], v[15], wa, wb, sl);
    }
}

// ---- 32-register rounds (five DIF stages per thread) with quarter-turn twiddle sharing ---------------------------------------
// The second half of a stage's table is the quarter turn of the first, W[k + 2^(s-1)] = (W[k].im, -W[k].re)
// (rom_twiddle_int.vhd:177-183; it holds for the Taylor stages too -- the planner verifies it on the generated tables
// before choosing these kernels), so a round keeps only the BASE twiddles: 8 for the top stage (register offset 16),
// then 4 / 2 / 1 / 1 for the offsets 8 / 4 / 2 / 1.
struct RoundTwQ {
    u32 wa8[4], wb8[4], wa4[2], wb4[2], wa2[1], wb2[1], wa1[1], wb1[1];
};

// offset 16: pairs (j, j + 16), twiddle index j (0..15): base for j < 8, quarter turn of base[j - 8] for j >= 8.
// P0: PREMASK of the inputs (0 unshifted, 0xF already X >> 1), or VARSH with a per-thread shift amount.
template <bool FASTX, int P0, bool VARSH, int ROUND = 0>
__device__ __forceinline__ void dif_top16(u32 (&v)[32], const u32 (&wa)[8], const u32 (&wb)[8], const Slice &sl, v2s shv)
{
    const u32 wa0[4] = {wa[0], wa[1], wa[2], wa[3]}, wb0[4] = {wb[0], wb[1], wb[2], wb[3]};
    const u32 wa1[4] = {wa[4], wa[5], wa[6], wa[7]}, wb1[4] = {wb[4], wb[5], wb[6], wb[7]};
    if constexpr (ROUND != 0) { // RNDMODE = 1: plain values everywhere (no kinds, no pre-shifted outputs), exact extraction
        static_assert(!FASTX, "round mode uses the exact extraction");
        group4<ROUND, 0, false, false, false, 0>(v[0], v[16], v[1], v[17], v[2], v[18], v[3], v[19], wa0, wb0, sl);
        group4<ROUND, 0, false, false, false, 0>(v[4], v[20], v[5], v[21], v[6], v[22], v[7], v[23], wa1, wb1, sl);
        group4<ROUND, 0, true, false, false, 0>(v[8], v[24], v[9], v[25], v[10], v[26], v[11], v[27], wa0, wb0, sl);
        group4<ROUND, 0, true, false, false, 0>(v[12], v[28], v[13], v[29], v[14], v[30], v[15], v[31], wa1, wb1, sl);
        return;
    }
    group4<false, FASTX, false, true, false, P0, VARSH>(v[0], v[16], v[1], v[17], v[2], v[18], v[3], v[19], wa0, wb0, sl, shv);
    group4<false, FASTX, false, true, false, P0, VARSH>(v[4], v[20], v[5], v[21], v[6], v[22], v[7], v[23], wa1, wb1, sl, shv);
    group4<false, FASTX, true, true, false, P0, VARSH>(v[8], v[24], v[9], v[25], v[10], v[26], v[11], v[27], wa0, wb0, sl, shv);
    group4<false, FASTX, true, true, false, P0, VARSH>(v[12], v[28], v[13], v[29], v[14], v[30], v[15], v[31], wa1, wb1, sl, shv);
}

// four DIF stages on registers v[B .. B+15], offsets 8, 4, 2, 1 (stage numbers s0+3 .. s0).  NS < 4 runs only the last NS.
// The inputs of the FIRST executed stage: PREMASK P0 (0 / 0xF) or VARSH0 (per-thread shift); later stages follow the
// rule "the upper output of a butterfly is Y >> 1".
template <bool FASTX, int B, int P0, bool VARSH0, int NS = 4, int ROUND = 0>
__device__ __forceinline__ void dif_round_q(u32 (&v)[32], const RoundTwQ &t, const Slice &sl, v2s shv)
{
    const v2s none = {0, 0};
    if constexpr (ROUND != 0) { // RNDMODE = 1: the same pairings and twiddle sharing on plain values
        static_assert(!FASTX, "round mode uses the exact extraction");
        if constexpr (NS >= 4) {
            group4<ROUND, 0, false, false, false, 0>(v[B + 0], v[B + 8], v[B + 1], v[B + 9], v[B + 2], v[B + 10], v[B + 3], v[B + 11], t.wa8, t.wb8, sl);
            group4<ROUND, 0, true, false, false, 0>(v[B + 4], v[B + 12], v[B + 5], v[B + 13], v[B + 6], v[B + 14], v[B + 7], v[B + 15], t.wa8, t.wb8, sl);
        }
        if constexpr (NS >= 3) {
            const u32 wa[4] = {t.wa4[0], t.wa4[1], t.wa4[0], t.wa4[1]}, wb[4] = {t.wb4[0], t.wb4[1], t.wb4[0], t.wb4[1]};
            group4<ROUND, 0, false, false, false, 0>(v[B + 0], v[B + 4], v[B + 1], v[B + 5], v[B + 8], v[B + 12], v[B + 9], v[B + 13], wa, wb, sl);
            group4<ROUND, 0, true, false, false, 0>(v[B + 2], v[B + 6], v[B + 3], v[B + 7], v[B + 10], v[B + 14], v[B + 11], v[B + 15], wa, wb, sl);
        }
        if constexpr (NS >= 2) {
            const u32 wa[4] = {t.wa2[0], t.wa2[0], t.wa2[0], t.wa2[0]}, wb[4] = {t.wb2[0], t.wb2[0], t.wb2[0], t.wb2[0]};
            group4<ROUND, 0, false, false, false, 0>(v[B + 0], v[B + 2], v[B + 4], v[B + 6], v[B + 8], v[B + 10], v[B + 12], v[B + 14], wa, wb, sl);
            group4<ROUND, 0, true, false, false, 0>(v[B + 1], v[B + 3], v[B + 5], v[B + 7], v[B + 9], v[B + 11], v[B + 13], v[B + 15], wa, wb, sl);
        }
        if constexpr (NS >= 1) {
            const u32 wa[4] = {t.wa1[0], t.wa1[0], t.wa1[0], t.wa1[0]}, wb[4] = {t.wb1[0], t.wb1[0], t.wb1[0], t.wb1[0]};
            group4<ROUND, 0, false, false, false, 0>(v[B + 0], v[B + 1], v[B + 4], v[B + 5], v[B + 8], v[B + 9], v[B + 12], v[B + 13], wa, wb, sl);
            group4<ROUND, 0, false, false, false, 0>(v[B + 2], v[B + 3], v[B + 6], v[B + 7], v[B + 10], v[B + 11], v[B + 14], v[B + 15], wa, wb, sl);
        }
        (void)shv;
        return;
    }
    if constexpr (NS >= 4) { // offset 8: twiddle j & 7: base j < 4, quarter turn j >= 4
        group4<false, FASTX, false, true, false, P0, VARSH0>(v[B + 0], v[B + 8], v[B + 1], v[B + 9], v[B + 2], v[B + 10], v[B + 3], v[B + 11], t.wa8, t.wb8, sl, shv);
        group4<false, FASTX, true, true, false, P0, VARSH0>(v[B + 4], v[B + 12], v[B + 5], v[B + 13], v[B + 6], v[B + 14], v[B + 7], v[B + 15], t.wa8, t.wb8, sl, shv);
    }
    if constexpr (NS >= 3) { // offset 4: pairs (j, j + 4), j in {0..3, 8..11}; twiddle j & 3: base 0, 1; quarter turn 2, 3
        constexpr bool first = NS == 3;
        constexpr int PM = first ? P0 : 0xC; // butterflies 2, 3 of each group come from the upper half of the offset-8 stage
        const u32 wa[4] = {t.wa4[0], t.wa4[1], t.wa4[0], t.wa4[1]}, wb[4] = {t.wb4[0], t.wb4[1], t.wb4[0], t.wb4[1]};
        group4<false, FASTX, false, true, false, PM, first && VARSH0>(v[B + 0], v[B + 4], v[B + 1], v[B + 5], v[B + 8], v[B + 12], v[B + 9], v[B + 13], wa, wb, sl, first ? shv : none);
        group4<false, FASTX, true, true, false, PM, first && VARSH0>(v[B + 2], v[B + 6], v[B + 3], v[B + 7], v[B + 10], v[B + 14], v[B + 11], v[B + 15], wa, wb, sl, first ? shv : none);
    }
    if constexpr (NS >= 2) { // offset 2: pairs (j, j + 2), j in {0,1,4,5,8,9,12,13}; twiddle j & 1: base 0; quarter turn 1
        constexpr bool first = NS == 2;
        constexpr int PM = first ? P0 : 0xA; // j = 4, 12 (butterflies 1, 3) are upper outputs of the offset-4 stage
        const u32 wa[4] = {t.wa2[0], t.wa2[0], t.wa2[0], t.wa2[0]}, wb[4] = {t.wb2[0], t.wb2[0], t.wb2[0], t.wb2[0]};
        group4<false, FASTX, false, true, false, PM, first && VARSH0>(v[B + 0], v[B + 2], v[B + 4], v[B + 6], v[B + 8], v[B + 10], v[B + 12], v[B + 14], wa, wb, sl, first ? shv : none);
        group4<false, FASTX, true, true, false, PM, first && VARSH0>(v[B + 1], v[B + 3], v[B + 5], v[B + 7], v[B + 9], v[B + 11], v[B + 13], v[B + 15], wa, wb, sl, first ? shv : none);
    }
    if constexpr (NS >= 1) { // offset 1: pairs (j, j + 1), j even; one twiddle; j & 2 = upper output of the offset-2 stage
        constexpr bool first = NS == 1;
        const u32 wa[4] = {t.wa1[0], t.wa1[0], t.wa1[0], t.wa1[0]}, wb[4] = {t.wb1[0], t.wb1[0], t.wb1[0], t.wb1[0]};
        group4<false, FASTX, false, true, false, first ? P0 : 0, first && VARSH0>(v[B + 0], v[B + 1], v[B + 4], v[B + 5], v[B + 8], v[B + 9], v[B + 12], v[B + 13], wa, wb, sl, first ? shv : none);
        group4<false, FASTX, false, true, false, first ? P0 : 0xF, first && VARSH0>(v[B + 2], v[B + 3], v[B + 6], v[B + 7], v[B + 10], v[B + 11], v[B + 14], v[B + 15], wa, wb, sl, first ? shv : none);
    }
}

// ---- 32-register DIT rounds with quarter-turn sharing (mirror of dif_round_q / dif_top16); twiddles in the DIT packing ----
// four DIT stages on registers v[B .. B+15], offsets 1, 2, 4, 8 (stage numbers s0 .. s0+3)
template <bool FASTX, int B, int ROUND = 0>
__device__ __forceinline__ void dit_round_q(u32 (&v)[32], const RoundTwQ &t, const Slice &sl)
{
    { // offset 1: one twiddle
        const u32 wa[4] = {t.wa1[0], t.wa1[0], t.wa1[0], t.wa1[0]}, wb[4] = {t.wb1[0], t.wb1[0], t.wb1[0], t.wb1[0]};
        group4_dit<FASTX, false, ROUND, true>(v[B + 0], v[B + 1], v[B + 2], v[B + 3], v[B + 4], v[B + 5], v[B + 6], v[B + 7], wa, wb, sl);
        group4_dit<FASTX, false, ROUND, true>(v[B + 8], v[B + 9], v[B + 10], v[B + 11], v[B + 12], v[B + 13], v[B + 14], v[B + 15], wa, wb, sl);
    }
    { // offset 2: twiddle index j & 1: base 0, quarter turn 1
        const u32 wa[4] = {t.wa2[0], t.wa2[0], t.wa2[0], t.wa2[0]}, wb[4] = {t.wb2[0], t.wb2[0], t.wb2[0], t.wb2[0]};
        group4_dit<FASTX, false, ROUND, true>(v[B + 0], v[B + 2], v[B + 4], v[B + 6], v[B + 8], v[B + 10], v[B + 12], v[B + 14], wa, wb, sl);
        group4_dit<FASTX, false, ROUND, true, true>(v[B + 1], v[B + 3], v[B + 5], v[B + 7], v[B + 9], v[B + 11], v[B + 13], v[B + 15], wa, wb, sl);
    }
    { // offset 4: twiddle index j & 3: base 0, 1; quarter turn 2, 3
        const u32 wa[4] = {t.wa4[0], t.wa4[1], t.wa4[0], t.wa4[1]}, wb[4] = {t.wb4[0], t.wb4[1], t.wb4[0], t.wb4[1]};
        group4_dit<FASTX, false, ROUND, true>(v[B + 0], v[B + 4], v[B + 1], v[B + 5], v[B + 8], v[B + 12], v[B + 9], v[B + 13], wa, wb, sl);
        group4_dit<FASTX, false, ROUND, true, true>(v[B + 2], v[B + 6], v[B + 3], v[B + 7], v[B + 10], v[B + 14], v[B + 11], v[B + 15], wa, wb, sl);
    }
    { // offset 8: twiddle index j & 7: base j < 4, quarter turn j >= 4
        group4_dit<FASTX, false, ROUND, true>(v[B + 0], v[B + 8], v[B + 1], v[B + 9], v[B + 2], v[B + 10], v[B + 3], v[B + 11], t.wa8, t.wb8, sl);
        group4_dit<FASTX, false, ROUND, true, true>(v[B + 4], v[B + 12], v[B + 5], v[B + 13], v[B + 6], v[B + 14], v[B + 7], v[B + 15], t.wa8, t.wb8, sl);
    }
}
// offset 16: pairs (j, j + 16), twiddle index j: base for j < 8, quarter turn of base[j - 8] for j >= 8
template <bool FASTX, int ROUND = 0>
__device__ __forceinline__ void dit_top16(u32 (&v)[32], const u32 (&wa)[8], const u32 (&wb)[8], const Slice &sl)
{
    const u32 wa0[4] = {wa[0], wa[1], wa[2], wa[3]}, wb0[4] = {wb[0], wb[1], wb[2], wb[3]};
    const u32 wa1[4] = {wa[4], wa[5], wa[6], wa[7]}, wb1[4] = {wb[4], wb[5], wb[6], wb[7]};
    group4_dit<FASTX, false, ROUND, true>(v[0], v[16], v[1], v[17], v[2], v[18], v[3], v[19], wa0, wb0, sl);
    group4_dit<FASTX, false, ROUND, true>(v[4], v[20], v[5], v[21], v[6], v[22], v[7], v[23], wa1, wb1, sl);
    group4_dit<FASTX, false, ROUND, true, true>(v[8], v[24], v[9], v[25], v[10], v[26], v[11], v[27], wa0, wb0, sl);
    group4_dit<FASTX, false, ROUND, true, true>(v[12], v[28], v[13], v[29], v[14], v[30], v[15], v[31], wa1, wb1, sl);
}

// ---- four DIT stages on register offsets 1, 2, 4, 8 (NS < 4: only the first NS) ---------------------
template <bool FASTX, int NS = 4, int ROUND = 0, bool DITPACK = false>
__device__ __forceinline__ void dit_round(u32 (&v)[16], const RoundTw &tw, const Slice &sl)
{
    if constexpr (NS >= 1) {
        const u32 wa[4] = {tw.wa1[0], tw.wa1[0], tw.wa1[0], tw.wa1[0]}, wb[4] = {tw.wb1[0], tw.wb1[0], tw.wb1[0], tw.wb1[0]};
        group4_dit<FASTX, false, ROUND, DITPACK>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], wa, wb, sl);
        group4_dit<FASTX, false, ROUND, DITPACK>(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], wa, wb, sl);
    }
    if constexpr (NS >= 2) {
        const u32 wa[4] = {tw.wa2[0], tw.wa2[1], tw.wa2[0], tw.wa2[1]}, wb[4] = {tw.wb2[0], tw.wb2[1], tw.wb2[0], tw.wb2[1]};
        group4_dit<FASTX, false, ROUND, DITPACK>(v[0], v[2], v[1], v[3], v[4], v[6], v[5], v[7], wa, wb, sl);
        group4_dit<FASTX, false, ROUND, DITPACK>(v[8], v[10], v[9], v[11], v[12], v[14], v[13], v[15], wa, wb, sl);
    }
    if constexpr (NS >= 3) {
        group4_dit<FASTX, false, ROUND, DITPACK>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], tw.wa4, tw.wb4, sl);
        group4_dit<FASTX, false, ROUND, DITPACK>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], tw.wa4, tw.wb4, sl);
    }
    if constexpr (NS >= 4) {
        const u32 wa0[4] = {tw.wa8[0], tw.wa8[1], tw.wa8[2], tw.wa8[3]}, wb0[4] = {tw.wb8[0], tw.wb8[1], tw.wb8[2], tw.wb8[3]};
        const u32 wa1[4] = {tw.wa8[4], tw.wa8[5], tw.wa8[6], tw.wa8[7]}, wb1[4] = {tw.wb8[4], tw.wb8[5], tw.wb8[6], tw.wb8[7]};
        group4_dit<FASTX, false, ROUND, DITPACK>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa0, wb0, sl);
        group4_dit<FASTX, false, ROUND, DITPACK>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa1, wb1, sl);
    }
}

// ---- round C: DIF stages 3,2,1,0 / DIT stages 0,1,2,3 on reg = n3..0, uniform twiddles ----------------
template <bool FASTX, int ROUND = 0, bool DPK = false> __device__ __forceinline__ void dif_round_c(u32 (&v)[16], const RoundCConsts &c, const Slice &sl, v2s shv)
{
    const u32 wa0[4] = {c.wa3[0], c.wa3[1], c.wa3[2], c.wa3[3]}, wb0[4] = {c.wb3[0], c.wb3[1], c.wb3[2], c.wb3[3]};
    const u32 wa1[4] = {c.wa3[4], c.wa3[5], c.wa3[6], c.wa3[7]}, wb1[4] = {c.wb3[4], c.wb3[5], c.wb3[6], c.wb3[7]};
    if constexpr (ROUND) { // RNDMODE = 1: full-width values, exact extraction; STAGE 1 / 0 on rhu2 sums (int_dif2_fly.vhd:167-219)
        static_assert(!FASTX, "round mode uses the exact extraction");
        group4<ROUND, false, false, false, true, 0, false, DPK>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa0, wb0, sl);
        group4<ROUND, false, false, false, true, 0, false, DPK>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa1, wb1, sl);
        group4<ROUND, false, false, false, true, 0, false, DPK>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], c.wa2, c.wb2, sl);
        group4<ROUND, false, false, false, true, 0, false, DPK>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], c.wa2, c.wb2, sl);
        round_stages10<16, ROUND == 2>(v, sl);
        return;
    }
    group4<false, FASTX, false, true, true, 0, true, DPK>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa0, wb0, sl, shv);
    group4<false, FASTX, false, true, true, 0, true, DPK>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa1, wb1, sl, shv);
    group4<false, FASTX, false, true, true, 0, false, DPK>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], c.wa2, c.wb2, sl);
    group4<false, FASTX, false, true, true, 0xF, false, DPK>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], c.wa2, c.wb2, sl);
#pragma unroll
    for (int g = 0; g < 16; g += 8) { // stage 1: kind = r & 4
        bfly_triv<false, false>(v[g], v[g + 2]);
        bfly_mj<false, false>(v[g + 1], v[g + 3]);
        bfly_triv<false, true>(v[g + 4], v[g + 6]);
        bfly_mj<false, true>(v[g + 5], v[g + 7]);
    }
#pragma unroll
    for (int g = 0; g < 16; g += 2) bfly_triv<false, false>(v[g], v[g + 1]);
}

template <bool FASTX, int ROUND = 0, bool DITPACK = false> __device__ __forceinline__ void dit_round_c(u32 (&v)[16], const RoundCConsts &c, const Slice &sl)
{
#pragma unroll
    for (int g = 0; g < 16; g += 2) bfly_triv<(ROUND != 0), false>(v[g], v[g + 1]); // STAGE 0: T = B
    if constexpr (ROUND) {
        if constexpr (ROUND == 2) {
#pragma unroll
            for (int g = 1; g < 16; g += 2) v[g] = wrap_w(v[g], sl.wd);
        }
    }
#pragma unroll
    for (int g = 0; g < 16; g += 4) { // STAGE 1: even positions T = B, odd positions T = +j B (quirk)
        bfly_triv<(ROUND != 0), false>(v[g], v[g + 2]);
        bfly_pj_dit<(ROUND != 0)>(v[g + 1], v[g + 3]);
    }
    if constexpr (ROUND) {
        if constexpr (ROUND == 2) {
#pragma unroll
            for (int g = 0; g < 16; g += 4) v[g + 2] = wrap_w(v[g + 2], sl.wd), v[g + 3] = wrap_w(v[g + 3], sl.wd);
        }
    }
    group4_dit<FASTX, true, ROUND, DITPACK>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], c.wa2, c.wb2, sl);
    group4_dit<FASTX, true, ROUND, DITPACK>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], c.wa2, c.wb2, sl);
    const u32 wa0[4] = {c.wa3[0], c.wa3[1], c.wa3[2], c.wa3[3]}, wb0[4] = {c.wb3[0], c.wb3[1], c.wb3[2], c.wb3[3]};
    const u32 wa1[4] = {c.wa3[4], c.wa3[5], c.wa3[6], c.wa3[7]}, wb1[4] = {c.wb3[4], c.wb3[5], c.wb3[6], c.wb3[7]};
    group4_dit<FASTX, true, ROUND, DITPACK>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa0, wb0, sl);
    group4_dit<FASTX, true, ROUND, DITPACK>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa1, wb1, sl);
}


} // namespace intfft
