// intfft_u32.hpp -- unpacked int32 butterflies and the forward transform of the unscaled wave kernels
// (intfft_fast1024u.hip: int_fftNk; intfft_fast1024ux.hip: int_ifftNk and the pair).  See intfft_fast1024u.hip
// for the arithmetic and the reference citations.
#pragma once
#include "intfft_internal.hpp"

#include <cstdlib>

namespace intfft {

using u32 = uint32_t;

struct UConsts {
    int wr3[8], wi3[8]; // STAGE 3 twiddles (uniform)
    int wr2[4], wi2[4]; // STAGE 2
};

// Short frames (L < 10), 8-byte outputs: ONE lane swap after the last stage (lane bit 5 = a(L-1) <-> reg bit 3)
// gives every lane two consecutive outputs = one dwordx4 store, and lane bits 0, 1, .. carry a(L-2), a(L-3), ..
// (output bits 1, 2, ..) so that adjacent lanes write adjacent 16-byte pieces; then the frame bits.
template <int L> __host__ __device__ constexpr int lane_bit_u(int k)
{
    if (L == 10) return 9 - k;
    if (k == L - 1) return 5;
    if (k < L - 1) return (L - 2) - k;
    return (L - 5) + (k - L);
}

// LC lane mapping of the transform: MAP 0 = lane_bit_u<L> (forward kernel, store-optimised), MAP 1 = natural
// (lane bit i = a(9-i): the pair, which never does I/O from LC), MAP 2 = lane_bit<L> (inverse alone: the mirror
// of the packed kernels' short-frame store, intfft_fast1024.hip)
template <int L, int MAP> __host__ __device__ constexpr int ulb(int k)
{
    return MAP == 0 ? lane_bit_u<L>(k) : MAP == 1 ? 9 - k : lane_bit<L>(k);
}

constexpr int ROWU = 20; // LDS row stride in dwords, per plane (re plane then im plane)

// one general DIF butterfly, unscaled; WO = output width of the stage
template <bool WRAP, int WO, bool UNIFORM_W = false>
__device__ __forceinline__ void ufly(int &are, int &aim, int &bre, int &bim, int wr, int wi, int sh)
{
    // wave-uniform twiddles stay in SGPRs; the empty asm keeps the compiler from hoisting their 64-bit sign
    // extension out of the frame loop (which turns every product into a 3-instruction 64 x 32 multiply)
    if (UNIFORM_W) asm volatile("" : "+s"(wr), "+s"(wi));
    else asm volatile("" : "+v"(wr), "+v"(wi)); // per-lane twiddles: same hazard (seen in the DIT kernels)
    const int dre = are - bre, dim = aim - bim, ndim = bim - aim;
    are += bre;
    aim += bim;
    const long long xr = (long long)dre * wr + (long long)ndim * wi; // M2 - M1 (int_cmult_dsp48.vhd:192-207)
    const long long xi = (long long)dre * wi + (long long)dim * wr;  // M2 + M1 (:209-224)
    // low 32 bits of (x >> sh), 1 <= sh <= 15: one v_alignbit_b32
    int yr = (int)__builtin_amdgcn_alignbit((u32)(xr >> 32), (u32)xr, (u32)sh);
    int yi = (int)__builtin_amdgcn_alignbit((u32)(xi >> 32), (u32)xi, (u32)sh);
    if (WRAP) {
        yr = __builtin_amdgcn_sbfe(yr, 0, WO);
        yi = __builtin_amdgcn_sbfe(yi, 0, WO);
    }
    bre = yr;
    bim = yi;
}
// STAGE 0 and even positions of STAGE 1
__device__ __forceinline__ void ufly_triv(int &are, int &aim, int &bre, int &bim)
{
    const int dre = are - bre, dim = aim - bim;
    are += bre;
    aim += bim;
    bre = dre;
    bim = dim;
}
// odd positions of STAGE 1: Y.re = D.im, Y.im = D.re >= 0 ? -D.re : ~D.re  (int_dif2_fly.vhd:297-304)
__device__ __forceinline__ void ufly_mj(int &are, int &aim, int &bre, int &bim)
{
    const int dre = are - bre, dim = aim - bim;
    are += bre;
    aim += bim;
    bre = dim;
    bim = (dre >> 31) - dre; // -x for x >= 0, -x - 1 = ~x for x < 0
}

__device__ __forceinline__ void uswap32(int &a, int &b)
{
    const auto r = __builtin_amdgcn_permlane32_swap((u32)a, (u32)b, false, false);
    a = (int)r[0];
    b = (int)r[1];
}
__device__ __forceinline__ void uswap16(int &a, int &b)
{
    const auto r = __builtin_amdgcn_permlane16_swap((u32)a, (u32)b, false, false);
    a = (int)r[0];
    b = (int)r[1];
}

template <int L, bool WRAP, int MAP = 0>
__device__ __forceinline__ void utransform(int (&re)[16], int (&im)[16], const int (&w9r)[8], const int (&w9i)[8],
                                           const int (&w8r)[4], const int (&w8i)[4], const int (&w7r)[2],
                                           const int (&w7i)[2], int w6r, int w6i, int w5r, int w5i, int w4r, int w4i,
                                           const UConsts &c, int sh, u32 *wr_base, const uint4 *rd_base)
{
    // STAGE s is stage ii = L - 1 - s of the pipeline: output width 17 + ii = 16 + L - s (int_fftNk.vhd:187-207)
    if constexpr (L >= 10) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ufly<WRAP, 16 + L - 9>(re[j], im[j], re[j + 8], im[j + 8], w9r[j], w9i[j], sh);
    }
    if constexpr (L >= 9) {
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                ufly<WRAP, 16 + L - 8>(re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w8r[j], w8i[j], sh);
    }
    if constexpr (L >= 8) {
#pragma unroll
        for (int g = 0; g < 16; g += 4)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                ufly<WRAP, 16 + L - 7>(re[g + j], im[g + j], re[g + j + 2], im[g + j + 2], w7r[j], w7i[j], sh);
    }
    if constexpr (L >= 7) {
#pragma unroll
        for (int g = 0; g < 16; g += 2) ufly<WRAP, 16 + L - 6>(re[g], im[g], re[g + 1], im[g + 1], w6r, w6i, sh);
    }
    // lane bit 5 <-> reg bit 3, stage 5
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uswap32(re[j], re[j + 8]);
        uswap32(im[j], im[j + 8]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) ufly<WRAP, 16 + L - 5>(re[j], im[j], re[j + 8], im[j + 8], w5r, w5i, sh);
    // lane bit 4 <-> reg bit 2, stage 4
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uswap16(re[g + j], re[g + j + 4]);
            uswap16(im[g + j], im[g + j + 4]);
        }
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int j = 0; j < 4; ++j) ufly<WRAP, 16 + L - 4>(re[g + j], im[g + j], re[g + j + 4], im[g + j + 4], w4r, w4i, sh);

    // LDS transpose (re plane, im plane): regs become n3..0, lane bit i = n(9-i)
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int j0 = j & 1, j1 = (j >> 1) & 1, j2 = (j >> 2) & 1, j3 = (j >> 3) & 1;
        // reg j3 = a5, j2 = a4, j1 = a7, j0 = a6 here (intfft_fast1024.hip)
        const int row_j = (j1 << ulb<L, MAP>(7)) + (j0 << ulb<L, MAP>(6)) + (j3 << ulb<L, MAP>(5)) + (j2 << ulb<L, MAP>(4));
        wr_base[ROWU * row_j] = (u32)re[j];
        wr_base[64 * ROWU + ROWU * row_j] = (u32)im[j];
    }
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 x = rd_base[q], y = rd_base[q + 16 * ROWU];
        re[4 * q + 0] = (int)x.x;
        re[4 * q + 1] = (int)x.y;
        re[4 * q + 2] = (int)x.z;
        re[4 * q + 3] = (int)x.w;
        im[4 * q + 0] = (int)y.x;
        im[4 * q + 1] = (int)y.y;
        im[4 * q + 2] = (int)y.z;
        im[4 * q + 3] = (int)y.w;
    }
    wave_lds_fence();

    // stages 3, 2 (uniform twiddles), 1, 0
#pragma unroll
    for (int r = 0; r < 8; ++r) ufly<WRAP, 16 + L - 3, true>(re[r], im[r], re[r + 8], im[r + 8], c.wr3[r], c.wi3[r], sh);
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int r = 0; r < 4; ++r) ufly<WRAP, 16 + L - 2, true>(re[g + r], im[g + r], re[g + r + 4], im[g + r + 4], c.wr2[r], c.wi2[r], sh);
#pragma unroll
    for (int g = 0; g < 16; g += 4) {
        ufly_triv(re[g], im[g], re[g + 2], im[g + 2]);
        ufly_mj(re[g + 1], im[g + 1], re[g + 3], im[g + 3]);
    }
#pragma unroll
    for (int g = 0; g < 16; g += 2) ufly_triv(re[g], im[g], re[g + 1], im[g + 1]);
}

// ---- general-width butterflies (intfft_fastw32.hip, intfft_fast4096w.hip) ----------------------------------
enum { W_TRUNC = 0, W_ROUND = 1, W_UNSCALED = 2 };

// MASKED = false: every stage of the plan is in a single-DSP regime (a = 0): exact sum first, chained v_mad_i64_i32
template <int MODE, bool UNIFORM_W = false, bool MASKED = true>
__device__ __forceinline__ void gfly(int &are, int &aim, int &bre, int &bim, int wr, int wi, const W32Stage &s)
{
    if (UNIFORM_W) asm volatile("" : "+s"(wr), "+s"(wi)); // see intfft_fast1024u.hip
    else asm volatile("" : "+v"(wr), "+v"(wi));            // same for per-lane twiddles (see gfly_dit)
    int dre, dim;
    if (MODE == W_UNSCALED) {
        dre = are - bre, dim = aim - bim;
        are += bre, aim += bim;
    } else if (MODE == W_TRUNC) {
        const int ar = are >> 1, ai = aim >> 1, br = bre >> 1, bi = bim >> 1;
        dre = ar - br, dim = ai - bi;
        are = ar + br, aim = ai + bi;
    } else { // rhu2(A +/- B) on the exact sum, wrapped to DTW bits (int_dif2_fly.vhd:173-218)
        // rhu2(A + B) = (A | B) - ((A ^ B) >> 1): always inside the DTW bits of its operands, no wrap; rhu2(A - B) = rhu2(A + B) - B, which
        // leaves the range only for A = max, B = min: the RTL's wrap (derivation: sumdiff, intfft_pk16.hpp).  7 operations per component
        // instead of 17, all of them in the fast-issue VOP2 class.
#ifdef INTFFT_RHU2_LONG // (A/B: the form of rounds 1-5)
        const int ar = are >> 1, ai = aim >> 1, br = bre >> 1, bi = bim >> 1;
        dre = (int)((u32)(ar - br + (are & ~bre & 1)) << s.wosh) >> s.wosh;
        dim = (int)((u32)(ai - bi + (aim & ~bim & 1)) << s.wosh) >> s.wosh;
        are = (int)((u32)(ar + br + ((are | bre) & 1)) << s.wosh) >> s.wosh;
        aim = (int)((u32)(ai + bi + ((aim | bim) & 1)) << s.wosh) >> s.wosh;
#else
        const int sr = (are | bre) - ((are ^ bre) >> 1), si = (aim | bim) - ((aim ^ bim) >> 1);
        dre = (int)((u32)(sr - bre) << s.wosh) >> s.wosh;
        dim = (int)((u32)(si - bim) << s.wosh) >> s.wosh;
        are = sr, aim = si;
#endif
    }
    unsigned long long xr, xi;
    if (MASKED) {
        const unsigned long long m2r = (unsigned long long)((long long)dre * wr), m1r = (unsigned long long)((long long)dim * wi);
        const unsigned long long m2i = (unsigned long long)((long long)dre * wi), m1i = (unsigned long long)((long long)dim * wr);
        const unsigned long long k = 0xFFFFFFFF00000000ull | s.keep;
        xr = (m2r & k) - (m1r & k), xi = (m2i & k) + (m1i & k);
    } else {
        const int ndim = -dim;
        xr = (unsigned long long)((long long)dre * wr + (long long)ndim * wi);
        xi = (unsigned long long)((long long)dre * wi + (long long)dim * wr);
    }
    bre = (int)(__builtin_amdgcn_alignbit((u32)(xr >> 32), (u32)xr, (u32)s.sh) << s.wsh) >> s.wsh;
    bim = (int)(__builtin_amdgcn_alignbit((u32)(xi >> 32), (u32)xi, (u32)s.sh) << s.wsh) >> s.wsh;
}
// STAGE 0 (ODD = false), STAGE 1: even positions Y = D, odd positions Y = -j D with the negation quirk
template <int MODE, bool ODD>
__device__ __forceinline__ void gfly_triv(int &are, int &aim, int &bre, int &bim, const W32Stage &s)
{
    int dre, dim;
    if (MODE == W_UNSCALED) {
        dre = are - bre, dim = aim - bim;
        are += bre, aim += bim;
    } else if (MODE == W_TRUNC) {
        const int ar = are >> 1, ai = aim >> 1, br = bre >> 1, bi = bim >> 1;
        dre = ar - br, dim = ai - bi;
        are = ar + br, aim = ai + bi;
    } else {
        // rhu2(A + B) = (A | B) - ((A ^ B) >> 1): always inside the DTW bits of its operands, no wrap; rhu2(A - B) = rhu2(A + B) - B, which
        // leaves the range only for A = max, B = min: the RTL's wrap (derivation: sumdiff, intfft_pk16.hpp).  7 operations per component
        // instead of 17, all of them in the fast-issue VOP2 class.
#ifdef INTFFT_RHU2_LONG // (A/B: the form of rounds 1-5)
        const int ar = are >> 1, ai = aim >> 1, br = bre >> 1, bi = bim >> 1;
        dre = (int)((u32)(ar - br + (are & ~bre & 1)) << s.wosh) >> s.wosh;
        dim = (int)((u32)(ai - bi + (aim & ~bim & 1)) << s.wosh) >> s.wosh;
        are = (int)((u32)(ar + br + ((are | bre) & 1)) << s.wosh) >> s.wosh;
        aim = (int)((u32)(ai + bi + ((aim | bim) & 1)) << s.wosh) >> s.wosh;
#else
        const int sr = (are | bre) - ((are ^ bre) >> 1), si = (aim | bim) - ((aim ^ bim) >> 1);
        dre = (int)((u32)(sr - bre) << s.wosh) >> s.wosh;
        dim = (int)((u32)(si - bim) << s.wosh) >> s.wosh;
        are = sr, aim = si;
#endif
    }
    if (ODD) {
        bre = dim;
        bim = (dre >> 31) - dre; // -x for x >= 0 (fits: x < 2^(w-1)), ~x for x < 0   (int_dif2_fly.vhd:297-304)
    } else {
        bre = dre;
        bim = dim;
    }
}


// ---- general-width DIT butterflies (intfft_w32inv.hip): int_dit2_fly.vhd:142-325 ------------------------------------
// T = B * conj(W) through the re/im-swapped multiplier feed at width DTW (wsh = 32 - DTW), then the sum / difference
// of (A, T) in the plan's scaling mode (wosh = 32 - output width)
template <int MODE> __device__ __forceinline__ void gsumdiff(int &are, int &aim, int &bre, int &bim, int tr, int ti, const W32Stage &s)
{
    if (MODE == W_UNSCALED) {
        bre = are - tr, bim = aim - ti;
        are += tr, aim += ti;
    } else if (MODE == W_TRUNC) {
        const int ar = are >> 1, ai = aim >> 1, xr = tr >> 1, xi = ti >> 1;
        bre = ar - xr, bim = ai - xi;
        are = ar + xr, aim = ai + xi;
    } else {
#ifdef INTFFT_RHU2_LONG
        const int ar = are >> 1, ai = aim >> 1, xr = tr >> 1, xi = ti >> 1;
        bre = (int)((u32)(ar - xr + (are & ~tr & 1)) << s.wosh) >> s.wosh;
        bim = (int)((u32)(ai - xi + (aim & ~ti & 1)) << s.wosh) >> s.wosh;
        are = (int)((u32)(ar + xr + ((are | tr) & 1)) << s.wosh) >> s.wosh;
        aim = (int)((u32)(ai + xi + ((aim | ti) & 1)) << s.wosh) >> s.wosh;
#else
        const int sr = (are | tr) - ((are ^ tr) >> 1), si = (aim | ti) - ((aim ^ ti) >> 1); // rhu2(A + T), then rhu2(A - T) = rhu2(A + T) - T (see gfly)
        bre = (int)((u32)(sr - tr) << s.wosh) >> s.wosh;
        bim = (int)((u32)(si - ti) << s.wosh) >> s.wosh;
        are = sr, aim = si;
#endif
    }
}
template <int MODE, bool UNIFORM_W = false, bool MASKED = true>
__device__ __forceinline__ void gfly_dit(int &are, int &aim, int &bre, int &bim, int wr, int wi, const W32Stage &s)
{
    // keep hipcc from hoisting the 64-bit sign extension of the (frame-invariant) twiddles out of the frame loop, which
    // turns each v_mad_i64_i32 into a four-instruction 64 x 32 multiply (seen on the per-lane twiddles of the DIT path)
    if (UNIFORM_W) asm volatile("" : "+s"(wr), "+s"(wi));
    else asm volatile("" : "+v"(wr), "+v"(wi));
    unsigned long long xr, xi;
    if (MASKED) {
        const unsigned long long m2i = (unsigned long long)((long long)bim * wr), m1i = (unsigned long long)((long long)bre * wi);
        const unsigned long long m2r = (unsigned long long)((long long)bim * wi), m1r = (unsigned long long)((long long)bre * wr);
        const unsigned long long k = 0xFFFFFFFF00000000ull | s.keep;
        xi = (m2i & k) - (m1i & k); // DO_RE of the swapped feed = T.im
        xr = (m2r & k) + (m1r & k); // DO_IM = T.re
    } else {
        const int nbre = -bre;
        xi = (unsigned long long)((long long)bim * wr + (long long)nbre * wi);
        xr = (unsigned long long)((long long)bim * wi + (long long)bre * wr);
    }
    const int tr = (int)(__builtin_amdgcn_alignbit((u32)(xr >> 32), (u32)xr, (u32)s.sh) << s.wsh) >> s.wsh;
    const int ti = (int)(__builtin_amdgcn_alignbit((u32)(xi >> 32), (u32)xi, (u32)s.sh) << s.wsh) >> s.wsh;
    gsumdiff<MODE>(are, aim, bre, bim, tr, ti, s);
}
// STAGE 0 (ODD = false), STAGE 1: even positions T = B, odd positions T = +j B with the negation quirk
template <int MODE, bool ODD>
__device__ __forceinline__ void gfly_dit_triv(int &are, int &aim, int &bre, int &bim, const W32Stage &s)
{
    const int tr = ODD ? (bim >> 31) - bim : bre, ti = ODD ? bre : bim; // int_dit2_fly.vhd:264-276
    gsumdiff<MODE>(are, aim, bre, bim, tr, ti, s);
}

// ---- unscaled results of 33 / 34 bits (intfft_fastw32.hip, intfft_fast4096w.hip: W32Args::out64) --------------------
// When DATA_WIDTH + NFFT exceeds 32 by at most 2, every multiplier stage (STAGE >= 2) still works within 32 bits; only
// the two multiplier-free stages 1 and 0 produce the 33rd and 34th bit.  They run on sign-extended 64-bit registers
// (exact sums, no wrap needed: int_dif2_fly.vhd:222-318) and the results are stored in int64 containers.
__device__ __forceinline__ void tail64_stages10(long long (&xr)[16], long long (&xi)[16]);
__device__ __forceinline__ void tail64_unscaled(const int (&re)[16], const int (&im)[16], long long (&xr)[16], long long (&xi)[16])
{
#pragma unroll
    for (int r = 0; r < 16; ++r) xr[r] = re[r], xi[r] = im[r];
    tail64_stages10(xr, xi);
}
__device__ __forceinline__ void tail64_stages10(long long (&xr)[16], long long (&xi)[16])
{
#pragma unroll
    for (int g = 0; g < 16; g += 4) { // STAGE 1: even positions Y = D; odd positions Y.re = D.im, Y.im = D.re >= 0 ? -D.re : ~D.re
        long long dr = xr[g] - xr[g + 2], di = xi[g] - xi[g + 2];
        xr[g] += xr[g + 2], xi[g] += xi[g + 2];
        xr[g + 2] = dr, xi[g + 2] = di;
        dr = xr[g + 1] - xr[g + 3], di = xi[g + 1] - xi[g + 3];
        xr[g + 1] += xr[g + 3], xi[g + 1] += xi[g + 3];
        xr[g + 3] = di, xi[g + 3] = (dr >> 63) - dr;
    }
#pragma unroll
    for (int g = 0; g < 16; g += 2) { // STAGE 0
        const long long dr = xr[g] - xr[g + 1], di = xi[g] - xi[g + 1];
        xr[g] += xr[g + 1], xi[g] += xi[g + 1];
        xr[g + 1] = dr, xi[g + 1] = di;
    }
}

// ---- unscaled results of 35 / 36 bits (intfft_fast4096w.hip: W32Args::out64 = 2) -----------------------------------------------
// DATA_WIDTH + NFFT - 4 <= 32: every stage down to STAGE 4 still fits 32-bit registers; the whole last register round
// (STAGE 3, 2 with wave-uniform twiddles, then 1, 0) runs on 64-bit registers.  A general butterfly of that round: exact
// 64-bit products d * w = mad_i64_i32(dL, w) + (mul_lo(dH, w) << 32) (|d| < 2^39, |w| < 2^23: width + TWDL_WIDTH <= 63),
// the regime's truncation points as in gfly (masked form for a > 0), the wo-bit slice of the sum by two 64-bit shifts.
__device__ __forceinline__ unsigned long long mul64x32(int dl, int dh, int w)
{
    return (unsigned long long)((long long)dl * w) + ((unsigned long long)((u32)dh * (u32)w) << 32);
}
template <bool MASKED>
__device__ __forceinline__ void gfly64(long long &are, long long &aim, long long &bre, long long &bim, int wr, int wi, const W32Stage &s)
{
    asm volatile("" : "+s"(wr), "+s"(wi)); // wave-uniform twiddles: see intfft_fast1024u.hip
    const long long dre = are - bre, dim = aim - bim; // unscaled: exact (int_dif2_fly.vhd:222-240)
    are += bre;
    aim += bim;
    const int rl = (int)dre, rh = (int)(dre >> 32) - (rl >> 31);
    const int il = (int)dim, ih = (int)(dim >> 32) - (il >> 31);
    unsigned long long m2r = mul64x32(rl, rh, wr), m1r = mul64x32(il, ih, wi);
    unsigned long long m2i = mul64x32(rl, rh, wi), m1i = mul64x32(il, ih, wr);
    if (MASKED) {
        const unsigned long long k = 0xFFFFFFFF00000000ull | s.keep;
        m2r &= k, m1r &= k, m2i &= k, m1i &= k;
    }
    const unsigned long long xr = m2r - m1r, xi = m2i + m1i;
    const int wo = 32 - s.wsh; // the multiplier's width (33 .. 38 here)
    bre = (long long)(xr << (64 - s.sh - wo)) >> (64 - wo);
    bim = (long long)(xi << (64 - s.sh - wo)) >> (64 - wo);
}

} // namespace intfft
