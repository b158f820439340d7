// intfft_wide16.hip -- two-pass kernels for BASELINE config 3: int_fftNk with NFFT = 16, DATA_WIDTH = 24,
// FORMAT = 1 (full bit growth: 24-bit samples in int32 containers -> 40-bit results in int64 containers),
// 16 <= TWDL_WIDTH <= 24, natural order in and out.
//
// The sixteen DIF stages split where the growing width crosses 32 bits (int_fftNk.vhd:187-207: stage ii has
// DTW = 24 + ii inputs and 25 + ii bit outputs):
//   pass 1  k_wide16_p1   STAGE 15..8  (ii 0..7, widths <= 32)  int32 registers, user array -> plan scratch
//   pass 2  k_wide16_p2   STAGE 7..0   (ii 8..15, widths 33..40) 64-bit registers, plan scratch -> user array
// Index n = 256 r + c.  Pass 1 transforms over r (a 256-thread workgroup owns 16 adjacent columns c of one
// frame: 4096 samples, 16 per thread, 128-byte runs in global memory); pass 2 transforms over c (a workgroup
// owns the 16 rows r = 16 j + r0, j = 0..15, i.e. the rows whose bit-reversed indices are adjacent, so that
// each store instruction writes 256-byte runs of the natural-order output brev16(n)).  The scratch layout
// [r0][c7..4][j][c3..0] makes the 4096 samples of a pass-2 workgroup contiguous AND turns every register store of pass 1 and every
// register load of pass 2 into one 2 KiB run (round 3; round 2's [r0][j][c] left pass 1 with 128-byte pieces 2 KiB apart).
//
// N = 2^13 .. 2^15 (template L; round 2): the same two kernels on VIRTUAL 2^16-point frames of G = 2^(16-L) consecutive real
// frames.  The top 16 - L bits of r are then frame-number bits g: their stages (STAGE >= L) are skipped, the twiddle index
// of every other stage (position mod 2^s) is unchanged.  A pass-2 unit is (g, low) with r = g 2^(L-8) + t4 2^(L-12) + low:
// its 16 rows t4 = 0..15 are the rows of ONE real frame whose bit-reversed indices are adjacent, scratch layout
// [g, low][t4][c].  Pass-2 stages whose width is still <= 32 (24 + L - 7 .. at the short lengths) take the general form
// of the result slice (two 64-bit shifts instead of v_alignbit + v_bfe).  DATA_WIDTH is a kernel argument.
//
// Each pass = two in-register rounds of four stages (registers carry 4 index bits) with one block-wide LDS
// transpose between them, like intfft_fast4096.hip; arithmetic on unpacked registers like
// intfft_fast1024u.hip.  All twiddles that depend on the thread are frame invariant and live in VGPRs
// (pass 1: a workgroup keeps its column tile over its frame loop).
//
// Multiplier (int_cmult_dsp48.vhd:182-434, every regime): result = wrap_w(((M2 >> a) -/+ (M1 >> a)) >> b).
//   a = 0 : exact sum first -> chained v_mad_i64_i32
//   a > 0 : (M >> a) << a == M & ~(2^a - 1), so  ((M2 & K) -/+ (M1 & K)) >> (a + b)  with K = ~(2^a - 1)
// and bits [a+b, a+b+w) of the 64-bit sum are sliced with v_alignbit_b32 (+ v_ashrrev / v_bfe for the sign).
// 64-bit data d = dH * 2^32 + dL (dL signed): d * w = v_mad_i64_i32(dL, w) + (v_mul_lo_u32(dH, w) << 32),
// exact because |d| < 2^39 and |w| < 2^23 (width 40 + 24 <= 64).
// multi-pass kernels: non-temporal loads measure 4-14 % faster here (the single-pass kernels gain 4-30 % from PLAIN loads): intfft_device.hpp
#define INTFFT_NT_LOADS 1
#include "intfft_internal.hpp"

#include <cstdlib>

namespace intfft {

using u32 = uint32_t;
using u64 = unsigned long long;
using i64 = long long;

constexpr int ROWW = 20;              // LDS row stride in dwords (16 data + 4 pad)
constexpr int PLANEW = 256 * ROWW;    // dwords per transpose plane
#ifndef SCHED_GROUP
#define SCHED_GROUP 2 // butterflies the scheduler may interleave (bounds the live 64-bit products)
#endif

__device__ __forceinline__ constexpr int rev4w(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

// ---- the cores' own beat orders on the two-pass wide classes (round 5; NAT instantiations, WideArgs::native bit 0: HALVES on the time side, bit 1: BITREV on
// the frequency side; int_fftNk.vhd:15-21 / int_ifftNk.vhd:15-21).  Time side (first pass of the forward core, last of the inverse): thread (hi4, c) holds the rows
// r = 16 j + hi4; a HALVES beat (x[i], x[i + N/2]) is the register pair (j, j | 2^(L-13)) -- adjacent samples in memory, ONE 16-byte (32-byte) access.
// Frequency side: BITREV order is the core position itself, p = (t4 << (L-4)) + 256 low + c in a unit: coalesced with thread = c, registers = t4; a 16 x 16
// exchange through the transpose planes (rows of the same 20-dword stride) hands it to / takes it from the round layout (thread = (c7..4, t4), registers = c3..0).
// time-side pair of register j (HALVES): its partner and the 16-byte index of the pair inside the virtual frame: g N/2 + 4096 (j's bits below the pair bit) + thread part
template <int L> __device__ __forceinline__ constexpr int halves_pair_bit() { return 1 << (L - 13); }
template <int L> __device__ __forceinline__ constexpr int halves_pair_index(int j) // j with the pair bit clear
{
    return ((j >> (L - 12)) << (L - 1)) + 4096 * (j & ((1 << (L - 12)) - 1));
}

// ---- 32-bit butterflies (pass 1) --------------------------------------------------------------------
// s2 = a + b + wo - 32 (alignbit amount that leaves the wo result bits top-aligned), s3 = 32 - wo
template <bool AZ>
__device__ __forceinline__ void wfly32(int &are, int &aim, int &bre, int &bim, int wr, int wi, const WideStage &s)
{
    asm volatile("" : "+v"(wr), "+v"(wi)); // keep the twiddles' 64-bit sign extension from being hoisted (intfft_u32.hpp)
    const int dre = are - bre, dim = aim - bim; // unscaled: exact, one bit of growth (int_dif2_fly.vhd:222-240)
    are += bre;
    aim += bim;
    u64 xr, xi;
    if (AZ) {
        const int nd = -dim;
        xr = (u64)((i64)dre * wr + (i64)nd * wi); // M2 - M1
        xi = (u64)((i64)dre * wi + (i64)dim * wr); // M2 + M1
    } else {
        const u64 m2r = (u64)((i64)dre * wr), m1r = (u64)((i64)dim * wi);
        const u64 m2i = (u64)((i64)dre * wi), m1i = (u64)((i64)dim * wr);
        const u64 k = 0xFFFFFFFF00000000ull | s.keep;
        xr = (m2r & k) - (m1r & k);
        xi = (m2i & k) + (m1i & k);
    }
    bre = (int)__builtin_amdgcn_alignbit((u32)(xr >> 32), (u32)xr, (u32)s.s2) >> s.s3;
    bim = (int)__builtin_amdgcn_alignbit((u32)(xi >> 32), (u32)xi, (u32)s.s2) >> s.s3;
}

template <int H, bool AZ>
__device__ __forceinline__ void wstage32x(int (&re)[16], int (&im)[16], const int (&wr)[H], const int (&wi)[H],
                                          const WideStage &s)
{
#pragma unroll
    for (int g = 0; g < 16; g += 2 * H)
#pragma unroll
        for (int j = 0; j < H; ++j) {
            wfly32<AZ>(re[g + j], im[g + j], re[g + j + H], im[g + j + H], wr[j], wi[j], s);
            if (SCHED_GROUP && ((g / 2 + j) % SCHED_GROUP) == SCHED_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        }
}
template <int H>
__device__ __forceinline__ void wstage32(int (&re)[16], int (&im)[16], const int (&wr)[H], const int (&wi)[H],
                                         const WideStage &s)
{
    wstage32x<H, false>(re, im, wr, wi, s); // the masked form covers a = 0 too (keep = ~0)
}

// XS > 0 (round 5, intfft_widelong.hip): the frames are the 2^16-point blocks of N = 2^(16 + XS)-point frames whose STAGE 16 + XS - 1 .. 16 ran in k_wide_pre:
// the same kernel with its stage descriptors XS entries further down (the caller passes block counts and a.dw = the width behind the pre-pass)
template <int L, bool NAT = false, int XS = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_wide16_p1(const int2 *in, int2 *scr, const int2 *__restrict__ twt,
                                                   const WideArgs a, size_t nframes_user)
{
    static_assert(L >= 13 && L <= 16, "virtual 2^16-point frames");
    static_assert(XS == 0 || (L == 16 && !NAT && XS <= 4), "long frames: whole 2^16-point blocks, natural order");
    constexpr int G = 1 << (16 - L);                     // real frames per virtual frame
    const size_t nframes = (nframes_user + G - 1) / G;   // virtual frames
    constexpr int X = L - 16 + XS;                       // a.st[] is indexed by L - 1 - STAGE: STAGE 15 - k is entry X + k
    __shared__ __attribute__((aligned(16))) u32 lds[2 * PLANEW];
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;
    const int tile = blockIdx.x & 15;
    const int c = 16 * tile + lo4;

    // frame-invariant twiddles; table of STAGE s starts at twt + 2^s - 1, index = n mod 2^s, n = c + 256 r
    // round 1: thread = (r3..0 = hi4, c3..0), registers r7..4
    int w15r[8], w15i[8], w14r[4], w14i[4], w13r[2], w13i[2], w12r[1], w12i[1];
    // round 2: thread = (r7..4, c3..0), registers r3..0: indices depend on c only
    int w11r[8], w11i[8], w10r[4], w10i[4], w9r[2], w9i[2], w8r[1], w8i[1];
    {
        const int base = c + 256 * hi4;
        // (the index of STAGE s is n mod 2^s: frame-number bits of r above bit s - 8 do not enter; stages >= L have no table)
        if constexpr (L > 15) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int2 w = twt[32767 + base + 4096 * j];
                w15r[j] = w.x, w15i[j] = w.y;
            }
        }
        if constexpr (L > 14) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int2 w = twt[16383 + ((base + 4096 * j) & 16383)];
                w14r[j] = w.x, w14i[j] = w.y;
            }
        }
        if constexpr (L > 13) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int2 w = twt[8191 + ((base + 4096 * j) & 8191)];
                w13r[j] = w.x, w13i[j] = w.y;
            }
        }
        int2 w = twt[4095 + base];
        w12r[0] = w.x, w12i[0] = w.y;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            w = twt[2047 + c + 256 * j];
            w11r[j] = w.x, w11i[j] = w.y;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w = twt[1023 + c + 256 * j];
            w10r[j] = w.x, w10i[j] = w.y;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            w = twt[511 + c + 256 * j];
            w9r[j] = w.x, w9i[j] = w.y;
        }
        w = twt[255 + c];
        w8r[0] = w.x, w8i[0] = w.y;
    }
    // transpose: element (thread (hi4, lo4), register j) -> row 16 j + lo4, column hi4; thread t reads row t
    u32 *const wr_re = lds + ROWW * lo4 + hi4;
    u32 *const wr_im = wr_re + PLANEW;
    const uint4 *const rd_re = reinterpret_cast<const uint4 *>(lds + ROWW * tid);
    const uint4 *const rd_im = reinterpret_cast<const uint4 *>(lds + PLANEW + ROWW * tid);

    const size_t fstep = gridDim.x >> 4;
    for (size_t f = blockIdx.x >> 4; f < nframes; f += fstep) {
        int re[16], im[16];
        const int2 *src = in + f * 65536; // wave-uniform; the thread's part is toff (at32: SGPR base + 32-bit VGPR offset, no 64-bit VALU address arithmetic)
        unsigned toff = (unsigned)(c + 256 * hi4);
        asm volatile("" : "+v"(toff)); // opaque per iteration: hoisted out of the frame loop the zero-extended offset becomes a VGPR pair again
        const bool partial = L < 16 && (f + 1) * G > nframes_user; // last group: rows of absent frames read as 0, are not stored
        typedef int v2i __attribute__((ext_vector_type(2)));
        if (NAT && (a.native & 1)) { // HALVES order in: one 16-byte load per register pair (j, j | 2^(L-13))
            typedef int v4i __attribute__((ext_vector_type(4)));
            const v4i *src4 = reinterpret_cast<const v4i *>(in) + f * 32768;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j & halves_pair_bit<L>()) continue;
                v4i x = {0, 0, 0, 0};
                if (!partial || f * G + (size_t)(j >> (L - 12)) < nframes_user) x = INTFFT_LD(at32(src4 + halves_pair_index<L>(j), toff));
                re[j] = __builtin_amdgcn_sbfe(x.x, 0, a.dw), im[j] = __builtin_amdgcn_sbfe(x.y, 0, a.dw);
                re[j | halves_pair_bit<L>()] = __builtin_amdgcn_sbfe(x.z, 0, a.dw), im[j | halves_pair_bit<L>()] = __builtin_amdgcn_sbfe(x.w, 0, a.dw);
            }
        } else if (!partial) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v2i x = INTFFT_LD(at32(reinterpret_cast<const v2i *>(src + 4096 * j), toff));
                re[j] = __builtin_amdgcn_sbfe(x.x, 0, a.dw); // conv_std_logic_vector(.., DATA_WIDTH): wrap on load
                im[j] = __builtin_amdgcn_sbfe(x.y, 0, a.dw);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) { // r = 16 j + hi4: real frame g = r >> (L - 8) = j >> (L - 12)
                v2i x = {0, 0};
                if (f * G + (size_t)(j >> (L - 12)) < nframes_user) x = *at32(reinterpret_cast<const v2i *>(src + 4096 * j), toff);
                re[j] = __builtin_amdgcn_sbfe(x.x, 0, a.dw);
                im[j] = __builtin_amdgcn_sbfe(x.y, 0, a.dw);
            }
        }
        if constexpr (L > 15) wstage32<8>(re, im, w15r, w15i, a.st[X + 0]);
        if constexpr (L > 14) wstage32<4>(re, im, w14r, w14i, a.st[X + 1]);
        if constexpr (L > 13) wstage32<2>(re, im, w13r, w13i, a.st[X + 2]);
        wstage32<1>(re, im, w12r, w12i, a.st[X + 3]);
        __syncthreads(); // the previous frame's reads are done
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            wr_re[ROWW * 16 * j] = (u32)re[j];
            wr_im[ROWW * 16 * j] = (u32)im[j];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = rd_re[q], y = rd_im[q];
            re[4 * q + 0] = (int)x.x, re[4 * q + 1] = (int)x.y, re[4 * q + 2] = (int)x.z, re[4 * q + 3] = (int)x.w;
            im[4 * q + 0] = (int)y.x, im[4 * q + 1] = (int)y.y, im[4 * q + 2] = (int)y.z, im[4 * q + 3] = (int)y.w;
        }
        wstage32<8>(re, im, w11r, w11i, a.st[X + 4]);
        wstage32<4>(re, im, w10r, w10i, a.st[X + 5]);
        wstage32<2>(re, im, w9r, w9i, a.st[X + 6]);
        wstage32<1>(re, im, w8r, w8i, a.st[X + 7]);
        // now thread = (r7..4 = hi4, c3..0), register q = r3..0, r = 16 hi4 + q = g 2^(L-8) + t4 2^(L-12) + low:
        // scratch [unit = (g, low)][t4][c]   (L = 16: [r0 = q][j = hi4][c])
        // with r = 16 hi4 + q: g = hi4 >> (L-12), t4 = ((hi4 << (16-L)) & 15) | (q >> (L-12)), low = q mod 2^(L-12): a thread part
        // plus a compile-time part per register
        const int g = hi4 >> (L - 12);
        if (partial && f * G + (size_t)g >= nframes_user) continue; // uniform per (L-12 .. 3 bits of hi4): the barriers above are passed
        // within a unit the 4096 samples are laid out [column tile c7..4][t4][c3..0] (round 3; [t4][c] before): the 256 threads of this
        // workgroup then store 2 KiB runs (L = 16: one run per register) instead of 128-byte pieces 2 KiB apart, and pass 2 reads the
        // unit as sixteen 2 KiB runs, one per register
        int2 *dst = scr + f * 65536;
        unsigned toff2 = (unsigned)(256 * tile + lo4 + 4096 * (g << (L - 12)) + 16 * ((hi4 << (16 - L)) & 15));
        asm volatile("" : "+v"(toff2));
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const v2i y = {re[q], im[q]};
            *at32(reinterpret_cast<v2i *>(dst + 4096 * (q & ((1 << (L - 12)) - 1)) + 16 * (q >> (L - 12))), toff2) = y;
        }
    }
}

// ---- 64-bit butterflies (pass 2) --------------------------------------------------------------------
struct W2Consts {
    int wr3[8], wi3[8]; // STAGE 3 twiddles (wave uniform)
    int wr2[4], wi2[4]; // STAGE 2
};

__device__ __forceinline__ u64 mul64x32(int dl, int dh, int w)
{
    return (u64)((i64)dl * w) + ((u64)((u32)dh * (u32)w) << 32);
}

// sh = a + b, w32 = wo - 32 (1..8)
template <bool AZ, bool UNIFORM_W>
__device__ __forceinline__ void wfly64(i64 &are, i64 &aim, i64 &bre, i64 &bim, int wr, int wi, const WideStage &s)
{
    if (UNIFORM_W) asm volatile("" : "+s"(wr), "+s"(wi)); // see intfft_fast1024u.hip
    else asm volatile("" : "+v"(wr), "+v"(wi));
    const i64 dre = are - bre, dim = aim - bim;
    are += bre;
    aim += bim;
    const int rl = (int)dre, rh = (int)(dre >> 32) - (rl >> 31);
    const int il = (int)dim, ih = (int)(dim >> 32) - (il >> 31);
    u64 m2r = mul64x32(rl, rh, wr), m1r = mul64x32(il, ih, wi);
    u64 m2i = mul64x32(rl, rh, wi), m1i = mul64x32(il, ih, wr);
    if (!AZ) {
        const u64 k = 0xFFFFFFFF00000000ull | s.keep;
        m2r &= k, m1r &= k, m2i &= k, m1i &= k;
    }
    const u64 xr = m2r - m1r, xi = m2i + m1i;
    if (s.w32 < 1) { // width <= 32 (the leading pass-2 stages of the shorter lengths): the general form of the slice
        const int wo = 32 + s.w32;
        bre = (i64)(xr << (64 - s.sh - wo)) >> (64 - wo);
        bim = (i64)(xi << (64 - s.sh - wo)) >> (64 - wo);
        return;
    }
    const u32 lr = __builtin_amdgcn_alignbit((u32)(xr >> 32), (u32)xr, (u32)s.sh);
    const u32 li = __builtin_amdgcn_alignbit((u32)(xi >> 32), (u32)xi, (u32)s.sh);
    const int hr = __builtin_amdgcn_sbfe((int)(xr >> 32), s.sh, s.w32);
    const int hi = __builtin_amdgcn_sbfe((int)(xi >> 32), s.sh, s.w32);
    bre = (i64)(((u64)(u32)hr << 32) | lr);
    bim = (i64)(((u64)(u32)hi << 32) | li);
}
__device__ __forceinline__ void wfly64_triv(i64 &are, i64 &aim, i64 &bre, i64 &bim)
{
    const i64 dre = are - bre, dim = aim - bim;
    are += bre;
    aim += bim;
    bre = dre;
    bim = dim;
}
// odd positions of STAGE 1: Y.re = D.im, Y.im = D.re >= 0 ? -D.re : ~D.re  (int_dif2_fly.vhd:297-304)
__device__ __forceinline__ void wfly64_mj(i64 &are, i64 &aim, i64 &bre, i64 &bim)
{
    const i64 dre = are - bre, dim = aim - bim;
    are += bre;
    aim += bim;
    bre = dim;
    bim = (dre >> 63) - dre;
}

template <int H, bool AZ, bool UNIFORM_W>
__device__ __forceinline__ void wstage64x(i64 (&re)[16], i64 (&im)[16], const int (&wr)[H], const int (&wi)[H],
                                          const WideStage &s)
{
#pragma unroll
    for (int g = 0; g < 16; g += 2 * H)
#pragma unroll
        for (int j = 0; j < H; ++j) {
            wfly64<AZ, UNIFORM_W>(re[g + j], im[g + j], re[g + j + H], im[g + j + H], wr[j], wi[j], s);
            if (SCHED_GROUP && ((g / 2 + j) % SCHED_GROUP) == SCHED_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        }
}
template <int H, bool UNIFORM_W = false>
__device__ __forceinline__ void wstage64(i64 (&re)[16], i64 (&im)[16], const int (&wr)[H], const int (&wi)[H],
                                         const WideStage &s)
{
    wstage64x<H, false, UNIFORM_W>(re, im, wr, wi, s);
}

// XS > 0 (round 5, intfft_widelong.hip): N = 2^LX, LX = 16 + XS, as B = 2^XS blocks of 2^16 points (n = 65536 b + 256 r + c) whose STAGE LX-1 .. 16 ran in
// k_wide_pre and STAGE 15 .. 8 in k_wide16_p1<16, false, XS> block by block.  A unit is then the 16 rows (b, jtop) -- every block b x the rows
// r = (jtop << (4 + XS)) | ulow -- whose natural-order output indices brev_LX(n) are adjacent: rev4(t4 = (b << (4 - XS)) | jtop) is again the low nibble of the
// output index, so the store below is the L = 16 one with LX in place of L; the loads gather 128-byte row pieces from the blocks' [r0][c7..4][j][c3..0] layouts.
// R32 (with XS > 0): STAGE 7 .. 4 still within 32 bits (16-bit data: DATA_WIDTH + LX - 4 <= 32) -- round 1 on the int32 butterflies of pass 1, two transpose
// planes instead of three; only round 2 (STAGE 3 .. 0, two of them multiplier-free) pays for 64-bit words
template <int L, bool IN64 = false, bool NAT = false, int XS = 0, bool R32 = false> // IN64: the scratch holds 64-bit words (k_wide64_p1 in front: DATA_WIDTH 25 .. 32, round 5)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_wide16_p2(const int2 *scr, i64 *out, const int2 *__restrict__ twt,
                                                   const WideArgs a, const W2Consts k, size_t nframes_user)
{
    static_assert(XS == 0 || (L == 16 && XS <= 4), "long frames: whole 2^16-point blocks");
    static_assert(!R32 || (XS > 0 && !IN64), "int32 first round: instantiated for the long frames only");
    constexpr int G = 1 << (16 - L);
    const size_t nframes = (nframes_user + G - 1) / G; // virtual frames
    constexpr int X = L - 16 + XS;                     // a.st[] entry of STAGE 15 - k is X + k
    constexpr int LX = L + XS;                         // the output frame
    __shared__ __attribute__((aligned(16))) u32 lds[2 * PLANEW];
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;

    // round 1: thread = (j = hi4, c3..0 = lo4), registers c7..4: STAGE 7 index 16 jj + lo4, ... STAGE 4 index lo4
    int w7r[8], w7i[8], w6r[4], w6i[4], w5r[2], w5i[2], w4r[1], w4i[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int2 w = twt[127 + 16 * j + lo4];
        w7r[j] = w.x, w7i[j] = w.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int2 w = twt[63 + 16 * j + lo4];
        w6r[j] = w.x, w6i[j] = w.y;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int2 w = twt[31 + 16 * j + lo4];
        w5r[j] = w.x, w5i[j] = w.y;
    }
    {
        const int2 w = twt[15 + lo4];
        w4r[0] = w.x, w4i[0] = w.y;
    }
    // transpose: element (thread (j = hi4, c3..0 = lo4), register c7..4 = q) -> row 16 q + j, column c3..0;
    // thread t reads row t, so afterwards thread = (c7..4 = hi4', j = lo4'), registers c3..0
    u32 *const wr0 = lds + ROWW * hi4 + lo4;
    u32 *const wr1 = wr0 + PLANEW;
    const uint4 *const rd0 = reinterpret_cast<const uint4 *>(lds + ROWW * tid);
    const uint4 *const rd1 = reinterpret_cast<const uint4 *>(lds + PLANEW + ROWW * tid);

    // work unit u = 16 f + (g, low)   (XS > 0: u = (f << (4 + XS)) + ulow)
    const size_t units = nframes << (4 + XS);
    for (size_t u = blockIdx.x; u < units; u += gridDim.x) {
        const size_t f = u >> (4 + XS);
        const int r0 = (int)(u & ((16u << XS) - 1u));
        const int ug = XS ? 0 : r0 >> (L - 12), ulow = XS ? r0 : r0 & ((1 << (L - 12)) - 1);
        const size_t real = f * G + (size_t)ug; // the real frame these 16 rows belong to
        if (L < 16 && real >= nframes_user) continue;
        i64 re[16], im[16];
        // unit layout [c7..4 = q][t4 = hi4][c3..0]: 2 KiB per register; thread part = tid
        // (XS > 0: block (f << XS) + b, row r = 16 j + r0' with r0' = ulow & 15, j = (jtop << XS) | (ulow >> 4), layout [r0'][q][j][c3..0] per block)
        const int2 *src = XS ? scr + (f << (16 + XS)) + 4096 * (r0 & 15) + 16 * (r0 >> 4) : scr + f * 65536 + 4096 * r0;
        typedef int v2i __attribute__((ext_vector_type(2)));
        unsigned tid_l = XS ? (unsigned)((hi4 >> (4 - XS)) * 65536 + 16 * ((hi4 & ((1 << (4 - XS)) - 1)) << XS) + lo4) : (unsigned)tid;
        asm volatile("" : "+v"(tid_l)); // (opaque per iteration, see k_wide16_p1)
        if constexpr (IN64) { // 16 bytes per sample
            typedef i64 v2li __attribute__((ext_vector_type(2)));
            const v2li *src64 = XS ? reinterpret_cast<const v2li *>(scr) + (f << (16 + XS)) + 4096 * (r0 & 15) + 16 * (r0 >> 4) : reinterpret_cast<const v2li *>(scr) + f * 65536 + 4096 * r0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const v2li x = *at32(src64 + 256 * q, tid_l);
                re[q] = x.x;
                im[q] = x.y;
            }
        } else if constexpr (!R32) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const v2i x = *at32(reinterpret_cast<const v2i *>(src + 256 * q), tid_l);
                re[q] = x.x;
                im[q] = x.y;
            }
        }
        if constexpr (R32) { // round 1 on int32, two planes
            int re32[16], im32[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const v2i x = *at32(reinterpret_cast<const v2i *>(src + 256 * q), tid_l);
                re32[q] = x.x;
                im32[q] = x.y;
            }
            wstage32<8>(re32, im32, w7r, w7i, a.st[X + 8]);
            wstage32<4>(re32, im32, w6r, w6i, a.st[X + 9]);
            wstage32<2>(re32, im32, w5r, w5i, a.st[X + 10]);
            wstage32<1>(re32, im32, w4r, w4i, a.st[X + 11]);
            __syncthreads(); // the previous unit's reads are done
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                wr0[ROWW * 16 * q] = (u32)re32[q];
                wr1[ROWW * 16 * q] = (u32)im32[q];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 x = rd0[q], y = rd1[q];
                re[4 * q + 0] = (int)x.x, re[4 * q + 1] = (int)x.y, re[4 * q + 2] = (int)x.z, re[4 * q + 3] = (int)x.w;
                im[4 * q + 0] = (int)y.x, im[4 * q + 1] = (int)y.y, im[4 * q + 2] = (int)y.z, im[4 * q + 3] = (int)y.w;
            }
        } else {
        wstage64<8>(re, im, w7r, w7i, a.st[X + 8]);
        wstage64<4>(re, im, w6r, w6i, a.st[X + 9]);
        wstage64<2>(re, im, w5r, w5i, a.st[X + 10]);
        wstage64<1>(re, im, w4r, w4i, a.st[X + 11]);
        // 36-bit values through three dword planes: re.lo, im.lo, then (re.hi & 0xFFFF) | (im.hi << 16)
        u32 hp[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) hp[q] = ((u32)((u64)re[q] >> 32) & 0xFFFFu) | ((u32)((u64)im[q] >> 32) << 16);
        u32 rlo[16], ilo[16];
        __syncthreads(); // the previous unit's reads are done
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            wr0[ROWW * 16 * q] = (u32)re[q];
            wr1[ROWW * 16 * q] = (u32)im[q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = rd0[q], y = rd1[q];
            rlo[4 * q + 0] = x.x, rlo[4 * q + 1] = x.y, rlo[4 * q + 2] = x.z, rlo[4 * q + 3] = x.w;
            ilo[4 * q + 0] = y.x, ilo[4 * q + 1] = y.y, ilo[4 * q + 2] = y.z, ilo[4 * q + 3] = y.w;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) wr0[ROWW * 16 * q] = hp[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = rd0[q];
            hp[4 * q + 0] = x.x, hp[4 * q + 1] = x.y, hp[4 * q + 2] = x.z, hp[4 * q + 3] = x.w;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            re[q] = (i64)(((u64)(u32)__builtin_amdgcn_sbfe((int)hp[q], 0, 16) << 32) | rlo[q]);
            im[q] = (i64)(((u64)(u32)((int)hp[q] >> 16) << 32) | ilo[q]);
        }
        }
        // round 2: registers c3..0: STAGE 3, 2 (uniform twiddles), 1, 0
        wstage64<8, true>(re, im, k.wr3, k.wi3, a.st[X + 12]);
        wstage64<4, true>(re, im, k.wr2, k.wi2, a.st[X + 13]);
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
            wfly64_triv(re[g], im[g], re[g + 2], im[g + 2]);
            wfly64_mj(re[g + 1], im[g + 1], re[g + 3], im[g + 3]);
        }
#pragma unroll
        for (int g = 0; g < 16; g += 2) wfly64_triv(re[g], im[g], re[g + 1], im[g + 1]);
        // natural-order output index within the real frame: brev_L(256 r' + c), r' = 2^(L-12) t4 + low
        //   = 2^(L-4) rev4(c3..0) + 2^(L-8) rev4(c7..4) + 16 brev_(L-12)(low) + rev4(t4)      (L = 16: low = r0, t4 = j)
        typedef i64 v2l __attribute__((ext_vector_type(2)));
        if (NAT && (a.native & 2)) { // BITREV order out: memory index = core position p = (t4 << (L-4)) + 256 low + c
            // the thread's 16 results c3..0 of row t4 -> thread = c, registers = t4: a 16 x 16 exchange through the planes (re.lo / im.lo, then the packed high halves)
            u32 hq[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) hq[q] = ((u32)((u64)re[q] >> 32) & 0xFFFFu) | ((u32)((u64)im[q] >> 32) << 16);
            uint4 *const xw0 = reinterpret_cast<uint4 *>(lds + ROWW * tid), *const xw1 = reinterpret_cast<uint4 *>(lds + PLANEW + ROWW * tid);
            const u32 *const xr0 = lds + ROWW * (16 * hi4) + lo4, *const xr1 = xr0 + PLANEW; // + ROWW * t4
            u32 xl[16], yl[16];
            __syncthreads(); // the round's transpose reads are done
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xw0[q] = make_uint4((u32)re[4 * q], (u32)re[4 * q + 1], (u32)re[4 * q + 2], (u32)re[4 * q + 3]);
                xw1[q] = make_uint4((u32)im[4 * q], (u32)im[4 * q + 1], (u32)im[4 * q + 2], (u32)im[4 * q + 3]);
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 16; ++t) xl[t] = xr0[ROWW * t], yl[t] = xr1[ROWW * t];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) xw0[q] = make_uint4(hq[4 * q], hq[4 * q + 1], hq[4 * q + 2], hq[4 * q + 3]);
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 16; ++t) hq[t] = xr0[ROWW * t];
            v2l *dstp = reinterpret_cast<v2l *>(out) + (real << LX) + 256 * ulow; // wave-uniform
            unsigned tp = (unsigned)tid;
            asm volatile("" : "+v"(tp));
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const v2l y = {(i64)(((u64)(u32)(int)__builtin_amdgcn_sbfe((int)hq[t], 0, 16) << 32) | xl[t]), (i64)(((u64)(u32)((int)hq[t] >> 16) << 32) | yl[t])};
                // row t4 = t of the unit (XS > 0: block b = t >> (4 - XS), row r = ((t mod 2^(4-XS)) << (4 + XS)) | ulow of that block)
                const size_t row_off = XS ? ((size_t)(t >> (4 - XS)) << 16) + ((size_t)(t & ((1 << (4 - XS)) - 1)) << (12 + XS)) : (size_t)t << (L - 4);
                __builtin_nontemporal_store(y, at32(dstp + row_off, tp));
            }
            continue;
        }
        const int rlow = (int)(__brev((unsigned)ulow) >> (32 - (LX - 12)));
        v2l *dst = reinterpret_cast<v2l *>(out) + (real << LX) + 16 * rlow; // wave-uniform
        unsigned toff = (unsigned)((rev4w(hi4) << (LX - 8)) + rev4w(lo4));
        asm volatile("" : "+v"(toff));
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const v2l y = {re[q], im[q]};
            __builtin_nontemporal_store(y, at32(dst + ((size_t)rev4w(q) << (LX - 4)), toff));
        }
    }
}

// ---- the inverse: int_ifftNk on the same class (round 4) --------------------------------------------------------------------------------
// int_ifftNk (src/vhdl/fft/int_ifftNk.vhd:183-341), DATA_WIDTH 17 .. 24 in int32 containers, FORMAT = 1, natural order in and out: DIT
// STAGE s pairs the positions that differ in bit s (position p takes X[brev_L(p)]), multiplies the INPUT B at its own width 24 + s
// (int_dit2_fly.vhd:290-325: the re/im-swapped multiplier feed, T.im = the multiplier's re output, T.re = its im output) and grows one
// bit in X = A + T, Y = A - T.  The widths cross 32 bits behind STAGE 7, so the passes mirror the forward ones with the word sizes exchanged:
//   pass 1  k_wide16_q1   STAGE 0..7 over c = p7..p0 on int32 registers: the geometry of k_wide16_p2 (a workgroup = one unit of 16 rows whose
//                         bit-reversed indices are adjacent; the bit-reversed gather of the natural-order input in 128-byte runs), user array ->
//                         plan scratch (the forward layout [unit][c7..4][t4][c3..0], 2 KiB per register)
//   pass 2  k_wide16_q2   STAGE 8..L-1 over r = p15..p8 on 64-bit registers: the geometry of k_wide16_p1 (16 adjacent columns of a virtual frame),
//                         three-plane LDS transpose of the <= 36-bit values, natural-order store in 256-byte runs
// WideArgs::st[s] is STAGE s here (processing order = stage number); the slice width of a stage is its multiplier width mw = 24 + s.
template <bool UNIFORM_W>
__device__ __forceinline__ void wdit32(int &are, int &aim, int &bre, int &bim, int wr, int wi, const WideStage &s)
{
    if (UNIFORM_W) asm volatile("" : "+s"(wr), "+s"(wi));
    else asm volatile("" : "+v"(wr), "+v"(wi)); // (see wfly32)
    const u64 m2r = (u64)((i64)bim * wr), m1r = (u64)((i64)bre * wi); // the multiplier's re output on (DI_RE, DI_IM) = (B.im, B.re): T.im
    const u64 m2i = (u64)((i64)bim * wi), m1i = (u64)((i64)bre * wr); // its im output: T.re
    const u64 k = 0xFFFFFFFF00000000ull | s.keep;
    const u64 xi = (m2r & k) - (m1r & k), xr = (m2i & k) + (m1i & k);
    const int tre = (int)__builtin_amdgcn_alignbit((u32)(xr >> 32), (u32)xr, (u32)s.s2) >> s.s3;
    const int tim = (int)__builtin_amdgcn_alignbit((u32)(xi >> 32), (u32)xi, (u32)s.s2) >> s.s3;
    bre = are - tre, bim = aim - tim; // unscaled: exact, one bit of growth (int_dit2_fly.vhd:142-162)
    are += tre, aim += tim;
}
template <int H, bool UNIFORM_W = false>
__device__ __forceinline__ void wdstage32(int (&re)[16], int (&im)[16], const int (&wr)[H], const int (&wi)[H], const WideStage &s)
{
#pragma unroll
    for (int g = 0; g < 16; g += 2 * H)
#pragma unroll
        for (int j = 0; j < H; ++j) {
            wdit32<UNIFORM_W>(re[g + j], im[g + j], re[g + j + H], im[g + j + H], wr[j], wi[j], s);
            if (SCHED_GROUP && ((g / 2 + j) % SCHED_GROUP) == SCHED_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        }
}
template <bool UNIFORM_W>
__device__ __forceinline__ void wdit64(i64 &are, i64 &aim, i64 &bre, i64 &bim, int wr, int wi, const WideStage &s)
{
    if (UNIFORM_W) asm volatile("" : "+s"(wr), "+s"(wi));
    else asm volatile("" : "+v"(wr), "+v"(wi));
    const int rl = (int)bre, rh = (int)(bre >> 32) - (rl >> 31);
    const int il = (int)bim, ih = (int)(bim >> 32) - (il >> 31);
    u64 m2r = mul64x32(il, ih, wr), m1r = mul64x32(rl, rh, wi);
    u64 m2i = mul64x32(il, ih, wi), m1i = mul64x32(rl, rh, wr);
    const u64 k = 0xFFFFFFFF00000000ull | s.keep;
    m2r &= k, m1r &= k, m2i &= k, m1i &= k;
    const u64 xi = m2r - m1r, xr = m2i + m1i;
    i64 tre, tim;
    if (s.w32 < 1) { // a 32-bit multiplier (STAGE 8 of 24-bit data): the general form of the slice
        const int wo = 32 + s.w32;
        tre = (i64)(xr << (64 - s.sh - wo)) >> (64 - wo);
        tim = (i64)(xi << (64 - s.sh - wo)) >> (64 - wo);
    } else {
        const u32 lr = __builtin_amdgcn_alignbit((u32)(xr >> 32), (u32)xr, (u32)s.sh);
        const u32 li = __builtin_amdgcn_alignbit((u32)(xi >> 32), (u32)xi, (u32)s.sh);
        const int hr = __builtin_amdgcn_sbfe((int)(xr >> 32), s.sh, s.w32);
        const int hi = __builtin_amdgcn_sbfe((int)(xi >> 32), s.sh, s.w32);
        tre = (i64)(((u64)(u32)hr << 32) | lr);
        tim = (i64)(((u64)(u32)hi << 32) | li);
    }
    bre = are - tre, bim = aim - tim;
    are += tre, aim += tim;
}
template <int H>
__device__ __forceinline__ void wdstage64(i64 (&re)[16], i64 (&im)[16], const int (&wr)[H], const int (&wi)[H], const WideStage &s)
{
#pragma unroll
    for (int g = 0; g < 16; g += 2 * H)
#pragma unroll
        for (int j = 0; j < H; ++j) {
            wdit64<false>(re[g + j], im[g + j], re[g + j + H], im[g + j + H], wr[j], wi[j], s);
            if (SCHED_GROUP && ((g / 2 + j) % SCHED_GROUP) == SCHED_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        }
}

// XS > 0 (round 5, intfft_widelong.hip): N = 2^LX, LX = 16 + XS -- the unit's 16 rows (b, jtop) across the B = 2^XS blocks as in k_wide16_p2<.., XS>: the gather is
// the L = 16 one with LX in place of L, the store goes to the blocks' own [r0][c7..4][j][c3..0] layouts (what k_wide16_q2<16> reads block by block).  IN16: int16 containers
// (DATA_WIDTH <= 16).
template <int L, bool NAT = false, int XS = 0, bool IN16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_wide16_q1(const int2 *in, int2 *scr, const int2 *__restrict__ twt, const WideArgs a,
                                                                                             const W2Consts k, size_t nframes_user)
{
    static_assert(L >= 13 && L <= 16, "virtual 2^16-point frames");
    static_assert((XS == 0 && !IN16) || (L == 16 && XS >= 1 && XS <= 4), "long frames: whole 2^16-point blocks");
    constexpr int G = 1 << (16 - L);
    constexpr int LX = L + XS;
    const size_t nframes = (nframes_user + G - 1) / G; // virtual frames
    __shared__ __attribute__((aligned(16))) u32 lds[2 * PLANEW];
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;
    // round 2: thread = (j = hi4, c3..0 = lo4), registers c7..4: STAGE 4 index lo4, STAGE 5 index 16 jj + lo4, ... (frame invariant)
    int w7r[8], w7i[8], w6r[4], w6i[4], w5r[2], w5i[2], w4r[1], w4i[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int2 w = twt[127 + 16 * j + lo4];
        w7r[j] = w.x, w7i[j] = w.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int2 w = twt[63 + 16 * j + lo4];
        w6r[j] = w.x, w6i[j] = w.y;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int2 w = twt[31 + 16 * j + lo4];
        w5r[j] = w.x, w5i[j] = w.y;
    }
    {
        const int2 w = twt[15 + lo4];
        w4r[0] = w.x, w4i[0] = w.y;
    }
    // transpose: round-1 thread t = (c7..4 = hi4, j = lo4) holds registers c3..0 -> row t; round-2 thread (j = hi4, c3..0 = lo4), register
    // q = c7..4, reads (row 16 q + j, column c3..0) -- k_wide16_p2's transpose run backwards
    uint4 *const wr0 = reinterpret_cast<uint4 *>(lds + ROWW * tid);
    uint4 *const wr1 = reinterpret_cast<uint4 *>(lds + PLANEW + ROWW * tid);
    const u32 *const rd0 = lds + ROWW * hi4 + lo4;
    const u32 *const rd1 = rd0 + PLANEW;

    const size_t units = nframes << (4 + XS); // work unit u = 16 f + (g, low)   (XS > 0: u = (f << (4 + XS)) + ulow)
    for (size_t u = blockIdx.x; u < units; u += gridDim.x) {
        const size_t f = u >> (4 + XS);
        const int r0 = (int)(u & ((16u << XS) - 1u));
        const int ug = XS ? 0 : r0 >> (L - 12), ulow = XS ? r0 : r0 & ((1 << (L - 12)) - 1);
        const size_t real = f * G + (size_t)ug; // the real frame these 16 rows belong to
        if (L < 16 && real >= nframes_user) continue;
        typedef int v2i __attribute__((ext_vector_type(2)));
        // position (r' = 2^(L-12) t4 + low, c) takes X[brev_L] = in[2^(L-4) rev4(c3..0) + 2^(L-8) rev4(c7..4) + 16 brev_(L-12)(low) + rev4(t4)]
        const int rlow = (int)(__brev((unsigned)ulow) >> (32 - (LX - 12)));
        const v2i *src = reinterpret_cast<const v2i *>(in) + (real << LX) + 16 * rlow; // wave-uniform
        unsigned toff = (unsigned)((rev4w(hi4) << (LX - 8)) + rev4w(lo4));
        // the thread's part of the scratch index: tid in the unit's own 4096-sample region; XS > 0: block b = hi4 >> (4 - XS), row j = (jtop << XS) | (ulow >> 4)
        unsigned tid_l = XS ? (unsigned)((hi4 >> (4 - XS)) * 65536 + 16 * ((hi4 & ((1 << (4 - XS)) - 1)) << XS) + lo4) : (unsigned)tid;
        asm volatile("" : "+v"(toff), "+v"(tid_l)); // (opaque per iteration, see k_wide16_p1)
        int re[16], im[16];
        // BITREV order: row t4 = t of the unit sits t << (L-4) positions up (XS > 0: block b = t >> (4 - XS), row ((t mod 2^(4-XS)) << (4 + XS)) | ulow of that block)
        auto row_off = [&](int t) -> size_t { return XS ? ((size_t)(t >> (4 - XS)) << 16) + ((size_t)(t & ((1 << (4 - XS)) - 1)) << (12 + XS)) : (size_t)t << (L - 4); };
        unsigned tid_c = (unsigned)tid;
        asm volatile("" : "+v"(tid_c));
        if (NAT && IN16 && (a.native & 2)) { // BITREV order in, int16 containers: one plane of packed samples through the exchange
            const u32 *srcp = reinterpret_cast<const u32 *>(in) + (real << LX) + 256 * ulow; // wave-uniform
            u32 *const xw0 = lds + ROWW * (16 * hi4) + lo4; // + ROWW * t4
            const uint4 *const xr0 = reinterpret_cast<const uint4 *>(lds + ROWW * tid);
            __syncthreads(); // the previous unit's transpose reads are done
#pragma unroll
            for (int t = 0; t < 16; ++t) xw0[ROWW * t] = INTFFT_LD(at32(srcp + row_off(t), tid_c));
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 x = xr0[q];
                re[4 * q] = __builtin_amdgcn_sbfe((int)x.x, 0, a.dw), im[4 * q] = __builtin_amdgcn_sbfe((int)x.x, 16, a.dw);
                re[4 * q + 1] = __builtin_amdgcn_sbfe((int)x.y, 0, a.dw), im[4 * q + 1] = __builtin_amdgcn_sbfe((int)x.y, 16, a.dw);
                re[4 * q + 2] = __builtin_amdgcn_sbfe((int)x.z, 0, a.dw), im[4 * q + 2] = __builtin_amdgcn_sbfe((int)x.z, 16, a.dw);
                re[4 * q + 3] = __builtin_amdgcn_sbfe((int)x.w, 0, a.dw), im[4 * q + 3] = __builtin_amdgcn_sbfe((int)x.w, 16, a.dw);
            }
        } else if (NAT && !IN16 && (a.native & 2)) { // BITREV order in: memory index = core position; coalesced loads (thread = c, registers = t4), then the 16 x 16 exchange
            const v2i *srcp = reinterpret_cast<const v2i *>(in) + (real << LX) + 256 * ulow; // wave-uniform
            u32 *const xw0 = lds + ROWW * (16 * hi4) + lo4, *const xw1 = xw0 + PLANEW; // + ROWW * t4
            const uint4 *const xr0 = reinterpret_cast<const uint4 *>(lds + ROWW * tid), *const xr1 = reinterpret_cast<const uint4 *>(lds + PLANEW + ROWW * tid);
            __syncthreads(); // the previous unit's transpose reads are done
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const v2i x = INTFFT_LD(at32(srcp + row_off(t), tid_c));
                xw0[ROWW * t] = (u32)x.x, xw1[ROWW * t] = (u32)x.y;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 x = xr0[q], y = xr1[q];
                re[4 * q] = __builtin_amdgcn_sbfe((int)x.x, 0, a.dw), re[4 * q + 1] = __builtin_amdgcn_sbfe((int)x.y, 0, a.dw);
                re[4 * q + 2] = __builtin_amdgcn_sbfe((int)x.z, 0, a.dw), re[4 * q + 3] = __builtin_amdgcn_sbfe((int)x.w, 0, a.dw);
                im[4 * q] = __builtin_amdgcn_sbfe((int)y.x, 0, a.dw), im[4 * q + 1] = __builtin_amdgcn_sbfe((int)y.y, 0, a.dw);
                im[4 * q + 2] = __builtin_amdgcn_sbfe((int)y.z, 0, a.dw), im[4 * q + 3] = __builtin_amdgcn_sbfe((int)y.w, 0, a.dw);
            }
        } else if constexpr (IN16) {
            const u32 *src16 = reinterpret_cast<const u32 *>(in) + (real << LX) + 16 * rlow;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const u32 x = *at32(src16 + ((size_t)rev4w(q) << (LX - 4)), toff);
                re[q] = __builtin_amdgcn_sbfe((int)x, 0, a.dw);
                im[q] = __builtin_amdgcn_sbfe((int)x, 16, a.dw);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const v2i x = *at32(src + ((size_t)rev4w(q) << (LX - 4)), toff); // (plain: the sixteen 8-byte pieces of a 128-byte line come from sixteen lanes of one instruction)
                re[q] = __builtin_amdgcn_sbfe(x.x, 0, a.dw); // conv_std_logic_vector(.., DATA_WIDTH): wrap on load
                im[q] = __builtin_amdgcn_sbfe(x.y, 0, a.dw);
            }
        }
        // STAGE 0: T = B (int_dit2_fly.vhd:221-230); STAGE 1: T = B on even positions, +j B with the negation quirk on odd ones (:234-286)
#pragma unroll
        for (int g = 0; g < 16; g += 2) {
            const int xr = re[g] + re[g + 1], xi = im[g] + im[g + 1];
            re[g + 1] = re[g] - re[g + 1], im[g + 1] = im[g] - im[g + 1];
            re[g] = xr, im[g] = xi;
        }
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
            { // even position
                const int xr = re[g] + re[g + 2], xi = im[g] + im[g + 2];
                re[g + 2] = re[g] - re[g + 2], im[g + 2] = im[g] - im[g + 2];
                re[g] = xr, im[g] = xi;
            }
            { // odd position: T.im = B.re, T.re = B.im >= 0 ? -B.im : ~B.im
                const int tre = (im[g + 3] >> 31) - im[g + 3], tim = re[g + 3];
                re[g + 3] = re[g + 1] - tre, im[g + 3] = im[g + 1] - tim;
                re[g + 1] += tre, im[g + 1] += tim;
            }
        }
        wdstage32<4, true>(re, im, k.wr2, k.wi2, a.st[2]);
        wdstage32<8, true>(re, im, k.wr3, k.wi3, a.st[3]);
        __syncthreads(); // the previous unit's reads are done
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wr0[q] = make_uint4((u32)re[4 * q], (u32)re[4 * q + 1], (u32)re[4 * q + 2], (u32)re[4 * q + 3]);
            wr1[q] = make_uint4((u32)im[4 * q], (u32)im[4 * q + 1], (u32)im[4 * q + 2], (u32)im[4 * q + 3]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            re[q] = (int)rd0[ROWW * 16 * q];
            im[q] = (int)rd1[ROWW * 16 * q];
        }
        wdstage32<1>(re, im, w4r, w4i, a.st[4]);
        wdstage32<2>(re, im, w5r, w5i, a.st[5]);
        wdstage32<4>(re, im, w6r, w6i, a.st[6]);
        wdstage32<8>(re, im, w7r, w7i, a.st[7]);
        // scratch, the forward layout: unit r0 = [c7..4 = q][t4 = hi4][c3..0 = lo4]: 2 KiB per register
        v2i *dst = XS ? reinterpret_cast<v2i *>(scr) + (f << (16 + XS)) + 4096 * (r0 & 15) + 16 * (r0 >> 4) : reinterpret_cast<v2i *>(scr) + f * 65536 + 4096 * r0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const v2i y = {re[q], im[q]};
            *at32(dst + 256 * q, tid_l) = y;
        }
    }
}

template <int L, bool IN64 = false, bool NAT = false> // IN64: the scratch holds 64-bit words (k_wide64_q1 in front)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_wide16_q2(const int2 *scr, i64 *out, const int2 *__restrict__ twt, const WideArgs a,
                                                                                             size_t nframes_user)
{
    static_assert(L >= 13 && L <= 16, "virtual 2^16-point frames");
    constexpr int G = 1 << (16 - L);
    const size_t nframes = (nframes_user + G - 1) / G; // virtual frames
    __shared__ __attribute__((aligned(16))) u32 lds[2 * PLANEW];
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;
    const int tile = blockIdx.x & 15;
    const int c = 16 * tile + lo4;
    // round 1: thread = (r7..4 = hi4, c3..0 = lo4), registers r3..0: STAGE 8 + b on register bit b, index (rr << 8) | c: per column
    int w11r[8], w11i[8], w10r[4], w10i[4], w9r[2], w9i[2], w8r[1], w8i[1];
    // round 2: thread = (r3..0 = hi4, c3..0), registers r7..4: STAGE 12 + b, index n mod 2^s with n = c + 256 hi4 + 4096 jj (stages >= L: frame-number bits, skipped)
    int w15r[8], w15i[8], w14r[4], w14i[4], w13r[2], w13i[2], w12r[1], w12i[1];
    {
        int2 w;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            w = twt[2047 + c + 256 * j];
            w11r[j] = w.x, w11i[j] = w.y;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w = twt[1023 + c + 256 * j];
            w10r[j] = w.x, w10i[j] = w.y;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            w = twt[511 + c + 256 * j];
            w9r[j] = w.x, w9i[j] = w.y;
        }
        w = twt[255 + c];
        w8r[0] = w.x, w8i[0] = w.y;
        const int base = c + 256 * hi4;
        w = twt[4095 + base];
        w12r[0] = w.x, w12i[0] = w.y;
        if constexpr (L > 13) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                w = twt[8191 + ((base + 4096 * j) & 8191)];
                w13r[j] = w.x, w13i[j] = w.y;
            }
        }
        if constexpr (L > 14) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w = twt[16383 + ((base + 4096 * j) & 16383)];
                w14r[j] = w.x, w14i[j] = w.y;
            }
        }
        if constexpr (L > 15) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                w = twt[32767 + base + 4096 * j];
                w15r[j] = w.x, w15i[j] = w.y;
            }
        }
    }
    // transpose: round-1 thread t = (r7..4, c3..0) holds registers r3..0 -> row t; round-2 thread (r3..0 = hi4, c3..0 = lo4), register j = r7..4,
    // reads (row 16 j + lo4, column hi4) -- k_wide16_p1's transpose run backwards; <= 36-bit values through three dword planes
    uint4 *const wr0 = reinterpret_cast<uint4 *>(lds + ROWW * tid);
    uint4 *const wr1 = reinterpret_cast<uint4 *>(lds + PLANEW + ROWW * tid);
    const u32 *const rd0 = lds + ROWW * lo4 + hi4;
    const u32 *const rd1 = rd0 + PLANEW;

    const size_t fstep = gridDim.x >> 4;
    for (size_t f = blockIdx.x >> 4; f < nframes; f += fstep) {
        const bool partial = L < 16 && (f + 1) * G > nframes_user; // last group: rows of absent frames read as 0, are not stored
        typedef int v2i __attribute__((ext_vector_type(2)));
        const int g = hi4 >> (L - 12); // the real frame (within the group) of this thread's round-1 rows
        const bool present = !partial || f * G + (size_t)g < nframes_user;
        const v2i *src = reinterpret_cast<const v2i *>(scr) + f * 65536; // wave-uniform
        unsigned toff2 = (unsigned)(256 * tile + lo4 + 4096 * (g << (L - 12)) + 16 * ((hi4 << (16 - L)) & 15));
        unsigned toff = (unsigned)(c + 256 * hi4);
        asm volatile("" : "+v"(toff), "+v"(toff2));
        i64 re[16], im[16];
        if (present && IN64) { // 16 bytes per sample
            typedef i64 v2li __attribute__((ext_vector_type(2)));
            const v2li *src64 = reinterpret_cast<const v2li *>(scr) + f * 65536;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const v2li x = INTFFT_LD(at32(src64 + 4096 * (q & ((1 << (L - 12)) - 1)) + 16 * (q >> (L - 12)), toff2));
                re[q] = x.x, im[q] = x.y;
            }
        } else if (present) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const v2i x = INTFFT_LD(at32(src + 4096 * (q & ((1 << (L - 12)) - 1)) + 16 * (q >> (L - 12)), toff2));
                re[q] = x.x, im[q] = x.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) re[q] = 0, im[q] = 0;
        }
        wdstage64<1>(re, im, w8r, w8i, a.st[8]);
        wdstage64<2>(re, im, w9r, w9i, a.st[9]);
        wdstage64<4>(re, im, w10r, w10i, a.st[10]);
        wdstage64<8>(re, im, w11r, w11i, a.st[11]);
        u32 hp[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) hp[q] = ((u32)((u64)re[q] >> 32) & 0xFFFFu) | ((u32)((u64)im[q] >> 32) << 16);
        u32 rlo[16], ilo[16];
        __syncthreads(); // the previous frame's reads are done
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wr0[q] = make_uint4((u32)re[4 * q], (u32)re[4 * q + 1], (u32)re[4 * q + 2], (u32)re[4 * q + 3]);
            wr1[q] = make_uint4((u32)im[4 * q], (u32)im[4 * q + 1], (u32)im[4 * q + 2], (u32)im[4 * q + 3]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            rlo[j] = rd0[ROWW * 16 * j];
            ilo[j] = rd1[ROWW * 16 * j];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) wr0[q] = make_uint4(hp[4 * q], hp[4 * q + 1], hp[4 * q + 2], hp[4 * q + 3]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) hp[j] = rd0[ROWW * 16 * j];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            re[j] = (i64)(((u64)(u32)__builtin_amdgcn_sbfe((int)hp[j], 0, 16) << 32) | rlo[j]);
            im[j] = (i64)(((u64)(u32)((int)hp[j] >> 16) << 32) | ilo[j]);
        }
        wdstage64<1>(re, im, w12r, w12i, a.st[12]);
        if constexpr (L > 13) wdstage64<2>(re, im, w13r, w13i, a.st[13]);
        if constexpr (L > 14) wdstage64<4>(re, im, w14r, w14i, a.st[14]);
        if constexpr (L > 15) wdstage64<8>(re, im, w15r, w15i, a.st[15]);
        // natural order: x[(16 j + hi4) 256 + c] of the virtual frame; register j's real frame is j >> (L - 12)
        typedef i64 v2l __attribute__((ext_vector_type(2)));
        if (NAT && (a.native & 1)) { // HALVES order out: one 32-byte store per register pair (j, j | 2^(L-13))
            typedef i64 v4l __attribute__((ext_vector_type(4)));
            v4l *dst4 = reinterpret_cast<v4l *>(out) + f * 32768;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j & halves_pair_bit<L>()) continue;
                const v4l y = {re[j], im[j], re[j | halves_pair_bit<L>()], im[j | halves_pair_bit<L>()]};
                if (!partial || f * G + (size_t)(j >> (L - 12)) < nframes_user) __builtin_nontemporal_store(y, at32(dst4 + halves_pair_index<L>(j), toff));
            }
            continue;
        }
        v2l *dst = reinterpret_cast<v2l *>(out) + f * 65536;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const v2l y = {re[j], im[j]};
            if (!partial || f * G + (size_t)(j >> (L - 12)) < nframes_user) __builtin_nontemporal_store(y, at32(dst + 4096 * j, toff));
        }
    }
}


// ---- DATA_WIDTH 25 .. 32 with bit growth (round 5): the FIRST pass on 64-bit words too -------------------------------------------------------
// int_fft_single_path.vhd:15 documents DATA_WIDTH 8 .. 32; with FORMAT = 1 at N >= 8192 such data leave int32 in the first stages (DATA_WIDTH + NFFT - 8
// > 32), so pass 1 cannot be k_wide16_p1 / k_wide16_q1.  k_wide64_p1 / k_wide64_q1 are those kernels on 64-bit registers -- the same tiles, loads and
// scratch layout (in 16-byte samples), the butterflies of pass 2 (wfly64 / wdit64: exact 64-bit products, mw + TWDL_WIDTH <= 64, checked by the planner),
// the three-plane LDS transpose of the <= 36-bit values between the rounds -- and k_wide16_p2 / k_wide16_q2 read their scratch as 64-bit words
// (IN64).  Results up to 48 bits (the 16-bit
// high plane of the second pass's transpose).
template <int L, bool NAT = false, int XS = 0> // XS > 0: the 2^16-point blocks of N = 2^(16 + XS) behind k_wide_pre (intfft_widelong.hip), as in k_wide16_p1
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_wide64_p1(const int2 *in, i64 *scr, const int2 *__restrict__ twt, const WideArgs a,
                                                                                             size_t nframes_user)
{
    static_assert(L >= 13 && L <= 16, "virtual 2^16-point frames");
    static_assert(XS == 0 || (L == 16 && !NAT && XS <= 4), "long frames: whole 2^16-point blocks, natural order");
    constexpr int G = 1 << (16 - L);
    const size_t nframes = (nframes_user + G - 1) / G;
    constexpr int X = L - 16 + XS;
    __shared__ __attribute__((aligned(16))) u32 lds[2 * PLANEW];
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;
    const int tile = blockIdx.x & 15;
    const int c = 16 * tile + lo4;
    int w15r[8], w15i[8], w14r[4], w14i[4], w13r[2], w13i[2], w12r[1], w12i[1];
    int w11r[8], w11i[8], w10r[4], w10i[4], w9r[2], w9i[2], w8r[1], w8i[1];
    {
        const int base = c + 256 * hi4;
        if constexpr (L > 15) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int2 w = twt[32767 + base + 4096 * j];
                w15r[j] = w.x, w15i[j] = w.y;
            }
        }
        if constexpr (L > 14) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int2 w = twt[16383 + ((base + 4096 * j) & 16383)];
                w14r[j] = w.x, w14i[j] = w.y;
            }
        }
        if constexpr (L > 13) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int2 w = twt[8191 + ((base + 4096 * j) & 8191)];
                w13r[j] = w.x, w13i[j] = w.y;
            }
        }
        int2 w = twt[4095 + base];
        w12r[0] = w.x, w12i[0] = w.y;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            w = twt[2047 + c + 256 * j];
            w11r[j] = w.x, w11i[j] = w.y;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w = twt[1023 + c + 256 * j];
            w10r[j] = w.x, w10i[j] = w.y;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            w = twt[511 + c + 256 * j];
            w9r[j] = w.x, w9i[j] = w.y;
        }
        w = twt[255 + c];
        w8r[0] = w.x, w8i[0] = w.y;
    }
    // transpose as in k_wide16_p1: element (thread (hi4, lo4), register j) -> row 16 j + lo4, column hi4; thread t reads row t; three dword planes
    u32 *const wr0 = lds + ROWW * lo4 + hi4;
    u32 *const wr1 = wr0 + PLANEW;
    const uint4 *const rd0 = reinterpret_cast<const uint4 *>(lds + ROWW * tid);
    const uint4 *const rd1 = reinterpret_cast<const uint4 *>(lds + PLANEW + ROWW * tid);

    const size_t fstep = gridDim.x >> 4;
    for (size_t f = blockIdx.x >> 4; f < nframes; f += fstep) {
        i64 re[16], im[16];
        const int2 *src = in + f * 65536;
        unsigned toff = (unsigned)(c + 256 * hi4);
        asm volatile("" : "+v"(toff));
        const bool partial = L < 16 && (f + 1) * G > nframes_user;
        typedef int v2i __attribute__((ext_vector_type(2)));
        // conv_std_logic_vector(.., DATA_WIDTH): wrap on load (v_bfe_i32 takes the width modulo 32: nothing to wrap at 32 bits; the builtin returns unsigned)
        auto wrap = [&](int x) { return a.dw >= 32 ? x : (int)__builtin_amdgcn_sbfe(x, 0, a.dw); };
        if (NAT && (a.native & 1)) { // HALVES order in (see k_wide16_p1)
            typedef int v4i __attribute__((ext_vector_type(4)));
            const v4i *src4 = reinterpret_cast<const v4i *>(in) + f * 32768;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j & halves_pair_bit<L>()) continue;
                v4i x = {0, 0, 0, 0};
                if (!partial || f * G + (size_t)(j >> (L - 12)) < nframes_user) x = INTFFT_LD(at32(src4 + halves_pair_index<L>(j), toff));
                re[j] = wrap(x.x), im[j] = wrap(x.y);
                re[j | halves_pair_bit<L>()] = wrap(x.z), im[j | halves_pair_bit<L>()] = wrap(x.w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                v2i x = {0, 0};
                if (!partial || f * G + (size_t)(j >> (L - 12)) < nframes_user) x = INTFFT_LD(at32(reinterpret_cast<const v2i *>(src + 4096 * j), toff));
                re[j] = wrap(x.x);
                im[j] = wrap(x.y);
            }
        }
        if constexpr (L > 15) wstage64<8>(re, im, w15r, w15i, a.st[X + 0]);
        if constexpr (L > 14) wstage64<4>(re, im, w14r, w14i, a.st[X + 1]);
        if constexpr (L > 13) wstage64<2>(re, im, w13r, w13i, a.st[X + 2]);
        wstage64<1>(re, im, w12r, w12i, a.st[X + 3]);
        u32 hp[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) hp[q] = ((u32)((u64)re[q] >> 32) & 0xFFFFu) | ((u32)((u64)im[q] >> 32) << 16);
        u32 rlo[16], ilo[16];
        __syncthreads(); // the previous frame's reads are done
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            wr0[ROWW * 16 * j] = (u32)re[j];
            wr1[ROWW * 16 * j] = (u32)im[j];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = rd0[q], y = rd1[q];
            rlo[4 * q + 0] = x.x, rlo[4 * q + 1] = x.y, rlo[4 * q + 2] = x.z, rlo[4 * q + 3] = x.w;
            ilo[4 * q + 0] = y.x, ilo[4 * q + 1] = y.y, ilo[4 * q + 2] = y.z, ilo[4 * q + 3] = y.w;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) wr0[ROWW * 16 * j] = hp[j];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = rd0[q];
            hp[4 * q + 0] = x.x, hp[4 * q + 1] = x.y, hp[4 * q + 2] = x.z, hp[4 * q + 3] = x.w;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            re[q] = (i64)(((u64)(u32)__builtin_amdgcn_sbfe((int)hp[q], 0, 16) << 32) | rlo[q]);
            im[q] = (i64)(((u64)(u32)((int)hp[q] >> 16) << 32) | ilo[q]);
        }
        if constexpr (L == 16) { // round-2 twiddles (they depend on the column only) re-read per frame from the L2-resident table instead of living in 30
            // VGPRs over the frame loop: at L = 16 (all eight stages, 30 + 30 twiddle registers) that removes the kernel's 7 spilled VGPRs: 85 -> 94 Gsample/s;
            // at L < 16 nothing spills and holding them is 2-3 % faster; two waves per SIMD: 90, four with the reload: 71
            int cc = c;
            asm volatile("" : "+v"(cc));
            int2 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                w = twt[2047 + cc + 256 * j];
                w11r[j] = w.x, w11i[j] = w.y;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w = twt[1023 + cc + 256 * j];
                w10r[j] = w.x, w10i[j] = w.y;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                w = twt[511 + cc + 256 * j];
                w9r[j] = w.x, w9i[j] = w.y;
            }
            w = twt[255 + cc];
            w8r[0] = w.x, w8i[0] = w.y;
        }
        wstage64<8>(re, im, w11r, w11i, a.st[X + 4]);
        wstage64<4>(re, im, w10r, w10i, a.st[X + 5]);
        wstage64<2>(re, im, w9r, w9i, a.st[X + 6]);
        wstage64<1>(re, im, w8r, w8i, a.st[X + 7]);
        const int g = hi4 >> (L - 12);
        if (partial && f * G + (size_t)g >= nframes_user) continue;
        // scratch: 16 bytes per sample in the layout of k_wide16_p1's (a 12-byte form -- low dword pairs + a plane of packed 16-bit high halves, 56 -> 48
        // bytes of traffic per sample for the plan -- measured SLOWER forward: N = 2^16 77 against 89 Gsample/s, N = 2^13 99 against 127; inverse unchanged)
        typedef i64 v2l __attribute__((ext_vector_type(2)));
        v2l *dst = reinterpret_cast<v2l *>(scr) + f * 65536;
        unsigned toff2 = (unsigned)(256 * tile + lo4 + 4096 * (g << (L - 12)) + 16 * ((hi4 << (16 - L)) & 15));
        asm volatile("" : "+v"(toff2));
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const v2l y = {re[q], im[q]};
            *at32(dst + 4096 * (q & ((1 << (L - 12)) - 1)) + 16 * (q >> (L - 12)), toff2) = y;
        }
    }
}

template <int H, bool UNIFORM_W>
__device__ __forceinline__ void wdstage64u(i64 (&re)[16], i64 (&im)[16], const int (&wr)[H], const int (&wi)[H], const WideStage &s)
{
#pragma unroll
    for (int g = 0; g < 16; g += 2 * H)
#pragma unroll
        for (int j = 0; j < H; ++j) {
            wdit64<UNIFORM_W>(re[g + j], im[g + j], re[g + j + H], im[g + j + H], wr[j], wi[j], s);
            if (SCHED_GROUP && ((g / 2 + j) % SCHED_GROUP) == SCHED_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        }
}

template <int L, bool NAT = false, int XS = 0> // XS > 0: N = 2^(16 + XS), the rows of a unit across the blocks (see k_wide16_q1)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_wide64_q1(const int2 *in, i64 *scr, const int2 *__restrict__ twt, const WideArgs a,
                                                                                             const W2Consts k, size_t nframes_user)
{
    static_assert(L >= 13 && L <= 16, "virtual 2^16-point frames");
    static_assert(XS == 0 || (L == 16 && XS <= 4), "long frames: whole 2^16-point blocks");
    constexpr int G = 1 << (16 - L);
    constexpr int LX = L + XS;
    const size_t nframes = (nframes_user + G - 1) / G;
    __shared__ __attribute__((aligned(16))) u32 lds[2 * PLANEW];
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;
    int w7r[8], w7i[8], w6r[4], w6i[4], w5r[2], w5i[2], w4r[1], w4i[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int2 w = twt[127 + 16 * j + lo4];
        w7r[j] = w.x, w7i[j] = w.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int2 w = twt[63 + 16 * j + lo4];
        w6r[j] = w.x, w6i[j] = w.y;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int2 w = twt[31 + 16 * j + lo4];
        w5r[j] = w.x, w5i[j] = w.y;
    }
    {
        const int2 w = twt[15 + lo4];
        w4r[0] = w.x, w4i[0] = w.y;
    }
    // k_wide16_q1's transpose (round-1 thread t writes row t; round-2 thread (j = hi4, c3..0 = lo4), register q, reads (row 16 q + j, column c3..0)), three planes
    uint4 *const wr0 = reinterpret_cast<uint4 *>(lds + ROWW * tid);
    uint4 *const wr1 = reinterpret_cast<uint4 *>(lds + PLANEW + ROWW * tid);
    const u32 *const rd0 = lds + ROWW * hi4 + lo4;
    const u32 *const rd1 = rd0 + PLANEW;

    const size_t units = nframes << (4 + XS);
    for (size_t u = blockIdx.x; u < units; u += gridDim.x) {
        const size_t f = u >> (4 + XS);
        const int r0 = (int)(u & ((16u << XS) - 1u));
        const int ug = XS ? 0 : r0 >> (L - 12), ulow = XS ? r0 : r0 & ((1 << (L - 12)) - 1);
        const size_t real = f * G + (size_t)ug;
        if (L < 16 && real >= nframes_user) continue;
        typedef int v2i __attribute__((ext_vector_type(2)));
        const int rlow = (int)(__brev((unsigned)ulow) >> (32 - (LX - 12)));
        const v2i *src = reinterpret_cast<const v2i *>(in) + (real << LX) + 16 * rlow;
        unsigned toff = (unsigned)((rev4w(hi4) << (LX - 8)) + rev4w(lo4));
        unsigned tid_l = XS ? (unsigned)((hi4 >> (4 - XS)) * 65536 + 16 * ((hi4 & ((1 << (4 - XS)) - 1)) << XS) + lo4) : (unsigned)tid;
        asm volatile("" : "+v"(toff), "+v"(tid_l));
        i64 re[16], im[16];
        auto wrap = [&](int x) { return a.dw >= 32 ? x : (int)__builtin_amdgcn_sbfe(x, 0, a.dw); }; // (the builtin returns unsigned)
        if (NAT && (a.native & 2)) { // BITREV order in (see k_wide16_q1)
            auto row_off = [&](int t) -> size_t { return XS ? ((size_t)(t >> (4 - XS)) << 16) + ((size_t)(t & ((1 << (4 - XS)) - 1)) << (12 + XS)) : (size_t)t << (L - 4); };
            unsigned tid_c = (unsigned)tid;
            asm volatile("" : "+v"(tid_c));
            const v2i *srcp = reinterpret_cast<const v2i *>(in) + (real << LX) + 256 * ulow;
            u32 *const xw0 = lds + ROWW * (16 * hi4) + lo4, *const xw1 = xw0 + PLANEW;
            const uint4 *const xr0 = reinterpret_cast<const uint4 *>(lds + ROWW * tid), *const xr1 = reinterpret_cast<const uint4 *>(lds + PLANEW + ROWW * tid);
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const v2i x = INTFFT_LD(at32(srcp + row_off(t), tid_c));
                xw0[ROWW * t] = (u32)x.x, xw1[ROWW * t] = (u32)x.y;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 x = xr0[q], y = xr1[q];
                re[4 * q] = wrap((int)x.x), re[4 * q + 1] = wrap((int)x.y), re[4 * q + 2] = wrap((int)x.z), re[4 * q + 3] = wrap((int)x.w);
                im[4 * q] = wrap((int)y.x), im[4 * q + 1] = wrap((int)y.y), im[4 * q + 2] = wrap((int)y.z), im[4 * q + 3] = wrap((int)y.w);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const v2i x = *at32(src + ((size_t)rev4w(q) << (LX - 4)), toff);
                re[q] = wrap(x.x);
                im[q] = wrap(x.y);
            }
        }
        // STAGE 0: T = B; STAGE 1: T = B on even positions, +j B with the negation quirk on odd ones (int_dit2_fly.vhd:221-286)
#pragma unroll
        for (int g = 0; g < 16; g += 2) {
            const i64 xr = re[g] + re[g + 1], xi = im[g] + im[g + 1];
            re[g + 1] = re[g] - re[g + 1], im[g + 1] = im[g] - im[g + 1];
            re[g] = xr, im[g] = xi;
        }
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
            {
                const i64 xr = re[g] + re[g + 2], xi = im[g] + im[g + 2];
                re[g + 2] = re[g] - re[g + 2], im[g + 2] = im[g] - im[g + 2];
                re[g] = xr, im[g] = xi;
            }
            {
                const i64 tre = (im[g + 3] >> 63) - im[g + 3], tim = re[g + 3];
                re[g + 3] = re[g + 1] - tre, im[g + 3] = im[g + 1] - tim;
                re[g + 1] += tre, im[g + 1] += tim;
            }
        }
        wdstage64u<4, true>(re, im, k.wr2, k.wi2, a.st[2]);
        wdstage64u<8, true>(re, im, k.wr3, k.wi3, a.st[3]);
        u32 hp[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) hp[q] = ((u32)((u64)re[q] >> 32) & 0xFFFFu) | ((u32)((u64)im[q] >> 32) << 16);
        u32 rlo[16], ilo[16];
        __syncthreads(); // the previous unit's reads are done
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wr0[q] = make_uint4((u32)re[4 * q], (u32)re[4 * q + 1], (u32)re[4 * q + 2], (u32)re[4 * q + 3]);
            wr1[q] = make_uint4((u32)im[4 * q], (u32)im[4 * q + 1], (u32)im[4 * q + 2], (u32)im[4 * q + 3]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            rlo[q] = rd0[ROWW * 16 * q];
            ilo[q] = rd1[ROWW * 16 * q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) wr0[q] = make_uint4(hp[4 * q], hp[4 * q + 1], hp[4 * q + 2], hp[4 * q + 3]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) hp[q] = rd0[ROWW * 16 * q];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            re[q] = (i64)(((u64)(u32)__builtin_amdgcn_sbfe((int)hp[q], 0, 16) << 32) | rlo[q]);
            im[q] = (i64)(((u64)(u32)((int)hp[q] >> 16) << 32) | ilo[q]);
        }
        wdstage64<1>(re, im, w4r, w4i, a.st[4]);
        wdstage64<2>(re, im, w5r, w5i, a.st[5]);
        wdstage64<4>(re, im, w6r, w6i, a.st[6]);
        wdstage64<8>(re, im, w7r, w7i, a.st[7]);
        typedef i64 v2l __attribute__((ext_vector_type(2)));
        v2l *dst = XS ? reinterpret_cast<v2l *>(scr) + (f << (16 + XS)) + 4096 * (r0 & 15) + 16 * (r0 >> 4) : reinterpret_cast<v2l *>(scr) + f * 65536 + 4096 * r0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const v2l y = {re[q], im[q]};
            *at32(dst + 256 * q, tid_l) = y;
        }
    }
}

// orders the two-pass wide kernels take: natural, or the core's own beat order on either side (NAT instantiations, round 5)
#ifndef INTFFT_WIDE16_TEMPLATES_ONLY /* intfft_widelong.hip includes the kernel templates above and stops here */
static bool wide_orders_ok(int direction, int in_order, int out_order)
{
    if (direction == 0) return (in_order == 0 || in_order == 2) && (out_order == 0 || out_order == 1); // int_fftNk: NATURAL | HALVES in, NATURAL | BITREV out
    return (in_order == 0 || in_order == 1) && (out_order == 0 || out_order == 2);                     // int_ifftNk: NATURAL | BITREV in, NATURAL | HALVES out
}

// which first pass a configuration takes: 0 none, 1 int32 (k_wide16_p1 / q1), 2 64-bit words (k_wide64_p1 / q1; round 5)
int wide16_class(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order, int out_order)
{
    if (wide16_supported(log2n, data_width, twdl_width, format, direction, use_fly, in_order, out_order)) return 1;
    // DATA_WIDTH up to 32 in int32 containers, results of 33 .. 48 bits in int64 containers; the per-stage conditions (exact 64-bit products, the
    // slice inside one dword pair) are checked by the planner on the stage list
    const bool common = log2n >= 13 && log2n <= 16 && data_width >= 17 && data_width <= 32 && data_width + log2n > 32 && data_width + log2n <= 48 &&
                        twdl_width >= 8 && twdl_width <= 24 && format == 1 && use_fly == 1 && (direction == 0 || direction == 1) &&
                        wide_orders_ok(direction, in_order, out_order);
    return common && (direction == 0 || direction == 1) ? 2 : 0;
}

bool wide16_supported(int log2n, int data_width, int twdl_width, int format, int direction, int use_fly, int in_order,
                      int out_order)
{
    // int32 containers in (DATA_WIDTH 17..32), pass 1 within 32 bits (DATA_WIDTH + NFFT - 8 <= 32), results in int64 containers
    // of at most 40 bits (the 64-bit products and the three-plane transpose of pass 2)
    // the inverse (round 4): STAGE 0..7 on int32 (DATA_WIDTH + 8 <= 32), STAGE 8..L-1 on 64-bit words, the same result widths
    if (direction == 1)
        return log2n >= 13 && log2n <= 16 && data_width >= 17 && data_width + 8 <= 32 && data_width + log2n > 32 && data_width + log2n <= 40 &&
               twdl_width >= 16 && twdl_width <= 24 && format == 1 && use_fly == 1 && wide_orders_ok(direction, in_order, out_order);
    return log2n >= 13 && log2n <= 16 && data_width >= 17 && data_width + log2n - 8 <= 32 && data_width + log2n > 32 &&
           data_width + log2n <= 40 && twdl_width >= 16 && twdl_width <= 24 && format == 1 && direction == 0 && use_fly == 1 &&
           wide_orders_ok(direction, in_order, out_order);
}

const char *wide16_kernel_name(int direction, int w64)
{
    if (w64) return direction == 1 ? "k_wide64_q1+k_wide16_q2" : "k_wide64_p1+k_wide16_p2";
    return direction == 1 ? "k_wide16_q1+q2" : "k_wide16_p1+p2";
}

template <int L, bool NAT>
static hipError_t launch_wide_ln(const WideArgs &a, const W2Consts &k, const void *in, void *out, void *scratch, const int2 *tw_all,
                                 size_t nframes, hipStream_t stream, int direction)
{
    const size_t nvf = (nframes + ((size_t)1 << (16 - L)) - 1) >> (16 - L); // virtual 2^16-point frames
    const size_t units = nvf * 16;
    auto tiles16 = [&](const void *kern) { // a multiple of the 16 column tiles
        size_t g = resident_blocks(kern, 256, 2) & ~(size_t)15;
        if (g < 16) g = 16;
        return g > units ? units : g;
    };
    auto anyg = [&](const void *kern) {
        const size_t g = resident_blocks(kern, 256, 2);
        return g > units ? units : g;
    };
    const int2 *pin = static_cast<const int2 *>(in);
    i64 *pout = static_cast<i64 *>(out);
    if (a.w64 && direction == 1) {
        hipLaunchKernelGGL((k_wide64_q1<L, NAT>), dim3((unsigned)anyg(kptr(k_wide64_q1<L, NAT>))), dim3(256), 0, stream, pin, static_cast<i64 *>(scratch), tw_all, a, k, nframes);
        hipLaunchKernelGGL((k_wide16_q2<L, true, NAT>), dim3((unsigned)tiles16(kptr(k_wide16_q2<L, true, NAT>))), dim3(256), 0, stream, static_cast<const int2 *>(scratch), pout, tw_all, a, nframes);
    } else if (a.w64) {
        hipLaunchKernelGGL((k_wide64_p1<L, NAT>), dim3((unsigned)tiles16(kptr(k_wide64_p1<L, NAT>))), dim3(256), 0, stream, pin, static_cast<i64 *>(scratch), tw_all, a, nframes);
        hipLaunchKernelGGL((k_wide16_p2<L, true, NAT>), dim3((unsigned)anyg(kptr(k_wide16_p2<L, true, NAT>))), dim3(256), 0, stream, static_cast<const int2 *>(scratch), pout, tw_all, a, k, nframes);
    } else if (direction == 1) {
        hipLaunchKernelGGL((k_wide16_q1<L, NAT>), dim3((unsigned)anyg(kptr(k_wide16_q1<L, NAT>))), dim3(256), 0, stream, pin, static_cast<int2 *>(scratch), tw_all, a, k, nframes);
        hipLaunchKernelGGL((k_wide16_q2<L, false, NAT>), dim3((unsigned)tiles16(kptr(k_wide16_q2<L, false, NAT>))), dim3(256), 0, stream, static_cast<const int2 *>(scratch), pout, tw_all, a, nframes);
    } else {
        hipLaunchKernelGGL((k_wide16_p1<L, NAT>), dim3((unsigned)tiles16(kptr(k_wide16_p1<L, NAT>))), dim3(256), 0, stream, pin, static_cast<int2 *>(scratch), tw_all, a, nframes);
        hipLaunchKernelGGL((k_wide16_p2<L, false, NAT>), dim3((unsigned)anyg(kptr(k_wide16_p2<L, false, NAT>))), dim3(256), 0, stream, static_cast<const int2 *>(scratch), pout, tw_all, a, k, nframes);
    }
    return hipGetLastError();
}
template <int L>
static hipError_t launch_wide_l(const WideArgs &a, const W2Consts &k, const void *in, void *out, void *scratch, const int2 *tw_all,
                                size_t nframes, hipStream_t stream, int direction)
{
    // the natural-order instantiations carry none of the native-order code
    return a.native ? launch_wide_ln<L, true>(a, k, in, out, scratch, tw_all, nframes, stream, direction)
                    : launch_wide_ln<L, false>(a, k, in, out, scratch, tw_all, nframes, stream, direction);
}

hipError_t launch_wide16(int log2n, const WideArgs &a, const void *in, void *out, void *scratch, const int2 *tw_all,
                         const int2 *h_tw, size_t nframes, hipStream_t stream, int direction)
{
    if (nframes == 0) return hipSuccess;
    W2Consts k;
    for (int i = 0; i < 8; ++i) k.wr3[i] = h_tw[7 + i].x, k.wi3[i] = h_tw[7 + i].y;
    for (int i = 0; i < 4; ++i) k.wr2[i] = h_tw[3 + i].x, k.wi2[i] = h_tw[3 + i].y;
    switch (log2n) {
    case 13: return launch_wide_l<13>(a, k, in, out, scratch, tw_all, nframes, stream, direction);
    case 14: return launch_wide_l<14>(a, k, in, out, scratch, tw_all, nframes, stream, direction);
    case 15: return launch_wide_l<15>(a, k, in, out, scratch, tw_all, nframes, stream, direction);
    default: return launch_wide_l<16>(a, k, in, out, scratch, tw_all, nframes, stream, direction);
    }
}

#endif // INTFFT_WIDE16_TEMPLATES_ONLY
} // namespace intfft
