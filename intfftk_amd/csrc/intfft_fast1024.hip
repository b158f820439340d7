// intfft_fast1024.hip -- packed-int16 wave kernel for the headline configuration:
// int_fftNk with NFFT = 10 (N = 1024), DATA_WIDTH = 16, TWDL_WIDTH <= 16, scaled (FORMAT = 0),
// natural-order input, NATURAL or BITREV output  (src/vhdl/main/int_fft_single_path.vhd:157-268).
//
// One wave64 owns one frame: 16 VGPRs of packed (re | im << 16) int16 per lane, persistent loop
// over frames.  The ten radix-2 DIF stages (src/vhdl/fft/int_dif2_fly.vhd:144-373) are evaluated
// literally -- per-stage truncation forbids any algebraic stage fusion -- but the DATAFLOW is
// regrouped so that every butterfly is lane-local:
//
//   index bit:        9 8 7 6 | 5      | 4      | 3 2 1 0
//   phase 1 (regs)    j3..j0  | lane5  | lane4  | lane3..0     stages 9,8,7,6 in registers
//   v_permlane32_swap lane5   | j3     |                       stage 5
//   v_permlane16_swap         |        | j2                    stage 4
//   LDS transpose (5 KiB/wave, conflict-free b32 writes, b128 reads): regs = bits 3..0
//   phase 3 (regs)                                 r3..r0      stages 3,2 (wave-uniform twiddles
//                                                              in SGPRs), 1, 0 (multiplier-free)
//
// The cross-commutators (src/vhdl/delay/int_delay_line.vhd:60-104) are exactly this regrouping;
// the final bit-reversal (src/vhdl/buffers/int_bitrev_order.vhd:82-104) is folded into the LDS
// transpose so that every global store instruction writes 256 contiguous bytes.
//
// Arithmetic per general butterfly (SURVEY.md section 9.2, 9.4 "sngl" regime, w = 16):
//   A1 = A >> 1, B1 = B >> 1            v_pk_ashrrev_i16 x2   (LSB dropped BEFORE the add)
//   S = A1 + B1, D = A1 - B1            v_pk_add_u16, v_pk_sub_i16
//   re = D.re*wr - D.im*wi              v_dot2_i32_i16 with W packed as (wr, -wi)   (exact in int32)
//   im = D.re*wi + D.im*wr              v_dot2_i32_i16 with W packed as (wi,  wr)
//   Y  = { im[t+14:t-1], re[t+14:t-1] } 2 x v_bfe_i32 + v_perm_b32   (floor, wrap to 16 bits)
// In truncate mode the next stage only ever reads Y >> 1, so the multiplier emits
// { sext(im[t+14:t]), sext(re[t+14:t]) } directly and that stage skips its input shift.
//
// The multiplies are issued from inline asm (VOP3P v_dot2_i32_i16 with an inline-constant 0
// accumulator; the builtin lowers to v_dot2c + v_mov 0).  hipcc does not pad hazards inside an asm
// statement, so each statement interleaves TWO butterflies: every DOT result is read >= 3 and
// overwritten >= 4 instructions after the DOT that produced it (gfx940-class DOT->VALU hazards).
#include "intfft_internal.hpp"

#include <cstdlib>

namespace intfft {

using u32 = uint32_t;
typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s as_v2s(u32 x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ u32 as_u32(v2s x) { return __builtin_bit_cast(u32, x); }

// wave-uniform twiddles of the in-register stages 3 and 2, both packings (kernel argument -> SGPRs)
struct Fast1024Consts {
    u32 wa3[8], wb3[8]; // STAGE 3: table index r & 7
    u32 wa2[4], wb2[4]; // STAGE 2: table index r & 3
};

// S, D of one butterfly.  PRE: the inputs already hold A >> 1, B >> 1 (truncate mode only).
template <bool ROUND, bool PRE> __device__ __forceinline__ void sumdiff(u32 a, u32 b, u32 &s, u32 &d)
{
    const v2s A = as_v2s(a), B = as_v2s(b);
    if (!ROUND) { // int_dif2_fly.vhd:144-164
        const v2s A1 = PRE ? A : A >> (short)1, B1 = PRE ? B : B >> (short)1;
        s = as_u32(A1 + B1);
        d = as_u32(A1 - B1);
    } else { // :167-219  rhu2(A+B) = (A|B) - ((A^B)>>1);  rhu2(A-B) = (A>>1) - (B>>1) + (A & ~B & 1)
        s = as_u32((A | B) - ((A ^ B) >> (short)1));
        const v2s one = {1, 1};
        d = as_u32((A >> (short)1) - (B >> (short)1) + ((A & ~B) & one));
    }
}

// Two complex multiplies cmult_{16,t}(D, W) in the single-DSP regime (int_cmult_dsp48.vhd:184-225).
// off/WIDTH select the result slice of the exact 32-bit sums: (t-1, 16) = Y, (t, 15) = Y >> 1.
// Twiddles in VGPRs (lane-dependent stages).
template <int WIDTH>
__device__ __forceinline__ void cmul2_v(u32 d0, u32 wa0, u32 wb0, u32 d1, u32 wa1, u32 wb1, int off, u32 sel,
                                        u32 &y0, u32 &y1)
{
    u32 r0, i0, r1, i1;
    asm("v_dot2_i32_i16 %[r0], %[d0], %[wa0], 0\n\t"
        "v_dot2_i32_i16 %[i0], %[d0], %[wb0], 0\n\t"
        "v_dot2_i32_i16 %[r1], %[d1], %[wa1], 0\n\t"
        "v_dot2_i32_i16 %[i1], %[d1], %[wb1], 0\n\t"
        "v_bfe_i32 %[y0], %[r0], %[off], %[wd]\n\t"
        "v_bfe_i32 %[r0], %[i0], %[off], %[wd]\n\t"
        "v_bfe_i32 %[y1], %[r1], %[off], %[wd]\n\t"
        "v_bfe_i32 %[i0], %[i1], %[off], %[wd]\n\t"
        "v_perm_b32 %[y0], %[r0], %[y0], %[sel]\n\t"
        "v_perm_b32 %[y1], %[i0], %[y1], %[sel]"
        : [y0] "=&v"(y0), [y1] "=&v"(y1), [r0] "=&v"(r0), [i0] "=&v"(i0), [r1] "=&v"(r1), [i1] "=&v"(i1)
        : [d0] "v"(d0), [wa0] "v"(wa0), [wb0] "v"(wb0), [d1] "v"(d1), [wa1] "v"(wa1), [wb1] "v"(wb1),
          [off] "s"(off), [wd] "n"(WIDTH), [sel] "s"(sel));
}
// Twiddles in SGPRs (wave-uniform stages 3 and 2).
template <int WIDTH>
__device__ __forceinline__ void cmul2_s(u32 d0, u32 wa0, u32 wb0, u32 d1, u32 wa1, u32 wb1, int off, u32 sel,
                                        u32 &y0, u32 &y1)
{
    u32 r0, i0, r1, i1;
    asm("v_dot2_i32_i16 %[r0], %[d0], %[wa0], 0\n\t"
        "v_dot2_i32_i16 %[i0], %[d0], %[wb0], 0\n\t"
        "v_dot2_i32_i16 %[r1], %[d1], %[wa1], 0\n\t"
        "v_dot2_i32_i16 %[i1], %[d1], %[wb1], 0\n\t"
        "v_bfe_i32 %[y0], %[r0], %[off], %[wd]\n\t"
        "v_bfe_i32 %[r0], %[i0], %[off], %[wd]\n\t"
        "v_bfe_i32 %[y1], %[r1], %[off], %[wd]\n\t"
        "v_bfe_i32 %[i0], %[i1], %[off], %[wd]\n\t"
        "v_perm_b32 %[y0], %[r0], %[y0], %[sel]\n\t"
        "v_perm_b32 %[y1], %[i0], %[y1], %[sel]"
        : [y0] "=&v"(y0), [y1] "=&v"(y1), [r0] "=&v"(r0), [i0] "=&v"(i0), [r1] "=&v"(r1), [i1] "=&v"(i1)
        : [d0] "v"(d0), [wa0] "s"(wa0), [wb0] "s"(wb0), [d1] "v"(d1), [wa1] "s"(wa1), [wb1] "s"(wb1),
          [off] "s"(off), [wd] "n"(WIDTH), [sel] "s"(sel));
}

// Slicing parameters of a plan: truncate mode pre-shifts the multiplier outputs where the consumer
// stage is register-static (OUT_PRE), round mode never does.
struct Slice {
    int off_y;  // t - 1: Y = sum[t+14 : t-1]
    int off_y1; // t    : Y >> 1 = sext(sum[t+14 : t])
    u32 sel;    // v_perm_b32 selector {S0.b1, S0.b0, S1.b1, S1.b0}
};

// two general butterflies (a0,b0), (a1,b1): a <- S, b <- cmult(D, W)
template <bool ROUND, bool IN_PRE, bool OUT_PRE>
__device__ __forceinline__ void bfly2_v(u32 &a0, u32 &b0, u32 wa0, u32 wb0, u32 &a1, u32 &b1, u32 wa1, u32 wb1,
                                        const Slice &sl)
{
    u32 d0, d1;
    sumdiff<ROUND, IN_PRE>(a0, b0, a0, d0);
    sumdiff<ROUND, IN_PRE>(a1, b1, a1, d1);
    if (OUT_PRE) cmul2_v<15>(d0, wa0, wb0, d1, wa1, wb1, sl.off_y1, sl.sel, b0, b1);
    else cmul2_v<16>(d0, wa0, wb0, d1, wa1, wb1, sl.off_y, sl.sel, b0, b1);
}
template <bool ROUND, bool IN_PRE, bool OUT_PRE>
__device__ __forceinline__ void bfly2_s(u32 &a0, u32 &b0, u32 wa0, u32 wb0, u32 &a1, u32 &b1, u32 wa1, u32 wb1,
                                        const Slice &sl)
{
    u32 d0, d1;
    sumdiff<ROUND, IN_PRE>(a0, b0, a0, d0);
    sumdiff<ROUND, IN_PRE>(a1, b1, a1, d1);
    if (OUT_PRE) cmul2_s<15>(d0, wa0, wb0, d1, wa1, wb1, sl.off_y1, sl.sel, b0, b1);
    else cmul2_s<16>(d0, wa0, wb0, d1, wa1, wb1, sl.off_y, sl.sel, b0, b1);
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

// lane-half / row exchanges; the leading s_nop covers "VALU write -> v_permlane read" (2 wait
// states) for producers hipcc cannot see (the asm multiplies above)
__device__ __forceinline__ void swap32(u32 &a, u32 &b)
{
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16(u32 &a, u32 &b)
{
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

__device__ __forceinline__ u32 pack_wa(int2 w) { return ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16); }
__device__ __forceinline__ u32 pack_wb(int2 w) { return ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16); }

constexpr int ROW_DW = 20; // LDS row stride in dwords: 16 data + 4 pad (16-B aligned, conflict-free)

template <bool ROUND, bool OUT_BITREV, bool PIPE>
__global__ __launch_bounds__(256) void k_fft1024_i16(const u32 *in, u32 *out, const int2 *__restrict__ tw,
                                                     const Fast1024Consts c, size_t nframes, const Slice sl)
{
    __shared__ __attribute__((aligned(16))) u32 lds_all[4 * 64 * ROW_DW];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform: frame addresses stay scalar
    u32 *lds = lds_all + wv * 64 * ROW_DW;

    // ---- per-lane twiddles of the lane-dependent stages (frame invariant) --------------------
    // stage s table starts at tw + 2^s - 1; index = position mod 2^s (rom_twiddle_int.vhd:187-202)
    u32 wa9[8], wb9[8], wa8[4], wb8[4], wa7[2], wb7[2], wa6, wb6, wa5, wb5, wa4, wb4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int2 w = tw[511 + 64 * j + lane];
        wa9[j] = pack_wa(w);
        wb9[j] = pack_wb(w);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int2 w = tw[255 + 64 * j + lane];
        wa8[j] = pack_wa(w);
        wb8[j] = pack_wb(w);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int2 w = tw[127 + 64 * j + lane];
        wa7[j] = pack_wa(w);
        wb7[j] = pack_wb(w);
    }
    {
        int2 w = tw[63 + lane];
        wa6 = pack_wa(w);
        wb6 = pack_wb(w);
        w = tw[31 + (lane & 31)];
        wa5 = pack_wa(w);
        wb5 = pack_wb(w);
        w = tw[15 + (lane & 15)];
        wa4 = pack_wa(w);
        wb4 = pack_wb(w);
    }

    // ---- LDS transpose addressing -------------------------------------------------------------
    // after the two lane swaps: lane5 = n9, lane4 = n8, lane3..0 = n3..0; reg j3 = n5, j2 = n4,
    // j1 = n7, j0 = n6.  Destination row = new lane:
    //   NATURAL: row bit i = n(9-i)  (so that X index = rev4(r) * 64 + row, contiguous in row)
    //   BITREV : row = n9..n4        (memory index = n = row * 16 + r)
    const int t5 = lane >> 5, t4 = (lane >> 4) & 1;
    const int wr_lane = OUT_BITREV ? ROW_DW * (32 * t5 + 16 * t4) + (lane & 15)
                                   : ROW_DW * (t5 + 2 * t4) + (lane & 15);
    u32 *wr_base = lds + wr_lane;
    const uint4 *rd_base = reinterpret_cast<const uint4 *>(lds + ROW_DW * lane);

    // ---- one frame: registers v[] (lane = n5..0, j = n9..6) -> transformed frame f in memory ----
#ifndef INTFFT_ABLATE
#define INTFFT_ABLATE 0
#endif
    const size_t wave0_ = (size_t)blockIdx.x * 4 + wv;
    (void)wave0_;
    auto transform_store = [&](u32(&v)[16], size_t f) {
    // P = truncate mode: multiplier outputs are emitted pre-shifted (Y >> 1) whenever the stage
            // that consumes them pairs registers of one kind; after a stage with register offset h the
            // registers with (j & h) != 0 hold Y >> 1, the others hold S.
            constexpr bool P = !ROUND;

            // ---- phase 1: stages 9, 8, 7, 6 (register offsets 8, 4, 2, 1) ----
#pragma unroll
            for (int j = 0; j < 8; j += 2)
                bfly2_v<ROUND, false, P>(v[j], v[j + 8], wa9[j], wb9[j], v[j + 1], v[j + 9], wa9[j + 1], wb9[j + 1], sl);
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                bfly2_v<ROUND, false, P>(v[j], v[j + 4], wa8[j], wb8[j], v[j + 1], v[j + 5], wa8[j + 1], wb8[j + 1], sl);
                bfly2_v<ROUND, P, P>(v[8 + j], v[12 + j], wa8[j], wb8[j], v[9 + j], v[13 + j], wa8[j + 1], wb8[j + 1], sl);
            }
#pragma unroll
            for (int g = 0; g < 16; g += 8) {
                bfly2_v<ROUND, false, P>(v[g], v[g + 2], wa7[0], wb7[0], v[g + 1], v[g + 3], wa7[1], wb7[1], sl);
                bfly2_v<ROUND, P, P>(v[g + 4], v[g + 6], wa7[0], wb7[0], v[g + 5], v[g + 7], wa7[1], wb7[1], sl);
            }
#pragma unroll
            for (int g = 0; g < 16; g += 4) {
                u32 d0, d1;
                sumdiff<ROUND, false>(v[g], v[g + 1], v[g], d0);
                sumdiff<ROUND, P>(v[g + 2], v[g + 3], v[g + 2], d1);
                if (P) cmul2_v<15>(d0, wa6, wb6, d1, wa6, wb6, sl.off_y1, sl.sel, v[g + 1], v[g + 3]);
                else cmul2_v<16>(d0, wa6, wb6, d1, wa6, wb6, sl.off_y, sl.sel, v[g + 1], v[g + 3]);
            }

            // ---- lane bit 5 <-> reg bit 3, stage 5 (kind of v[j] still given by j & 1) ----
#pragma unroll
            for (int j = 0; j < 8; ++j) swap32(v[j], v[j + 8]);
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                u32 d0, d1;
                sumdiff<ROUND, false>(v[j], v[j + 8], v[j], d0);
                sumdiff<ROUND, P>(v[j + 1], v[j + 9], v[j + 1], d1);
                if (P) cmul2_v<15>(d0, wa5, wb5, d1, wa5, wb5, sl.off_y1, sl.sel, v[j + 8], v[j + 9]);
                else cmul2_v<16>(d0, wa5, wb5, d1, wa5, wb5, sl.off_y, sl.sel, v[j + 8], v[j + 9]);
            }

            // ---- lane bit 4 <-> reg bit 2, stage 4 (kind given by j & 8); outputs NOT pre-shifted:
            //      after the transpose their kind would depend on the lane ----
#pragma unroll
            for (int g = 0; g < 16; g += 8)
#pragma unroll
                for (int j = 0; j < 4; ++j) swap16(v[g + j], v[g + j + 4]);
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                bfly2_v<ROUND, false, false>(v[j], v[j + 4], wa4, wb4, v[j + 1], v[j + 5], wa4, wb4, sl);
                bfly2_v<ROUND, P, false>(v[8 + j], v[12 + j], wa4, wb4, v[9 + j], v[13 + j], wa4, wb4, sl);
            }

            // ---- LDS transpose: regs become n3..0 ----
            asm volatile("" ::: "memory"); // keep the previous frame's reads ahead of these writes
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int j0 = j & 1, j1 = (j >> 1) & 1, j2 = (j >> 2) & 1, j3 = (j >> 3) & 1;
                const int row_j = OUT_BITREV ? (8 * j1 + 4 * j0 + 2 * j3 + j2) : (4 * j1 + 8 * j0 + 16 * j3 + 32 * j2);
                wr_base[ROW_DW * row_j] = v[j];
            }
            asm volatile("" ::: "memory"); // LDS ops of one wave execute in order: no barrier needed
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 x = rd_base[q];
                v[4 * q + 0] = x.x;
                v[4 * q + 1] = x.y;
                v[4 * q + 2] = x.z;
                v[4 * q + 3] = x.w;
            }
            asm volatile("" ::: "memory");

            // ---- phase 3: stages 3, 2 (uniform twiddles), 1, 0 ----
#pragma unroll
            for (int r = 0; r < 8; r += 2)
                bfly2_s<ROUND, false, P>(v[r], v[r + 8], c.wa3[r], c.wb3[r], v[r + 1], v[r + 9], c.wa3[r + 1],
                                         c.wb3[r + 1], sl);
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                bfly2_s<ROUND, false, P>(v[r], v[r + 4], c.wa2[r], c.wb2[r], v[r + 1], v[r + 5], c.wa2[r + 1],
                                         c.wb2[r + 1], sl);
                bfly2_s<ROUND, P, P>(v[8 + r], v[12 + r], c.wa2[r], c.wb2[r], v[9 + r], v[13 + r], c.wa2[r + 1],
                                     c.wb2[r + 1], sl);
            }
#pragma unroll
            for (int g = 0; g < 16; g += 8) { // stage 1: kind given by r & 4
                bfly_triv<ROUND, false>(v[g], v[g + 2]);
                bfly_mj<ROUND, false>(v[g + 1], v[g + 3]);
                bfly_triv<ROUND, P>(v[g + 4], v[g + 6]);
                bfly_mj<ROUND, P>(v[g + 5], v[g + 7]);
            }
#pragma unroll
            for (int g = 0; g < 16; g += 2) bfly_triv<ROUND, false>(v[g], v[g + 1]);

            // ---- store ----
#if INTFFT_ABLATE & 2 // diagnostic build: only the last frames are really stored
            if (f + (size_t)gridDim.x * 4 < nframes) {
                u32 acc = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc ^= v[r];
                if (acc == 0x12345679u) out[f] = acc; // keeps the arithmetic live
                return;
            }
#endif
            if (OUT_BITREV) {
                uint4 *dst = reinterpret_cast<uint4 *>(out + f * 1024 + lane * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            } else {
                u32 *dst = out + f * 1024 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); // rev4
                    __builtin_nontemporal_store(v[r], dst + 64 * rr);
                }
            }
    };
    auto load_frame = [&](u32(&v)[16], size_t f) {
#if INTFFT_ABLATE & 1 // diagnostic build: only the first frame is really loaded
        if (f != wave0_) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = v[j] * 3u + (u32)f;
            return;
        }
#endif
        const u32 *src = in + f * 1024 + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(src + 64 * j);
    };

    // ---- persistent frame loop, software-pipelined: frame f+1 is in flight while f is computed ----
    const size_t wave0 = (size_t)blockIdx.x * 4 + wv;
    const size_t nwaves = (size_t)gridDim.x * 4;
    if (!PIPE) { // one frame at a time: fewer VGPRs, more waves per SIMD
        for (size_t f = wave0; f < nframes; f += nwaves) {
            u32 v[16];
            load_frame(v, f);
            transform_store(v, f);
        }
        return;
    }
    u32 va[16], vb[16];
    size_t f = wave0;
    if (f < nframes) load_frame(va, f);
    while (f < nframes) {
        const size_t f1 = f + nwaves;
        if (f1 < nframes) load_frame(vb, f1);
        transform_store(va, f);
        if (f1 >= nframes) break;
        f = f1 + nwaves;
        if (f < nframes) load_frame(va, f);
        transform_store(vb, f1);
    }
}

bool fast1024_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int direction,
                        int use_fly, int in_order, int out_order)
{
    (void)rndmode;
    return log2n == 10 && data_width == 16 && twdl_width >= 8 && twdl_width <= 16 && format == 0 &&
           direction == 0 && use_fly == 1 && in_order == 0 && (out_order == 0 || out_order == 1);
}

const char *fast1024_kernel_name() { return "k_fft1024_i16"; }

template <bool ROUND, bool OUT_BITREV, bool PIPE>
static hipError_t launch_p(const u32 *in, u32 *out, const int2 *tw, const Fast1024Consts &c, size_t nframes,
                           const Slice &sl, hipStream_t stream)
{
    // persistent waves: exactly the resident grid (occupancy x CUs), so no block waits for a slot
    static int per_cu = 0, cus = 0;
    if (!per_cu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fft1024_i16<ROUND, OUT_BITREV, PIPE>, 256, 0) != hipSuccess ||
            per_cu <= 0)
            per_cu = 4;
        if (const char *e = getenv("INTFFT_BLOCKS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
    }
    const size_t need = (nframes + 3) / 4;
    const size_t cap = (size_t)cus * (size_t)per_cu;
    const unsigned blocks = (unsigned)(need < cap ? need : cap);
    hipLaunchKernelGGL((k_fft1024_i16<ROUND, OUT_BITREV, PIPE>), dim3(blocks), dim3(256), 0, stream, in, out, tw, c,
                       nframes, sl);
    return hipGetLastError();
}

template <bool ROUND, bool OUT_BITREV>
static hipError_t launch_t(const u32 *in, u32 *out, const int2 *tw, const Fast1024Consts &c, size_t nframes,
                           const Slice &sl, hipStream_t stream)
{
    static int pipe = -1; // software-pipelined (2 frames in flight per wave) unless INTFFT_FAST_PIPE=0
    if (pipe < 0) {
        const char *e = getenv("INTFFT_FAST_PIPE");
        pipe = e ? atoi(e) != 0 : 1;
    }
    return pipe ? launch_p<ROUND, OUT_BITREV, true>(in, out, tw, c, nframes, sl, stream)
                : launch_p<ROUND, OUT_BITREV, false>(in, out, tw, c, nframes, sl, stream);
}

hipError_t launch_fast1024(const Fast1024Args &a, const void *in, void *out, const int2 *tw_all,
                           const int2 *h_tw, size_t nframes, hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    Fast1024Consts c;
    for (int k = 0; k < 8; ++k) { // stage 3 table at offset 2^3 - 1
        const int2 w = h_tw[7 + k];
        c.wa3[k] = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        c.wb3[k] = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    }
    for (int k = 0; k < 4; ++k) { // stage 2 table at offset 3
        const int2 w = h_tw[3 + k];
        c.wa2[k] = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        c.wb2[k] = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    }
    const Slice sl{a.twd - 1, a.twd, 0x05040100u};
    const u32 *pin = static_cast<const u32 *>(in);
    u32 *pout = static_cast<u32 *>(out);
    if (a.rnd == RND_ROUND)
        return a.out_bitrev ? launch_t<true, true>(pin, pout, tw_all, c, nframes, sl, stream)
                            : launch_t<true, false>(pin, pout, tw_all, c, nframes, sl, stream);
    return a.out_bitrev ? launch_t<false, true>(pin, pout, tw_all, c, nframes, sl, stream)
                        : launch_t<false, false>(pin, pout, tw_all, c, nframes, sl, stream);
}

} // namespace intfft
