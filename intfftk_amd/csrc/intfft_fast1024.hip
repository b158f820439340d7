// intfft_fast1024.hip -- packed-int16 wave kernel for the headline configuration:
// int_fftNk with NFFT = 10 (N = 1024), DATA_WIDTH = 16, TWDL_WIDTH <= 16, scaled (FORMAT = 0),
// natural-order input, NATURAL or BITREV output  (src/vhdl/main/int_fft_single_path.vhd:157-268),
// and, as template instances L = 6..9, for 64 <= N < 1024 in natural order (2^(10-L) frames per wave; see lane_bit<L>()
// in intfft_internal.hpp; the reference testbench ships NFFT = 7).  The description below is for N = 1024.
//
// One wave64 owns one frame: 16 VGPRs of packed (re | im << 16) int16 per lane, persistent loop
// over frames, no barrier.  The ten radix-2 DIF stages (src/vhdl/fft/int_dif2_fly.vhd:144-373) are
// evaluated literally -- per-stage truncation forbids any algebraic stage fusion -- but the
// DATAFLOW is regrouped so that every butterfly is lane-local:
//
//   index bit:        9 8 7 6 | 5      | 4      | 3 2 1 0
//   phase 1 (regs)    j3..j0  | lane5  | lane4  | lane3..0     stages 9,8,7,6 in registers
//   v_permlane32_swap lane5   | j3     |                       stage 5
//   v_permlane16_swap         |        | j2                    stage 4
//   LDS transpose (5 KiB/wave, b32 writes, conflict-free b128 reads): regs = bits 3..0
//   phase 3 (regs)                                 r3..r0      stages 3,2 (wave-uniform twiddles
//                                                              in SGPRs), 1, 0 (multiplier-free)
//
// The cross-commutators (src/vhdl/delay/int_delay_line.vhd:60-104) are exactly this regrouping;
// the final bit-reversal (src/vhdl/buffers/int_bitrev_order.vhd:82-104) is folded into the LDS
// transpose so that every global store instruction writes 256 contiguous bytes.
//
// Arithmetic per general butterfly (SURVEY.md section 9.2, 9.4 "sngl" regime, w = 16):
//   A1 = A >> 1, B1 = B >> 1            v_pk_ashrrev_i16 x<=2 (LSB dropped BEFORE the add)
//   S = A1 + B1, D = A1 - B1            v_pk_add_u16, v_pk_sub_i16
//   re = D.re*wr - D.im*wi              VOP3P v_dot2_i32_i16, W packed as (wr, -wi)  (exact in int32)
//   im = D.re*wi + D.im*wr              VOP3P v_dot2_i32_i16, W packed as (wi,  wr)
//   Y  = { im[t+14:t-1], re[t+14:t-1] } floor + wrap to 16 bits
// Truncate mode only ever reads Y >> 1 downstream, so the multiplier emits
//   Y >> 1 = { sext(im[t+14:t]), sext(re[t+14:t]) }  directly (2 x v_bfe_i32 + v_perm_b32),
// and when a frame's samples all carry one guard bit (|re|, |im| <= 2^14: proven below to exclude
// any 31-bit overflow of re/im at every stage) and t = 16, Y >> 1 is simply the high halves of
// re and im: ONE v_perm_b32 ("fast extraction").  Every other frame takes the exact extraction.
//
// VALU issue is the co-bottleneck of this kernel (tools/valubench.hip: every VOP3/VOP3P op used
// here issues at ~4.3 cycles per wave-instruction on gfx950), hence the attention to op count.
//
// Inline asm: the builtin for v_dot2 lowers to v_dot2c + v_mov 0 (+2 ops per butterfly), so the
// multiplies are issued from asm.  hipcc pads no hazards inside an asm statement; each statement
// therefore interleaves 2 (exact) or 4 (fast) butterflies so that every DOT result is read >= 3
// and overwritten >= 4 instructions after the DOT that produced it (gfx940-class DOT -> VALU
// hazards), and the asm v_permlane*_swap carry their own s_nop 1.
#include "intfft_pk16.hpp"

#include <cstdlib>
#include <type_traits>

namespace intfft {

using u32 = uint32_t;

// wave-uniform twiddles of the in-register stages 3 and 2, both packings (kernel argument -> SGPRs)
struct Fast1024Consts {
    u32 wa3[8], wb3[8]; // STAGE 3: table index r & 7
    u32 wa2[4], wb2[4]; // STAGE 2: table index r & 3
};

// per-lane twiddles of the lane-dependent stages.  Truncate mode keeps only the first half of the
// stage 9/8/7 tables: the second half is the quarter turn W' = (wi, -wr) of the first
// (rom_twiddle_int.vhd:177-183), i.e. Wa' = Wb and Wb' = -Wa, evaluated as dot(D, Wb), dot(-D, Wa).
// (-D is exact there: |D| <= 2^15 - 1.  Round mode can produce D = -2^15, so it keeps full tables.)
template <bool ROUND> struct Twiddles {
    static constexpr int NQ = ROUND ? 1 : 2;
    u32 wa9[8 / NQ], wb9[8 / NQ], wa8[4 / NQ], wb8[4 / NQ], wa7[2 / NQ], wb7[2 / NQ];
    u32 wa6[1], wb6[1], wa5[1], wb5[1], wa4[1], wb4[1];
};

constexpr int ROW_DW = 20; // LDS row stride in dwords: 16 data + 4 pad (16-B aligned, conflict-free)

// ---- one frame: v[] (lane = n5..0, j = n9..6) -> transformed, stored as frame f -------------------
// ROUND: 0 truncate, 1 round, 2 round on narrow data (the w-bit wraps of intfft_pk16.hpp)
// phase 1: stages 9, 8, 7, 6 in registers
template <int L, int ROUND, int FASTX>
__device__ __forceinline__ void transform_phase1(u32 (&v)[16], const Twiddles<(ROUND != 0)> &tw, const Slice &sl)
{
    static_assert(L >= 6 && L <= 10, "wave kernel: 64 <= N <= 1024");
    // P (truncate mode): multiplier outputs are emitted pre-shifted (Y >> 1); after a stage with
    // register offset h the registers with (j & h) != 0 hold Y >> 1, the others hold S.
    constexpr bool P = !ROUND;
    constexpr bool Q = !ROUND;
    constexpr int M0 = 0;                                                        // PREMASKs
    constexpr int MH = (P && L >= 10) ? 0xC : 0;                                 // stage 8 after stage 9
    constexpr int MODD7 = (P && L >= 9) ? 0xA : 0, MODD6 = (P && L >= 8) ? 0xA : 0;

    // ---- phase 1: stages 9, 8, 7, 6 (register offsets 8, 4, 2, 1) ----
    if constexpr (Q) {
        if constexpr (L >= 10) {
            group4<ROUND, FASTX, false, P, false, M0>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], tw.wa9, tw.wb9, sl);
            group4<ROUND, FASTX, Q, P, false, M0>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], tw.wa9, tw.wb9, sl);
        }
        // stage 8: (j, j+4); kind of the inputs = j & 8
        if constexpr (L >= 9) {
            const u32 wa[4] = {tw.wa8[0], tw.wa8[1], tw.wa8[0], tw.wa8[1]}, wb[4] = {tw.wb8[0], tw.wb8[1], tw.wb8[0], tw.wb8[1]};
            group4<ROUND, FASTX, false, P, false, MH>(v[0], v[4], v[1], v[5], v[8], v[12], v[9], v[13], wa, wb, sl);
            group4<ROUND, FASTX, Q, P, false, MH>(v[2], v[6], v[3], v[7], v[10], v[14], v[11], v[15], wa, wb, sl);
        }
        // stage 7: (j, j+2); kind = j & 4; pairs with j odd use the quarter turn of wa7[0]
        if constexpr (L >= 8) {
            const u32 wa[4] = {tw.wa7[0], tw.wa7[0], tw.wa7[0], tw.wa7[0]}, wb[4] = {tw.wb7[0], tw.wb7[0], tw.wb7[0], tw.wb7[0]};
            group4<ROUND, FASTX, false, P, false, MODD7>(v[0], v[2], v[4], v[6], v[8], v[10], v[12], v[14], wa, wb, sl);
            group4<ROUND, FASTX, Q, P, false, MODD7>(v[1], v[3], v[5], v[7], v[9], v[11], v[13], v[15], wa, wb, sl);
        }
    } else {
        const u32 wa9a[4] = {tw.wa9[0], tw.wa9[1], tw.wa9[2], tw.wa9[3]}, wb9a[4] = {tw.wb9[0], tw.wb9[1], tw.wb9[2], tw.wb9[3]};
        const u32 wa9b[4] = {tw.wa9[4 % (8 / Twiddles<(ROUND != 0)>::NQ)], tw.wa9[5 % (8 / Twiddles<(ROUND != 0)>::NQ)],
                             tw.wa9[6 % (8 / Twiddles<(ROUND != 0)>::NQ)], tw.wa9[7 % (8 / Twiddles<(ROUND != 0)>::NQ)]};
        const u32 wb9b[4] = {tw.wb9[4 % (8 / Twiddles<(ROUND != 0)>::NQ)], tw.wb9[5 % (8 / Twiddles<(ROUND != 0)>::NQ)],
                             tw.wb9[6 % (8 / Twiddles<(ROUND != 0)>::NQ)], tw.wb9[7 % (8 / Twiddles<(ROUND != 0)>::NQ)]};
        if constexpr (L >= 10) {
            group4<ROUND, false, false, P, false, M0>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa9a, wb9a, sl);
            group4<ROUND, false, false, P, false, M0>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa9b, wb9b, sl);
        }
        constexpr int N8 = 4 / Twiddles<(ROUND != 0)>::NQ, N7 = 2 / Twiddles<(ROUND != 0)>::NQ;
        const u32 wa8a[4] = {tw.wa8[0], tw.wa8[1 % N8], tw.wa8[0], tw.wa8[1 % N8]}, wb8a[4] = {tw.wb8[0], tw.wb8[1 % N8], tw.wb8[0], tw.wb8[1 % N8]};
        const u32 wa8b[4] = {tw.wa8[2 % N8], tw.wa8[3 % N8], tw.wa8[2 % N8], tw.wa8[3 % N8]}, wb8b[4] = {tw.wb8[2 % N8], tw.wb8[3 % N8], tw.wb8[2 % N8], tw.wb8[3 % N8]};
        if constexpr (L >= 9) {
            group4<ROUND, false, false, P, false, MH>(v[0], v[4], v[1], v[5], v[8], v[12], v[9], v[13], wa8a, wb8a, sl);
            group4<ROUND, false, false, P, false, MH>(v[2], v[6], v[3], v[7], v[10], v[14], v[11], v[15], wa8b, wb8b, sl);
        }
        const u32 wa7a[4] = {tw.wa7[0], tw.wa7[0], tw.wa7[0], tw.wa7[0]}, wb7a[4] = {tw.wb7[0], tw.wb7[0], tw.wb7[0], tw.wb7[0]};
        const u32 wa7b[4] = {tw.wa7[1 % N7], tw.wa7[1 % N7], tw.wa7[1 % N7], tw.wa7[1 % N7]}, wb7b[4] = {tw.wb7[1 % N7], tw.wb7[1 % N7], tw.wb7[1 % N7], tw.wb7[1 % N7]};
        if constexpr (L >= 8) {
            group4<ROUND, false, false, P, false, MODD7>(v[0], v[2], v[4], v[6], v[8], v[10], v[12], v[14], wa7a, wb7a, sl);
            group4<ROUND, false, false, P, false, MODD7>(v[1], v[3], v[5], v[7], v[9], v[11], v[13], v[15], wa7b, wb7b, sl);
        }
    }
    if constexpr (L >= 7) { // stage 6: (j, j+1), j even; kind = j & 2; one twiddle per lane
        const u32 wa[4] = {tw.wa6[0], tw.wa6[0], tw.wa6[0], tw.wa6[0]}, wb[4] = {tw.wb6[0], tw.wb6[0], tw.wb6[0], tw.wb6[0]};
        group4<ROUND, FASTX, false, P, false, MODD6>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], wa, wb, sl);
        group4<ROUND, FASTX, false, P, false, MODD6>(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], wa, wb, sl);
    }

}

// the rest: lane swaps with stages 5 and 4, the LDS transpose, stages 3..0, the store
template <int L, int ROUND, int OUT_BITREV, int FASTX>
__device__ __forceinline__ void transform_tail(u32 (&v)[16], u32 *out, size_t f, int lane, const Twiddles<(ROUND != 0)> &tw,
                                               const Fast1024Consts &c, const Slice &sl, u32 *wr_base,
                                               const uint4 *rd_base, v2s sh3, int lane_off, bool st_ok, size_t nframes_user)
{
    constexpr bool out_lanes = OUT_BITREV == 2; // the BITREV instantiation with the serial-stream store map
    static_assert(L >= 7 || !OUT_BITREV, "native orders need N >= 128");
    constexpr bool P = !ROUND;
    constexpr int M0 = 0, MA = P ? 0xF : 0;
    constexpr int MODD5 = (P && L >= 7) ? 0xA : 0;
    // ---- lane bit 5 <-> reg bit 3, stage 5: (j, j+8); kind = j & 1 ----
    swap_guard(v);
#pragma unroll
    for (int j = 0; j < 8; ++j) swap32(v[j], v[j + 8]);
    {
        const u32 wa[4] = {tw.wa5[0], tw.wa5[0], tw.wa5[0], tw.wa5[0]}, wb[4] = {tw.wb5[0], tw.wb5[0], tw.wb5[0], tw.wb5[0]};
        group4<ROUND, FASTX, false, P, false, MODD5>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa, wb, sl);
        group4<ROUND, FASTX, false, P, false, MODD5>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa, wb, sl);
    }

    // ---- lane bit 4 <-> reg bit 2, stage 4: (j, j+4); kind = j & 8 ----
    swap_guard(v);
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int j = 0; j < 4; ++j) swap16(v[g + j], v[g + j + 4]);
    {
        const u32 wa[4] = {tw.wa4[0], tw.wa4[0], tw.wa4[0], tw.wa4[0]}, wb[4] = {tw.wb4[0], tw.wb4[0], tw.wb4[0], tw.wb4[0]};
        group4<ROUND, FASTX, false, P, false, M0>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], wa, wb, sl);
        group4<ROUND, FASTX, false, P, false, MA>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], wa, wb, sl);
    }

    // ---- LDS transpose: regs become n3..0.  The kind of a value (S or Y >> 1) is now j & 4, and
    //      reg bit 2 holds n4, which becomes a LANE bit: stage 3 shifts by a per-lane amount ----
    wave_lds_fence(); // keep the previous frame's reads ahead of these writes
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int j0 = j & 1, j1 = (j >> 1) & 1, j2 = (j >> 2) & 1, j3 = (j >> 3) & 1;
        // at this point reg j3 = a5, j2 = a4, j1 = a7, j0 = a6
        const int row_j = OUT_BITREV ? (8 * j1 + 4 * j0 + 2 * j3 + j2)
                                     : ((j1 << lane_bit<L>(7)) + (j0 << lane_bit<L>(6)) + (j3 << lane_bit<L>(5)) +
                                        (j2 << lane_bit<L>(4)));
        wr_base[ROW_DW * row_j] = v[j];
    }
    wave_lds_fence(); // LDS ops of one wave execute in order: no barrier needed
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 x = rd_base[q];
        v[4 * q + 0] = x.x;
        v[4 * q + 1] = x.y;
        v[4 * q + 2] = x.z;
        v[4 * q + 3] = x.w;
    }
    wave_lds_fence();

    // ---- phase 3: stages 3, 2 (uniform twiddles), 1, 0 ----
    {
        const u32 wa0[4] = {c.wa3[0], c.wa3[1], c.wa3[2], c.wa3[3]}, wb0[4] = {c.wb3[0], c.wb3[1], c.wb3[2], c.wb3[3]};
        const u32 wa1[4] = {c.wa3[4], c.wa3[5], c.wa3[6], c.wa3[7]}, wb1[4] = {c.wb3[4], c.wb3[5], c.wb3[6], c.wb3[7]};
        group4<ROUND, FASTX, false, P, true, M0, P>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa0, wb0, sl, sh3);
        group4<ROUND, FASTX, false, P, true, M0, P>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa1, wb1, sl, sh3);
        // stage 2: (r, r+4); kind = r & 8
        group4<ROUND, FASTX, false, P, true, M0>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], c.wa2, c.wb2, sl);
        group4<ROUND, FASTX, false, P, true, MA>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], c.wa2, c.wb2, sl);
    }
    if constexpr (ROUND) {
        round_stages10<16, ROUND == 2>(v, sl);
    } else {
#pragma unroll
        for (int g = 0; g < 16; g += 8) { // stage 1: kind = r & 4
            bfly_triv<(ROUND != 0), false>(v[g], v[g + 2]);
            bfly_mj<(ROUND != 0), false>(v[g + 1], v[g + 3]);
            bfly_triv<(ROUND != 0), P>(v[g + 4], v[g + 6]);
            bfly_mj<(ROUND != 0), P>(v[g + 5], v[g + 7]);
        }
#pragma unroll
        for (int g = 0; g < 16; g += 2) bfly_triv<(ROUND != 0), false>(v[g], v[g + 1]);
    }

    // ---- store ----
    if (OUT_BITREV) {
        // memory index = n.  Two lane swaps turn (regs n3..0, lane n9..4) into (regs n9 n8 n1 n0,
        // lane n3 n2 n7..4): every lane then owns 4 consecutive n and a store instruction covers 1 KiB
        swap_guard(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) swap32(v[r], v[r + 8]);
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int r = 0; r < 4; ++r) swap16(v[g + r], v[g + r + 4]);
        typedef u32 v4u __attribute__((ext_vector_type(4)));
        v4u *dst = reinterpret_cast<v4u *>(out + f * 1024) + (((lane & 15) << 2) | (lane >> 4));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // short frames: the vector's frame within the chunk = bits a9..aL of (q, lane & 15); st_ok = "chunk is full"
            if (L < 10 && !st_ok && f * (size_t)(1 << (10 - L)) + (size_t)(((q << 8) | ((lane & 15) << 4)) >> L) >= nframes_user)
                continue;
            const v4u x = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            if constexpr (out_lanes) {
                // BITREV_LANES (outbuf_half_path.vhd:160-172: the serial stream [lane 0 frame ; lane 1 frame] that int_bitrev_order reads):
                // core position n of a frame goes to memory index (n & 1) * N/2 + (n >> 1).  The vector holds four consecutive n:
                // the even ones are two consecutive words of the first half, the odd ones of the second -- two 8-byte stores,
                // 512 contiguous bytes per wave instruction each.
                typedef u32 v2u __attribute__((ext_vector_type(2)));
                const int n0 = 4 * (64 * q + (((lane & 15) << 2) | (lane >> 4))), nl = n0 & ((1 << L) - 1);
                u32 *const d = out + f * 1024 + (n0 - nl) + (nl >> 1);
                const v2u ev = {x.x, x.z}, od = {x.y, x.w};
                __builtin_nontemporal_store(ev, reinterpret_cast<v2u *>(d));
                __builtin_nontemporal_store(od, reinterpret_cast<v2u *>(d + (1 << (L - 1))));
                continue;
            }
            __builtin_nontemporal_store(x, dst + 64 * q);
        }
    } else if constexpr (L < 10) {
        // regs a3..a0, lane bits 5, 4 = a(L-1), a(L-2): after the swaps reg bit 3 = a(L-1), bit 2 = a(L-2) and the
        // four registers {q, q+8, q+4, q+12} are four consecutive outputs (q = a1 a0)
        swap_guard(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) swap32(v[r], v[r + 8]);
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int r = 0; r < 4; ++r) swap16(v[g + r], v[g + r + 4]);
        typedef u32 v4u __attribute__((ext_vector_type(4)));
        if constexpr (L == 6) {
            // N = 64: a lane's four vectors lie in four quarters of its 256-byte frame and a store instruction would write sixteen 64-byte
            // runs.  The chunk goes through the wave's (now idle) LDS tile in memory order instead and leaves as 1 KiB per instruction.
            u32 *const tile = const_cast<u32 *>(reinterpret_cast<const u32 *>(rd_base)) - ROW_DW * lane; // the wave's 64 x ROW_DW dwords
            wave_lds_fence();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4u x = {v[q], v[q + 8], v[q + 4], v[q + 12]};
                *reinterpret_cast<v4u *>(tile + lane_off + (q & 1) * out_weight<L>(0) + (q >> 1) * out_weight<L>(1)) = x;
            }
            wave_lds_fence();
            v4u *dst4 = reinterpret_cast<v4u *>(out + f * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 64 * i + lane; // 16-byte piece of the chunk: frame (4 e) >> 6 within it
                if (f * (size_t)(1 << (10 - L)) + (size_t)(e >> (L - 2)) >= nframes_user) continue;
                __builtin_nontemporal_store(*reinterpret_cast<const v4u *>(tile + 4 * e), dst4 + e);
            }
            wave_lds_fence(); // the next frame's transposition writes this tile again
        } else {
            u32 *dst = out + f * 1024 + lane_off;
            if (st_ok) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4u x = {v[q], v[q + 8], v[q + 4], v[q + 12]};
                    __builtin_nontemporal_store(x, reinterpret_cast<v4u *>(dst + (q & 1) * out_weight<L>(0) + (q >> 1) * out_weight<L>(1)));
                }
            }
        }
    } else {
        u32 *dst = out + f * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); // rev4
            __builtin_nontemporal_store(v[r], dst + 64 * rr);
        }
    }
}

template <int L, int ROUND, int OUT_BITREV, int FASTX>
__device__ __forceinline__ void transform_store(u32 (&v)[16], u32 *out, size_t f, int lane, const Twiddles<(ROUND != 0)> &tw,
                                                const Fast1024Consts &c, const Slice &sl, u32 *wr_base,
                                                const uint4 *rd_base, v2s sh3, int lane_off, bool st_ok, size_t nframes_user)
{
    transform_phase1<L, ROUND, FASTX>(v, tw, sl);
    transform_tail<L, ROUND, OUT_BITREV, FASTX>(v, out, f, lane, tw, c, sl, wr_base, rd_base, sh3, lane_off, st_ok, nframes_user);
}

// Magnitude votes of the 16-bit fast path (N >= 256).  Fast extraction needs every 32-bit dot-product sum inside [-2^30, 2^30), i.e.
// |D| |W| < 2^30 with |W| <= 32767.71: |D| <= 32752 suffices.  Through a scaled-truncate stage the complex magnitude bound M of a
// frame's values grows by at most 1.42 (see frame_has_guard_bit in intfft_pk16.hpp), and |D| <= M + 0.71.  So a frame whose values
// all satisfy |re|, |im| < T = 23100 (M <= 32669) is safe for ten -- and for twenty-four -- further stages: 32669 + 1.42 * 24 + 0.71
// = 32704.  The power-of-two test of the other kernels (T = 2^14) is this one with a bit trick; T = 23100 costs the same two
// operations per register (v_pk_add_u16 + v_pk_max_u16: biased values compared unsigned) and admits frames up to 70 % of full scale.
// Behind phase 1 the odd registers hold Y >> 1 (|Y >> 1| < T / 2 <=> |Y| < T); full-scale random input passes that second vote
// with probability 0.998 (four scaled stages average 16 inputs per value: T = 4.9 sigma), T = 2^14 would pass 33 % of such frames.
typedef unsigned short v2us __attribute__((ext_vector_type(2)));
constexpr unsigned GUARD_T = 23100;
__device__ __forceinline__ bool frame_within_T(const u32 (&v)[16])
{
    const v2us bias = {(unsigned short)GUARD_T, (unsigned short)GUARD_T};
    v2us acc = {0, 0};
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_elementwise_max(acc, __builtin_bit_cast(v2us, v[j]) + bias);
    return __builtin_amdgcn_ballot_w64(acc.x >= 2 * GUARD_T || acc.y >= 2 * GUARD_T) == 0;
}
__device__ __forceinline__ bool frame_within_T_after_phase1(const u32 (&v)[16])
{
    const v2us b0 = {(unsigned short)GUARD_T, (unsigned short)GUARD_T}, b1 = {(unsigned short)(GUARD_T / 2), (unsigned short)(GUARD_T / 2)};
    v2us a0 = {0, 0}, a1 = {0, 0};
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        a0 = __builtin_elementwise_max(a0, __builtin_bit_cast(v2us, v[j]) + b0);         // S-type
        a1 = __builtin_elementwise_max(a1, __builtin_bit_cast(v2us, v[j + 1]) + b1);     // Y >> 1-type
    }
    return __builtin_amdgcn_ballot_w64(a0.x >= 2 * GUARD_T || a0.y >= 2 * GUARD_T || a1.x >= GUARD_T || a1.y >= GUARD_T) == 0;
}

// ROUND: 0 truncate, 1 round, 2 round on narrow data (its own instantiation: the w-bit wraps of intfft_pk16.hpp)
template <int L, int ROUND, int OUT_BITREV, bool PIPE, bool FAST_OK>
#ifdef INTFFT_WPE_F /* A/B: tools/build_variant.sh fwpe5 intfft_fast1024.hip -DINTFFT_WPE_F=5 */
#define INTFFT_WPE_F_ATTR __attribute__((amdgpu_waves_per_eu(INTFFT_WPE_F)))
#else
#define INTFFT_WPE_F_ATTR
#endif
__global__ __launch_bounds__(256) INTFFT_WPE_F_ATTR void k_fft1024_i16(const u32 *in, u32 *out, const int2 *__restrict__ twt,
                                                     const Fast1024Consts c, size_t nframes_user, const Slice sl,
                                                     int io_flags)
{
    const int in_halves = io_flags & 1;                    // HALVES beats in
    constexpr bool out_lanes = OUT_BITREV == 2;            // BITREV_LANES instead of BITREV out: its own instantiations (OUT_BITREV = 2)
    constexpr int FP = 1 << (10 - L);                         // frames per 1024-sample chunk
    const size_t nframes = (nframes_user + FP - 1) / FP;      // chunks ("frames" of the wave loop below)
    __shared__ __attribute__((aligned(16))) u32 lds_all[4 * 64 * ROW_DW];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform: frame addresses stay scalar
    u32 *lds = lds_all + wv * 64 * ROW_DW;

    // ---- per-lane twiddles of the lane-dependent stages (frame invariant) --------------------
    // stage s table starts at twt + 2^s - 1; index = position mod 2^s (rom_twiddle_int.vhd:187-202)
    Twiddles<(ROUND != 0)> tw;
    constexpr int NQ = Twiddles<(ROUND != 0)>::NQ;
    if constexpr (L >= 10) {
#pragma unroll
        for (int j = 0; j < 8 / NQ; ++j) {
            const int2 w = twt[511 + 64 * j + lane];
            tw.wa9[j] = pack_wa(w);
            tw.wb9[j] = pack_wb(w);
        }
    }
    if constexpr (L >= 9) {
#pragma unroll
        for (int j = 0; j < 4 / NQ; ++j) {
            const int2 w = twt[255 + 64 * j + lane];
            tw.wa8[j] = pack_wa(w);
            tw.wb8[j] = pack_wb(w);
        }
    }
    if constexpr (L >= 8) {
#pragma unroll
        for (int j = 0; j < 2 / NQ; ++j) {
            const int2 w = twt[127 + 64 * j + lane];
            tw.wa7[j] = pack_wa(w);
            tw.wb7[j] = pack_wb(w);
        }
    }
    {
        int2 w;
        if constexpr (L >= 7) {
            w = twt[63 + lane];
            tw.wa6[0] = pack_wa(w);
            tw.wb6[0] = pack_wb(w);
        }
        w = twt[31 + (lane & 31)];
        tw.wa5[0] = pack_wa(w);
        tw.wb5[0] = pack_wb(w);
        w = twt[15 + (lane & 15)];
        tw.wa4[0] = pack_wa(w);
        tw.wb4[0] = pack_wb(w);
    }

    // ---- LDS transpose addressing -------------------------------------------------------------
    // after the two lane swaps: lane5 = n9, lane4 = n8, lane3..0 = n3..0; reg j3 = n5, j2 = n4,
    // j1 = n7, j0 = n6.  Destination row = new lane:
    //   NATURAL: row bit i = n(9-i)  (so that X index = rev4(r) * 64 + row, contiguous in row)
    //   BITREV : row = n9..n4        (memory index = n = row * 16 + r)
    const int t5 = lane >> 5, t4 = (lane >> 4) & 1;
    const int wr_lane = OUT_BITREV ? ROW_DW * (32 * t5 + 16 * t4) + (lane & 15)
                                   : ROW_DW * ((t5 << lane_bit<L>(9)) + (t4 << lane_bit<L>(8))) + (lane & 15);
    u32 *wr_base = lds + wr_lane;
    const uint4 *rd_base = reinterpret_cast<const uint4 *>(lds + ROW_DW * lane);
    // stage 3 per-lane shift: lanes whose n4 = 1 hold Y >> 1 already (n4 = row bit 5 / bit 0)
    const int n4 = OUT_BITREV ? (lane & 1) : ((lane >> lane_bit<L>(4)) & 1);
    const v2s sh3 = {(short)(1 - n4), (short)(1 - n4)};
    // N < 1024: output offset (dwords) of this lane's dwordx4 stores within the chunk and its frame within
    // the chunk; after the output swaps lane bit 5 = a3, lane bit 4 = a2, the rest as lane_bit<L>()
    int lane_off = 0, lane_frame = 0;
    if constexpr (L < 10) {
        lane_off = ((lane >> 5) & 1) * out_weight<L>(3) + ((lane >> 4) & 1) * out_weight<L>(2);
#pragma unroll
        for (int k = 4; k < 10; ++k) {
            if (k == L - 1 || k == L - 2) continue;
            const int bit = (lane >> lane_bit<L>(k)) & 1;
            lane_off += bit * out_weight<L>(k);
            if (k >= L) lane_frame += bit << (k - L);
        }
    }

    // Vote prediction (round 6): a frame that fails the input vote sends the wave's next VOTE_SKIP frames straight to the exact phase 1
    // without voting (always correct -- the exact extraction needs no precondition -- and they still take the second vote behind stage 6);
    // a stream of full-scale frames then pays one vote per frame instead of two (each is 32 packed operations + a ballot, ~4 % of the
    // frame), and a stream that turns quiet again is back on the fast phase 1 within VOTE_SKIP frames.  Wave-uniform (from a ballot).
#ifndef INTFFT_VOTE_SKIP
#define INTFFT_VOTE_SKIP 7 // (0: every frame takes the input vote -- the behaviour up to round 5; tools/build_variant.sh A/B)
#endif
    constexpr int VOTE_SKIP = INTFFT_VOTE_SKIP;
    int vote_skip = 0;
    auto run = [&](u32(&v)[16], size_t f) {
        // partial last chunk: natural order -> per-lane predicate; BITREV order -> "chunk is full" + the frame count
        const bool st_ok = L == 10 || (OUT_BITREV ? (f + 1) * FP <= nframes_user : f * FP + (size_t)lane_frame < nframes_user);
        if constexpr (FAST_OK && L >= 8) {
            // one instance of each phase body: fast or exact phase 1, then -- for a frame that failed the input vote -- a second vote
            // on the values behind stage 6, then the fast or exact tail
            if (sl.wd == 16) {
                bool fast = false;
                if (vote_skip > 0) --vote_skip;
                else if (!(fast = frame_within_T(v))) vote_skip = VOTE_SKIP;
                if (fast) transform_phase1<L, ROUND, 1>(v, tw, sl);
                else {
                    transform_phase1<L, ROUND, 2>(v, tw, sl); // the t = 16 exact form (mul2x_t16)
                    fast = frame_within_T_after_phase1(v);
                }
                if (fast) transform_tail<L, ROUND, OUT_BITREV, 1>(v, out, f, lane, tw, c, sl, wr_base, rd_base, sh3, lane_off, st_ok, nframes_user);
                else transform_tail<L, ROUND, OUT_BITREV, 2>(v, out, f, lane, tw, c, sl, wr_base, rd_base, sh3, lane_off, st_ok, nframes_user);
                return;
            }
        }
        if (FAST_OK && frame_has_guard_bit(v, sl.gbias, sl.gmask)) {
            transform_store<L, ROUND, OUT_BITREV, FAST_OK>(v, out, f, lane, tw, c, sl, wr_base, rd_base, sh3, lane_off, st_ok, nframes_user);
            return;
        }
        // exact path.  DATA_WIDTH < 16: containers wrapped to w bits, w-bit exact extraction (and w-bit rhu2 wraps in round mode).
        // A FAST_OK kernel is launched for 16-bit twiddles only: its 16-bit exact path is the t = 16 form (mul2x_t16).
        if (sl.wd != 16) wrap_inputs(v, sl.wd);
        if (FAST_OK && sl.wd == 16)
            transform_store<L, ROUND, OUT_BITREV, FAST_OK ? 2 : 0>(v, out, f, lane, tw, c, sl, wr_base, rd_base, sh3, lane_off, st_ok, nframes_user);
        else
            transform_store<L, ROUND, OUT_BITREV, 0>(v, out, f, lane, tw, c, sl, wr_base, rd_base, sh3, lane_off, st_ok, nframes_user);
    };
    auto load_frame = [&](u32(&v)[16], size_t f) {
        const bool partial = L < 10 && (f + 1) * FP > nframes_user; // last chunk: samples of absent frames read as 0
        if (in_halves) {
            // HALVES: beat i of a frame holds (x[i], x[i + N/2]).  Beat q = 64 jj + lane of the chunk belongs to frame
            // q >> (L-1); its two samples are chunk positions a and a + N/2, i.e. lane `lane` of the registers
            // j0 = (frame << (L-6)) | (i >> 6) and j0 | 2^(L-7)   (N = 1024: (jj, jj + 8))
            if constexpr (L >= 7) {
                typedef u32 v2u __attribute__((ext_vector_type(2)));
                const v2u *src2 = reinterpret_cast<const v2u *>(in + f * 1024) + lane;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    constexpr int HB = 1 << (L - 7);
                    const int j0 = ((jj >> (L - 7)) << (L - 6)) | (jj & (HB - 1));
                    v2u w = {0u, 0u};
                    if (!partial || f * FP + (size_t)(jj >> (L - 7)) < nframes_user) w = INTFFT_LD(src2 + 64 * jj);
                    v[j0] = w.x;
                    v[j0 | HB] = w.y;
                }
            }
            return;
        }
        if (partial) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const size_t fr = f * FP + (size_t)((64 * j + lane) >> L);
                v[j] = fr < nframes_user ? in[f * 1024 + 64 * j + lane] : 0u;
            }
            return;
        }
        const u32 *src = in + f * 1024 + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = INTFFT_LD(src + 64 * j);
    };

    const size_t wave0 = (size_t)blockIdx.x * 4 + wv;
    const size_t nwaves = (size_t)gridDim.x * 4;
    if (!PIPE) { // one frame at a time: fewer VGPRs, more waves per SIMD
        for (size_t f = wave0; f < nframes; f += nwaves) {
            u32 v[16];
            load_frame(v, f);
            run(v, f);
        }
        return;
    }
    // software-pipelined: frame f+1 is in flight while f is computed
    u32 va[16], vb[16];
    size_t f = wave0;
    if (f < nframes) load_frame(va, f);
    while (f < nframes) {
        const size_t f1 = f + nwaves;
        if (f1 < nframes) load_frame(vb, f1);
        run(va, f);
        if (f1 >= nframes) break;
        f = f1 + nwaves;
        if (f < nframes) load_frame(va, f);
        run(vb, f1);
    }
}

bool packed_width_ok(int data_width, int format, int rndmode)
{
    const bool narrow = diag_env("INTFFT_NO_NARROW16") == nullptr; // read per plan: the tests switch it
    (void)rndmode; // both sum / difference modes
    return data_width == 16 || (narrow && data_width >= 9 && data_width <= 15 && format == 0);
}

bool fast1024_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int direction,
                        int use_fly, int in_order, int out_order)
{
    if (!(packed_width_ok(data_width, format, rndmode) && twdl_width >= 8 && twdl_width <= 16 && format == 0 && direction == 0 && use_fly == 1))
        return false;
    // N >= 128: NATURAL or HALVES (native int_fftNk beats) in, NATURAL, BITREV (native) or BITREV_LANES (the serial form) out; N = 64: natural only
    if (log2n >= 7 && log2n <= 10) return (in_order == 0 || in_order == 2) && (out_order == 0 || out_order == 1 || out_order == 3);
    return log2n == 6 && in_order == 0 && out_order == 0;
}

const char *fast1024_kernel_name() { return "k_fft1024_i16"; }

static int env_int(const char *name, int dflt)
{
    const char *e = diag_env(name);
    return e ? atoi(e) : dflt;
}

template <int L, int ROUND, int OUT_BITREV, bool PIPE, bool FAST_OK>
static hipError_t launch_k(const u32 *in, u32 *out, const int2 *tw, const Fast1024Consts &c, size_t nframes,
                           const Slice &sl, int in_halves, hipStream_t stream)
{
    // persistent waves: the resident grid (occupancy x CUs), so no block waits for a slot.  Neither the grid size nor the
    // occupancy is a lever any more: 4 (= resident, 105 VGPRs) / 5 / 6 / 8 / 12 / 16 / 32 blocks per CU all measure 90-94 us per
    // 65536 frames, inside the run-to-run spread (tools/tune_headline.sh, INTFFT_BLOCKS_PER_CU), and a variant with every
    // per-lane twiddle parked in LDS (90 VGPRs, five waves per SIMD) measured 91.2-91.5 us against 91.1-91.5
    const size_t cap = resident_blocks(kptr(k_fft1024_i16<L, ROUND, OUT_BITREV, PIPE, FAST_OK>), 256, 4, PIPE ? 8 : 0);
    const size_t chunks = (nframes + ((size_t)1 << (10 - L)) - 1) >> (10 - L); // 1024-sample chunks, one per wave pass
    const size_t need = (chunks + 3) / 4;
    const unsigned blocks = (unsigned)(need < cap ? need : cap);
    hipLaunchKernelGGL((k_fft1024_i16<L, ROUND, OUT_BITREV, PIPE, FAST_OK>), dim3(blocks), dim3(256), 0, stream, in, out,
                       tw, c, nframes, sl, in_halves);
    return hipGetLastError();
}

template <int L, int ROUND, int OUT_BITREV>
static hipError_t launch_t(const u32 *in, u32 *out, const int2 *tw, const Fast1024Consts &c, size_t nframes,
                           const Slice &sl, bool fast_ok, int in_halves, hipStream_t stream)
{
    static const int pipe_env = env_int("INTFFT_FAST_PIPE", 1);      // 1: two frames in flight per wave
    static const int allow_fast = env_int("INTFFT_FAST_EXTRACT", 1); // 0: always the exact extraction
    constexpr bool ONLY_PIPE = L < 10;                               // short frames: pipelined variant only
    const bool pipe = ONLY_PIPE || pipe_env;
    if constexpr (!ROUND) {
        if (fast_ok && allow_fast) {
            if constexpr (!ONLY_PIPE)
                if (!pipe) return launch_k<L, ROUND, OUT_BITREV, false, true>(in, out, tw, c, nframes, sl, in_halves, stream);
            return launch_k<L, ROUND, OUT_BITREV, true, true>(in, out, tw, c, nframes, sl, in_halves, stream);
        }
    }
    if constexpr (!ONLY_PIPE)
        if (!pipe) return launch_k<L, ROUND, OUT_BITREV, false, false>(in, out, tw, c, nframes, sl, in_halves, stream);
    return launch_k<L, ROUND, OUT_BITREV, true, false>(in, out, tw, c, nframes, sl, in_halves, stream);
}

template <int L>
static hipError_t launch_short(int round, int out_bitrev, int in_halves, const u32 *in, u32 *out, const int2 *tw,
                               const Fast1024Consts &c, size_t nframes, const Slice &sl, bool fast_ok, hipStream_t stream)
{
    if constexpr (L >= 7) { // out_bitrev: 1 BITREV, 2 BITREV_LANES -> the OUT_BITREV = 1 / 2 instantiations
        if (out_bitrev == 2)
            return round == 2   ? launch_t<L, 2, 2>(in, out, tw, c, nframes, sl, false, in_halves, stream)
                   : round == 1 ? launch_t<L, 1, 2>(in, out, tw, c, nframes, sl, false, in_halves, stream)
                                : launch_t<L, 0, 2>(in, out, tw, c, nframes, sl, fast_ok, in_halves, stream);
        if (out_bitrev)
            return round == 2   ? launch_t<L, 2, 1>(in, out, tw, c, nframes, sl, false, in_halves, stream)
                   : round == 1 ? launch_t<L, 1, 1>(in, out, tw, c, nframes, sl, false, in_halves, stream)
                                : launch_t<L, 0, 1>(in, out, tw, c, nframes, sl, fast_ok, in_halves, stream);
    }
    return round == 2   ? launch_t<L, 2, 0>(in, out, tw, c, nframes, sl, false, in_halves, stream)
           : round == 1 ? launch_t<L, 1, 0>(in, out, tw, c, nframes, sl, false, in_halves, stream)
                        : launch_t<L, 0, 0>(in, out, tw, c, nframes, sl, fast_ok, in_halves, stream);
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
    Slice sl{a.twd - 1, a.twd, 0x05040100u, 0x07060302u};
    if (a.dw != 16) sl.set_width(a.dw);
    const bool fast_ok = a.twd == 16; // high halves == bits [31:16] only for t = 16
    const u32 *pin = static_cast<const u32 *>(in);
    u32 *pout = static_cast<u32 *>(out);
    const int round = a.rnd == RND_ROUND ? (a.dw != 16 ? 2 : 1) : 0; // round mode on narrow data: its own instantiation
    switch (a.log2n) {
    case 6: return launch_short<6>(round, a.out_bitrev, a.in_halves, pin, pout, tw_all, c, nframes, sl, fast_ok, stream);
    case 7: return launch_short<7>(round, a.out_bitrev, a.in_halves, pin, pout, tw_all, c, nframes, sl, fast_ok, stream);
    case 8: return launch_short<8>(round, a.out_bitrev, a.in_halves, pin, pout, tw_all, c, nframes, sl, fast_ok, stream);
    case 9: return launch_short<9>(round, a.out_bitrev, a.in_halves, pin, pout, tw_all, c, nframes, sl, fast_ok, stream);
    default: break;
    }
    return launch_short<10>(round, a.out_bitrev, a.in_halves, pin, pout, tw_all, c, nframes, sl, fast_ok, stream);
}

} // namespace intfft
