// intfft_big20.hip -- multi-pass packed-int16 kernels for N = 2^13 .. 2^20 (BASELINE config 4 is N = 2^20); three passes
// for N >= 2^17 and for the pair, TWO passes for N = 2^13 .. 2^16 forward / inverse (2^16 = 256 x 256, 8192-sample tiles):
//   forward  k_big20_p1<L, ., LOWB = 8>  stages L-1..8 on virtual 2^16-point frames (the pass-1 kernel below with 8 low bits)
//            k_mid_p2 (natural order out: stages 7..0 + bit-reversed store) | k_mid_c<DIF> (BITREV out: 256-point groups)
//   inverse  k_mid_q1 (natural order in) | k_mid_c<DIT> (BITREV in): STAGE 0..7;  k_big20_q1<L, ., 8>: STAGE 8..L-1
//   pair     k_big20_p1<L, ., 8>, k_mid_pair (STAGE 7..0 / 0..7 on every 256-point group, in place), k_big20_q1<L, ., 8>
// The three-pass scheme:
// int_fftNk with NFFT = 13..20, DATA_WIDTH = 16, TWDL_WIDTH <= 16, scaled-truncate, natural in -> natural out
// (stages >= 11 read the Taylor twiddle tables of row_twiddle_tay.vhd, built by k_twiddle_stage).
// The description below is for N = 2^20; shorter frames (template parameter L = log2 N):
//   pass 1  L >= 17: groups of 2^(20-L) consecutive frames form one virtual 2^20-point frame whose top index bits
//           number the frame; their stages are skipped (twiddle index = position mod 2^s is unaffected).
//           L <= 16: groups of 2^(16-L) frames form a virtual 2^16-point frame; stages L-1..12 are one register
//           round (regs = n15..12 at stride 4096, thread = 512 consecutive n: 2 KiB runs, no LDS).
//   pass 2  unchanged (4096 consecutive points); pass 3 uses the top 8 in-frame bits n(L-1)..n(L-8) as its 256
//           rows: brev_L sends them to the 8 lowest output bits (1 KiB runs per store instruction).
//
// A 2^20-point frame is 4 MiB: three passes through a plan-owned scratch, every global access a full
// 128-byte line, every butterfly lane-local (same packed arithmetic as intfft_fast1024.hip):
//
//   pass 1  stages 19..12   tile = 256 values of n19..12  x  32 consecutive n (128-B rows, stride 4096)
//                           512 threads, regs = n19..16 -> LDS transpose -> regs = n15..12, in: user, out: scratch
//   pass 2  stages 11..4    tile = 4096 consecutive n (16 KiB), 256 threads, regs = n11..8 -> LDS -> regs = n7..4
//                           -> LDS back -> coalesced store, scratch in place; twiddles frame invariant
//   pass 3  stages 3..0 + bit reversal (int_bitrev_order.vhd:82-104)
//                           tile = 256 values of n19..12 x 32 consecutive n (n4..0), 512 threads, LDS transpose to
//                           regs = n3..0, thread = (n4, rev8(n19..12)): every store writes 256 contiguous bytes
//
// Values travel between passes in the packed form the next stage consumes: multiplier outputs are
// emitted as Y >> 1 (truncate mode only ever reads Y >> 1), so an element's "kind" (S or Y >> 1) after
// pass 1 is index bit 12 and after pass 2 is index bit 4; the consuming pass shifts by a per-thread amount.
// multi-pass kernels: non-temporal loads measure 4-14 % faster here (the single-pass kernels gain 4-30 % from PLAIN loads): intfft_device.hpp
#define INTFFT_NT_LOADS 1
#include "intfft_pk16.hpp"

#include <cstdlib>

namespace intfft {

constexpr int ROWB = 17; // LDS row stride in dwords: odd -> conflict-free b32 rows and columns

__device__ __forceinline__ constexpr int rev4b(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

__device__ __forceinline__ void ld_tw(const uint2 *__restrict__ t, unsigned idx, u32 &wa, u32 &wb)
{
    const uint2 w = t[idx];
    wa = w.x;
    wb = w.y;
}

// ---- pass 1: stages 19..12 ---------------------------------------------------------------------------
// grid = 128 chunks x G frame groups; a workgroup keeps its chunk's 30 twiddle pairs in registers and walks
// the frames g, g + G, ... of the launch (the twiddles depend on the chunk, not on the frame)
// LOWB = 12: the three-pass split above.  LOWB = 8 (N = 2^13 .. 2^16, two-pass split): the same tile walk on a virtual
// 2^16-point frame, stages 15..8, rows n15..8 at stride 256, 8 chunks of 32 consecutive n7..0; k_mid_p2 finishes.
template <int L, bool FAST_OK, int LOWB = 12>
__global__ __launch_bounds__(512) void k_big20_p1(const u32 *in, u32 *scr, const uint2 *__restrict__ twf, size_t nframes_user,
                                                  unsigned groups, const Slice sl, int halves)
{
    static_assert(L >= LOWB + 5 && L <= LOWB + 8, "two-round pass 1");
    constexpr int LV = LOWB + 8, NS1 = L - (LOWB + 4), G = 1 << (LV - L); // virtual frame; executed stages of round 1; frames per virtual frame
    constexpr unsigned ROW = 1u << LOWB, ROW16 = 16u << LOWB;
    const size_t nframes = (nframes_user + G - 1) / G;   // virtual 2^20-point frames
    __shared__ u32 lds[512 * ROWB];
    const int tid = threadIdx.x, l = tid & 31, hx = tid >> 5; // hx = n15..12 (round 1) / n19..16 (round 2)
    const unsigned chunk = blockIdx.x / groups, grp = blockIdx.x % groups; // chunk 0..2^LOWB/32 - 1
    const unsigned lfull = chunk * 32 + l;                                   // n(LOWB-1)..0

    // round 1: stages 19..16; twiddle index = (n mod 2^s) = ((j mod 2^i) * 16 + hx) * 4096 + lfull
    // round 2: stages 15..12; twiddle index = (reg mod 2^i) * 4096 + lfull
    RoundTw t1, t2;
    {
        const unsigned b = hx * ROW + lfull;
        if constexpr (NS1 >= 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ld_tw(twf, (1u << (LV - 1)) - 1u + b + (unsigned)j * ROW16, t1.wa8[j], t1.wb8[j]);
        }
        if constexpr (NS1 >= 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ld_tw(twf, (1u << (LV - 2)) - 1u + b + (unsigned)j * ROW16, t1.wa4[j], t1.wb4[j]);
        }
        if constexpr (NS1 >= 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) ld_tw(twf, (1u << (LV - 3)) - 1u + b + (unsigned)j * ROW16, t1.wa2[j], t1.wb2[j]);
        }
        ld_tw(twf, (1u << (LV - 4)) - 1u + b, t1.wa1[0], t1.wb1[0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) ld_tw(twf, (1u << (LOWB + 3)) - 1u + lfull + (unsigned)j * ROW, t2.wa8[j], t2.wb8[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) ld_tw(twf, (1u << (LOWB + 2)) - 1u + lfull + (unsigned)j * ROW, t2.wa4[j], t2.wb4[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) ld_tw(twf, (1u << (LOWB + 1)) - 1u + lfull + (unsigned)j * ROW, t2.wa2[j], t2.wb2[j]);
        ld_tw(twf, ROW - 1u + lfull, t2.wa1[0], t2.wb1[0]);
    }
    const v2s none = {0, 0};
    const short s2 = (short)(1 - (hx & 1)); // round 2: kind of the inputs = n(LOWB+4) = (tid >> 5) & 1
    const v2s sh2 = {s2, s2};

    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    for (size_t frame = grp; frame < nframes; frame += groups) {
        const u32 *src = in + frame * ((size_t)1 << LV) + lfull;
        u32 *dst = scr + frame * ((size_t)1 << LV) + lfull;
        // the full-group path: wave-uniform base + 32-bit thread offset (at32: SGPR base, no 64-bit VALU address arithmetic per access)
        const u32 *srcu = in + frame * ((size_t)1 << LV);
        u32 *dstu = scr + frame * ((size_t)1 << LV);
        unsigned toff_a = ((unsigned)hx << LOWB) + lfull, toff_b = ((unsigned)hx << (LOWB + 4)) + lfull;
        asm volatile("" : "+v"(toff_a), "+v"(toff_b));
        u32 v[16];
        const bool partial = L < LV && (frame + 1) * G > nframes_user; // last group: absent frames read as 0, not stored
        if (halves) { // HALVES beats (x[i], x[i + N/2]): beat 65536 jj + 4096 hx + n11..0 of the group -> regs (j0, j0 | 2^(L-17))
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            const v2u *src2 = reinterpret_cast<const v2u *>(in + frame * ((size_t)1 << LV)) + ((size_t)hx << LOWB) + lfull;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                constexpr int HS = L - (LOWB + 5), HB = 1 << HS;
                const int j0 = ((jj >> HS) << (HS + 1)) | (jj & (HB - 1));
                v2u w = {0u, 0u};
                if (!partial || frame * G + (size_t)(jj >> HS) < nframes_user) w = INTFFT_LD(src2 + (size_t)jj * ROW16);
                v[j0] = w.x;
                v[j0 | HB] = w.y;
            }
        } else if (partial) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[j] = frame * G + (size_t)((16 * j + hx) >> (L - LOWB)) < nframes_user ? src[(size_t)(16 * j + hx) << LOWB] : 0u;
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = INTFFT_LD(at32(srcu + ((size_t)j << (LOWB + 4)), toff_a)); // regs = n19..16
        }
        // guard-bit vote of the tile (it is closed under stages 19..12, so its own inputs bound every sum);
        // the barrier also orders the previous frame's LDS reads before this frame's writes
        const bool fast = FAST_OK && !block_any(vote_flags, vote_phase, guard_acc(v, sl.gbias, sl.gmask) != 0);
        if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd); // DATA_WIDTH < 16, exact path: containers wrapped to w bits
        if (!FAST_OK) __syncthreads();
        if (fast) dif_round<FAST_OK, false, NS1>(v, t1, sl, none);
        else if (!FAST_OK && sl.round == 1) dif_round<false, false, NS1, 1>(v, t1, sl, none); // RNDMODE = 1: plain values in every pass, exact extraction
        else if (!FAST_OK && sl.round == 2) dif_round<false, false, NS1, 2>(v, t1, sl, none); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
        else dif_round<false, false, NS1>(v, t1, sl, none);
        // transpose: (thread (hx = n15..12, l), reg j = n19..16) -> (thread (j, l), reg hx)
#pragma unroll
        for (int j = 0; j < 16; ++j) lds[ROWB * (32 * j + l) + hx] = v[j];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = lds[ROWB * tid + r]; // now tid >> 5 = n19..16, regs = n15..12
        if (fast) dif_round<FAST_OK, true>(v, t2, sl, sh2);
        else if (!FAST_OK && sl.round == 1) dif_round<false, true, 4, 1>(v, t2, sl, sh2); // RNDMODE = 1: plain values in every pass, exact extraction
        else if (!FAST_OK && sl.round == 2) dif_round<false, true, 4, 2>(v, t2, sl, sh2); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
        else dif_round<false, true>(v, t2, sl, sh2);
        if (partial) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (frame * G + (size_t)((16 * hx + r) >> (L - LOWB)) < nframes_user) dst[(size_t)(16 * hx + r) << LOWB] = v[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) *at32(dstu + ((size_t)r << LOWB), toff_b) = v[r];
        }
    }
}

// ---- pass 1 for L <= 16: stages L-1..12 as one register round on a virtual 2^16-point frame ---------------
// grid = 8 chunks (512 consecutive n11..0 each) x G frame groups; regs = n15..12, no LDS, wave-level guard vote
template <int L, bool FAST_OK>
__global__ __launch_bounds__(512) void k_big16_p1(const u32 *in, u32 *scr, const uint2 *__restrict__ twf, size_t nframes_user,
                                                  unsigned groups, const Slice sl, int halves)
{
    static_assert(L >= 13 && L <= 16, "one-round pass 1");
    constexpr int NS = L - 12, G = 1 << (16 - L);
    const size_t nframes = (nframes_user + G - 1) / G; // virtual 2^16-point frames
    const unsigned chunk = blockIdx.x / groups, grp = blockIdx.x % groups; // chunk 0..7
    const unsigned lfull = chunk * 512 + threadIdx.x;                        // n11..0
    RoundTw t;
    if constexpr (NS >= 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ld_tw(twf, (1u << 15) - 1u + lfull + (unsigned)j * 4096u, t.wa8[j], t.wb8[j]);
    }
    if constexpr (NS >= 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ld_tw(twf, (1u << 14) - 1u + lfull + (unsigned)j * 4096u, t.wa4[j], t.wb4[j]);
    }
    if constexpr (NS >= 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) ld_tw(twf, (1u << 13) - 1u + lfull + (unsigned)j * 4096u, t.wa2[j], t.wb2[j]);
    }
    ld_tw(twf, (1u << 12) - 1u + lfull, t.wa1[0], t.wb1[0]);
    const v2s none = {0, 0};
    for (size_t frame = grp; frame < nframes; frame += groups) {
        const u32 *src = in + frame * 65536 + lfull;
        u32 *dst = scr + frame * 65536 + lfull;
        u32 v[16];
        const bool partial = L < 16 && (frame + 1) * G > nframes_user;
        if (halves) { // HALVES beats: beat 4096 jj + n11..0 of the group -> regs (j0, j0 | 2^(L-13))
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            const v2u *src2 = reinterpret_cast<const v2u *>(in + frame * 65536) + lfull;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                constexpr int HB = 1 << (L - 13);
                const int j0 = ((jj >> (L - 13)) << (L - 12)) | (jj & (HB - 1));
                v2u w = {0u, 0u};
                if (!partial || frame * G + (size_t)(jj >> (L - 13)) < nframes_user) w = INTFFT_LD(src2 + (size_t)jj * 4096);
                v[j0] = w.x;
                v[j0 | HB] = w.y;
            }
        } else if (partial) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = frame * G + (size_t)(j >> (L - 12)) < nframes_user ? src[(size_t)j << 12] : 0u;
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = INTFFT_LD(src + ((size_t)j << 12));
        }
        const bool fast = FAST_OK && frame_has_guard_bit(v, sl.gbias, sl.gmask);
        if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd); // DATA_WIDTH < 16, exact path: containers wrapped to w bits
        if (fast) dif_round<FAST_OK, false, NS>(v, t, sl, none);
        else if (!FAST_OK && sl.round == 1) dif_round<false, false, NS, 1>(v, t, sl, none); // RNDMODE = 1: plain values in every pass, exact extraction
        else if (!FAST_OK && sl.round == 2) dif_round<false, false, NS, 2>(v, t, sl, none); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
        else dif_round<false, false, NS>(v, t, sl, none);
        if (partial) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (frame * G + (size_t)(j >> (L - 12)) < nframes_user) dst[(size_t)j << 12] = v[j];
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) dst[(size_t)j << 12] = v[j];
        }
    }
}

// ---- inverse pass 1 (DIT mirror of pass 1): STAGE 12..L-1 of int_ifftNk, plan scratch -> user array -----------------
// Same tiles, frame groups and register-resident twiddles as k_big16_p1 / k_big20_p1; the DIT butterflies reuse the
// DIF twiddle packing through the re/im-swapped multiplier feed (int_dit2_fly.vhd:304-322).  Inputs are full-width
// values (DIT outputs are never pre-shifted).
template <int L, bool FAST_OK>
__global__ __launch_bounds__(512) void k_big16_q1(const u32 *scr, u32 *out, const uint2 *__restrict__ twf, size_t nframes_user,
                                                  unsigned groups, const Slice sl, int halves)
{
    static_assert(L >= 13 && L <= 16, "one-round pass");
    constexpr int NS = L - 12, G = 1 << (16 - L);
    const size_t nframes = (nframes_user + G - 1) / G;
    const unsigned chunk = blockIdx.x / groups, grp = blockIdx.x % groups;
    const unsigned lfull = chunk * 512 + threadIdx.x;
    RoundTw t;
    if constexpr (NS >= 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ld_tw(twf, (1u << 15) - 1u + lfull + (unsigned)j * 4096u, t.wa8[j], t.wb8[j]);
    }
    if constexpr (NS >= 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ld_tw(twf, (1u << 14) - 1u + lfull + (unsigned)j * 4096u, t.wa4[j], t.wb4[j]);
    }
    if constexpr (NS >= 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) ld_tw(twf, (1u << 13) - 1u + lfull + (unsigned)j * 4096u, t.wa2[j], t.wb2[j]);
    }
    ld_tw(twf, (1u << 12) - 1u + lfull, t.wa1[0], t.wb1[0]);
    for (size_t frame = grp; frame < nframes; frame += groups) {
        const u32 *src = scr + frame * 65536 + lfull;
        u32 *dst = out + frame * 65536 + lfull;
        u32 v[16];
        const bool partial = L < 16 && (frame + 1) * G > nframes_user;
        if (partial) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = frame * G + (size_t)(j >> (L - 12)) < nframes_user ? src[(size_t)j << 12] : 0u;
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = src[(size_t)j << 12];
        }
        if (FAST_OK && frame_has_guard_bit(v, sl.gbias, sl.gmask)) dit_round<FAST_OK, NS>(v, t, sl);
        else if (!FAST_OK && sl.round == 1) dit_round<false, NS, 1>(v, t, sl); // RNDMODE = 1: plain values in every pass, exact extraction
        else if (!FAST_OK && sl.round == 2) dit_round<false, NS, 2>(v, t, sl); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
        else dit_round<false, NS>(v, t, sl);
        if (halves) {
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            v2u *d2 = reinterpret_cast<v2u *>(out + frame * 65536) + lfull;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                constexpr int HB = 1 << (L - 13);
                const int j0 = ((jj >> (L - 13)) << (L - 12)) | (jj & (HB - 1));
                const v2u w = {v[j0], v[j0 | HB]};
                if (!partial || frame * G + (size_t)(jj >> (L - 13)) < nframes_user) __builtin_nontemporal_store(w, d2 + (size_t)jj * 4096);
            }
        } else if (partial) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (frame * G + (size_t)(j >> (L - 12)) < nframes_user) dst[(size_t)j << 12] = v[j];
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) __builtin_nontemporal_store(v[j], dst + ((size_t)j << 12));
        }
    }
}

// LOWB = 8: the inverse two-pass split (N = 2^13 .. 2^16), STAGE 8..L-1 after k_mid_q1
template <int L, bool FAST_OK, int LOWB = 12>
__global__ __launch_bounds__(512) void k_big20_q1(const u32 *scr, u32 *out, const uint2 *__restrict__ twf, size_t nframes_user,
                                                  unsigned groups, const Slice sl, int halves)
{
    static_assert(L >= LOWB + 5 && L <= LOWB + 8, "two-round pass");
    constexpr int LV = LOWB + 8, NS1 = L - (LOWB + 4), G = 1 << (LV - L);
    constexpr unsigned ROW = 1u << LOWB, ROW16 = 16u << LOWB;
    const size_t nframes = (nframes_user + G - 1) / G;
    __shared__ u32 lds[512 * ROWB];
    const int tid = threadIdx.x, l = tid & 31, hx = tid >> 5; // hx = n19..16 (round 1: regs n15..12) / n15..12 (round 2)
    const unsigned chunk = blockIdx.x / groups, grp = blockIdx.x % groups;
    const unsigned lfull = chunk * 32 + l;
    RoundTw t1, t2; // t2: STAGE 15..12 (index depends on n11..0 only); t1: STAGE 19..16 with hx = n15..12
    {
        const unsigned b = hx * ROW + lfull;
        if constexpr (NS1 >= 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ld_tw(twf, (1u << (LV - 1)) - 1u + b + (unsigned)j * ROW16, t1.wa8[j], t1.wb8[j]);
        }
        if constexpr (NS1 >= 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ld_tw(twf, (1u << (LV - 2)) - 1u + b + (unsigned)j * ROW16, t1.wa4[j], t1.wb4[j]);
        }
        if constexpr (NS1 >= 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) ld_tw(twf, (1u << (LV - 3)) - 1u + b + (unsigned)j * ROW16, t1.wa2[j], t1.wb2[j]);
        }
        ld_tw(twf, (1u << (LV - 4)) - 1u + b, t1.wa1[0], t1.wb1[0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) ld_tw(twf, (1u << (LOWB + 3)) - 1u + lfull + (unsigned)j * ROW, t2.wa8[j], t2.wb8[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) ld_tw(twf, (1u << (LOWB + 2)) - 1u + lfull + (unsigned)j * ROW, t2.wa4[j], t2.wb4[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) ld_tw(twf, (1u << (LOWB + 1)) - 1u + lfull + (unsigned)j * ROW, t2.wa2[j], t2.wb2[j]);
        ld_tw(twf, ROW - 1u + lfull, t2.wa1[0], t2.wb1[0]);
    }
    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    for (size_t frame = grp; frame < nframes; frame += groups) {
        const u32 *src = scr + frame * ((size_t)1 << LV) + lfull;
        u32 *dst = out + frame * ((size_t)1 << LV) + lfull;
        const u32 *srcu = scr + frame * ((size_t)1 << LV); // (the full-group path: at32, see k_big20_p1)
        u32 *dstu = out + frame * ((size_t)1 << LV);
        unsigned toff_a = ((unsigned)hx << LOWB) + lfull, toff_b = ((unsigned)hx << (LOWB + 4)) + lfull;
        asm volatile("" : "+v"(toff_a), "+v"(toff_b));
        const bool partial = L < LV && (frame + 1) * G > nframes_user;
        u32 v[16];
        if (partial) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                v[r] = frame * G + (size_t)((16 * hx + r) >> (L - LOWB)) < nframes_user ? src[(size_t)(16 * hx + r) << LOWB] : 0u;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) // thread hx = n19..16, regs = n15..12; (two-pass split: non-temporal loads +3 %, three-pass: -2 %)
                v[r] = LOWB == 8 ? INTFFT_LD(at32(srcu + ((size_t)r << LOWB), toff_b)) : *at32(srcu + ((size_t)r << LOWB), toff_b);
        }
        const bool fast = FAST_OK && !block_any(vote_flags, vote_phase, guard_acc(v, sl.gbias, sl.gmask) != 0); // also orders the previous LDS reads
        if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd); // DATA_WIDTH < 16, exact path: containers wrapped to w bits
        if (!FAST_OK) __syncthreads();
        if (fast) dit_round<FAST_OK>(v, t2, sl);
        else if (!FAST_OK && sl.round == 1) dit_round<false, 4, 1>(v, t2, sl); // RNDMODE = 1: plain values in every pass, exact extraction
        else if (!FAST_OK && sl.round == 2) dit_round<false, 4, 2>(v, t2, sl); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
        else dit_round<false>(v, t2, sl);
        // transpose: (thread (hx = n19..16, l), reg r = n15..12) -> (thread (r, l), reg hx)
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[ROWB * (32 * r + l) + hx] = v[r];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = lds[ROWB * tid + j]; // now tid >> 5 = n15..12, regs = n19..16
        if (fast) dit_round<FAST_OK, NS1>(v, t1, sl);
        else if (!FAST_OK && sl.round == 1) dit_round<false, NS1, 1>(v, t1, sl); // RNDMODE = 1: plain values in every pass, exact extraction
        else if (!FAST_OK && sl.round == 2) dit_round<false, NS1, 2>(v, t1, sl); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
        else dit_round<false, NS1>(v, t1, sl);
        if (halves) {
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            v2u *d2 = reinterpret_cast<v2u *>(out + frame * ((size_t)1 << LV)) + ((size_t)hx << LOWB) + lfull;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                constexpr int HS = L - (LOWB + 5), HB = 1 << HS;
                const int j0 = ((jj >> HS) << (HS + 1)) | (jj & (HB - 1));
                const v2u w = {v[j0], v[j0 | HB]};
                if (!partial || frame * G + (size_t)(jj >> HS) < nframes_user) __builtin_nontemporal_store(w, d2 + (size_t)jj * ROW16);
            }
        } else if (partial) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (frame * G + (size_t)((16 * j + hx) >> (L - LOWB)) < nframes_user) dst[(size_t)(16 * j + hx) << LOWB] = v[j];
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) __builtin_nontemporal_store(v[j], at32(dstu + ((size_t)j << (LOWB + 4)), toff_a));
        }
    }
}

// twiddles of a four-stage register round from the int2 ROM/Taylor table: STAGE s0+3 .. s0 with regs = n(s0+3)..n(s0) and
// `low` = n(s0-1)..n0 of the thread (table index = position mod 2^s)
template <int S0> __device__ __forceinline__ void load_round_tw(const int2 *__restrict__ twt, int low, RoundTw &t)
{
    auto ld = [&](int idx, u32 &wa, u32 &wb) {
        const int2 w = twt[idx];
        wa = pack_wa(w);
        wb = pack_wb(w);
    };
    constexpr int R = 1 << S0;
#pragma unroll
    for (int j = 0; j < 8; ++j) ld((8 * R - 1) + R * j + low, t.wa8[j], t.wb8[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) ld((4 * R - 1) + R * j + low, t.wa4[j], t.wb4[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) ld((2 * R - 1) + R * j + low, t.wa2[j], t.wb2[j]);
    ld((R - 1) + low, t.wa1[0], t.wb1[0]);
}

// ---- pass 2: stages 11..4 on 4096 consecutive points, in place ----------------------------------------
template <bool FAST_OK>
__global__ __launch_bounds__(256) void k_big20_p2(u32 *scr, const int2 *__restrict__ twt, size_t nblocks4k, const Slice sl)
{
    __shared__ u32 lds[2 * 256 * ROWB];
    u32 *const reg0 = lds, *const reg1 = lds + 256 * ROWB;
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;
    RoundTw ta, tb;
    load_round_tw<8>(twt, tid, ta);   // STAGE 11..8, low = n7..0
    load_round_tw<4>(twt, lo4, tb);   // STAGE 7..4, low = n3..0
    const short sb = (short)(1 - (hi4 & 1)); // LB: kind = n8 = t'4
    const v2s sh_b = {sb, sb};

    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    for (size_t b = blockIdx.x; b < nblocks4k; b += gridDim.x) {
        u32 *p = scr + b * 4096;
        // kind of this block's inputs = n12 = block index bit 0 (pass 1 left Y >> 1 where n12 = 1)
        const short sa = (short)(1 - (int)(b & 1));
        const v2s sh_a = {sa, sa};
        u32 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = p[256 * j + tid]; // LA: regs = n11..8, thread = n7..0
        // guard-bit vote on this block's own inputs (S-type: |v| < 2^14; Y >> 1 type: |v| < 2^13)
        bool fast = false;
        if (FAST_OK) {
            const u32 addc = (b & 1) ? sl.gbias1 : sl.gbias, maskc = (b & 1) ? sl.gmask1 : sl.gmask;
            u32 acc = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc |= v[j] + addc;
            fast = !block_any(vote_flags, vote_phase, (acc & maskc) != 0);
        }
        if (fast) dif_round<FAST_OK, true>(v, ta, sl, sh_a);
        else if (!FAST_OK && sl.round == 1) dif_round<false, true, 4, 1>(v, ta, sl, sh_a); // RNDMODE = 1: plain values in every pass, exact extraction
        else if (!FAST_OK && sl.round == 2) dif_round<false, true, 4, 2>(v, ta, sl, sh_a); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
        else dif_round<false, true>(v, ta, sl, sh_a);
        // LA -> LB: row = 16 j + n3..0, column = n7..4
#pragma unroll
        for (int j = 0; j < 16; ++j) reg0[ROWB * (16 * j + lo4) + hi4] = v[j];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = reg0[ROWB * tid + r]; // LB: regs = n7..4, thread = (n11..8, n3..0)
        if (fast) dif_round<FAST_OK, true>(v, tb, sl, sh_b);
        else if (!FAST_OK && sl.round == 1) dif_round<false, true, 4, 1>(v, tb, sl, sh_b); // RNDMODE = 1: plain values in every pass, exact extraction
        else if (!FAST_OK && sl.round == 2) dif_round<false, true, 4, 2>(v, tb, sl, sh_b); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
        else dif_round<false, true>(v, tb, sl, sh_b);
        // LB -> LA for a coalesced store: element (t' = (n11..8, n3..0), reg j' = n7..4) -> row n7..0 = 16 j' + n3..0,
        // column n11..8
#pragma unroll
        for (int j = 0; j < 16; ++j) reg1[ROWB * (16 * j + lo4) + hi4] = v[j];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) p[256 * r + tid] = reg1[ROWB * tid + r];
        // the next iteration's writes to reg0 / reg1 are ordered behind these reads by its own barriers:
        // reg0 is rewritten only after every thread passed the second barrier above (all reg0 reads done);
        // reg1 is rewritten only after the next first barrier (all reg1 reads done).
    }
}

// ---- pass 3: stages 3..0 and the bit-reversed (natural-order) store ----------------------------------
template <bool FAST_OK>
__global__ __launch_bounds__(512) void k_big20_p3(const u32 *scr, u32 *out, const RoundCConsts c, const Slice sl,
                                                  int L)
{
    __shared__ u32 lds[512 * ROWB];
    const int tid = threadIdx.x, e = tid & 31, px = tid >> 5; // e = n4..0, px = n(L-5)..n(L-8)
    const size_t frame = blockIdx.x >> (L - 13); // frame-major: the tiles of one frame are neighbouring blocks (adjacent rows / runs)
    const unsigned mid = (unsigned)(blockIdx.x & ((1u << (L - 13)) - 1u));
    const u32 *src = scr + (frame << L) + mid * 32 + e;
    u32 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = src[(size_t)(16 * j + px) << (L - 8)]; // reg j = n(L-1)..n(L-4)

    // transpose -> regs = n3..0, thread = (n4, rev8(n19..12)); rev8(16 j + px) = 16 rev4(px) + rev4(j)
    const int rpx = ((px & 1) << 3) | ((px & 2) << 1) | ((px & 4) >> 1) | ((px & 8) >> 3);
    {
        u32 *w = lds + ROWB * ((e >> 4) * 256 + 16 * rpx) + (e & 15);
#pragma unroll
        for (int j = 0; j < 16; ++j) w[ROWB * rev4b(j)] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = lds[ROWB * tid + r];

    // stages 3..0; kind of the inputs = n4 = tid >> 8 (pass 2 left Y >> 1 where n4 = 1)
    const short s3 = (short)(1 - (tid >> 8));
    const v2s sh3 = {s3, s3};
    bool fast = false;
    if (FAST_OK) { // vote on the tile's own inputs; the kind (hence the threshold) depends on n4 = tid >> 8
        const u32 addc = (tid >> 8) ? sl.gbias1 : sl.gbias, maskc = (tid >> 8) ? sl.gmask1 : sl.gmask;
        u32 acc = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc |= v[r] + addc;
    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
        fast = !block_any(vote_flags, vote_phase, (acc & maskc) != 0);
    }
    if (fast) dif_round_c<FAST_OK>(v, c, sl, sh3);
    else if (!FAST_OK && sl.round == 1) dif_round_c<false, 1>(v, c, sl, sh3); // RNDMODE = 1: plain values in every pass, exact extraction
    else if (!FAST_OK && sl.round == 2) dif_round_c<false, 2>(v, c, sl, sh3); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
    else dif_round_c<false>(v, c, sl, sh3);

    // natural order: X index = brev_L(n) = rev4(r) << (L-4) | n4 << (L-5) | brev(mid) << 8 | rev8(n(L-1)..n(L-8))
    const unsigned rmid = L > 13 ? __brev(mid) >> (32 - (L - 13)) : 0u;
    u32 *dst = out + (frame << L) + ((size_t)(tid >> 8) << (L - 5)) + ((size_t)rmid << 8) + (tid & 255);
#pragma unroll
    for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(v[r], dst + ((size_t)rev4b(r) << (L - 4)));
}

// ---- two-pass split, N = 2^13 .. 2^16: pass 2 = stages 7..0 and the bit-reversed (natural-order) store --------------
// After k_big20_p1<L, ., 8> (stages L-1..8) every 256 consecutive points are an independent 256-point DIF whose twiddles
// depend on n7..0 only.  tile = 32 rows R = n(L-1)..n(L-5) x 256 consecutive n7..0 (mid = n(L-6)..n8 fixed), 512 threads:
//   round 1  thread = (R, n3..0), regs = n7..4 (64-B runs per row), stages 7..4, twiddles frame- and row-invariant
//   LDS transpose -> thread = (n7..4, rev5(R)), regs = n3..0, stages 3..0 (constant twiddles)
//   store    X index = brev_L(n) = rev4(n3..0) << (L-4) | rev4(n7..4) << (L-8) | rev(mid) << 5 | rev5(R): 128-B runs
// kind of the inputs = n8 (pass 1 left Y >> 1 where n8 = 1) = mid bit 0, or R bit 0 when L = 13.
template <bool FAST_OK>
__global__ __launch_bounds__(512) void k_mid_p2(const u32 *scr, u32 *out, const int2 *__restrict__ twt, const RoundCConsts c,
                                                const Slice sl, int L)
{
    __shared__ u32 lds[512 * ROWB];
    const int tid = threadIdx.x, lo4 = tid & 15, R = tid >> 4;
    const size_t frame = blockIdx.x >> (L - 13); // frame-major: the tiles of one frame are neighbouring blocks (adjacent rows / runs)
    const unsigned mid = (unsigned)(blockIdx.x & ((1u << (L - 13)) - 1u));
    RoundTw tb;
    load_round_tw<4>(twt, lo4, tb); // STAGE 7..4, low = n3..0
    const u32 *src = scr + (frame << L) + ((size_t)R << (L - 5)) + mid * 256 + lo4;
    u32 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = src[16 * j]; // regs = n7..4
    const int k8 = L > 13 ? (int)(mid & 1u) : (R & 1);
    const short sa = (short)(1 - k8);
    const v2s sh_a = {sa, sa};
    bool fast = false;
    if (FAST_OK) { // vote on the tile's own inputs (closed under stages 7..0); the threshold depends on the kind
        const u32 addc = k8 ? sl.gbias1 : sl.gbias, maskc = k8 ? sl.gmask1 : sl.gmask;
        u32 acc = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc |= v[j] + addc;
    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
        fast = !block_any(vote_flags, vote_phase, (acc & maskc) != 0);
    }
    if (fast) dif_round<FAST_OK, true>(v, tb, sl, sh_a);
    else if (!FAST_OK && sl.round == 1) dif_round<false, true, 4, 1>(v, tb, sl, sh_a); // RNDMODE = 1: plain values in every pass, exact extraction
    else if (!FAST_OK && sl.round == 2) dif_round<false, true, 4, 2>(v, tb, sl, sh_a); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
    else dif_round<false, true>(v, tb, sl, sh_a);
    // transpose: (thread (R, n3..0), reg j = n7..4) -> (thread 32 j + rev5(R), reg n3..0)
    {
        const int rR = (int)(__brev((unsigned)R) >> 27);
        u32 *w = lds + ROWB * rR + lo4;
#pragma unroll
        for (int j = 0; j < 16; ++j) w[ROWB * 32 * j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = lds[ROWB * tid + r];
    const int hi4 = tid >> 5;                     // n7..4
    const short s3 = (short)(1 - (hi4 & 1));      // kind = n4
    const v2s sh3 = {s3, s3};
    if (fast) dif_round_c<FAST_OK>(v, c, sl, sh3);
    else if (!FAST_OK && sl.round == 1) dif_round_c<false, 1>(v, c, sl, sh3); // RNDMODE = 1: plain values in every pass, exact extraction
    else if (!FAST_OK && sl.round == 2) dif_round_c<false, 2>(v, c, sl, sh3); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
    else dif_round_c<false>(v, c, sl, sh3);
    const unsigned rmid = L > 13 ? __brev(mid) >> (32 - (L - 13)) : 0u;
    u32 *dst = out + (frame << L) + ((size_t)rmid << 5); // wave-uniform; thread part through at32 (SGPR base + 32-bit offset)
    const unsigned toff = ((unsigned)rev4b(hi4) << (L - 8)) + (unsigned)(tid & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(v[r], at32(dst + ((size_t)rev4b(r) << (L - 4)), toff));
}

// ---- inverse two-pass split (mirror of k_mid_p2): bit-reversed load of the natural-order input + DIT STAGE 0..7 -------
// thread = (n7..4, rev5(R)), regs = n3..0 -> STAGE 0..3 -> LDS -> thread = (R, n3..0), regs = n7..4 -> STAGE 4..7 -> scratch
template <bool FAST_OK>
__global__ __launch_bounds__(512) void k_mid_q1(const u32 *in, u32 *scr, const int2 *__restrict__ twt, const RoundCConsts c,
                                                const Slice sl, int L)
{
    __shared__ u32 lds[512 * ROWB];
    const int tid = threadIdx.x, lo4 = tid & 15, R = tid >> 4, hi4 = tid >> 5;
    const size_t frame = blockIdx.x >> (L - 13); // frame-major: the tiles of one frame are neighbouring blocks (adjacent rows / runs)
    const unsigned mid = (unsigned)(blockIdx.x & ((1u << (L - 13)) - 1u));
    RoundTw tb;
    load_round_tw<4>(twt, lo4, tb); // STAGE 7..4, low = n3..0
    const unsigned rmid = L > 13 ? __brev(mid) >> (32 - (L - 13)) : 0u;
    const u32 *src = in + (frame << L) + ((size_t)rev4b(hi4) << (L - 8)) + ((size_t)rmid << 5) + (tid & 31);
    u32 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = INTFFT_LD(src + ((size_t)rev4b(r) << (L - 4)));
    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    const bool fast = FAST_OK && !block_any(vote_flags, vote_phase, guard_acc(v, sl.gbias, sl.gmask) != 0); // the tile is closed under STAGE 0..7
    if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd); // DATA_WIDTH < 16, exact path: containers wrapped to w bits
    if (fast) dit_round_c<FAST_OK>(v, c, sl);
    else if (!FAST_OK && sl.round == 1) dit_round_c<false, 1>(v, c, sl); // RNDMODE = 1: plain values in every pass, exact extraction
    else if (!FAST_OK && sl.round == 2) dit_round_c<false, 2>(v, c, sl); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
    else dit_round_c<false>(v, c, sl);
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[ROWB * tid + r] = v[r];
    __syncthreads();
    {
        const int rR = (int)(__brev((unsigned)R) >> 27);
        const u32 *w = lds + ROWB * rR + lo4;
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = w[ROWB * 32 * j]; // thread = (R, n3..0), regs = n7..4
    }
    if (fast) dit_round<FAST_OK>(v, tb, sl);
    else if (!FAST_OK && sl.round == 1) dit_round<false, 4, 1>(v, tb, sl); // RNDMODE = 1: plain values in every pass, exact extraction
    else if (!FAST_OK && sl.round == 2) dit_round<false, 4, 2>(v, tb, sl); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
    else dit_round<false>(v, tb, sl);
    // thread = (R, n3..0), regs n7..4: a store instruction would write four 64-byte pieces of four rows.  Two lane swaps
    // (reg bit 0 = n4 <-> lane bit 4 = R bit 0, reg bit 1 = n5 <-> lane bit 5 = R bit 1) make the lanes n5..0: 256-byte runs
    swap_guard(v);
#pragma unroll
    for (int j = 0; j < 16; j += 2) swap16(v[j], v[j + 1]);
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (!(j & 2)) swap32(v[j], v[j + 2]);
    u32 *dst = scr + (frame << L) + ((size_t)(R & ~3) << (L - 5)) + mid * 256 + (tid & 63);
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[((size_t)(r & 3) << (L - 5)) + ((r >> 2) << 6)] = v[r]; // row (R & ~3) | (r & 3), n7..6 = r >> 2
}

// ---- inverse pass 3 (mirror of pass 3): bit-reversed load of the natural-order input + DIT STAGE 0..3 -----------------
template <bool FAST_OK>
__global__ __launch_bounds__(512) void k_big20_q3(const u32 *in, u32 *scr, const RoundCConsts c, const Slice sl, int L)
{
    __shared__ u32 lds[512 * ROWB];
    const int tid = threadIdx.x;
    const size_t frame = blockIdx.x >> (L - 13); // frame-major: the tiles of one frame are neighbouring blocks (adjacent rows / runs)
    const unsigned mid = (unsigned)(blockIdx.x & ((1u << (L - 13)) - 1u));
    // thread = (n4 = tid >> 8, rev8(n(L-1)..n(L-8)) = tid & 255), regs = n3..0: X index = brev_L(n) (1 KiB runs)
    const unsigned rmid = L > 13 ? __brev(mid) >> (32 - (L - 13)) : 0u;
    const u32 *src = in + (frame << L) + ((size_t)(tid >> 8) << (L - 5)) + ((size_t)rmid << 8) + (tid & 255);
    u32 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = INTFFT_LD(src + ((size_t)rev4b(r) << (L - 4)));
    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    const bool fast = FAST_OK && !block_any(vote_flags, vote_phase, guard_acc(v, sl.gbias, sl.gmask) != 0);
    if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd); // DATA_WIDTH < 16, exact path: containers wrapped to w bits
    if (fast) dit_round_c<FAST_OK>(v, c, sl);
    else if (!FAST_OK && sl.round == 1) dit_round_c<false, 1>(v, c, sl); // RNDMODE = 1: plain values in every pass, exact extraction
    else if (!FAST_OK && sl.round == 2) dit_round_c<false, 2>(v, c, sl); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
    else dit_round_c<false>(v, c, sl);
    // transpose to thread = (px = n(L-5)..n(L-8), e = n4..0), regs j = n(L-1)..n(L-4): rev8(16 j + px) = tid & 255
    const int px = rev4b((tid >> 4) & 15), j = rev4b(tid & 15);
    {
        u32 *w = lds + ROWB * (32 * px + 16 * (tid >> 8)) + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) w[ROWB * r] = v[r];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = lds[ROWB * tid + q];
    u32 *dst = scr + (frame << L) + mid * 32 + (tid & 31);
#pragma unroll
    for (int q = 0; q < 16; ++q) dst[(size_t)(16 * q + (tid >> 5)) << (L - 8)] = v[q];
}

// ---- inverse pass 2 (mirror of pass 2): DIT STAGE 4..11 on 4096 consecutive points, in place ------------------------------
template <bool FAST_OK>
__global__ __launch_bounds__(256) void k_big20_q2(u32 *scr, const int2 *__restrict__ twt, size_t nblocks4k, const Slice sl)
{
    __shared__ u32 lds[2 * 256 * ROWB];
    u32 *const reg0 = lds, *const reg1 = lds + 256 * ROWB;
    const int tid = threadIdx.x, lo4 = tid & 15, hi4 = tid >> 4;
    RoundTw ta, tb;
    load_round_tw<8>(twt, tid, ta);   // STAGE 11..8, low = n7..0
    load_round_tw<4>(twt, lo4, tb);   // STAGE 7..4, low = n3..0

    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    for (size_t b = blockIdx.x; b < nblocks4k; b += gridDim.x) {
        u32 *p = scr + b * 4096;
        u32 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = p[256 * j + tid]; // LA: regs = n11..8, thread = n7..0
        const bool fast = FAST_OK && !block_any(vote_flags, vote_phase, guard_acc(v, sl.gbias, sl.gmask) != 0);
        if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd); // DATA_WIDTH < 16, exact path: containers wrapped to w bits
        if (!FAST_OK) __syncthreads(); // orders the previous block's reg1 reads before this block's writes
        // LA -> LB: row = 16 j + n3..0, column = n7..4
#pragma unroll
        for (int j = 0; j < 16; ++j) reg0[ROWB * (16 * j + lo4) + hi4] = v[j];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = reg0[ROWB * tid + r]; // LB: regs = n7..4, thread = (n11..8, n3..0)
        if (fast) dit_round<FAST_OK>(v, tb, sl);
        else if (!FAST_OK && sl.round == 1) dit_round<false, 4, 1>(v, tb, sl); // RNDMODE = 1: plain values in every pass, exact extraction
        else if (!FAST_OK && sl.round == 2) dit_round<false, 4, 2>(v, tb, sl); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
        else dit_round<false>(v, tb, sl);
        // LB -> LA: element (t' = (n11..8, n3..0), reg j' = n7..4) -> row n7..0 = 16 j' + n3..0, column n11..8
#pragma unroll
        for (int j = 0; j < 16; ++j) reg1[ROWB * (16 * j + lo4) + hi4] = v[j];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = reg1[ROWB * tid + r]; // LA again
        if (fast) dit_round<FAST_OK>(v, ta, sl);
        else if (!FAST_OK && sl.round == 1) dit_round<false, 4, 1>(v, ta, sl); // RNDMODE = 1: plain values in every pass, exact extraction
        else if (!FAST_OK && sl.round == 2) dit_round<false, 4, 2>(v, ta, sl); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
        else dit_round<false>(v, ta, sl);
#pragma unroll
        for (int r = 0; r < 16; ++r) p[256 * r + tid] = v[r];
    }
}

// ---- BITREV order on the frequency side (native int_fftNk output / int_ifftNk input beats): memory index = core index, so
// the last four DIF stages (or the first four DIT stages) act on 16 consecutive samples.  One wave per 1024 consecutive
// samples: dwordx4 loads, two lane swaps -> regs n3..0 (lane = n9..n4), stages, two lane swaps, dwordx4 stores.  No LDS.
template <bool DIT, bool FAST_OK>
__global__ __launch_bounds__(256) void k_big_c(const u32 *src, u32 *dst, const RoundCConsts c, size_t nchunks, const Slice sl)
{
    const int lane = threadIdx.x & 63;
    const size_t wave0 = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const int unit = ((lane & 15) << 2) | (lane >> 4); // x4 unit of the lane's vector q = (n9 n8): (q | n7..n4 | n3 n2)
    // DIF: pass 2 left Y >> 1 where n4 = 1; after the swaps lane bit 0 = n4
    const short s3 = (short)(1 - (lane & 1));
    const v2s sh3 = {s3, s3};
    for (size_t ch = wave0; ch < nchunks; ch += nwaves) {
        const v4u *s4 = reinterpret_cast<const v4u *>(src + ch * 1024) + unit;
        u32 v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4u x = INTFFT_LD(s4 + 64 * q);
            v[4 * q] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
        }
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int r = 0; r < 4; ++r) swap16(v[g + r], v[g + r + 4]);
#pragma unroll
        for (int r = 0; r < 8; ++r) swap32(v[r], v[r + 8]);
        bool fast = false;
        if (FAST_OK) {
            u32 acc = 0;
            const u32 addc = (!DIT && (lane & 1)) ? sl.gbias1 : sl.gbias, maskc = (!DIT && (lane & 1)) ? sl.gmask1 : sl.gmask;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc |= v[r] + addc;
            fast = __builtin_amdgcn_ballot_w64((acc & maskc) != 0) == 0;
        }
        if (DIT && !fast && sl.wd != 16) wrap_inputs(v, sl.wd);
        if (DIT) {
            if (fast) dit_round_c<FAST_OK>(v, c, sl);
            else if (!FAST_OK && sl.round == 1) dit_round_c<false, 1>(v, c, sl); // RNDMODE = 1: plain values in every pass, exact extraction
            else if (!FAST_OK && sl.round == 2) dit_round_c<false, 2>(v, c, sl); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
            else dit_round_c<false>(v, c, sl);
        } else {
            if (fast) dif_round_c<FAST_OK>(v, c, sl, sh3);
            else if (!FAST_OK && sl.round == 1) dif_round_c<false, 1>(v, c, sl, sh3); // RNDMODE = 1: plain values in every pass, exact extraction
            else if (!FAST_OK && sl.round == 2) dif_round_c<false, 2>(v, c, sl, sh3); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
            else dif_round_c<false>(v, c, sl, sh3);
        }
        swap_guard(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) swap32(v[r], v[r + 8]);
#pragma unroll
        for (int g = 0; g < 16; g += 8)
#pragma unroll
            for (int r = 0; r < 4; ++r) swap16(v[g + r], v[g + r + 4]);
        v4u *d4 = reinterpret_cast<v4u *>(dst + ch * 1024) + unit;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4u x = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            __builtin_nontemporal_store(x, d4 + 64 * q);
        }
    }
}

// ---- two-pass split with BITREV order on the frequency side, N = 2^13 .. 2^16: the last eight DIF stages (first eight DIT
// stages) act on 256 consecutive samples.  One wave per 1024 consecutive samples (q = n9 n8 numbers the four 256-point groups):
//   DIF  regs = n7..4, lane = (q, n3..0): 64-B runs from the scratch; stages 7..4; wave-private LDS transpose to the
//        k_big_c layout (regs = n3..0, lane = n9..n4); stages 3..0; two lane swaps; dwordx4 stores
//   DIT  the same walk backwards: dwordx4 loads of the BITREV-ordered input, STAGE 0..3, transpose, STAGE 4..7, scratch
// kind of the DIF inputs = n8 (k_big20_p1<., ., 8> left Y >> 1 where n8 = 1) = lane bit 4.
template <bool DIT, bool FAST_OK>
__global__ __launch_bounds__(256) void k_mid_c(const u32 *src, u32 *dst, const int2 *__restrict__ twt, const RoundCConsts c,
                                               size_t nchunks, const Slice sl)
{
    __shared__ u32 lds_all[4 * 64 * ROWB];
    u32 *const lds = lds_all + (threadIdx.x >> 6) * 64 * ROWB;
    const int lane = threadIdx.x & 63, lo4 = lane & 15, q = lane >> 4;
    const size_t wave0 = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const int unit = ((lane & 15) << 2) | (lane >> 4); // as in k_big_c
    RoundTw tb;
    load_round_tw<4>(twt, lo4, tb); // STAGE 7..4, low = n3..0
    const short sa = (short)(1 - (q & 1)), s3 = (short)(1 - (lane & 1));
    const v2s sh_a = {sa, sa}, sh3 = {s3, s3};
    for (size_t ch = wave0; ch < nchunks; ch += nwaves) {
        u32 v[16];
        if (!DIT) {
            const u32 *p = src + ch * 1024 + q * 256 + lo4;
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = p[16 * j];
            bool fast = false;
            if (FAST_OK) {
                const u32 addc = (q & 1) ? sl.gbias1 : sl.gbias, maskc = (q & 1) ? sl.gmask1 : sl.gmask;
                u32 acc = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc |= v[j] + addc;
                fast = __builtin_amdgcn_ballot_w64((acc & maskc) != 0) == 0;
            }
            if (fast) dif_round<FAST_OK, true>(v, tb, sl, sh_a);
            else if (!FAST_OK && sl.round == 1) dif_round<false, true, 4, 1>(v, tb, sl, sh_a); // RNDMODE = 1: plain values in every pass, exact extraction
            else if (!FAST_OK && sl.round == 2) dif_round<false, true, 4, 2>(v, tb, sl, sh_a); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
            else dif_round<false, true>(v, tb, sl, sh_a);
            wave_lds_fence(); // keep the previous chunk's reads ahead of these writes
#pragma unroll
            for (int j = 0; j < 16; ++j) lds[ROWB * (16 * q + j) + lo4] = v[j];
            wave_lds_fence(); // LDS ops of one wave execute in order: no barrier needed
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = lds[ROWB * lane + r]; // regs = n3..0, lane = n9..n4
            wave_lds_fence();
            if (fast) dif_round_c<FAST_OK>(v, c, sl, sh3);
            else if (!FAST_OK && sl.round == 1) dif_round_c<false, 1>(v, c, sl, sh3); // RNDMODE = 1: plain values in every pass, exact extraction
            else if (!FAST_OK && sl.round == 2) dif_round_c<false, 2>(v, c, sl, sh3); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
            else dif_round_c<false>(v, c, sl, sh3);
            swap_guard(v);
#pragma unroll
            for (int r = 0; r < 8; ++r) swap32(v[r], v[r + 8]);
#pragma unroll
            for (int g = 0; g < 16; g += 8)
#pragma unroll
                for (int r = 0; r < 4; ++r) swap16(v[g + r], v[g + r + 4]);
            v4u *d4 = reinterpret_cast<v4u *>(dst + ch * 1024) + unit;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const v4u x = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
                __builtin_nontemporal_store(x, d4 + 64 * k);
            }
        } else {
            const v4u *s4 = reinterpret_cast<const v4u *>(src + ch * 1024) + unit;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const v4u x = INTFFT_LD(s4 + 64 * k);
                v[4 * k] = x.x, v[4 * k + 1] = x.y, v[4 * k + 2] = x.z, v[4 * k + 3] = x.w;
            }
#pragma unroll
            for (int g = 0; g < 16; g += 8)
#pragma unroll
                for (int r = 0; r < 4; ++r) swap16(v[g + r], v[g + r + 4]);
#pragma unroll
            for (int r = 0; r < 8; ++r) swap32(v[r], v[r + 8]);
            const bool fast = FAST_OK && frame_has_guard_bit(v, sl.gbias, sl.gmask); // the 1024 samples are closed under STAGE 0..7
            if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd);
            if (fast) dit_round_c<FAST_OK>(v, c, sl);
            else if (!FAST_OK && sl.round == 1) dit_round_c<false, 1>(v, c, sl); // RNDMODE = 1: plain values in every pass, exact extraction
            else if (!FAST_OK && sl.round == 2) dit_round_c<false, 2>(v, c, sl); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
            else dit_round_c<false>(v, c, sl);
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) lds[ROWB * lane + r] = v[r];
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = lds[ROWB * (16 * q + j) + lo4]; // regs = n7..4, lane = (q, n3..0)
            wave_lds_fence();
            if (fast) dit_round<FAST_OK>(v, tb, sl);
            else if (!FAST_OK && sl.round == 1) dit_round<false, 4, 1>(v, tb, sl); // RNDMODE = 1: plain values in every pass, exact extraction
            else if (!FAST_OK && sl.round == 2) dit_round<false, 4, 2>(v, tb, sl); // ... on narrow data: with the w-bit wraps (intfft_pk16.hpp)
            else dit_round<false>(v, tb, sl);
            u32 *p = dst + ch * 1024 + q * 256 + lo4;
#pragma unroll
            for (int j = 0; j < 16; ++j) p[16 * j] = v[j];
        }
    }
}

// ---- pair, 256 x 256 split (N = 2^13 .. 2^16): the whole pair of STAGE 7..0 / 0..7 on every 256-point group, in place.
// k_mid_c's forward and inverse halves back to back: the bit reversal between the cores cancels (int_fft_ifft_pair.vhd:242-280),
// so the DIF result at core index n is the DIT input at core index n.  One guard-bit vote on the inputs covers all 16 stages.
template <bool FAST_OK>
__global__ __launch_bounds__(256) void k_mid_pair(u32 *scr, const int2 *__restrict__ twt, const RoundCConsts c, size_t nchunks,
                                                  const Slice sl)
{
    __shared__ u32 lds_all[4 * 64 * ROWB];
    u32 *const lds = lds_all + (threadIdx.x >> 6) * 64 * ROWB;
    const int lane = threadIdx.x & 63, lo4 = lane & 15, q = lane >> 4;
    const size_t wave0 = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    RoundTw tb;
    load_round_tw<4>(twt, lo4, tb);
    // both cores run here: DIT packing, the forward core on D with exchanged halves (group4's DPK form); round mode keeps the DIF
    // packing (its forward core needs the unswapped D), as in the single-pass pair kernels
    const bool rnd = !FAST_OK && sl.round;
    if (!rnd) to_dit_packing(tb);
    const short sa = (short)(1 - (q & 1)), s3 = (short)(1 - (lane & 1)); // kinds: n8 = lane bit 4, then n4 = lane bit 0
    const v2s sh_a = {sa, sa}, sh3 = {s3, s3};
    for (size_t ch = wave0; ch < nchunks; ch += nwaves) {
        u32 *p = scr + ch * 1024 + q * 256 + lo4;
        u32 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = p[16 * j]; // regs = n7..4, lane = (q, n3..0)
        bool fast = false;
        if (FAST_OK) {
            const u32 addc = (q & 1) ? sl.gbias1 : sl.gbias, maskc = (q & 1) ? sl.gmask1 : sl.gmask;
            u32 acc = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc |= v[j] + addc;
            fast = __builtin_amdgcn_ballot_w64((acc & maskc) != 0) == 0;
        }
#define INTFFT_MIDPAIR(FX, RD, DP)                                                                               \
    {                                                                                                            \
        dif_round<FX, true, 4, RD, DP>(v, tb, sl, sh_a);                                                         \
        wave_lds_fence();                                                                           \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) lds[ROWB * (16 * q + j) + lo4] = v[j];                    \
        wave_lds_fence(); /* LDS ops of one wave execute in order */                                \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) v[r] = lds[ROWB * lane + r]; /* regs n3..0, lane n9..n4 */ \
        wave_lds_fence();                                                                           \
        dif_round_c<FX, RD, DP>(v, c, sl, sh3);                                                                  \
        dit_round_c<FX, RD, DP>(v, c, sl);                                                                       \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) lds[ROWB * lane + r] = v[r];                              \
        wave_lds_fence();                                                                           \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) v[j] = lds[ROWB * (16 * q + j) + lo4];                    \
        wave_lds_fence();                                                                           \
        dit_round<FX, 4, RD, DP>(v, tb, sl);                                                                     \
    }
        if (FAST_OK && fast) INTFFT_MIDPAIR(FAST_OK, false, true)
        else if (rnd && sl.round == 2) INTFFT_MIDPAIR(false, 2, false)
        else if (rnd) INTFFT_MIDPAIR(false, 1, false)
        else INTFFT_MIDPAIR(false, false, true)
#undef INTFFT_MIDPAIR
#pragma unroll
        for (int j = 0; j < 16; ++j) p[16 * j] = v[j];
    }
}

bool big20_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int direction, int use_fly,
                     int in_order, int out_order)
{
    // RNDMODE = 1: every direction; the 32-register two-pass plans of N = 2^17 / 2^18 are truncate-mode only (planner)
    if (rndmode && diag_env("INTFFT_NO_PACKED_ROUND")) return false;
    if (rndmode && data_width != 16 && direction == 2 && log2n > 16) return false; // narrow round-mode pair beyond N = 65536: generic kernels
    return log2n >= 13 && log2n <= 20 && packed_width_ok(data_width, format, rndmode) && twdl_width >= 8 && twdl_width <= 16 && format == 0 &&
           use_fly == 1 &&
           (direction == 0 ? (in_order == 0 || in_order == 2) && (out_order == 0 || out_order == 1)   // + HALVES in, BITREV out
            : direction == 1 ? (in_order == 0 || in_order == 1) && (out_order == 0 || out_order == 2) // + BITREV in, HALVES out
                             : in_order == 0 && out_order == 0);
}

const char *big20_kernel_name(int direction, int two_pass, int freq_bitrev)
{
    if (two_pass == 2 && direction == 2) return "k_big2p_a/k_mid_pair/k_big2p_q"; // N = 2^17, 2^18
    if (two_pass == 2)
        return direction == 1 ? (freq_bitrev ? "k_mid_c/k_big2p_q" : "k_mid_q1/k_big2p_q") : (freq_bitrev ? "k_big2p_a/k_mid_c" : "k_big2p_a/k_mid_p2");
    return two_pass ? (direction == 2 ? "k_big20_p1/k_mid_pair/q1"
                       : direction == 1 ? (freq_bitrev ? "k_mid_c/k_big20_q1" : "k_mid_q1/k_big20_q1")
                                      : (freq_bitrev ? "k_big20_p1/k_mid_c" : "k_big20_p1/k_mid_p2"))
                    : direction == 1 ? "k_big20_q3/q2/q1" : direction == 2 ? "k_big20_p1/k_fft4096_i16<MID>/q1" : "k_big20_p1/p2/p3";
}

template <int L>
static void launch_p1(bool fx, const u32 *pin, u32 *scr, const uint2 *tw16f, size_t nframes, const Slice &sl, hipStream_t stream,
                      int halves = 0)
{
    if constexpr (L >= 17) {
        const size_t nvf = (nframes + ((size_t)1 << (20 - L)) - 1) >> (20 - L);
        const unsigned groups = (unsigned)(nvf < 16 ? nvf : 16);
        if (fx) hipLaunchKernelGGL((k_big20_p1<L, true>), dim3(128u * groups), dim3(512), 0, stream, pin, scr, tw16f, nframes, groups, sl, halves);
        else hipLaunchKernelGGL((k_big20_p1<L, false>), dim3(128u * groups), dim3(512), 0, stream, pin, scr, tw16f, nframes, groups, sl, halves);
    } else {
        const size_t nvf = (nframes + ((size_t)1 << (16 - L)) - 1) >> (16 - L);
        const unsigned groups = (unsigned)(nvf < 128 ? nvf : 128);
        if (fx) hipLaunchKernelGGL((k_big16_p1<L, true>), dim3(8u * groups), dim3(512), 0, stream, pin, scr, tw16f, nframes, groups, sl, halves);
        else hipLaunchKernelGGL((k_big16_p1<L, false>), dim3(8u * groups), dim3(512), 0, stream, pin, scr, tw16f, nframes, groups, sl, halves);
    }
}

template <int L>
static void launch_q1(bool fx, const u32 *scr, u32 *pout, const uint2 *tw16f, size_t nframes, const Slice &sl, hipStream_t stream,
                      int halves = 0)
{
    if constexpr (L >= 17) {
        const size_t nvf = (nframes + ((size_t)1 << (20 - L)) - 1) >> (20 - L);
        const unsigned groups = (unsigned)(nvf < 16 ? nvf : 16);
        if (fx) hipLaunchKernelGGL((k_big20_q1<L, true>), dim3(128u * groups), dim3(512), 0, stream, scr, pout, tw16f, nframes, groups, sl, halves);
        else hipLaunchKernelGGL((k_big20_q1<L, false>), dim3(128u * groups), dim3(512), 0, stream, scr, pout, tw16f, nframes, groups, sl, halves);
    } else {
        const size_t nvf = (nframes + ((size_t)1 << (16 - L)) - 1) >> (16 - L);
        const unsigned groups = (unsigned)(nvf < 128 ? nvf : 128);
        if (fx) hipLaunchKernelGGL((k_big16_q1<L, true>), dim3(8u * groups), dim3(512), 0, stream, scr, pout, tw16f, nframes, groups, sl, halves);
        else hipLaunchKernelGGL((k_big16_q1<L, false>), dim3(8u * groups), dim3(512), 0, stream, scr, pout, tw16f, nframes, groups, sl, halves);
    }
}

// FFT -> IFFT pair for N = 2^13 .. 2^20 in three passes: DIF STAGE L-1..12 (pass 1 above), then the whole pair of
// STAGE 11..0 / 0..11 on every 4096-point block in place (k_fft4096_i16<MODE_MID>: the bit reversal between the cores
// cancels, int_fft_ifft_pair.vhd:242-280), then DIT STAGE 12..L-1 (k_big16_q1 / k_big20_q1).
hipError_t launch_bigpair(int log2n, int twd, int two_pass, const void *in, void *out, void *scratch, const int2 *tw_all,
                          const uint2 *tw16f, const int2 *h_tw, size_t nframes, hipStream_t stream, int data_width, int rndmode)
{
    if (nframes == 0) return hipSuccess;
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    if (data_width != 16) sl.set_width(data_width);
    const u32 *pin = static_cast<const u32 *>(in);
    u32 *pout = static_cast<u32 *>(out), *scr = static_cast<u32 *>(scratch);
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fx = twd == 16 && allow_fast && !rndmode;
    sl.round = rndmode ? (data_width != 16 ? 2 : 1) : 0;
    if (two_pass) { // 2^(L-8) x 256 split: DIF L-1..8, the pair of 7..0 / 0..7 per 256-point group, DIT 8..L-1
        RoundCConsts c;
        for (int k = 0; k < 8; ++k) {
            const int2 w = h_tw[7 + k];
            c.wa3[k] = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
            c.wb3[k] = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
        }
        for (int k = 0; k < 4; ++k) {
            const int2 w = h_tw[3 + k];
            c.wa2[k] = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
            c.wb2[k] = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
        }
        if (!rndmode) to_dit_packing_host(c); // k_mid_pair holds its twiddles in the DIT packing (round mode: DIF packing)
        const int vsh = log2n <= 16 ? 16 - log2n : 0;
        const size_t nvf = (nframes + ((size_t)1 << vsh) - 1) >> vsh;
        const unsigned groups = (unsigned)(nvf < 256 ? nvf : 256);
        const size_t nch = nframes << (log2n - 10), ccap = (size_t)device_cus() * 8;
        const unsigned gc = (unsigned)((nch + 3) / 4 < ccap ? (nch + 3) / 4 : ccap);
        if (log2n > 16) { // N = 2^17, 2^18: the 32-register passes on either side
            hipError_t e = launch_big2p_a(log2n, fx, pin, scr, tw16f, nframes, sl, 0, stream);
            if (e != hipSuccess) return e;
            if (fx) hipLaunchKernelGGL(k_mid_pair<true>, dim3(gc), dim3(256), 0, stream, scr, tw_all, c, nch, sl);
            else hipLaunchKernelGGL(k_mid_pair<false>, dim3(gc), dim3(256), 0, stream, scr, tw_all, c, nch, sl);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            return launch_big2p_q(log2n, fx, scr, pout, tw16f, nframes, sl, 0, stream);
        }
#define INTFFT_PAIR256(LL)                                                                                                       \
    if (fx) {                                                                                                                    \
        hipLaunchKernelGGL((k_big20_p1<LL, true, 8>), dim3(8u * groups), dim3(512), 0, stream, pin, scr, tw16f, nframes, groups, sl, 0); \
        hipLaunchKernelGGL(k_mid_pair<true>, dim3(gc), dim3(256), 0, stream, scr, tw_all, c, nch, sl);                           \
        hipLaunchKernelGGL((k_big20_q1<LL, true, 8>), dim3(8u * groups), dim3(512), 0, stream, scr, pout, tw16f, nframes, groups, sl, 0); \
    } else {                                                                                                                     \
        hipLaunchKernelGGL((k_big20_p1<LL, false, 8>), dim3(8u * groups), dim3(512), 0, stream, pin, scr, tw16f, nframes, groups, sl, 0); \
        hipLaunchKernelGGL(k_mid_pair<false>, dim3(gc), dim3(256), 0, stream, scr, tw_all, c, nch, sl);                          \
        hipLaunchKernelGGL((k_big20_q1<LL, false, 8>), dim3(8u * groups), dim3(512), 0, stream, scr, pout, tw16f, nframes, groups, sl, 0); \
    }
        switch (log2n) {
        case 13: INTFFT_PAIR256(13) break;
        case 14: INTFFT_PAIR256(14) break;
        case 15: INTFFT_PAIR256(15) break;
        default: INTFFT_PAIR256(16) break;
        }
#undef INTFFT_PAIR256
        return hipGetLastError();
    }
    switch (log2n) {
    case 13: launch_p1<13>(fx, pin, scr, tw16f, nframes, sl, stream); break;
    case 14: launch_p1<14>(fx, pin, scr, tw16f, nframes, sl, stream); break;
    case 15: launch_p1<15>(fx, pin, scr, tw16f, nframes, sl, stream); break;
    case 16: launch_p1<16>(fx, pin, scr, tw16f, nframes, sl, stream); break;
    case 17: launch_p1<17>(fx, pin, scr, tw16f, nframes, sl, stream); break;
    case 18: launch_p1<18>(fx, pin, scr, tw16f, nframes, sl, stream); break;
    case 19: launch_p1<19>(fx, pin, scr, tw16f, nframes, sl, stream); break;
    default: launch_p1<20>(fx, pin, scr, tw16f, nframes, sl, stream); break;
    }
    const hipError_t e = launch_fast4096_mid(twd, scr, nframes << (log2n - 12), tw_all, h_tw, stream, data_width, rndmode);
    if (e != hipSuccess) return e;
    switch (log2n) {
    case 13: launch_q1<13>(fx, scr, pout, tw16f, nframes, sl, stream); break;
    case 14: launch_q1<14>(fx, scr, pout, tw16f, nframes, sl, stream); break;
    case 15: launch_q1<15>(fx, scr, pout, tw16f, nframes, sl, stream); break;
    case 16: launch_q1<16>(fx, scr, pout, tw16f, nframes, sl, stream); break;
    case 17: launch_q1<17>(fx, scr, pout, tw16f, nframes, sl, stream); break;
    case 18: launch_q1<18>(fx, scr, pout, tw16f, nframes, sl, stream); break;
    case 19: launch_q1<19>(fx, scr, pout, tw16f, nframes, sl, stream); break;
    default: launch_q1<20>(fx, scr, pout, tw16f, nframes, sl, stream); break;
    }
    return hipGetLastError();
}

// int_ifftNk for N = 2^13 .. 2^20: the three passes mirrored (k_big20_q3, k_big20_q2, k_big16_q1 / k_big20_q1)
hipError_t launch_biginv(int log2n, int twd, int in_bitrev, int out_halves, int two_pass, const void *in, void *out, void *scratch,
                         const int2 *tw_all, const uint2 *tw16f, const int2 *h_tw, size_t nframes, hipStream_t stream, int data_width, int rndmode)
{
    if (nframes == 0) return hipSuccess;
    RoundCConsts c;
    for (int k = 0; k < 8; ++k) {
        const int2 w = h_tw[7 + k];
        c.wa3[k] = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        c.wb3[k] = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    }
    for (int k = 0; k < 4; ++k) {
        const int2 w = h_tw[3 + k];
        c.wa2[k] = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        c.wb2[k] = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    }
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    if (data_width != 16) sl.set_width(data_width);
    const u32 *pin = static_cast<const u32 *>(in);
    u32 *pout = static_cast<u32 *>(out), *scr = static_cast<u32 *>(scratch);
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fx = twd == 16 && allow_fast && !rndmode; // round mode: the exact-path instantiations, sl.round set
    sl.round = rndmode ? (data_width != 16 ? 2 : 1) : 0;
    const size_t nb3 = nframes << (log2n - 13), nb = nframes << (log2n - 12);
    if (nb3 > 0x7fffffffull) return hipErrorInvalidValue;
    if (two_pass && log2n >= 19) // N = 2^19, 2^20: the mirrors of the forward two-pass plan (intfft_big2x.hip); natural order in only (planner)
        return launch_big2x_inv(log2n, fx, pin, pout, scr, tw16f, h_tw, nframes, sl, out_halves, stream, in_bitrev);
    if (two_pass && log2n > 16) { // N = 2^17, 2^18: the same first pass, then STAGE 8..L-1 with 32 registers per thread
        if (in_bitrev) {
            const size_t nch = nframes << (log2n - 10), ccap = (size_t)device_cus() * 8;
            const unsigned gc = (unsigned)((nch + 3) / 4 < ccap ? (nch + 3) / 4 : ccap);
            if (fx) hipLaunchKernelGGL((k_mid_c<true, true>), dim3(gc), dim3(256), 0, stream, pin, scr, tw_all, c, nch, sl);
            else hipLaunchKernelGGL((k_mid_c<true, false>), dim3(gc), dim3(256), 0, stream, pin, scr, tw_all, c, nch, sl);
        } else if (fx) hipLaunchKernelGGL(k_mid_q1<true>, dim3((unsigned)nb3), dim3(512), 0, stream, pin, scr, tw_all, c, sl, log2n);
        else hipLaunchKernelGGL(k_mid_q1<false>, dim3((unsigned)nb3), dim3(512), 0, stream, pin, scr, tw_all, c, sl, log2n);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        return launch_big2p_q(log2n, fx, scr, pout, tw16f, nframes, sl, out_halves, stream);
    }
    if (two_pass && log2n <= 16) { // two-pass split: (bit-reversed) load + STAGE 0..7, then STAGE 8..L-1
        const size_t nvf = (nframes + ((size_t)1 << (16 - log2n)) - 1) >> (16 - log2n);
        const unsigned groups = (unsigned)(nvf < 256 ? nvf : 256);
        if (in_bitrev) {
            const size_t nch = nframes << (log2n - 10), ccap = (size_t)device_cus() * 8;
            const unsigned gc = (unsigned)((nch + 3) / 4 < ccap ? (nch + 3) / 4 : ccap);
            if (fx) hipLaunchKernelGGL((k_mid_c<true, true>), dim3(gc), dim3(256), 0, stream, pin, scr, tw_all, c, nch, sl);
            else hipLaunchKernelGGL((k_mid_c<true, false>), dim3(gc), dim3(256), 0, stream, pin, scr, tw_all, c, nch, sl);
        } else if (fx) hipLaunchKernelGGL(k_mid_q1<true>, dim3((unsigned)nb3), dim3(512), 0, stream, pin, scr, tw_all, c, sl, log2n);
        else hipLaunchKernelGGL(k_mid_q1<false>, dim3((unsigned)nb3), dim3(512), 0, stream, pin, scr, tw_all, c, sl, log2n);
#define INTFFT_Q1A(LL)                                                                                                           \
    if (fx) hipLaunchKernelGGL((k_big20_q1<LL, true, 8>), dim3(8u * groups), dim3(512), 0, stream, scr, pout, tw16f, nframes, groups, sl, out_halves); \
    else hipLaunchKernelGGL((k_big20_q1<LL, false, 8>), dim3(8u * groups), dim3(512), 0, stream, scr, pout, tw16f, nframes, groups, sl, out_halves)
        switch (log2n) {
        case 13: INTFFT_Q1A(13); break;
        case 14: INTFFT_Q1A(14); break;
        case 15: INTFFT_Q1A(15); break;
        default: INTFFT_Q1A(16); break;
        }
#undef INTFFT_Q1A
        return hipGetLastError();
    }
    const size_t cap = resident_blocks(kptr(k_big20_q2<true>), 256, 4, 0, false);
    const unsigned g2 = (unsigned)(nb < cap ? nb : cap);
    const size_t nch = nframes << (log2n - 10), ccap = (size_t)device_cus() * 8; // k_big_c: 1024-sample chunks, one per wave pass
    const unsigned gc = (unsigned)((nch + 3) / 4 < ccap ? (nch + 3) / 4 : ccap);
    if (fx) {
        if (in_bitrev) hipLaunchKernelGGL((k_big_c<true, true>), dim3(gc), dim3(256), 0, stream, pin, scr, c, nch, sl);
        else hipLaunchKernelGGL(k_big20_q3<true>, dim3((unsigned)nb3), dim3(512), 0, stream, pin, scr, c, sl, log2n);
        hipLaunchKernelGGL(k_big20_q2<true>, dim3(g2), dim3(256), 0, stream, scr, tw_all, nb, sl);
    } else {
        if (in_bitrev) hipLaunchKernelGGL((k_big_c<true, false>), dim3(gc), dim3(256), 0, stream, pin, scr, c, nch, sl);
        else hipLaunchKernelGGL(k_big20_q3<false>, dim3((unsigned)nb3), dim3(512), 0, stream, pin, scr, c, sl, log2n);
        hipLaunchKernelGGL(k_big20_q2<false>, dim3(g2), dim3(256), 0, stream, scr, tw_all, nb, sl);
    }
    switch (log2n) {
    case 13: launch_q1<13>(fx, scr, pout, tw16f, nframes, sl, stream, out_halves); break;
    case 14: launch_q1<14>(fx, scr, pout, tw16f, nframes, sl, stream, out_halves); break;
    case 15: launch_q1<15>(fx, scr, pout, tw16f, nframes, sl, stream, out_halves); break;
    case 16: launch_q1<16>(fx, scr, pout, tw16f, nframes, sl, stream, out_halves); break;
    case 17: launch_q1<17>(fx, scr, pout, tw16f, nframes, sl, stream, out_halves); break;
    case 18: launch_q1<18>(fx, scr, pout, tw16f, nframes, sl, stream, out_halves); break;
    case 19: launch_q1<19>(fx, scr, pout, tw16f, nframes, sl, stream, out_halves); break;
    default: launch_q1<20>(fx, scr, pout, tw16f, nframes, sl, stream, out_halves); break;
    }
    return hipGetLastError();
}

hipError_t launch_big20(int log2n, int twd, int in_halves, int out_bitrev, int two_pass, const void *in, void *out, void *scratch,
                        const int2 *tw_all, const uint2 *tw16f, const int2 *h_tw, size_t nframes, hipStream_t stream, int data_width, int rndmode)
{
    if (nframes == 0) return hipSuccess;
    RoundCConsts c;
    for (int k = 0; k < 8; ++k) {
        const int2 w = h_tw[7 + k];
        c.wa3[k] = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        c.wb3[k] = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    }
    for (int k = 0; k < 4; ++k) {
        const int2 w = h_tw[3 + k];
        c.wa2[k] = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        c.wb2[k] = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    }
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    if (data_width != 16) sl.set_width(data_width);
    const u32 *pin = static_cast<const u32 *>(in);
    u32 *pout = static_cast<u32 *>(out), *scr = static_cast<u32 *>(scratch);
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fx = twd == 16 && allow_fast && !rndmode; // round mode: the exact-path instantiations, sl.round set
    sl.round = rndmode ? (data_width != 16 ? 2 : 1) : 0;
    if (two_pass && log2n >= 19) // N = 2^19, 2^20: 1024 rows x 1024 columns, two ten-stage passes (intfft_big2x.hip); natural or BITREV order out
        return launch_big2x(log2n, fx, pin, pout, scr, tw16f, h_tw, nframes, sl, in_halves, stream, out_bitrev);
    if (two_pass && log2n > 16) { // N = 2^17, 2^18: the 32-register first pass (stages L-1..8), then the same second pass
        const size_t nb2 = nframes << (log2n - 13);
        if (nb2 > 0x7fffffffull) return hipErrorInvalidValue;
        const hipError_t e = launch_big2p_a(log2n, fx, pin, scr, tw16f, nframes, sl, in_halves, stream);
        if (e != hipSuccess) return e;
        if (out_bitrev) {
            const size_t nch = nframes << (log2n - 10), ccap = (size_t)device_cus() * 8;
            const unsigned gc = (unsigned)((nch + 3) / 4 < ccap ? (nch + 3) / 4 : ccap);
            if (fx) hipLaunchKernelGGL((k_mid_c<false, true>), dim3(gc), dim3(256), 0, stream, scr, pout, tw_all, c, nch, sl);
            else hipLaunchKernelGGL((k_mid_c<false, false>), dim3(gc), dim3(256), 0, stream, scr, pout, tw_all, c, nch, sl);
        } else if (fx) hipLaunchKernelGGL(k_mid_p2<true>, dim3((unsigned)nb2), dim3(512), 0, stream, scr, pout, tw_all, c, sl, log2n);
        else hipLaunchKernelGGL(k_mid_p2<false>, dim3((unsigned)nb2), dim3(512), 0, stream, scr, pout, tw_all, c, sl, log2n);
        return hipGetLastError();
    }
    if (two_pass && log2n <= 16) { // two-pass split: stages L-1..8, then stages 7..0 + the bit-reversed store (or none: BITREV out)
        const size_t nvf = (nframes + ((size_t)1 << (16 - log2n)) - 1) >> (16 - log2n);
        const unsigned groups = (unsigned)(nvf < 256 ? nvf : 256);
        const size_t nb2 = nframes << (log2n - 13);
        if (nb2 > 0x7fffffffull) return hipErrorInvalidValue;
#define INTFFT_P1A(LL)                                                                                                           \
    if (fx) hipLaunchKernelGGL((k_big20_p1<LL, true, 8>), dim3(8u * groups), dim3(512), 0, stream, pin, scr, tw16f, nframes, groups, sl, in_halves); \
    else hipLaunchKernelGGL((k_big20_p1<LL, false, 8>), dim3(8u * groups), dim3(512), 0, stream, pin, scr, tw16f, nframes, groups, sl, in_halves)
        switch (log2n) {
        case 13: INTFFT_P1A(13); break;
        case 14: INTFFT_P1A(14); break;
        case 15: INTFFT_P1A(15); break;
        default: INTFFT_P1A(16); break;
        }
#undef INTFFT_P1A
        if (out_bitrev) {
            const size_t nch = nframes << (log2n - 10), ccap = (size_t)device_cus() * 8;
            const unsigned gc = (unsigned)((nch + 3) / 4 < ccap ? (nch + 3) / 4 : ccap);
            if (fx) hipLaunchKernelGGL((k_mid_c<false, true>), dim3(gc), dim3(256), 0, stream, scr, pout, tw_all, c, nch, sl);
            else hipLaunchKernelGGL((k_mid_c<false, false>), dim3(gc), dim3(256), 0, stream, scr, pout, tw_all, c, nch, sl);
        } else if (fx) hipLaunchKernelGGL(k_mid_p2<true>, dim3((unsigned)nb2), dim3(512), 0, stream, scr, pout, tw_all, c, sl, log2n);
        else hipLaunchKernelGGL(k_mid_p2<false>, dim3((unsigned)nb2), dim3(512), 0, stream, scr, pout, tw_all, c, sl, log2n);
        return hipGetLastError();
    }
    switch (log2n) {
    case 13: launch_p1<13>(fx, pin, scr, tw16f, nframes, sl, stream, in_halves); break;
    case 14: launch_p1<14>(fx, pin, scr, tw16f, nframes, sl, stream, in_halves); break;
    case 15: launch_p1<15>(fx, pin, scr, tw16f, nframes, sl, stream, in_halves); break;
    case 16: launch_p1<16>(fx, pin, scr, tw16f, nframes, sl, stream, in_halves); break;
    case 17: launch_p1<17>(fx, pin, scr, tw16f, nframes, sl, stream, in_halves); break;
    case 18: launch_p1<18>(fx, pin, scr, tw16f, nframes, sl, stream, in_halves); break;
    case 19: launch_p1<19>(fx, pin, scr, tw16f, nframes, sl, stream, in_halves); break;
    default: launch_p1<20>(fx, pin, scr, tw16f, nframes, sl, stream, in_halves); break;
    }
    const size_t nb = nframes << (log2n - 12);
    const size_t cap = resident_blocks(kptr(k_big20_p2<true>), 256, 4, 0, false);
    const size_t nb3 = nframes << (log2n - 13);
    if (nb3 > 0x7fffffffull) return hipErrorInvalidValue;
    const unsigned g2 = (unsigned)(nb < cap ? nb : cap), g3 = (unsigned)nb3;
    const size_t nch = nframes << (log2n - 10), ccap = (size_t)device_cus() * 8;
    const unsigned gc = (unsigned)((nch + 3) / 4 < ccap ? (nch + 3) / 4 : ccap);
    if (fx) {
        hipLaunchKernelGGL(k_big20_p2<true>, dim3(g2), dim3(256), 0, stream, scr, tw_all, nb, sl);
        if (out_bitrev) hipLaunchKernelGGL((k_big_c<false, true>), dim3(gc), dim3(256), 0, stream, scr, pout, c, nch, sl);
        else hipLaunchKernelGGL(k_big20_p3<true>, dim3(g3), dim3(512), 0, stream, scr, pout, c, sl, log2n);
    } else {
        hipLaunchKernelGGL(k_big20_p2<false>, dim3(g2), dim3(256), 0, stream, scr, tw_all, nb, sl);
        if (out_bitrev) hipLaunchKernelGGL((k_big_c<false, false>), dim3(gc), dim3(256), 0, stream, scr, pout, c, nch, sl);
        else hipLaunchKernelGGL(k_big20_p3<false>, dim3(g3), dim3(512), 0, stream, scr, pout, c, sl, log2n);
    }
    return hipGetLastError();
}

} // namespace intfft
