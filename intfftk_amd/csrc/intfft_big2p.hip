// intfft_big2p.hip -- two-pass plans for N = 2^17 and 2^18: the pass of stages 8..L-1 with 32 registers per thread (forward
// from natural order: k_big2p_a; inverse: k_big2p_q).  int_fftNk, DATA_WIDTH = 16, TWDL_WIDTH <= 16, scaled-truncate (same packed arithmetic as intfft_fast1024.hip).
//
// N = 2^L = 2^(L-8) x 256.  The second pass is the one the N <= 2^16 two-pass plans already use: k_mid_p2 (stages 7..0 and
// the bit-reversed store on tiles of 32 rows x 256 points) or k_mid_c (BITREV order out).  What was missing for L > 16 is a
// first pass that does L - 8 = 9 or 10 stages (k_big20_p1 does at most 8): here a thread holds 32 samples, so one register
// round is FIVE stages, and two rounds around one block-wide LDS transpose cover 9 (5 + 4) or 10 (5 + 5) stages:
//
//   tile    2^(L-8) rows n(L-1)..n8 (stride 256 samples = 1 KiB)  x  32 consecutive n (128-byte rows); 8 column chunks per frame
//   round 1 thread = (hx = n(7+RB)..n8, l = n4..n0), regs j = n(L-1)..n(L-5): stages L-1..L-5
//   LDS     row (j << RB | hx), column l                          (RB = L - 13 = 4 or 5 stages in round 2)
//   round 2 L = 17: thread = (n16..n13, l), regs = (n12, n11..n8): two independent 4-stage rounds 11..8 (n12 rides along)
//           L = 18: thread = (n17..n13, l), regs = n12..n8: stages 12..8
//   store   in place (same positions in the plan scratch), Y >> 1 where n8 = 1 -- what k_mid_p2 / k_mid_c expect
//
// Twiddles: quarter-turn sharing (RoundTwQ, intfft_pk16.hpp) keeps a 5-stage round at 16 base pairs and a 4-stage round at 8.
// Round 2's set depends on the column only and is parked in LDS; round 1's set is per thread and is re-read from the
// L2-resident table in every frame (what fits 128 VGPRs = four waves per SIMD; see the notes in the kernel).
// L = 17: 512 threads, 68 KiB of LDS -> two workgroups per CU.  L = 18: 1024 threads, 136 KiB -> one.
// Measured (256 MiB of input): N = 2^17 332 Gsample/s (three passes: 252), N = 2^18 298 (247).
// k_big2p_q below is the inverse counterpart (DIT STAGE 8..L-1 after k_mid_q1 / k_mid_c): 325 (247) and 307 (243); the pair runs
// k_big2p_a, k_mid_pair, k_big2p_q: 202 (164) and 181 (162).
// multi-pass kernels: non-temporal loads measure 4-14 % faster here (the single-pass kernels gain 4-30 % from PLAIN loads): intfft_device.hpp
#define INTFFT_NT_LOADS 1
#include "intfft_pk16.hpp"

#include <cstdlib>

namespace intfft {

constexpr int ROW2P = 33; // LDS row stride in dwords (32 columns + 1): conflict-free rows and columns

template <int L, bool FAST_OK, int ROUND = 0> // ROUND: RNDMODE = 1 (2: on narrow data), its own instantiation (round 4: quarter turns through the negated twiddle)
__global__ __launch_bounds__(32 << (L - 13)) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_big2p_a(const u32 *in, u32 *scr, const uint2 *__restrict__ twf, size_t nframes,
                                                            unsigned groups, const Slice sl, int halves)
{
    static_assert(L == 17 || L == 18, "9 or 10 stages");
    static_assert(!ROUND || !FAST_OK, "round mode: exact extraction");
    constexpr int RB = L - 13;        // stages of round 2; thread bits hx
    constexpr int T = 32 << RB;
    extern __shared__ u32 lds[];      // (32 << RB) rows x ROW2P, then the round-2 twiddles: 8 (RB = 4) / 16 slots x 32 columns of {wa, wb}
    uint2 *const tw2 = reinterpret_cast<uint2 *>(lds + T * ROW2P + (T * ROW2P & 1));
    const int tid = threadIdx.x, l = tid & 31, hx = tid >> 5;
    const unsigned chunk = blockIdx.x & 7u, grp = blockIdx.x >> 3; // neighbouring blocks = neighbouring columns of one frame
    const unsigned lfull = chunk * 32 + l;                         // n7..n0
    const unsigned toff = ((unsigned)hx << 8) | lfull;             // this thread's offset inside a block of rows (n(7+RB)..n0)

    // every global access below is (wave-uniform pointer)[32-bit thread offset]: the compiler then addresses with an SGPR base
    // and ONE offset VGPR (global_load ... v_off, s[base]) instead of materialising a 64-bit address pair per access -- with
    // 32 loads, 32 stores and 16 twiddle loads per frame that was 160 VGPRs of addresses and 60 spilled dwords per lane
    auto ld = [&](unsigned uniform_idx, unsigned thread_off, u32 &wa, u32 &wb) {
        const uint2 w = ld2_at32b(twf + uniform_idx, thread_off); // thread_off: the thread's BYTE offset (opaque per frame)
        wa = w.x;
        wb = w.y;
    };
    // round 1: reg bit b <-> stage L-5+b; index of twiddle (stage s, low reg bits jj) = ((jj << RB | hx) << 8) | lfull
    u32 wa16[8], wb16[8];
    RoundTwQ t1;
    // round 2's twiddles depend on (chunk, l) only: the 32 threads with hx = 0 park them in LDS, every thread re-reads its
    // column's set in each frame (same address for all rows: broadcast reads) -- 32 VGPRs less than holding them
    if (hx == 0) {
        int slot = 0;
        auto park = [&](unsigned uniform_idx) { tw2[32 * slot++ + l] = (twf + uniform_idx)[lfull]; };
        if constexpr (RB == 5)
            for (int rr = 0; rr < 8; ++rr) park((1u << 12) - 1u + ((unsigned)rr << 8));
        for (int rr = 0; rr < 4; ++rr) park((1u << 11) - 1u + ((unsigned)rr << 8));
        for (int rr = 0; rr < 2; ++rr) park((1u << 10) - 1u + ((unsigned)rr << 8));
        park((1u << 9) - 1u);
        park((1u << 8) - 1u);
    }
    // round 1's 16 base twiddles are per thread (they depend on hx and l): re-read from the L2-resident table in every frame,
    // right behind the data loads.  (Held in VGPRs over the frame walk they push the kernel past 128 registers: measured
    // 309 vs 332 Gsample/s at N = 2^17, 260 vs 298 at 2^18.)
    auto round1_tw = [&](unsigned to) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) ld((1u << (L - 1)) - 1u + ((unsigned)jj << (RB + 8)), to, wa16[jj], wb16[jj]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ld((1u << (L - 2)) - 1u + ((unsigned)jj << (RB + 8)), to, t1.wa8[jj], t1.wb8[jj]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) ld((1u << (L - 3)) - 1u + ((unsigned)jj << (RB + 8)), to, t1.wa4[jj], t1.wb4[jj]);
        ld((1u << (L - 4)) - 1u, to, t1.wa2[0], t1.wb2[0]);
        ld((1u << (L - 5)) - 1u, to, t1.wa1[0], t1.wb1[0]);
    };
    // round 2 (stage 8 + b, table index (rr << 8) | lfull) reads its set from LDS
    auto round2_tw = [&](u32 (&wa2t)[8], u32 (&wb2t)[8], RoundTwQ &t2) {
        int slot = 0;
        auto get = [&](u32 &wa, u32 &wb) {
            const uint2 w = tw2[32 * slot++ + l];
            wa = w.x;
            wb = w.y;
        };
        if constexpr (RB == 5) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) get(wa2t[rr], wb2t[rr]);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) get(t2.wa8[rr], t2.wb8[rr]);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) get(t2.wa4[rr], t2.wb4[rr]);
        get(t2.wa2[0], t2.wb2[0]);
        get(t2.wa1[0], t2.wb1[0]);
    };
    u32 *const wr_base = lds + ROW2P * hx + l;              // transpose, write side: row (j << RB) + hx
    const u32 *const rd_base = lds + ROW2P * (hx << 5) + l; // read side: row (jx << 5) + q
    const unsigned toff2 = ((unsigned)hx << 13) | lfull; // store side: thread = (jx = tid >> 5, l): rows jx << 5
    const v2s none = {0, 0};
    // L = 18, round 2: the kind of its inputs is n13 = bit 0 of the new thread index (tid >> 5)
    const short s2 = (short)(1 - (hx & 1));
    const v2s sh2 = {s2, s2};

    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    for (size_t frame = grp; frame < nframes; frame += groups) {
        const u32 *src = in + (frame << L); // wave-uniform
        u32 *dst = scr + (frame << L);
        // opaque copies of the loop-invariant thread offsets: otherwise LICM hoists the 16 twiddle addresses (and more) out of
        // the frame loop into 40+ VGPRs that then spill; recomputing an address costs two VALU operations
        unsigned toff_l = toff, toff2_l = toff2;
        asm volatile("" : "+v"(toff_l), "+v"(toff2_l));
        u32 v[32];
        if (halves) { // HALVES order in: memory index = 2 * (n without n(L-1)) + n(L-1); registers j and j + 16 are one 8-byte load
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            const v2u *s2 = reinterpret_cast<const v2u *>(src);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v2u w = INTFFT_LD(at32(s2 + ((size_t)j << (RB + 8)), toff_l));
                v[j] = w.x;
                v[j + 16] = w.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = INTFFT_LD(at32(src + ((size_t)j << (RB + 8)), toff_l)); // regs = n(L-1)..n(L-5)
        }
        unsigned twb = toff_l * 8u;
        asm volatile("" : "+v"(twb));
        round1_tw(twb);
        // guard-bit vote of the tile (closed under stages L-1..8); the barrier also orders the previous frame's LDS reads
        bool fast = false;
        {
            u32 acc = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) acc |= v[j] + sl.gbias;
            const bool bad = block_any(vote_flags, vote_phase, (acc & sl.gmask) != 0);
            fast = FAST_OK && !bad;
        }
        if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd); // DATA_WIDTH < 16, exact path: containers wrapped to w bits
#define INTFFT_2P_ROUND1(FX)                                                                                  \
    {                                                                                                         \
        dif_top16<FX, 0, false, ROUND>(v, wa16, wb16, sl, none);                                              \
        dif_round_q<FX, 0, 0, false, 4, ROUND>(v, t1, sl, none);                                              \
        dif_round_q<FX, 16, 0xF, false, 4, ROUND>(v, t1, sl, none);                                           \
    }
        if (fast) INTFFT_2P_ROUND1(FAST_OK)
        else INTFFT_2P_ROUND1(false)
#undef INTFFT_2P_ROUND1
        // transpose: (thread (hx, l), reg j) -> row (j << RB | hx)
#pragma unroll
        for (int j = 0; j < 32; ++j) wr_base[ROW2P * (j << RB)] = v[j]; // per-thread base + compile-time offset (ds_write offset:)
        __syncthreads();
        u32 wa2t[8], wb2t[8]; // L = 18: top stage of round 2
        RoundTwQ t2;
        round2_tw(wa2t, wb2t, t2);
        // thread = (jx = n(L-1)..n13, l): rows (jx << 5 | q), regs q = n12..n8 (L = 17: n12 rides along as a passenger bit)
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = rd_base[ROW2P * q];
        if constexpr (RB == 4) { // two independent 4-stage rounds 11..8; the kind of their inputs is n12 = q bit 4
            if (fast) {
                dif_round_q<FAST_OK, 0, 0, false>(v, t2, sl, none);
                dif_round_q<FAST_OK, 16, 0xF, false>(v, t2, sl, none);
            } else {
                dif_round_q<false, 0, 0, false, 4, ROUND>(v, t2, sl, none);
                dif_round_q<false, 16, 0xF, false, 4, ROUND>(v, t2, sl, none);
            }
        } else { // stages 12..8; the kind of the inputs is n13 = jx bit 0 (a thread bit)
            if (fast) {
                dif_top16<FAST_OK, 0, true>(v, wa2t, wb2t, sl, sh2);
                dif_round_q<FAST_OK, 0, 0, false>(v, t2, sl, none);
                dif_round_q<FAST_OK, 16, 0xF, false>(v, t2, sl, none);
            } else {
                dif_top16<false, 0, true, ROUND>(v, wa2t, wb2t, sl, sh2);
                dif_round_q<false, 0, 0, false, 4, ROUND>(v, t2, sl, none);
                dif_round_q<false, 16, 0xF, false, 4, ROUND>(v, t2, sl, none);
            }
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) *at32(dst + ((size_t)q << 8), toff2_l) = v[q]; // row (jx << 5 | q): uniform q part + thread part
    }
    (void)T;
}

// ---- the inverse counterpart: DIT STAGE 8..L-1 after k_mid_q1 / k_mid_c<DIT> (which leave STAGE 0..7 done, in place) -----------
//   round 1 thread = (hx = n(L-1)..n13, l = n4..n0), regs q = n12..n8
//           L = 18: stages 8..12;  L = 17: two independent 4-stage rounds 8..11 (n12 rides along)
//   LDS     row rho = n(L-1)..n8 (natural), column l
//   round 2 thread = (p = n(L-6)..n8, l), regs j = n(L-1)..n(L-5): stages L-5..L-1
//   store   natural order, or HALVES (n(L-1) is the top register bit: the two halves are registers j and j + 16)
// Round 1's twiddles depend on the column only (parked in LDS, DIT packing); round 2's are per thread, re-read from the
// L2-resident table in every frame and converted to the DIT packing on arrival.
template <int L, bool FAST_OK>
__global__ __launch_bounds__(32 << (L - 13)) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_big2p_q(const u32 *scr, u32 *out, const uint2 *__restrict__ twf, size_t nframes,
                                                            unsigned groups, const Slice sl, int halves)
{
    static_assert(L == 17 || L == 18, "9 or 10 stages");
    constexpr int RB = L - 13; // thread bits hx of round 1 = stages of round 1 = thread bits p of round 2
    constexpr int T = 32 << RB;
    extern __shared__ u32 lds[];
    uint2 *const tw1 = reinterpret_cast<uint2 *>(lds + T * ROW2P + (T * ROW2P & 1));
    const int tid = threadIdx.x, l = tid & 31, hx = tid >> 5;
    const unsigned chunk = blockIdx.x & 7u, grp = blockIdx.x >> 3;
    const unsigned lfull = chunk * 32 + l;
    const unsigned toff = ((unsigned)hx << 13) | lfull; // load side: rows (hx << 5 | q)
    const unsigned toff2 = ((unsigned)hx << 8) | lfull; // store side and round-2 twiddles: p = tid >> 5
    // round 1 (stage 8 + b, table index (rr << 8) | lfull): parked in LDS in the DIT packing
    if (hx == 0) {
        int slot = 0;
        auto park = [&](unsigned uniform_idx) {
            uint2 w = (twf + uniform_idx)[lfull];
            to_dit_packing(w.x, w.y);
            tw1[32 * slot++ + l] = w;
        };
        park((1u << 8) - 1u);
        park((1u << 9) - 1u);
        for (int rr = 0; rr < 2; ++rr) park((1u << 10) - 1u + ((unsigned)rr << 8));
        for (int rr = 0; rr < 4; ++rr) park((1u << 11) - 1u + ((unsigned)rr << 8));
        if constexpr (RB == 5)
            for (int rr = 0; rr < 8; ++rr) park((1u << 12) - 1u + ((unsigned)rr << 8));
    }
    u32 *const wr_base = lds + ROW2P * (hx << 5) + l;  // write side: row (hx << 5) + q
    const u32 *const rd_base = lds + ROW2P * hx + l;   // read side: row (j << RB) + p
    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    for (size_t frame = grp; frame < nframes; frame += groups) {
        const u32 *src = scr + (frame << L);
        u32 *dst = out + (frame << L);
        unsigned toff_l = toff, toff2_l = toff2;
        asm volatile("" : "+v"(toff_l), "+v"(toff2_l));
        u32 v[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = INTFFT_LD(at32(src + ((size_t)q << 8), toff_l)); // (plain loads: 325 vs 332 Gsample/s)
        bool fast = false;
        {
            u32 acc = 0;
#pragma unroll
            for (int q = 0; q < 32; ++q) acc |= v[q] + sl.gbias;
            const bool bad = block_any(vote_flags, vote_phase, (acc & sl.gmask) != 0); // also orders the previous frame's LDS reads
            fast = FAST_OK && !bad;
        }
        const int rnd = FAST_OK ? 0 : sl.round; // RNDMODE = 1 on the exact-path instantiation (2: narrow data)
        if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd); // (first pass after k_mid_q1: already w-bit values; harmless)
        {
            RoundTwQ t1;
            u32 wa16[8], wb16[8];
            int slot = 0;
            auto get = [&](u32 &wa, u32 &wb) {
                const uint2 w = tw1[32 * slot++ + l];
                wa = w.x;
                wb = w.y;
            };
            get(t1.wa1[0], t1.wb1[0]);
            get(t1.wa2[0], t1.wb2[0]);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) get(t1.wa4[rr], t1.wb4[rr]);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) get(t1.wa8[rr], t1.wb8[rr]);
            if constexpr (RB == 5) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) get(wa16[rr], wb16[rr]);
            }
            if (fast) {
                dit_round_q<FAST_OK, 0>(v, t1, sl);
                dit_round_q<FAST_OK, 16>(v, t1, sl);
                if constexpr (RB == 5) dit_top16<FAST_OK>(v, wa16, wb16, sl);
            } else if (rnd == 1) {
                dit_round_q<false, 0, 1>(v, t1, sl);
                dit_round_q<false, 16, 1>(v, t1, sl);
                if constexpr (RB == 5) dit_top16<false, 1>(v, wa16, wb16, sl);
            } else if (rnd == 2) {
                dit_round_q<false, 0, 2>(v, t1, sl);
                dit_round_q<false, 16, 2>(v, t1, sl);
                if constexpr (RB == 5) dit_top16<false, 2>(v, wa16, wb16, sl);
            } else {
                dit_round_q<false, 0>(v, t1, sl);
                dit_round_q<false, 16>(v, t1, sl);
                if constexpr (RB == 5) dit_top16<false>(v, wa16, wb16, sl);
            }
        }
        // round 2's per-thread twiddles: stage L-5+b, index (jj << (L-5)) | (p << 8) | lfull
        RoundTwQ t2;
        u32 wa16[8], wb16[8];
        unsigned twb = toff2_l * 8u;
        asm volatile("" : "+v"(twb));
        auto ld = [&](unsigned uniform_idx, u32 &wa, u32 &wb) {
            const uint2 w = ld2_at32b(twf + uniform_idx, twb);
            wa = w.x;
            wb = w.y;
        };
        ld((1u << (L - 5)) - 1u, t2.wa1[0], t2.wb1[0]);
        ld((1u << (L - 4)) - 1u, t2.wa2[0], t2.wb2[0]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) ld((1u << (L - 3)) - 1u + ((unsigned)jj << (L - 5)), t2.wa4[jj], t2.wb4[jj]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ld((1u << (L - 2)) - 1u + ((unsigned)jj << (L - 5)), t2.wa8[jj], t2.wb8[jj]);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) ld((1u << (L - 1)) - 1u + ((unsigned)jj << (L - 5)), wa16[jj], wb16[jj]);
#pragma unroll
        for (int q = 0; q < 32; ++q) wr_base[ROW2P * q] = v[q];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = rd_base[ROW2P * (j << RB)];
        to_dit_packing(t2.wa1[0], t2.wb1[0]);
        to_dit_packing(t2.wa2[0], t2.wb2[0]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) to_dit_packing(t2.wa4[jj], t2.wb4[jj]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) to_dit_packing(t2.wa8[jj], t2.wb8[jj]);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) to_dit_packing(wa16[jj], wb16[jj]);
        if (fast) {
            dit_round_q<FAST_OK, 0>(v, t2, sl);
            dit_round_q<FAST_OK, 16>(v, t2, sl);
            dit_top16<FAST_OK>(v, wa16, wb16, sl);
        } else if (rnd == 1) {
            dit_round_q<false, 0, 1>(v, t2, sl);
            dit_round_q<false, 16, 1>(v, t2, sl);
            dit_top16<false, 1>(v, wa16, wb16, sl);
        } else if (rnd == 2) {
            dit_round_q<false, 0, 2>(v, t2, sl);
            dit_round_q<false, 16, 2>(v, t2, sl);
            dit_top16<false, 2>(v, wa16, wb16, sl);
        } else {
            dit_round_q<false, 0>(v, t2, sl);
            dit_round_q<false, 16>(v, t2, sl);
            dit_top16<false>(v, wa16, wb16, sl);
        }
        if (halves) { // memory index of (n(L-1), rest) = rest * 2 + n(L-1)
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            v2u *d2 = reinterpret_cast<v2u *>(dst);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v2u w = {v[j], v[j + 16]};
                __builtin_nontemporal_store(w, at32(d2 + ((size_t)j << (L - 5)), toff2_l));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) __builtin_nontemporal_store(v[j], at32(dst + ((size_t)j << (L - 5)), toff2_l));
        }
    }
    (void)T;
}

// the quarter-turn relation these kernels rely on, checked on the plan's generated tables (host copy)
bool big2p_tables_ok(int log2n, const int2 *h_tw, int twd)
{
    for (int s = 8; s < log2n; ++s) {
        const int2 *t = h_tw + ((size_t)1 << s) - 1;
        const size_t h = (size_t)1 << (s - 1);
        for (size_t k = 0; k < h; ++k) {
            const int neg = (int)(((long long)(-t[k].x) << (64 - twd)) >> (64 - twd));
            if (t[k + h].x != t[k].y || t[k + h].y != neg) return false;
            if (t[k].x == -32768 || t[k].y == -32768) return false; // k_big2p_q negates packed 16-bit twiddle halves
        }
    }
    return true;
}

bool big2p_supported(int log2n) { return (log2n == 17 || log2n == 18) && !diag_env("INTFFT_NO_BIG2P"); }

hipError_t launch_big2p_a(int log2n, bool fx, const u32 *pin, u32 *scr, const uint2 *tw16f, size_t nframes, const Slice &sl,
                          int halves, hipStream_t stream)
{
#define INTFFT_2P_LAUNCH(LL, FX, RD)                                                                                      \
    {                                                                                                                     \
        constexpr int TT = 32 << (LL - 13);                                                                               \
        const size_t ldsb = ((size_t)TT * ROW2P + 1) * sizeof(u32) + (LL == 18 ? 16 : 8) * 32 * sizeof(uint2);             \
        allow_max_lds(kptr(k_big2p_a<LL, FX, RD>));                                                                           \
        const size_t per_cu = LL == 17 ? 2 : 1, cap = (size_t)device_cus() * per_cu / 8;                                  \
        const unsigned groups = (unsigned)(nframes < cap ? nframes : (cap ? cap : 1));                                    \
        hipLaunchKernelGGL((k_big2p_a<LL, FX, RD>), dim3(8u * groups), dim3(TT), ldsb, stream, pin, scr, tw16f, nframes, groups, sl, halves); \
    }
    if (log2n == 17) {
        if (sl.round == 1) INTFFT_2P_LAUNCH(17, false, 1) else if (sl.round == 2) INTFFT_2P_LAUNCH(17, false, 2)
        else if (fx) INTFFT_2P_LAUNCH(17, true, 0) else INTFFT_2P_LAUNCH(17, false, 0)
    } else {
        if (sl.round == 1) INTFFT_2P_LAUNCH(18, false, 1) else if (sl.round == 2) INTFFT_2P_LAUNCH(18, false, 2)
        else if (fx) INTFFT_2P_LAUNCH(18, true, 0) else INTFFT_2P_LAUNCH(18, false, 0)
    }
#undef INTFFT_2P_LAUNCH
    return hipGetLastError();
}

hipError_t launch_big2p_q(int log2n, bool fx, const u32 *scr, u32 *pout, const uint2 *tw16f, size_t nframes, const Slice &sl,
                          int halves, hipStream_t stream)
{
#define INTFFT_2Q_LAUNCH(LL, FX)                                                                                          \
    {                                                                                                                     \
        constexpr int TT = 32 << (LL - 13);                                                                               \
        const size_t ldsb = ((size_t)TT * ROW2P + 1) * sizeof(u32) + (LL == 18 ? 16 : 8) * 32 * sizeof(uint2);             \
        allow_max_lds(kptr(k_big2p_q<LL, FX>));                                                                           \
        const size_t per_cu = LL == 17 ? 2 : 1, cap = (size_t)device_cus() * per_cu / 8;                                  \
        const unsigned groups = (unsigned)(nframes < cap ? nframes : (cap ? cap : 1));                                    \
        hipLaunchKernelGGL((k_big2p_q<LL, FX>), dim3(8u * groups), dim3(TT), ldsb, stream, scr, pout, tw16f, nframes, groups, sl, halves); \
    }
    if (log2n == 17) {
        if (fx) INTFFT_2Q_LAUNCH(17, true) else INTFFT_2Q_LAUNCH(17, false)
    } else {
        if (fx) INTFFT_2Q_LAUNCH(18, true) else INTFFT_2Q_LAUNCH(18, false)
    }
#undef INTFFT_2Q_LAUNCH
    return hipGetLastError();
}

} // namespace intfft
