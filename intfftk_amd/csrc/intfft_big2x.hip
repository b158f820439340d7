// intfft_big2x.hip -- TWO-pass plans for N = 2^19 and 2^20 (BASELINE config 4 is N = 2^20, Taylor twiddles): int_fftNk,
// DATA_WIDTH = 16 (or 9 .. 15 in int16 containers), TWDL_WIDTH <= 16, scaled-truncate, natural / HALVES order in -> natural / BITREV
// order out, and the inverse mirrors (src/vhdl/fft/int_fftNk.vhd:184-342; twiddles of STAGE >= 11 from row_twiddle_tay.vhd:123-268 via k_twiddle_stage).
// Same packed arithmetic as intfft_fast1024.hip.  N = 2^L = 2^(L-10) rows x 1024 columns:
//
//   pass A  k_big2x_a<L>  stages L-1..10 down the columns; user array -> plan scratch
//   pass B  k_big2x_b<L>  stages 9..0 along every 1024-point row + the bit-reversed (natural-order) store, or the core-order (BITREV) one;
//                         scratch -> user array
//
// A ten-stage pass on 128-byte rows needs a 1024 x 32 tile = 128 KiB: one workgroup per CU, nothing to overlap its load / compute /
// store phases with (measured: 3.8 TB/s on the 1024-row form of k_big2p_a).  Both passes here work on
// HALF lines instead -- 1024 rows x 16 columns (64-byte row pieces), 512 threads x 32 registers, 68 KiB of LDS: two workgroups
// per CU -- and the two workgroups that share every line of a tile are blocks b and b + 8: block b runs on XCD b % 8, so the
// pair sits on ONE XCD at the same time and the line is fetched into (written back from) that L2 once.  tools/xcdbench.hip,
// part 3: 64-byte pieces cost 3.1-3.7 TB/s when their halves are handled by unrelated workgroups, 3.8-4.6 with this pairing
// (strided read side) and 5.1-5.2 (strided write side) -- the rates of the 128-byte tiles, at twice the occupancy.
// The pairing is for speed only: correctness never depends on where a block runs.
//
// Scratch layout (per frame; n = row rho * 1024 + column, rho = (k = n(L-1)..n(L-4), rest = n(L-5)..n10), rest = (hi, q = n14..n10),
// column = (c = n9..n4, l = n3..n0)):   [q][c][hi][k][l]
//   pass A round 2 holds (thread = (n(L-1)..n15, l), regs = q): every register store of a workgroup is ONE 2 KiB run;
//   pass B's tile = the 16 rows k of one `rest` x all columns: 64 runs of 1 KiB; its bit-reversed store writes 64-byte pieces
//   whose other half belongs to the partner block (`rest` with its top bit flipped).
// Values travel between the passes as in the other multi-pass kernels: multiplier outputs pre-shifted (Y >> 1), so an element's
// kind after pass A is n10 = q bit 0, which is tile-uniform in pass B.
// multi-pass kernels: non-temporal loads measure 4-14 % faster here (the single-pass kernels gain 4-30 % from PLAIN loads): intfft_device.hpp
#define INTFFT_NT_LOADS 1
#include "intfft_pk16.hpp"

// memory policy of the two hot passes (round-4 A/B variants: tools/build_variant.sh <name> intfft_big2x.hip -DINTFFT_2XA_PLAIN ...; the
// counters per variant are in DESIGN.md section 4.2d).  Shipped: non-temporal loads of the user array in pass A, non-temporal stores of
// the user array in pass B.
#ifdef INTFFT_2XA_PLAIN
#define INTFFT_2XA_LD(p) (*(p))
#else
#define INTFFT_2XA_LD(p) INTFFT_LD(p)
#endif
#ifdef INTFFT_2XB_PLAIN
#define INTFFT_2XB_ST(x, p) (*(p) = (x))
#else
#define INTFFT_2XB_ST(x, p) __builtin_nontemporal_store(x, p)
#endif

#include <cstdlib>

namespace intfft {

constexpr int ROWX = 17; // pass A: LDS row stride in dwords (16 columns + 1)
constexpr int ROWY = 33; // pass B: LDS row stride in dwords (32 columns + 1)

// wave-uniform twiddles of pass B's second round (kernel argument -> SGPRs): STAGE 4, 3, 2 (STAGE 1, 0 are multiplier-free)
struct Round5Consts {
    u32 wa4[16], wb4[16]; // table index q & 15
    u32 wa3[8], wb3[8];   // q & 7
    u32 wa2[4], wb2[4];   // q & 3
};

__device__ __forceinline__ constexpr int rev5c(int r) { return ((r & 1) << 4) | ((r & 2) << 2) | (r & 4) | ((r & 8) >> 2) | ((r & 16) >> 4); }

// ---- pass A: stages L-1..10 ---------------------------------------------------------------------------------------------------
//   tile    2^(L-10) rows (stride 1024 samples = 4 KiB) x 16 consecutive columns; 64 column chunks per frame
//   round 1 thread = (hx = n(L-6)..n10, l = n3..n0), regs j = n(L-1)..n(L-5): stages L-1..L-5
//   LDS     row (j << RB | hx), column l                          (RB = L - 15 = 4 or 5 stages in round 2)
//   round 2 L = 20: thread = (n19..n15, l), regs = n14..n10: stages 14..10
//           L = 19: thread = (n18..n15, l), regs = (n14, n13..n10): two independent 4-stage rounds 13..10 (n14 rides along)
// Twiddles as in k_big2p_a: quarter-turn sharing; round 2's set depends on the column only and is parked in LDS, round 1's is
// per thread and re-read from the L2-resident table in every frame.
// CB = 5 (round 5, N = 2^19 only): FULL-line tiles -- 512 rows x 32 columns (128-byte row pieces) still fit 68 KiB, so two workgroups per CU read whole lines
// with no partner to meet in the L2 (tools/tilebench.hip, profiles/r05_tilebench_*.txt: 5.05 TB/s as a copy against 3.15-3.5 for the paired half lines).  At
// N = 2^20 the same tile is 128 KiB = one workgroup per CU (3.8 TB/s with its arithmetic, round 2): half lines (CB = 4) stay.
template <int L, bool FAST_OK, int ROUND = 0, int CB = 4> // ROUND: RNDMODE = 1 (2: on narrow data) -- its own instantiation, exact extraction only
__global__ __launch_bounds__((1 << CB) << (L - 15)) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_big2x_a(const u32 *in, u32 *scr, const uint2 *__restrict__ twf,
                                                                                                        size_t nframes, unsigned groups, const Slice sl, int halves)
{
    static_assert(L == 19 || L == 20, "9 or 10 stages");
    static_assert(!ROUND || !FAST_OK, "round mode: exact extraction");
    static_assert(CB == 4 || (CB == 5 && L == 19), "full-line tiles fit 68 KiB at 512 rows only");
    constexpr int RB = L - 15;
    constexpr int C = 1 << CB;        // columns of the tile
    constexpr int ROWX = C + 1;       // LDS row stride in dwords (shadows the 17 of the half-line kernels)
    constexpr int T = C << RB;
    extern __shared__ u32 lds[]; // (32 << RB) rows x ROWX, then the round-2 twiddles: 8 (RB = 4) / 16 slots x C columns of {wa, wb}
    uint2 *const tw2 = reinterpret_cast<uint2 *>(lds + (32 << RB) * ROWX + (((32 << RB) * ROWX) & 1));
    const int tid = threadIdx.x, l = tid & (C - 1), hx = tid >> CB;
    // blocks b and b + 8 (same XCD, same time) take the two 64-byte halves of the same lines: chunk = (column group g, part)
    // INTFFT_2XA_MAP (round-5 A/B variants, tools/build_variant.sh: which tiles an XCD holds at the same time; profiles/r05_c4_variants.md):
    //   0 (shipped)  XCD s takes the column lines s, s + 8, s + 16, s + 24 of every frame
    //   1            XCD s takes whole frames s, s + 8, ... (all 32 column lines of a frame at once; needs groups % 8 == 0)
    //   2            as 0 with the lines rotated by the frame number (an XCD sees every line offset across its 8 concurrent frames)
#ifndef INTFFT_2XA_MAP
#define INTFFT_2XA_MAP 0
#endif
    const unsigned slot = blockIdx.x & 7u, part = (blockIdx.x >> 3) & 1u;
    unsigned chunk, grp;
    if constexpr (CB == 5) { // full lines: no partner; neighbouring blocks = neighbouring column groups of one frame
        chunk = blockIdx.x & 31u, grp = blockIdx.x >> 5;
        (void)slot, (void)part;
    } else {
#if INTFFT_2XA_MAP == 1
        const unsigned jm = blockIdx.x >> 4;
        chunk = (jm & 31u) * 2u + part, grp = (jm >> 5) * 8u + slot;
#elif INTFFT_2XA_MAP == 2
        const unsigned G = (blockIdx.x >> 4) * 8u + slot;
        grp = G >> 5;
        chunk = ((G + grp) & 31u) * 2u + part;
#else
        const unsigned G = (blockIdx.x >> 4) * 8u + slot;
        chunk = (G & 31u) * 2u + part, grp = G >> 5;
#endif
    }
    const unsigned lfull = chunk * C + l;               // n9..n0
    const unsigned toff = ((unsigned)hx << 10) | lfull; // this thread's offset inside a block of rows (n(9+RB)..n0)
    auto ld = [&](unsigned uniform_idx, unsigned thread_boff, u32 &wa, u32 &wb) { // thread_boff: BYTE offset of the thread's entry
#ifdef INTFFT_2XA_ABL_TW /* ablation (results wrong by construction): every twiddle read lands in one 8 KiB window -- what do the 8 MiB tables cost? */
        thread_boff &= 0x1FF8u;
        uniform_idx &= 1023u;
#endif
        const uint2 w = ld2_at32b(twf + uniform_idx, thread_boff);
        wa = w.x;
        wb = w.y;
    };
    u32 wa16[8], wb16[8];
    RoundTwQ t1;
    if (hx == 0) { // round 2 (stage 10 + b, table index (rr << 10) | lfull): parked per column
        int s = 0;
        auto park = [&](unsigned uniform_idx) { tw2[C * s++ + l] = (twf + uniform_idx)[lfull]; };
        if constexpr (RB == 5)
            for (int rr = 0; rr < 8; ++rr) park((1u << 14) - 1u + ((unsigned)rr << 10));
        for (int rr = 0; rr < 4; ++rr) park((1u << 13) - 1u + ((unsigned)rr << 10));
        for (int rr = 0; rr < 2; ++rr) park((1u << 12) - 1u + ((unsigned)rr << 10));
        park((1u << 11) - 1u);
        park((1u << 10) - 1u);
    }
    // round 1: reg bit b <-> stage L-5+b; index of twiddle (stage s, low reg bits jj) = ((jj << RB | hx) << 10) | lfull
    auto round1_tw = [&](unsigned to) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) ld((1u << (L - 1)) - 1u + ((unsigned)jj << (RB + 10)), to, wa16[jj], wb16[jj]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ld((1u << (L - 2)) - 1u + ((unsigned)jj << (RB + 10)), to, t1.wa8[jj], t1.wb8[jj]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) ld((1u << (L - 3)) - 1u + ((unsigned)jj << (RB + 10)), to, t1.wa4[jj], t1.wb4[jj]);
        ld((1u << (L - 4)) - 1u, to, t1.wa2[0], t1.wb2[0]);
        ld((1u << (L - 5)) - 1u, to, t1.wa1[0], t1.wb1[0]);
    };
    auto round2_tw = [&](u32(&wa2t)[8], u32(&wb2t)[8], RoundTwQ &t2) {
        int s = 0;
        auto get = [&](u32 &wa, u32 &wb) {
            const uint2 w = tw2[C * s++ + l];
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
    u32 *const wr_base = lds + ROWX * hx + l;              // transpose, write side: row (j << RB) + hx
    const u32 *const rd_base = lds + ROWX * (hx << 5) + l; // read side: row (jx << 5) + q, jx = tid >> 4
    // store side: scratch [q][c][hi][k][l], thread = (jx = n(L-1)..n15, l): k = jx >> (RB - 4), hi = the low RB - 4 bits of jx
    // (c = n9..n4 = the tile's column group(s): chunk for 16-column tiles, 2 chunk + (l >> 4) for 32-column ones)
    const unsigned toff2 = (((chunk << (CB - 4)) + ((unsigned)l >> 4)) << (L - 11)) | (((unsigned)hx & ((1u << (RB - 4)) - 1u)) << 8) | (((unsigned)hx >> (RB - 4)) << 4) | ((unsigned)l & 15u);
    const v2s none = {0, 0};
    const short s2 = (short)(1 - (hx & 1)); // L = 20, round 2: the kind of its inputs is n15 = bit 0 of the new thread index
    const v2s sh2 = {s2, s2};
    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);

    for (size_t frame = grp; frame < nframes; frame += groups) {
        const u32 *src = in + (frame << L); // wave-uniform
        u32 *dst = scr + (frame << L);
        unsigned toff_l = toff, toff2_l = toff2; // opaque copies: see k_big2p_a (LICM would hoist 40+ VGPRs of addresses)
        asm volatile("" : "+v"(toff_l), "+v"(toff2_l));
        u32 v[32];
        __builtin_amdgcn_s_setprio(3); // the tile's 32 + 16 loads go out ahead of the other workgroup's arithmetic (C4 300-302 -> 305 Gsample/s)
        if (halves) { // HALVES order in: memory index = 2 * (n without n(L-1)) + n(L-1): registers j and j + 16 are one 8-byte load
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            const v2u *sh = reinterpret_cast<const v2u *>(src);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v2u w = INTFFT_2XA_LD(at32(sh + ((size_t)j << (RB + 10)), toff_l));
                v[j] = w.x;
                v[j + 16] = w.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = INTFFT_2XA_LD(at32(src + ((size_t)j << (RB + 10)), toff_l)); // regs = n(L-1)..n(L-5)
        }
        unsigned twb = toff_l * 8u;
        asm volatile("" : "+v"(twb));
        round1_tw(twb);
        __builtin_amdgcn_s_setprio(0);
        // guard-bit vote of the tile (closed under stages L-1..10); the barrier also orders the previous frame's LDS reads
        bool fast = false;
        {
            u32 acc = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) acc |= v[j] + sl.gbias;
            const bool bad = block_any(vote_flags, vote_phase, (acc & sl.gmask) != 0);
            fast = FAST_OK && !bad;
        }
        if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd); // DATA_WIDTH < 16, exact path: containers wrapped to w bits
#define INTFFT_2X_ROUND1(FX)                                                                                  \
    {                                                                                                         \
        dif_top16<FX, 0, false, ROUND>(v, wa16, wb16, sl, none);                                              \
        dif_round_q<FX, 0, 0, false, 4, ROUND>(v, t1, sl, none);                                              \
        dif_round_q<FX, 16, 0xF, false, 4, ROUND>(v, t1, sl, none);                                           \
    }
        if (fast) INTFFT_2X_ROUND1(FAST_OK)
        else INTFFT_2X_ROUND1(false)
#undef INTFFT_2X_ROUND1
#pragma unroll
        for (int j = 0; j < 32; ++j) wr_base[ROWX * (j << RB)] = v[j];
        __syncthreads();
        u32 wa2t[8], wb2t[8]; // L = 20: top stage of round 2
        RoundTwQ t2;
        round2_tw(wa2t, wb2t, t2);
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = rd_base[ROWX * q];
        if constexpr (RB == 4) { // two independent 4-stage rounds 13..10; the kind of their inputs is n14 = q bit 4
            if (fast) {
                dif_round_q<FAST_OK, 0, 0, false>(v, t2, sl, none);
                dif_round_q<FAST_OK, 16, 0xF, false>(v, t2, sl, none);
            } else {
                dif_round_q<false, 0, 0, false, 4, ROUND>(v, t2, sl, none);
                dif_round_q<false, 16, 0xF, false, 4, ROUND>(v, t2, sl, none);
            }
        } else { // stages 14..10; the kind of the inputs is n15 = jx bit 0 (a thread bit)
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
        for (int q = 0; q < 32; ++q) *at32(dst + ((size_t)q << (L - 5)), toff2_l) = v[q]; // [q][c][hi][k][l]: 2 KiB (L = 19: 1 KiB) per register
    }
    (void)T;
}

// ---- pass A at N = 2^20 on FULL lines, one pipelined workgroup per CU (round 5; an EXPERIMENT behind INTFFT_2XA_FULL20, not the shipped path) ---------------
// Result (profiles/r05_c4_variants.md): bit-exact, 141 us per 2^26 samples against the paired half lines' 145 on one stream -- no faster -- and it cannot
// share a CU with pass B of the other stream: C4 292 against 313 Gsample/s on the same box.  Kept so that the table regenerates.
// A 1024-row x 32-column tile (128-byte row pieces) is 128 KiB: one workgroup per CU, which -- load, ten stages, store in sequence -- ran at 3.8 TB/s in round 2
// and lost to the XCD-paired half lines (3.7 TB/s at two workgroups per CU, beside which pass B of the other stream can run).  What k_rows2k_tr showed
// (5.2 TB/s from one 135 KiB workgroup per CU with its eleven stages): the NEXT tile's 32 loads can be issued right behind the transpose's LDS writes -- the
// data registers are free from there to the next vote -- and fly during round 2 and the stores.  Same arithmetic, twiddles, votes and scratch layout as
// k_big2x_a<20> (thread = (hx, l = n4..n0): 1024 threads); round 2's twiddles are read from their LDS slots just before the stage that uses them (8 + 8
// pairs instead of 16 live beside the prefetched registers); the per-column set is re-parked per tile behind the vote's barrier.
template <bool FAST_OK>
__global__ __launch_bounds__(1024) void k_big2x_a1(const u32 *in, u32 *scr, const uint2 *__restrict__ twf, size_t nframes, const Slice sl)
{
    constexpr int L = 20, RB = 5, C = 32, ROWX = C + 1;
    extern __shared__ u32 lds[]; // 1024 rows x ROWX, then the round-2 twiddles: 16 slots x 32 columns of {wa, wb}
    uint2 *const tw2 = reinterpret_cast<uint2 *>(lds + 1024 * ROWX);
    const int tid = threadIdx.x, l = tid & (C - 1), hx = tid >> 5;
    u32 *const wr_base = lds + ROWX * hx + l;              // transpose, write side: row (j << RB) + hx
    const u32 *const rd_base = lds + ROWX * (hx << 5) + l; // read side: row (jx << 5) + q, jx = tid >> 5
    const v2s none = {0, 0};
    const short s2 = (short)(1 - (hx & 1)); // round 2: the kind of its inputs is n15 = bit 0 of the new thread index
    const v2s sh2 = {s2, s2};
    __shared__ __attribute__((aligned(256))) u32 vote_flags[64];
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    const size_t ntiles = nframes << 5; // 32 column groups per frame; neighbouring tiles = neighbouring column groups of one frame
    size_t t = blockIdx.x;
    bool have = t < ntiles;
    u32 v[32];
    auto load_tile = [&](size_t tt) {
        const u32 *src = in + ((tt >> 5) << L); // wave-uniform
        unsigned toff = ((unsigned)hx << 10) | (((unsigned)tt & 31u) * C + (unsigned)l);
        asm volatile("" : "+v"(toff));
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = INTFFT_2XA_LD(at32(src + ((size_t)j << (RB + 10)), toff)); // regs = n19..n15
    };
    if (have) load_tile(t);
    while (have) {
        const size_t ct = t;
        t += gridDim.x;
        const bool have_next = t < ntiles;
        const unsigned chunk = (unsigned)ct & 31u;
        const unsigned lfull = chunk * C + (unsigned)l;      // n9..n0
        const unsigned toff = ((unsigned)hx << 10) | lfull;
        // round 1's per-thread twiddles: reg bit b <-> stage L-5+b; index of (stage s, low reg bits jj) = ((jj << RB | hx) << 10) | lfull
        u32 wa16[8], wb16[8];
        RoundTwQ t1;
        {
            unsigned twb = toff * 8u;
            asm volatile("" : "+v"(twb));
            auto ld = [&](unsigned uniform_idx, u32 &wa, u32 &wb) {
                const uint2 w = ld2_at32b(twf + uniform_idx, twb);
                wa = w.x;
                wb = w.y;
            };
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) ld((1u << (L - 1)) - 1u + ((unsigned)jj << (RB + 10)), wa16[jj], wb16[jj]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) ld((1u << (L - 2)) - 1u + ((unsigned)jj << (RB + 10)), t1.wa8[jj], t1.wb8[jj]);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) ld((1u << (L - 3)) - 1u + ((unsigned)jj << (RB + 10)), t1.wa4[jj], t1.wb4[jj]);
            ld((1u << (L - 4)) - 1u, t1.wa2[0], t1.wb2[0]);
            ld((1u << (L - 5)) - 1u, t1.wa1[0], t1.wb1[0]);
        }
        bool fast = false;
        {
            u32 acc = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) acc |= v[j] + sl.gbias;
            const bool bad = block_any(vote_flags, vote_phase, (acc & sl.gmask) != 0); // also orders the previous tile's LDS reads (rows AND twiddle slots)
            fast = FAST_OK && !bad;
        }
        if (hx == 0) { // round 2's twiddles of this tile's columns (stage 10 + b, table index (rr << 10) | lfull): parked per column
            int sidx = 0;
            auto park = [&](unsigned uniform_idx) { tw2[C * sidx++ + l] = (twf + uniform_idx)[lfull]; };
            for (int rr = 0; rr < 8; ++rr) park((1u << 14) - 1u + ((unsigned)rr << 10));
            for (int rr = 0; rr < 4; ++rr) park((1u << 13) - 1u + ((unsigned)rr << 10));
            for (int rr = 0; rr < 2; ++rr) park((1u << 12) - 1u + ((unsigned)rr << 10));
            park((1u << 11) - 1u);
            park((1u << 10) - 1u);
        }
        if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd);
#define INTFFT_2XA1_ROUND1(FX)                                       \
    {                                                                \
        dif_top16<FX, 0, false>(v, wa16, wb16, sl, none);            \
        dif_round_q<FX, 0, 0, false, 4>(v, t1, sl, none);            \
        dif_round_q<FX, 16, 0xF, false, 4>(v, t1, sl, none);         \
    }
        if (fast) INTFFT_2XA1_ROUND1(FAST_OK)
        else INTFFT_2XA1_ROUND1(false)
#undef INTFFT_2XA1_ROUND1
#pragma unroll
        for (int j = 0; j < 32; ++j) wr_base[ROWX * (j << RB)] = v[j];
        asm volatile("" ::: "memory");
        if (have_next) load_tile(t); // flies during round 2 and the stores below (issued behind round 2's top stage instead: 273 against 292 Gsample/s)
        __syncthreads();
        u32 w[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) w[q] = rd_base[ROWX * q];
        {
            u32 wa2t[8], wb2t[8];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const uint2 x = tw2[C * rr + l];
                wa2t[rr] = x.x, wb2t[rr] = x.y;
            }
            if (fast) dif_top16<FAST_OK, 0, true>(w, wa2t, wb2t, sl, sh2);
            else dif_top16<false, 0, true>(w, wa2t, wb2t, sl, sh2);
        }
        __builtin_amdgcn_sched_barrier(0); // the lower stages' twiddles are fetched only now (register pressure beside the prefetched tile)
        {
            RoundTwQ t2;
            int sidx = 8;
            auto get = [&](u32 &wa, u32 &wb) {
                const uint2 x = tw2[C * sidx++ + l];
                wa = x.x, wb = x.y;
            };
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) get(t2.wa8[rr], t2.wb8[rr]);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) get(t2.wa4[rr], t2.wb4[rr]);
            get(t2.wa2[0], t2.wb2[0]);
            get(t2.wa1[0], t2.wb1[0]);
            if (fast) {
                dif_round_q<FAST_OK, 0, 0, false>(w, t2, sl, none);
                dif_round_q<FAST_OK, 16, 0xF, false>(w, t2, sl, none);
            } else {
                dif_round_q<false, 0, 0, false, 4>(w, t2, sl, none);
                dif_round_q<false, 16, 0xF, false, 4>(w, t2, sl, none);
            }
        }
        // store side: scratch [q][c][hi][k][l], thread = (jx = n19..n15, l): k = jx >> 1, hi = jx & 1; c = 2 chunk + (l >> 4)
        u32 *dst = scr + ((ct >> 5) << L);
        unsigned toff2 = (((chunk << 1) + ((unsigned)l >> 4)) << (L - 11)) | (((unsigned)hx & 1u) << 8) | (((unsigned)hx >> 1) << 4) | ((unsigned)l & 15u);
        asm volatile("" : "+v"(toff2));
#pragma unroll
        for (int q = 0; q < 32; ++q) *at32(dst + ((size_t)q << (L - 5)), toff2) = w[q];
        have = have_next;
    }
}

// ---- pass B's second round: DIF stages 4..0 on regs = n4..n0, wave-uniform twiddles -------------------------------------------
// inputs: per-thread kind (shv: 0 where the registers already hold X >> 1)
template <bool FASTX, int ROUND = 0> __device__ __forceinline__ void dif_round5_c(u32 (&v)[32], const Round5Consts &c, const Slice &sl, v2s shv)
{
    if constexpr (ROUND != 0) { // RNDMODE = 1 (int_dif2_fly.vhd:167-219): plain values, rhu2 sums, exact extraction; STAGE 1 / 0 in round_stages10
        static_assert(!FASTX, "round mode uses the exact extraction");
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
            const u32 wa[4] = {c.wa4[g], c.wa4[g + 1], c.wa4[g + 2], c.wa4[g + 3]}, wb[4] = {c.wb4[g], c.wb4[g + 1], c.wb4[g + 2], c.wb4[g + 3]};
            group4<ROUND, 0, false, false, true, 0>(v[g], v[g + 16], v[g + 1], v[g + 17], v[g + 2], v[g + 18], v[g + 3], v[g + 19], wa, wb, sl);
        }
        const u32 wa30[4] = {c.wa3[0], c.wa3[1], c.wa3[2], c.wa3[3]}, wb30[4] = {c.wb3[0], c.wb3[1], c.wb3[2], c.wb3[3]};
        const u32 wa31[4] = {c.wa3[4], c.wa3[5], c.wa3[6], c.wa3[7]}, wb31[4] = {c.wb3[4], c.wb3[5], c.wb3[6], c.wb3[7]};
#pragma unroll
        for (int B = 0; B < 32; B += 16) {
            group4<ROUND, 0, false, false, true, 0>(v[B + 0], v[B + 8], v[B + 1], v[B + 9], v[B + 2], v[B + 10], v[B + 3], v[B + 11], wa30, wb30, sl);
            group4<ROUND, 0, false, false, true, 0>(v[B + 4], v[B + 12], v[B + 5], v[B + 13], v[B + 6], v[B + 14], v[B + 7], v[B + 15], wa31, wb31, sl);
        }
#pragma unroll
        for (int B = 0; B < 32; B += 8)
            group4<ROUND, 0, false, false, true, 0>(v[B + 0], v[B + 4], v[B + 1], v[B + 5], v[B + 2], v[B + 6], v[B + 3], v[B + 7], c.wa2, c.wb2, sl);
        round_stages10<32, ROUND == 2>(v, sl);
        (void)shv;
        return;
    }
#pragma unroll
    for (int g = 0; g < 16; g += 4) { // stage 4: pairs (q, q + 16), twiddle index q
        const u32 wa[4] = {c.wa4[g], c.wa4[g + 1], c.wa4[g + 2], c.wa4[g + 3]}, wb[4] = {c.wb4[g], c.wb4[g + 1], c.wb4[g + 2], c.wb4[g + 3]};
        group4<false, FASTX, false, true, true, 0, true>(v[g], v[g + 16], v[g + 1], v[g + 17], v[g + 2], v[g + 18], v[g + 3], v[g + 19], wa, wb, sl, shv);
    }
    const u32 wa30[4] = {c.wa3[0], c.wa3[1], c.wa3[2], c.wa3[3]}, wb30[4] = {c.wb3[0], c.wb3[1], c.wb3[2], c.wb3[3]};
    const u32 wa31[4] = {c.wa3[4], c.wa3[5], c.wa3[6], c.wa3[7]}, wb31[4] = {c.wb3[4], c.wb3[5], c.wb3[6], c.wb3[7]};
    // stage 3: pairs (q, q + 8); kind = q & 16
    group4<false, FASTX, false, true, true, 0>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], wa30, wb30, sl);
    group4<false, FASTX, false, true, true, 0>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], wa31, wb31, sl);
    group4<false, FASTX, false, true, true, 0xF>(v[16], v[24], v[17], v[25], v[18], v[26], v[19], v[27], wa30, wb30, sl);
    group4<false, FASTX, false, true, true, 0xF>(v[20], v[28], v[21], v[29], v[22], v[30], v[23], v[31], wa31, wb31, sl);
#pragma unroll
    for (int B = 0; B < 32; B += 16) { // stage 2: pairs (q, q + 4); kind = q & 8
        group4<false, FASTX, false, true, true, 0>(v[B + 0], v[B + 4], v[B + 1], v[B + 5], v[B + 2], v[B + 6], v[B + 3], v[B + 7], c.wa2, c.wb2, sl);
        group4<false, FASTX, false, true, true, 0xF>(v[B + 8], v[B + 12], v[B + 9], v[B + 13], v[B + 10], v[B + 14], v[B + 11], v[B + 15], c.wa2, c.wb2, sl);
    }
#pragma unroll
    for (int g = 0; g < 32; g += 8) { // stage 1: kind = q & 4
        bfly_triv<false, false>(v[g], v[g + 2]);
        bfly_mj<false, false>(v[g + 1], v[g + 3]);
        bfly_triv<false, true>(v[g + 4], v[g + 6]);
        bfly_mj<false, true>(v[g + 5], v[g + 7]);
    }
#pragma unroll
    for (int g = 0; g < 32; g += 2) bfly_triv<false, false>(v[g], v[g + 1]);
}

// ---- pass B: stages 9..0 on 16 rows x 1024 columns + the bit-reversed store ---------------------------------------------------
//   tile    (frame, rest): rows rho = (k, rest), k = n(L-1)..n(L-4) = 0..15; scratch [q][c][hi][k][l] -> 64 runs of 1 KiB
//   round 1 thread = (n4, k, n3..n0), regs j = n9..n5: stages 9..5; twiddles per thread (they depend on n4..n0 only), frame invariant
//   LDS     row (j << 4 | k), column n4..n0
//   round 2 thread = (jj = n9..n5, kb = rev4(k)), regs q = n4..n0: stages 4..0, wave-uniform twiddles
//   store   X index = rev5(q) << (L-5) | rev5(jj) << (L-10) | rev(rest) << 4 | rev4(k): 16 consecutive lanes = one 64-byte piece;
//           the partner block (`rest` with its top bit flipped -> rev(rest) ^ 1) writes the other half of the line
//   OUT_BR  BITREV order out (the core's own order: memory index = core position, int_fftNk.vhd:184-342 without the reorder buffer):
//           the thread's 32 results are 32 consecutive positions of its row, every row of the tile one 4 KiB run -- staged through LDS
//           once more so that a wave writes 1 KiB runs non-temporally (stored straight from the registers the eight 16-byte pieces
//           of a line come from eight instructions: 257-270 Gsample/s with plain stores, 92 non-temporally; staged: 283-297)
template <int L, bool FAST_OK, bool OUT_BR = false, int ROUND = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_big2x_b(const u32 *scr, u32 *out, const uint2 *__restrict__ twf,
                                                                                             const Round5Consts c, size_t nframes, const Slice sl, int pre_all)
{
    static_assert(L == 19 || L == 20, "rows of 1024 points");
    static_assert(!ROUND || !FAST_OK, "round mode: exact extraction");
    constexpr int RL = L - 14; // bits of `rest`
    extern __shared__ u32 lds[];
    const int tid = threadIdx.x;
    const int m = ((tid >> 8) << 4) | (tid & 15), k = (tid >> 4) & 15;
    u32 wa16[8], wb16[8];
    RoundTwQ t1;
    {
        auto ld = [&](unsigned idx, u32 &wa, u32 &wb) {
            const uint2 w = twf[idx + (unsigned)m];
            wa = w.x;
            wb = w.y;
        };
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) ld(511u + ((unsigned)jj << 5), wa16[jj], wb16[jj]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ld(255u + ((unsigned)jj << 5), t1.wa8[jj], t1.wb8[jj]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) ld(127u + ((unsigned)jj << 5), t1.wa4[jj], t1.wb4[jj]);
        ld(63u, t1.wa2[0], t1.wb2[0]);
        ld(31u, t1.wa1[0], t1.wb1[0]);
    }
    const int jj = tid >> 4, kb = tid & 15;
    const int krow = ((kb & 1) << 3) | ((kb & 2) << 1) | ((kb & 4) >> 1) | ((kb & 8) >> 3);
    u32 *const wr_base = lds + ROWY * k + m;                       // row (j << 4) + k
    const u32 *const rd_base = lds + ROWY * ((jj << 4) | krow);    // row (jj << 4) + k, k = rev4(kb)
    const unsigned toff = (((unsigned)tid >> 8) << (L - 11)) | ((unsigned)tid & 255u);
    const unsigned rjj = __brev((unsigned)jj) >> 27;
    const unsigned toff2 = (rjj << (L - 10)) | (unsigned)kb;
    const short s5 = (short)(1 - (jj & 1)); // round 2: kind = n5
    const v2s sh5 = {s5, s5};
    const v2s none = {0, 0};
    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    // ONE block finishes both `rest` partners (the two 64-byte halves of every output line), one after the other: round 3 gave them to blocks
    // b and b + 8 (same XCD, same time), whose non-temporal half-line stores then left the L2 as 1.20 x the bytes (TCC_EA0_WRREQ, also in the
    // store-only skeleton tools/reqbench.hip: 1.29 x); from one block in sequence 1.02 x at the same or a slightly better rate
    for (size_t t = blockIdx.x;; t += gridDim.x) {
        const size_t G = t;
        const size_t frame = G >> (RL - 1);
        if (frame >= nframes) break;
        for (unsigned part = 0; part < 2; ++part) {
        const unsigned rest = (part << (RL - 1)) | ((unsigned)G & ((1u << (RL - 1)) - 1u));
        const unsigned q0 = rest & 31u, hi = rest >> 5;
        const u32 *src = scr + (frame << L) + ((size_t)q0 << (L - 5)) + (hi << 8); // wave-uniform
        u32 *dst = out + (frame << L) + ((__brev(rest) >> (32 - RL)) << 4);
        unsigned toff_l = toff, toff2_l = toff2;
        asm volatile("" : "+v"(toff_l), "+v"(toff2_l));
        u32 v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = INTFFT_LD(at32(src + ((size_t)j << (L - 10)), toff_l)); // c = (j, n4)
        // kind of this tile's inputs = n10 = q0 bit 0 (pass A left Y >> 1 there); vote on the tile's own inputs
        const unsigned k10 = pre_all ? 1u : (q0 & 1u); // (pre_all: the 2-D scheme's column pass left Y >> 1 everywhere)
        const short sa = (short)(1 - (int)k10);
        const v2s sh_a = {sa, sa};
        bool fast = false;
        {
            const u32 addc = k10 ? sl.gbias1 : sl.gbias, maskc = k10 ? sl.gmask1 : sl.gmask;
            u32 acc = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) acc |= v[j] + addc;
            const bool bad = block_any(vote_flags, vote_phase, (acc & maskc) != 0); // also orders the previous tile's LDS reads
            fast = FAST_OK && !bad;
        }
        if (fast) {
            dif_top16<FAST_OK, 0, true>(v, wa16, wb16, sl, sh_a);
            dif_round_q<FAST_OK, 0, 0, false>(v, t1, sl, none);
            dif_round_q<FAST_OK, 16, 0xF, false>(v, t1, sl, none);
        } else {
            dif_top16<false, 0, true, ROUND>(v, wa16, wb16, sl, sh_a);
            dif_round_q<false, 0, 0, false, 4, ROUND>(v, t1, sl, none);
            dif_round_q<false, 16, 0xF, false, 4, ROUND>(v, t1, sl, none);
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) wr_base[ROWY * (j << 4)] = v[j];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = rd_base[q];
        if (fast) dif_round5_c<FAST_OK>(v, c, sl, sh5);
        else dif_round5_c<false, ROUND>(v, c, sl, sh5);
        if constexpr (OUT_BR) { // position = ((k << RL | rest) << 10) | (jj << 5) | q: the tile is 16 rows of 4 KiB
            // through LDS once more (every thread rewrites the row it has just read), so that a wave writes 1 KiB runs non-temporally
            u32 *const own = lds + ROWY * ((jj << 4) | krow);
#pragma unroll
            for (int q = 0; q < 32; ++q) own[q] = v[q];
            __syncthreads();
            typedef u32 v4u __attribute__((ext_vector_type(4)));
            // flat [k][column] index of the tile: e = (512 i + tid) * 4 -> k = 2 i + (tid >> 8), column = 4 (tid & 255): the i part is wave-uniform
            // (at32: a uniform pointer + a 32-bit thread offset; as 64-bit per-access addresses these eight stores cost the kernel ten spilled VGPRs)
            unsigned tl = (unsigned)tid;
            asm volatile("" : "+v"(tl));
            const unsigned col = (tl & 255u) * 4u, k0 = tl >> 8;
            const u32 *sp0 = lds + ROWY * (((col >> 5) << 4) | k0) + (col & 31u);
            v4u *const d4 = reinterpret_cast<v4u *>(out + (frame << L) + ((size_t)rest << 10)); // wave-uniform
            const unsigned doff = (k0 << (RL + 8)) | (tl & 255u);                                // in 16-byte units
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u32 *sp = sp0 + ROWY * 2 * i;
                const v4u y = {sp[0], sp[1], sp[2], sp[3]};
                __builtin_nontemporal_store(y, at32(d4 + ((size_t)(2 * i) << (RL + 8)), doff));
            }
        } else {
#pragma unroll
            for (int q = 0; q < 32; ++q) INTFFT_2XB_ST(v[q], at32(dst + ((size_t)rev5c(q) << (L - 5)), toff2_l));
        }
        }
    }
}

// ---- the 2-D scheme at N = 2^21 = 1024 x 2048 in TWO launches (round 5): the row cores + the store of X[k1 + 1024 k2] in one kernel --------------
// k_big2x_c<11> leaves the products A[k1 = brev10(rho)][n2] W_N^(k1 n2) as plain values at [rho][n2]; the three-launch form then runs the 2048-point row
// cores as a 1-D sub-plan and a layout change ([rho][k2] -> X[brev10(rho) + 1024 k2]).  Here ONE workgroup of 1024 threads takes the 16 rows whose k1 are
// consecutive (rho = k << 6 | r6, k = 0..15: k1 = brev6(r6) << 4 | rev4(k)) -- 128 KiB, one workgroup per CU -- and writes 64-byte pieces of X itself:
//   round 1  wave = row k, lane = n5..n0, regs j = n10..n6: STAGE 10..6 (per-lane twiddles in registers, quarter-turn sharing)
//   stage 5  v_permlane32_swap exchanges n10 (register bit 4) with n5 (lane bit 5); STAGE 5 on the register pairs (j, j + 16), one twiddle per lane
//   LDS      row (n10..n5) << 4 | k, column n4..n0 (33 dwords apart): the block-wide transpose that also interleaves the 16 rows
//   round 2  thread = (jj = n10..n5, kb = rev4(k)), regs q = n4..n0: STAGE 4..0 on wave-uniform twiddles (dif_round5_c)
//   store    X[(rev5(q) << 6 | rev6(jj)) * 1024 + (brev6(r6) << 4) + kb]: 16 consecutive lanes = one 64-byte piece; the block finishes both r6 partners
//            (r6, r6 ^ 32: the two halves of every output line) one after the other, as k_big2x_b does
// (int_fftNk.vhd:184-342 for the 2048-point core; the scheme itself is this library's extension).  tools/tilebench.hip: this tile
// shape copies at 3.9-4.0 TB/s; the row sub-plan + layout change it replaces take 174 us per 2^25 samples.
// L1 = log2 N1 (the column length; 11: the 2048 x 2048 plan, round 5): the tile's rows are rho = k << (L1 - 4) | r, r of L1 - 4 bits
template <bool FAST_OK, int L1 = 10>
__global__ __launch_bounds__(1024) void k_rows2k_tr(const u32 *in, u32 *out, const uint2 *__restrict__ twf, const Round5Consts c, size_t nframes, const Slice sl)
{
    extern __shared__ u32 lds[]; // 1024 rows x ROWY
    const int tid = threadIdx.x, lane = tid & 63;
    const int k = __builtin_amdgcn_readfirstlane(tid >> 6); // the row of this wave (round 1)
    u32 wa5[4], wb5[4];
    {
        const uint2 w = twf[31u + (unsigned)(lane & 31)]; // STAGE 5: index n4..n0
        wa5[0] = wa5[1] = wa5[2] = wa5[3] = w.x;
        wb5[0] = wb5[1] = wb5[2] = wb5[3] = w.y;
    }
    // transpose, write side: element (register r = (n5, n9..n6), lane = (n10, n4..n0)) -> row ((n10, n9..n6, n5) << 4) | k
    u32 *const wr_base = lds + ROWY * ((((lane >> 5) << 5) << 4) | k) + (lane & 31);
    const int jj = tid >> 4, kb = tid & 15;
    const int krow = ((kb & 1) << 3) | ((kb & 2) << 1) | ((kb & 4) >> 1) | ((kb & 8) >> 3);
    const u32 *const rd_base = lds + ROWY * ((jj << 4) | krow);
    const unsigned rjj = __brev((unsigned)jj) >> 26;
    constexpr int RR = L1 - 5; // tiles per frame and partner: 2^RR
    const unsigned toff2 = (rjj << L1) | (unsigned)kb;
    const short s5 = (short)(1 - (jj & 1)); // round 2: kind = n5
    const v2s sh5 = {s5, s5};
    const v2s none = {0, 0};
    __shared__ __attribute__((aligned(256))) u32 vote_flags[64];
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    // One workgroup per CU has nobody to overlap its phases with, so the NEXT tile's 32 loads are issued right behind the transpose's LDS writes (the data
    // registers are free from there until the next vote) and fly during round 2 and the stores; round 1's 16 + 1 per-lane twiddle pairs are re-read from the
    // L2-resident table per tile so that the two register sets (next tile's inputs, this tile's round-2 values) fit 128 VGPRs.
    size_t t = blockIdx.x;
    unsigned part = 0;
    bool have = (t >> RR) < nframes;
    u32 v[32];
    auto load_tile = [&](size_t tt, unsigned pp) {
        const unsigned r6 = (pp << RR) | ((unsigned)tt & ((1u << RR) - 1u));
        const u32 *src = in + ((tt >> RR) << (L1 + 11)) + ((size_t)(((unsigned)k << (L1 - 4)) | r6) << 11); // wave-uniform: this wave's row
        unsigned lane_l = (unsigned)lane;
        asm volatile("" : "+v"(lane_l));
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = INTFFT_LD(at32(src + ((size_t)j << 6), lane_l)); // regs = n10..n6
    };
    if (have) load_tile(t, 0);
    while (have) {
        const size_t ct = t;
        const unsigned cpart = part;
        if (part == 0) part = 1;
        else part = 0, t += gridDim.x;
        const bool have_next = (t >> RR) < nframes;
        u32 wa16[8], wb16[8];
        RoundTwQ t1;
        {
            unsigned lb = (unsigned)lane * 8u; // the lane's byte offset into a stage table, opaque per tile
            asm volatile("" : "+v"(lb));
            auto ld = [&](unsigned idx, u32 &wa, u32 &wb) {
                const uint2 w = ld2_at32b(twf + idx, lb);
                wa = w.x;
                wb = w.y;
            };
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8) ld(1023u + ((unsigned)j8 << 6), wa16[j8], wb16[j8]);
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) ld(511u + ((unsigned)j4 << 6), t1.wa8[j4], t1.wb8[j4]);
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) ld(255u + ((unsigned)j2 << 6), t1.wa4[j2], t1.wb4[j2]);
            ld(127u, t1.wa2[0], t1.wb2[0]);
            ld(63u, t1.wa1[0], t1.wb1[0]);
        }
        bool fast = false;
        {
            u32 acc = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) acc |= v[j] + sl.gbias;
            const bool bad = block_any(vote_flags, vote_phase, (acc & sl.gmask) != 0); // also orders the previous tile's LDS reads
            fast = FAST_OK && !bad;
        }
#define INTFFT_R2K_ROUND1(FX)                                                                                                                  \
    {                                                                                                                                          \
        dif_top16<FX, 0, false>(v, wa16, wb16, sl, none);                                                                                      \
        dif_round_q<FX, 0, 0, false>(v, t1, sl, none);                                                                                         \
        dif_round_q<FX, 16, 0xF, false>(v, t1, sl, none);                                                                                      \
        swap_guard(*reinterpret_cast<u32(*)[16]>(&v[0])); /* VALU write -> v_permlane read (the asm multiplies are invisible to the padding) */ \
        swap_guard(*reinterpret_cast<u32(*)[16]>(&v[16]));                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) swap32(v[j], v[j + 16]); /* register bit 4: n10 -> n5 */                                \
        /* STAGE 5: pairs (j, j + 16); the kind of both inputs is n6 = j & 1 */                                                                \
        _Pragma("unroll") for (int j = 0; j < 16; j += 4)                                                                                      \
            group4<false, FX, false, true, false, 0xA>(v[j], v[j + 16], v[j + 1], v[j + 17], v[j + 2], v[j + 18], v[j + 3], v[j + 19], wa5, wb5, sl); \
    }
        if (fast) INTFFT_R2K_ROUND1(FAST_OK)
        else INTFFT_R2K_ROUND1(false)
#undef INTFFT_R2K_ROUND1
#pragma unroll
        for (int r = 0; r < 32; ++r) wr_base[ROWY * ((((r & 15) << 1) | (r >> 4)) << 4)] = v[r];
        asm volatile("" ::: "memory");
        if (have_next) load_tile(t, part); // flies during round 2 and the stores below
        __syncthreads();
        u32 w[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) w[q] = rd_base[q];
        if (fast) dif_round5_c<FAST_OK>(w, c, sl, sh5);
        else dif_round5_c<false>(w, c, sl, sh5);
        const unsigned r6 = (cpart << RR) | ((unsigned)ct & ((1u << RR) - 1u));
        u32 *dst = out + ((ct >> RR) << (L1 + 11)) + ((__brev(r6) >> (36 - L1)) << 4);
        unsigned toff2_l = toff2;
        asm volatile("" : "+v"(toff2_l));
#pragma unroll
        for (int q = 0; q < 32; ++q) INTFFT_2XB_ST(w[q], at32(dst + ((size_t)rev5c(q) << (L1 + 6)), toff2_l));
        have = have_next;
    }
}

// ---- the 2-D scheme at N = 2^22 = 2048 x 2048 in TWO launches (round 5): k_cols2k_c + k_rows2k_tr<., 11> --------------------------------------------
// The column cores (2048-point int_fftNk over n1 for every n2) + the multiplier on tiles of 2048 rows x 16 columns (128 KiB: one 1024-thread workgroup per CU),
// k_rows2k_tr's register rounds on k_big2x_c's tile:
//   round 1  wave = n3..n0, lane = (n5, n4, column kb), regs j = n10..n6: every load instruction reads four 64-byte pieces (the XCD partner block b ^ 8 takes the
//            other half of every line at the same time; non-temporal loads: 248 against 226 Gsample/s with plain ones); STAGE 10..6 on per-thread twiddles (index (j << 6) | n5..n0, re-read per tile), then
//            v_permlane32_swap (register bit 4 <-> lane bit 5: n10 <-> n5) and STAGE 5
//   LDS      row (n10..n5) << 4 | kb, column n4..n0 (33 dwords apart); the next tile's loads go out behind the writes
//   round 2  thread = (jj = n10..n5, kb), regs q = n4..n0: STAGE 4..0 on wave-uniform twiddles; position rho = jj << 5 | q holds A[k1 = brev11(rho)][n2]
//   multiply by W_N^(k1 n2) (table [chunk][rho][16] of (wr | wi << 16), read eight entries at a time: beside the next tile's 32 values a second set of 32 would
//            spill), plain results at [rho][n2] of the scratch: 64-byte pieces, the partner block fills the other half of the line
template <bool FAST_OK>
__global__ __launch_bounds__(1024) void k_cols2k_c(const u32 *in, u32 *scr, const uint2 *__restrict__ twf, const Round5Consts c, const u32 *__restrict__ tw2d, size_t nframes,
                                                   const Slice sl)
{
    extern __shared__ u32 lds[]; // 1024 rows x ROWY
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // n3..n0 of round 1
    const unsigned t6 = (((unsigned)lane >> 4) << 4) | (unsigned)wv; // n5..n0 of round 1
    u32 wa5[4], wb5[4];
    {
        const uint2 w = twf[31u + (t6 & 31u)]; // STAGE 5: index n4..n0
        wa5[0] = wa5[1] = wa5[2] = wa5[3] = w.x;
        wb5[0] = wb5[1] = wb5[2] = wb5[3] = w.y;
    }
    // transpose, write side: element (register r = (n5, n9..n6), lane = (n10, n4, kb), wave n3..n0) -> row ((n10, n9..n6, n5) << 4) | kb, column n4..n0
    u32 *const wr_base = lds + ROWY * ((((lane >> 5) << 5) << 4) | (lane & 15)) + ((((lane >> 4) & 1) << 4) | wv);
    const int jj = tid >> 4, kb = tid & 15;
    const u32 *const rd_base = lds + ROWY * ((jj << 4) | kb);
    const short s5 = (short)(1 - (jj & 1)); // round 2: kind = n5
    const v2s sh5 = {s5, s5};
    const v2s none = {0, 0};
    const unsigned loff = (((unsigned)lane >> 4) << 15) | ((unsigned)lane & 15u); // user side (round 1): rows n5 n4 (x 2048 samples), column kb
    const unsigned soff = ((unsigned)jj << 16) | (unsigned)kb;                    // scratch side (round 2): row jj << 5 (+ q), column kb
    const unsigned twoff = ((unsigned)jj << 9) + (unsigned)kb;                    // table: [rho = jj << 5 | q][kb]
    __shared__ __attribute__((aligned(256))) u32 vote_flags[64];
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    const unsigned slot = blockIdx.x & 7u, part = (blockIdx.x >> 3) & 1u;
    size_t t = blockIdx.x;
    auto tile_of = [&](size_t tt) { return (tt >> 4) * 8u + slot; }; // G: frame = G >> 6, chunk = (G & 63) * 2 + part
    bool have = (tile_of(t) >> 6) < nframes;
    u32 v[32];
    auto load_tile = [&](size_t G) {
        const unsigned chunk = ((unsigned)G & 63u) * 2u + part;
        const u32 *src = in + ((G >> 6) << 22) + ((size_t)wv << 11) + chunk * 16u; // wave-uniform
        unsigned lo = loff;
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = INTFFT_LD(at32(src + ((size_t)j << 17), lo)); // regs = n10..n6
    };
    if (have) load_tile(tile_of(t));
    while (have) {
        const size_t ct = tile_of(t);
        t += gridDim.x;
        const size_t nt = tile_of(t);
        const bool have_next = (nt >> 6) < nframes;
        const unsigned chunk = ((unsigned)ct & 63u) * 2u + part;
        u32 wa16[8], wb16[8];
        RoundTwQ t1;
        {
            unsigned lb = t6 * 8u; // the thread's byte offset into a stage table, opaque per tile
            asm volatile("" : "+v"(lb));
            auto ld = [&](unsigned idx, u32 &wa, u32 &wb) {
                const uint2 w = ld2_at32b(twf + idx, lb);
                wa = w.x;
                wb = w.y;
            };
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8) ld(1023u + ((unsigned)j8 << 6), wa16[j8], wb16[j8]);
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) ld(511u + ((unsigned)j4 << 6), t1.wa8[j4], t1.wb8[j4]);
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) ld(255u + ((unsigned)j2 << 6), t1.wa4[j2], t1.wb4[j2]);
            ld(127u, t1.wa2[0], t1.wb2[0]);
            ld(63u, t1.wa1[0], t1.wb1[0]);
        }
        bool fast = false;
        {
            u32 acc = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) acc |= v[j] + sl.gbias;
            const bool bad = block_any(vote_flags, vote_phase, (acc & sl.gmask) != 0); // also orders the previous tile's LDS reads
            fast = FAST_OK && !bad;
        }
#define INTFFT_C2K_ROUND1(FX)                                                                                                                  \
    {                                                                                                                                          \
        dif_top16<FX, 0, false>(v, wa16, wb16, sl, none);                                                                                      \
        dif_round_q<FX, 0, 0, false>(v, t1, sl, none);                                                                                         \
        dif_round_q<FX, 16, 0xF, false>(v, t1, sl, none);                                                                                      \
        swap_guard(*reinterpret_cast<u32(*)[16]>(&v[0]));                                                                                      \
        swap_guard(*reinterpret_cast<u32(*)[16]>(&v[16]));                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) swap32(v[j], v[j + 16]); /* register bit 4: n10 -> n5 */                                \
        _Pragma("unroll") for (int j = 0; j < 16; j += 4)                                                                                      \
            group4<false, FX, false, true, false, 0xA>(v[j], v[j + 16], v[j + 1], v[j + 17], v[j + 2], v[j + 18], v[j + 3], v[j + 19], wa5, wb5, sl); \
    }
        if (fast) INTFFT_C2K_ROUND1(FAST_OK)
        else INTFFT_C2K_ROUND1(false)
#undef INTFFT_C2K_ROUND1
#pragma unroll
        for (int r = 0; r < 32; ++r) wr_base[ROWY * ((((r & 15) << 1) | (r >> 4)) << 4)] = v[r];
        asm volatile("" ::: "memory");
        if (have_next) load_tile(nt); // flies during round 2, the multiplier and the stores below
        __syncthreads();
        u32 w[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) w[q] = rd_base[q];
        if (fast) dif_round5_c<FAST_OK>(w, c, sl, sh5);
        else dif_round5_c<false>(w, c, sl, sh5);
        // B = cmult(A, W_N^(k1 n2)): Wa = (wr, -wi), Wb = (wi, wr) from the packed entry; exact extraction (the row cores take the full Y)
        const u32 *const twu = tw2d + ((size_t)chunk << 15); // [chunk][rho][kb]
        unsigned two = twoff;
        asm volatile("" : "+v"(two));
        const gptr_t<const u32> twq = at32(twu, two);
        const v2s pm = {1, -1};
#pragma unroll
        for (int q0 = 0; q0 < 32; q0 += 8) {
            u32 tw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) tw[i] = twq[16 * (q0 + i)];
#pragma unroll
            for (int q = 0; q < 8; q += 4) {
                u32 wa[4], wb[4], y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    wa[i] = as_u32(as_v2s(tw[q + i]) * pm);
                    wb[i] = __builtin_amdgcn_alignbit(tw[q + i], tw[q + i], 16);
                }
                mul2x<16, false>(w[q0 + q], w[q0 + q], wa[0], wb[0], w[q0 + q + 1], w[q0 + q + 1], wa[1], wb[1], sl.off_y, sl.sel, y[0], y[1], sl.wd);
                mul2x<16, false>(w[q0 + q + 2], w[q0 + q + 2], wa[2], wb[2], w[q0 + q + 3], w[q0 + q + 3], wa[3], wb[3], sl.off_y, sl.sel, y[2], y[3], sl.wd);
#pragma unroll
                for (int i = 0; i < 4; ++i) w[q0 + q + i] = y[i];
            }
        }
        u32 *dst = scr + ((ct >> 6) << 22) + chunk * 16u; // wave-uniform
        unsigned so = soff;
        asm volatile("" : "+v"(so));
#pragma unroll
        for (int q = 0; q < 32; ++q) *at32(dst + ((size_t)q << 11), so) = w[q];
        have = have_next;
    }
}

// ---- the inverse: int_ifftNk at N = 2^19, 2^20 in two passes (mirrors of pass B and pass A) ------------------------------------------
// int_ifftNk (src/vhdl/fft/int_ifftNk.vhd:183-341) is DIT: core position p takes X[brev_L(p)], STAGE s pairs positions that differ
// in bit s (twiddle index p mod 2^s, re/im-swapped multiplier feed of int_dit2_fly.vhd:290-325), natural order out.
//   pass QB  k_big2x_qb<L>  the mirror of pass B: tile = the 16 rows k of one `rest` x 1024 columns; loads X with pass B's store
//            pattern (64-byte pieces, XCD-paired partner = `rest` with its top bit flipped), STAGE 0..4 on wave-uniform twiddles,
//            LDS transpose, STAGE 5..9 on per-thread twiddles held in registers; user array -> scratch [q][c][hi][k][l]
//   pass QA  k_big2x_qa<L>  the mirror of pass A: tile = 2^(L-10) rows x 16 columns; scratch (2 KiB runs) -> STAGE 10..14 (L = 19:
//            two 4-stage rounds 10..13) on per-column twiddles parked in LDS -> LDS transpose -> STAGE L-5..L-1 on per-thread twiddles
//            re-read per tile -> natural or HALVES order out (64-byte row pieces 4 KiB apart, XCD-paired)
// DIT butterflies have no pre-shifted kinds (A >> 1 and T >> 1 are formed inside every stage): the scratch holds plain values.
// Quarter turns negate the twiddle operand (group4_dit<.., QTURN>): the planner checks that no table entry is -2^15.

// five DIT stages 0..4 on regs = n4..n0, wave-uniform twiddles in the DIT packing {Wc, Wd} (host: to_dit_packing_host5)
template <bool FASTX, int ROUND = 0> __device__ __forceinline__ void dit_round5_c(u32 (&v)[32], const Round5Consts &c, const Slice &sl)
{
    constexpr bool RD = ROUND != 0; // RNDMODE = 1 (int_dit2_fly.vhd:164-217): rhu2 sums on full-width values; ROUND == 2: + the w-bit wrap of narrow data
#pragma unroll
    for (int g = 0; g < 32; g += 2) bfly_triv<RD, false>(v[g], v[g + 1]); // STAGE 0: T = B
    if constexpr (ROUND == 2) {
#pragma unroll
        for (int g = 1; g < 32; g += 2) v[g] = wrap_w(v[g], sl.wd);
    }
#pragma unroll
    for (int g = 0; g < 32; g += 4) { // STAGE 1: even positions T = B, odd positions T = +j B (quirk)
        bfly_triv<RD, false>(v[g], v[g + 2]);
        bfly_pj_dit<RD>(v[g + 1], v[g + 3]);
    }
    if constexpr (ROUND == 2) {
#pragma unroll
        for (int g = 0; g < 32; g += 4) v[g + 2] = wrap_w(v[g + 2], sl.wd), v[g + 3] = wrap_w(v[g + 3], sl.wd);
    }
#pragma unroll
    for (int B = 0; B < 32; B += 8) // STAGE 2: pairs (q, q + 4), twiddle q & 3
        group4_dit<FASTX, true, ROUND, true>(v[B + 0], v[B + 4], v[B + 1], v[B + 5], v[B + 2], v[B + 6], v[B + 3], v[B + 7], c.wa2, c.wb2, sl);
    const u32 wa30[4] = {c.wa3[0], c.wa3[1], c.wa3[2], c.wa3[3]}, wb30[4] = {c.wb3[0], c.wb3[1], c.wb3[2], c.wb3[3]};
    const u32 wa31[4] = {c.wa3[4], c.wa3[5], c.wa3[6], c.wa3[7]}, wb31[4] = {c.wb3[4], c.wb3[5], c.wb3[6], c.wb3[7]};
#pragma unroll
    for (int B = 0; B < 32; B += 16) { // STAGE 3: pairs (q, q + 8), twiddle q & 7
        group4_dit<FASTX, true, ROUND, true>(v[B + 0], v[B + 8], v[B + 1], v[B + 9], v[B + 2], v[B + 10], v[B + 3], v[B + 11], wa30, wb30, sl);
        group4_dit<FASTX, true, ROUND, true>(v[B + 4], v[B + 12], v[B + 5], v[B + 13], v[B + 6], v[B + 14], v[B + 7], v[B + 15], wa31, wb31, sl);
    }
#pragma unroll
    for (int g = 0; g < 16; g += 4) { // STAGE 4: pairs (q, q + 16), twiddle q
        const u32 wa[4] = {c.wa4[g], c.wa4[g + 1], c.wa4[g + 2], c.wa4[g + 3]}, wb[4] = {c.wb4[g], c.wb4[g + 1], c.wb4[g + 2], c.wb4[g + 3]};
        group4_dit<FASTX, true, ROUND, true>(v[g], v[g + 16], v[g + 1], v[g + 17], v[g + 2], v[g + 18], v[g + 3], v[g + 19], wa, wb, sl);
    }
}

// IN_BR: BITREV order in (int_ifftNk's own order: memory index = core position): the thread's 32 inputs are 32 consecutive positions of
// its row (eight plain 16-byte loads: 274 Gsample/s at N = 2^20; loaded as 1 KiB runs per wave and handed over through LDS: 240-260)
template <int L, bool FAST_OK, bool IN_BR = false, int ROUND = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_big2x_qb(const u32 *in, u32 *scr, const uint2 *__restrict__ twf,
                                                                                              const Round5Consts c, size_t nframes, const Slice sl)
{
    static_assert(L == 19 || L == 20, "rows of 1024 points");
    constexpr int RL = L - 14;
    extern __shared__ u32 lds[];
    const int tid = threadIdx.x;
    const int m = ((tid >> 8) << 4) | (tid & 15), k = (tid >> 4) & 15;
    u32 wa16[8], wb16[8];
    RoundTwQ t1; // STAGE 5 + b on reg bit b of round 2 (regs n9..n5), twiddle index (jj << 5) | m; DIT packing
    {
        auto ld = [&](unsigned idx, u32 &wa, u32 &wb) {
            const uint2 w = twf[idx + (unsigned)m];
            wa = w.x, wb = w.y;
            to_dit_packing(wa, wb);
        };
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) ld(511u + ((unsigned)jj << 5), wa16[jj], wb16[jj]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ld(255u + ((unsigned)jj << 5), t1.wa8[jj], t1.wb8[jj]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) ld(127u + ((unsigned)jj << 5), t1.wa4[jj], t1.wb4[jj]);
        ld(63u, t1.wa2[0], t1.wb2[0]);
        ld(31u, t1.wa1[0], t1.wb1[0]);
    }
    const int jj = tid >> 4, kb = tid & 15;
    const int krow = ((kb & 1) << 3) | ((kb & 2) << 1) | ((kb & 4) >> 1) | ((kb & 8) >> 3);
    u32 *const wr_base = lds + ROWY * ((jj << 4) | krow);     // round 1 thread (jj, kb): row (jj << 4) + k, column q
    const u32 *const rd_base = lds + ROWY * k + m;            // round 2 thread (k, m): row (j << 4) + k
    const unsigned toff = (((unsigned)tid >> 8) << (L - 11)) | ((unsigned)tid & 255u); // scratch side (round 2 thread)
    const unsigned rjj = __brev((unsigned)jj) >> 27;
    const unsigned toff2 = (rjj << (L - 10)) | (unsigned)kb;  // user side (round 1 thread)
    const unsigned slot = blockIdx.x & 7u, part = (blockIdx.x >> 3) & 1u;

    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    for (size_t t = blockIdx.x;; t += gridDim.x) {
        const size_t G = (t >> 4) * 8u + slot;
        const size_t frame = G >> (RL - 1);
        if (frame >= nframes) break;
        const unsigned rest = (part << (RL - 1)) | ((unsigned)G & ((1u << (RL - 1)) - 1u));
        const unsigned q0 = rest & 31u, hi = rest >> 5;
        const u32 *src = in + (frame << L) + ((__brev(rest) >> (32 - RL)) << 4);
        u32 *dst = scr + (frame << L) + ((size_t)q0 << (L - 5)) + (hi << 8);
        unsigned toff_l = toff, toff2_l = toff2;
        asm volatile("" : "+v"(toff_l), "+v"(toff2_l));
        u32 v[32];
        if constexpr (IN_BR) { // position = ((k << RL | rest) << 10) | (jj << 5) | q
            typedef u32 v4u __attribute__((ext_vector_type(4)));
            const v4u *sbr = reinterpret_cast<const v4u *>(in + (frame << L) + ((((size_t)krow << RL) | rest) << 10) + ((size_t)jj << 5));
#pragma unroll
            for (int q = 0; q < 32; q += 4) {
                const v4u x = sbr[q >> 2]; // (plain: the eight pieces of a line are asked for by eight instructions -- the line has to stay in cache)
                v[q] = x.x, v[q + 1] = x.y, v[q + 2] = x.z, v[q + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 32; ++q) v[q] = INTFFT_LD(at32(src + ((size_t)rev5c(q) << (L - 5)), toff2_l)); // core position (rho, jj << 5 | q) = X[brev_L]
        }
        bool fast = false;
        {
            u32 acc = 0;
#pragma unroll
            for (int q = 0; q < 32; ++q) acc |= v[q] + sl.gbias;
            const bool bad = block_any(vote_flags, vote_phase, (acc & sl.gmask) != 0); // also orders the previous tile's LDS reads
            fast = FAST_OK && !bad;
        }
        if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd);
        if (fast) dit_round5_c<FAST_OK>(v, c, sl);
        else dit_round5_c<false, ROUND>(v, c, sl);
#pragma unroll
        for (int q = 0; q < 32; ++q) wr_base[q] = v[q];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = rd_base[ROWY * (j << 4)];
        if (fast) {
            dit_round_q<FAST_OK, 0>(v, t1, sl);
            dit_round_q<FAST_OK, 16>(v, t1, sl);
            dit_top16<FAST_OK>(v, wa16, wb16, sl);
        } else {
            dit_round_q<false, 0, ROUND>(v, t1, sl);
            dit_round_q<false, 16, ROUND>(v, t1, sl);
            dit_top16<false, ROUND>(v, wa16, wb16, sl);
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) *at32(dst + ((size_t)j << (L - 10)), toff_l) = v[j];
    }
}

template <int L, bool FAST_OK, int ROUND = 0>
__global__ __launch_bounds__(16 << (L - 15)) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_big2x_qa(const u32 *scr, u32 *out, const uint2 *__restrict__ twf,
                                                                                                         size_t nframes, unsigned groups, const Slice sl, int halves)
{
    static_assert(L == 19 || L == 20, "9 or 10 stages");
    constexpr int RB = L - 15;
    extern __shared__ u32 lds[]; // (32 << RB) rows x ROWX, then round 1's twiddles: 8 (RB = 4) / 16 slots x 16 columns, DIT packing
    uint2 *const tw1 = reinterpret_cast<uint2 *>(lds + (32 << RB) * ROWX);
    const int tid = threadIdx.x, l = tid & 15, hx = tid >> 4;
    const unsigned slot = blockIdx.x & 7u, part = (blockIdx.x >> 3) & 1u, G = (blockIdx.x >> 4) * 8u + slot;
    const unsigned chunk = (G & 31u) * 2u + part, grp = G >> 5;
    const unsigned lfull = chunk * 16 + l;
    const unsigned toff = ((unsigned)hx << 10) | lfull; // user side and round-2 twiddles: thread (hx = n(L-6)..n10, l)
    const unsigned toff2 = (chunk << (L - 11)) | (((unsigned)hx & ((1u << (RB - 4)) - 1u)) << 8) | (((unsigned)hx >> (RB - 4)) << 4) | (unsigned)l; // scratch side
    if (hx == 0) { // round 1 (STAGE 10 + b, table index (rr << 10) | lfull): parked per column, DIT packing
        int s = 0;
        auto park = [&](unsigned uniform_idx) {
            uint2 w = (twf + uniform_idx)[lfull];
            to_dit_packing(w.x, w.y);
            tw1[16 * s++ + l] = w;
        };
        park((1u << 10) - 1u);
        park((1u << 11) - 1u);
        for (int rr = 0; rr < 2; ++rr) park((1u << 12) - 1u + ((unsigned)rr << 10));
        for (int rr = 0; rr < 4; ++rr) park((1u << 13) - 1u + ((unsigned)rr << 10));
        if constexpr (RB == 5)
            for (int rr = 0; rr < 8; ++rr) park((1u << 14) - 1u + ((unsigned)rr << 10));
    }
    u32 *const wr_base = lds + ROWX * (hx << 5) + l;  // round 1 thread (jx = tid >> 4, l): row (jx << 5) + q
    const u32 *const rd_base = lds + ROWX * hx + l;   // round 2 thread (hx, l): row (j << RB) + hx

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
        for (int q = 0; q < 32; ++q) v[q] = INTFFT_LD(at32(src + ((size_t)q << (L - 5)), toff2_l));
        bool fast = false;
        {
            u32 acc = 0;
#pragma unroll
            for (int q = 0; q < 32; ++q) acc |= v[q] + sl.gbias;
            const bool bad = block_any(vote_flags, vote_phase, (acc & sl.gmask) != 0); // also orders the previous frame's LDS reads
            fast = FAST_OK && !bad;
        }
        {
            RoundTwQ t1;
            u32 wa16[8], wb16[8];
            int s = 0;
            auto get = [&](u32 &wa, u32 &wb) {
                const uint2 w = tw1[16 * s++ + l];
                wa = w.x, wb = w.y;
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
            } else {
                dit_round_q<false, 0, ROUND>(v, t1, sl);
                dit_round_q<false, 16, ROUND>(v, t1, sl);
                if constexpr (RB == 5) dit_top16<false, ROUND>(v, wa16, wb16, sl);
            }
        }
        // round 2's per-thread twiddles: STAGE L-5+b, index (jj << (RB + 10)) | toff: re-read per tile, converted to the DIT packing
        RoundTwQ t2;
        u32 wa16[8], wb16[8];
        unsigned twb = toff_l * 8u; // the thread's byte offset into a stage table, opaque (see k_big2x_a)
        asm volatile("" : "+v"(twb));
        auto ld = [&](unsigned uniform_idx, u32 &wa, u32 &wb) {
            const uint2 w = ld2_at32b(twf + uniform_idx, twb);
            wa = w.x, wb = w.y;
        };
        ld((1u << (L - 5)) - 1u, t2.wa1[0], t2.wb1[0]);
        ld((1u << (L - 4)) - 1u, t2.wa2[0], t2.wb2[0]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) ld((1u << (L - 3)) - 1u + ((unsigned)jj << (RB + 10)), t2.wa4[jj], t2.wb4[jj]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ld((1u << (L - 2)) - 1u + ((unsigned)jj << (RB + 10)), t2.wa8[jj], t2.wb8[jj]);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) ld((1u << (L - 1)) - 1u + ((unsigned)jj << (RB + 10)), wa16[jj], wb16[jj]);
#pragma unroll
        for (int q = 0; q < 32; ++q) wr_base[ROWX * q] = v[q];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = rd_base[ROWX * (j << RB)];
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
        } else {
            dit_round_q<false, 0, ROUND>(v, t2, sl);
            dit_round_q<false, 16, ROUND>(v, t2, sl);
            dit_top16<false, ROUND>(v, wa16, wb16, sl);
        }
        if (halves) { // HALVES order out: memory index = 2 * (n without n(L-1)) + n(L-1): registers j and j + 16 are one 8-byte store
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            v2u *d2 = reinterpret_cast<v2u *>(dst);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v2u w = {v[j], v[j + 16]};
                __builtin_nontemporal_store(w, at32(d2 + ((size_t)j << (RB + 10)), toff_l));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) __builtin_nontemporal_store(v[j], at32(dst + ((size_t)j << (RB + 10)), toff_l));
        }
    }
}

// ---- the 2-D scheme at N = 2^20 = 1024 x 1024 in TWO launches: k_big2x_c + k_big2x_b<20> --------------
// forward, DATA_WIDTH = 16, TWDL_WIDTH <= 16, scaled-truncate, natural / HALVES order in -> natural order out:
//   k_big2x_c   the column cores (1024-point int_fftNk over n1 for every n2) on pass A's tiles -- 1024 rows x 16 columns, XCD-paired
//               half lines -- with the 1024-point core's own twiddles (index = row mod 2^s: per thread, frame and column invariant)
//               and STAGE 4..0 on wave-uniform twiddles; the result of row rho is A[k1 = brev10(rho)][n2], multiplied in place by
//               W_N^(k1 n2) (int_cmult_dsp48 at 16 x t bits, table [chunk][rho][16 columns] of (wr | wi << 16)), emitted as Y >> 1
//               (all the row core's first stage reads) into pass B's scratch layout
//   k_big2x_b   the row cores + the store of X[k1 + 1024 k2]: pass B as it is (its row number rho leaves as brev10(rho) = k1)
// instead of the five launches of the composite plan (layout change, column sub-plan, layout change + multiplier, row sub-plan,
// layout change).
// LR = log2 N2 (the row length).  LR = 10: the two-launch plan above (results as Y >> 1 in pass B's layout).  LR = 11 .. 14 (N = 2^21
// .. 2^24 as 1024 x N2): results as plain Y at [rho][n2] of the scratch, the N2-point row cores run as an ordinary 1-D sub-plan
// over 1024 frames per transform, and ONE layout change ([rho][k2] -> X[brev10(rho) + 1024 k2], any output order) finishes:
// three launches (four where the row sub-plan has two passes) instead of five (six).
template <bool FAST_OK, int LR>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_big2x_c(const u32 *in, u32 *scr, const uint2 *__restrict__ tw1k, const Round5Consts c,
                                                                                             const u32 *__restrict__ tw2d, size_t nframes, unsigned groups, const Slice sl,
                                                                                             int halves)
{
    static_assert(LR >= 10 && LR <= 14, "1024 rows x 2^LR columns");
    constexpr int L = 10 + LR, RB = 5;
    constexpr bool PASSB = LR == 10;
    constexpr unsigned CG = 1u << (LR - 5); // column groups of 128 bytes per frame
    extern __shared__ u32 lds[]; // 1024 rows x ROWX
    const int tid = threadIdx.x, l = tid & 15, hx = tid >> 4;
    const unsigned slot = blockIdx.x & 7u, part = (blockIdx.x >> 3) & 1u, G = (blockIdx.x >> 4) * 8u + slot;
    const unsigned chunk = (G & (CG - 1u)) * 2u + part, grp = G >> (LR - 5);
    const unsigned lfull = chunk * 16 + l;
    const unsigned toff = ((unsigned)hx << LR) | lfull;
    // round 1: regs j = rho9..5, thread hx = rho4..0: stages 9..5 of the 1024-point core, twiddle index (jj << 5 | hx).  The set is
    // frame invariant, but held over the loop it costs 32 VGPRs next to the 32 inter-core twiddles of round 2 (31 spilled dwords
    // per lane): it is re-read per tile instead -- 8 KiB of table, L1-resident
    u32 wa16[8], wb16[8];
    RoundTwQ t1;
    auto round1_tw = [&](unsigned h) {
        auto ld = [&](unsigned idx, u32 &wa, u32 &wb) {
            const uint2 w = (tw1k + idx)[h];
            wa = w.x;
            wb = w.y;
        };
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) ld(511u + ((unsigned)jj << 5), wa16[jj], wb16[jj]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ld(255u + ((unsigned)jj << 5), t1.wa8[jj], t1.wb8[jj]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) ld(127u + ((unsigned)jj << 5), t1.wa4[jj], t1.wb4[jj]);
        ld(63u, t1.wa2[0], t1.wb2[0]);
        ld(31u, t1.wa1[0], t1.wb1[0]);
    };
    u32 *const wr_base = lds + ROWX * hx + l;              // row (j << 5) + hx
    const u32 *const rd_base = lds + ROWX * (hx << 5) + l; // row (jx << 5) + q, jx = tid >> 4
    // store side, thread = (jx = rho9..5, l), regs q = rho4..0: pass B's layout [q][c][hi][k][l] (LR = 10) or natural [rho][n2]
    const unsigned toff2 = PASSB ? (chunk << (L - 11)) | (((unsigned)hx & 1u) << 8) | (((unsigned)hx >> 1) << 4) | (unsigned)l
                                 : ((unsigned)hx << (5 + LR)) | lfull;
    const u32 *const twu = tw2d + ((size_t)chunk << 14); // [chunk][rho = jx << 5 | q][l]
    const unsigned twoff = ((unsigned)hx << 9) + (unsigned)l;
    const v2s none = {0, 0};
    const short s5 = (short)(1 - (hx & 1)); // round 2: the kind of its inputs is rho5 = jx bit 0
    const v2s sh5 = {s5, s5};

    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    for (size_t frame = grp; frame < nframes; frame += groups) {
        const u32 *src = in + (frame << L);
        u32 *dst = scr + (frame << L);
        unsigned toff_l = toff, toff2_l = toff2, hx_l = (unsigned)hx, twoff_l = twoff;
        asm volatile("" : "+v"(toff_l), "+v"(toff2_l), "+v"(hx_l), "+v"(twoff_l));
        u32 v[32];
        if (halves) {
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            const v2u *sh = reinterpret_cast<const v2u *>(src);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v2u w = INTFFT_LD(at32(sh + ((size_t)j << (RB + LR)), toff_l));
                v[j] = w.x;
                v[j + 16] = w.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = INTFFT_LD(at32(src + ((size_t)j << (RB + LR)), toff_l));
        }
        round1_tw(hx_l);
        bool fast = false;
        {
            u32 acc = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) acc |= v[j] + sl.gbias;
            const bool bad = block_any(vote_flags, vote_phase, (acc & sl.gmask) != 0); // also orders the previous tile's LDS reads
            fast = FAST_OK && !bad;
        }
        if (fast) {
            dif_top16<FAST_OK, 0, false>(v, wa16, wb16, sl, none);
            dif_round_q<FAST_OK, 0, 0, false>(v, t1, sl, none);
            dif_round_q<FAST_OK, 16, 0xF, false>(v, t1, sl, none);
        } else {
            dif_top16<false, 0, false>(v, wa16, wb16, sl, none);
            dif_round_q<false, 0, 0, false>(v, t1, sl, none);
            dif_round_q<false, 16, 0xF, false>(v, t1, sl, none);
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) wr_base[ROWX * (j << RB)] = v[j];
        __syncthreads();
        u32 tw[32]; // the inter-core twiddles of this thread's 32 results (L2-resident slice of the table: one dword each)
        const gptr_t<const u32> twq = at32(twu, twoff_l);
#pragma unroll
        for (int q = 0; q < 32; ++q) tw[q] = twq[16 * q]; // one SGPR base + thread offset, 64-byte steps in the immediate
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = rd_base[ROWX * q];
        if (fast) dif_round5_c<FAST_OK>(v, c, sl, sh5);
        else dif_round5_c<false>(v, c, sl, sh5);
        // B = cmult(A, W_N^(k1 n2)): Wa = (wr, -wi), Wb = (wi, wr) from the packed entry; the operand is the plain 16-bit A.
        // LR = 10: emitted as Y >> 1 (all the row core's first stage reads; fast extraction where the tile voted for it);
        // longer rows: the full Y for the row sub-plan (the high halves alone cannot give it: exact extraction)
        const v2s pm = {1, -1};
#pragma unroll
        for (int q = 0; q < 32; q += 4) {
            u32 wa[4], wb[4], d[4], y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wa[i] = as_u32(as_v2s(tw[q + i]) * pm);
                wb[i] = __builtin_amdgcn_alignbit(tw[q + i], tw[q + i], 16);
                d[i] = v[q + i];
            }
            if constexpr (PASSB) {
                if (fast) mul4f<false>(d, d, wa, wb, sl.sel_hi, y);
                else {
                    mul2x<15, false>(d[0], d[0], wa[0], wb[0], d[1], d[1], wa[1], wb[1], sl.off_y1, sl.sel, y[0], y[1], sl.wd);
                    mul2x<15, false>(d[2], d[2], wa[2], wb[2], d[3], d[3], wa[3], wb[3], sl.off_y1, sl.sel, y[2], y[3], sl.wd);
                }
            } else {
                mul2x<16, false>(d[0], d[0], wa[0], wb[0], d[1], d[1], wa[1], wb[1], sl.off_y, sl.sel, y[0], y[1], sl.wd);
                mul2x<16, false>(d[2], d[2], wa[2], wb[2], d[3], d[3], wa[3], wb[3], sl.off_y, sl.sel, y[2], y[3], sl.wd);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) v[q + i] = y[i];
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) *at32(dst + ((size_t)q << (PASSB ? L - 5 : LR)), toff2_l) = v[q];
    }
}

// the table of k_big2x_c: entry [chunk][rho][l] = W_N^(k1 n2), k1 = brev10(rho), n2 = 16 chunk + l, as (wr | wi << 16)
__global__ void k_build_tw2d_tiles(u32 *__restrict__ out, int L, int twd, int l1)
{
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; // < 2^L
    const unsigned chunk = idx >> (l1 + 4), rho = (idx >> 4) & ((1u << l1) - 1u), l = idx & 15u;
    const unsigned k1 = __brev(rho) >> (32 - l1), n2 = chunk * 16 + l;
    int re, im;
    tw2d_eval(L, twd, (k1 * n2) & ((1u << L) - 1u), re, im);
    out[idx] = ((u32)re & 0xFFFFu) | ((u32)im << 16);
}

// returns 0 (not this path), 2 (N = 2^20: two launches, natural order out) or 3 (N = 2^21 .. 2^24 with N1 = 1024: column pass, row
// sub-plan, one layout change; any output order)
int fused2d_supported(int log2n, int l1, int data_width, int twdl_width, int format, int rndmode, int direction, int in_order, int out_order)
{
    // 8 (round 5): N = 2^22 = 2048 x 2048 in two launches (k_cols2k_c + k_rows2k_tr<., 11>), natural order in and out
    if (l1 == 11 && log2n == 22 && data_width == 16 && twdl_width >= 8 && twdl_width <= 16 && format == 0 && rndmode == 0 && direction == 0 && in_order == 0 &&
        out_order == 0 && !diag_env("INTFFT_2D_NO_FUSED_CORES") && !diag_env("INTFFT_2D_NO_ROWS2K"))
        return 8;
    if (!(l1 == 10 && data_width == 16 && twdl_width >= 8 && twdl_width <= 16 && format == 0 && rndmode == 0 && direction == 0 &&
          (in_order == 0 || in_order == 2)) || diag_env("INTFFT_2D_NO_FUSED_CORES"))
        return 0;
    if (log2n == 20) return out_order == 0 ? 2 : 0;
    return log2n >= 21 && log2n <= 24 ? 3 : 0;
}

// 1: N = 2^20 (k_big2x_qb + k_big2x_ci), 2: N = 2^21 (k_rows2k_qtr + k_big2x_ci<., 11>; round 5), 3: N = 2^21 .. 2^24 in three launches (one layout change, the
// N2-point inverse sub-plan, k_big2x_ci<., L2, ROWS>; round 5; N = 2^21 only with INTFFT_2D_NO_ROWS2K)
int fused2d_inv_supported(int log2n, int l1, int data_width, int twdl_width, int format, int rndmode, int direction, int in_order, int out_order)
{
    // 4 (round 5): N = 2^22 = 2048 x 2048 in two launches (k_rows2k_qtr<., 11, true> + k_cols2k_ci), natural order in and out
    if (l1 == 11 && log2n == 22 && data_width == 16 && twdl_width >= 8 && twdl_width <= 16 && format == 0 && rndmode == 0 && direction == 1 && in_order == 0 &&
        out_order == 0 && !diag_env("INTFFT_2D_NO_FUSED_CORES") && !diag_env("INTFFT_2D_NO_ROWS2K"))
        return 4;
    return (log2n == 20 ? 1 : log2n == 21 && !diag_env("INTFFT_2D_NO_ROWS2K") ? 2 : log2n >= 21 && log2n <= 24 ? 3 : 0) * (int)(l1 == 10 && data_width == 16 && twdl_width >= 8 && twdl_width <= 16 && format == 0 && rndmode == 0 && direction == 1 && in_order == 0 &&
           (out_order == 0 || out_order == 2) && !diag_env("INTFFT_2D_NO_FUSED_CORES"));
}

hipError_t build_fused2d_table(u32 *d_table, int log2n, int twd, hipStream_t stream, int l1)
{
    hipLaunchKernelGGL(k_build_tw2d_tiles, dim3(1u << (log2n - 8)), dim3(256), 0, stream, d_table, log2n, twd, l1);
    return hipGetLastError();
}

const char *fused2d_kernel_name() { return "2d[k_big2x_c|k_big2x_b]"; }

static void fused2d_consts(const int2 *h_tw1k, Round5Consts &c)
{
    auto pk = [&](int idx, u32 &wa, u32 &wb) {
        const int2 w = h_tw1k[idx];
        wa = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        wb = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    };
    for (int i = 0; i < 16; ++i) pk(15 + i, c.wa4[i], c.wb4[i]);
    for (int i = 0; i < 8; ++i) pk(7 + i, c.wa3[i], c.wb3[i]);
    for (int i = 0; i < 4; ++i) pk(3 + i, c.wa2[i], c.wb2[i]);
}

// the column pass alone (any supported row length): user array -> scratch; LR = 10 writes pass B's layout, longer rows [rho][n2]
hipError_t launch_fused2d_cols(int lr, int twd, const u32 *pin, u32 *scr, const uint2 *tw1k, const int2 *h_tw1k, const u32 *tw2d, size_t nframes, int halves,
                               hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    Round5Consts c;
    fused2d_consts(h_tw1k, c);
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fx = twd == 16 && allow_fast;
    const size_t ldsa = (size_t)1024 * ROWX * sizeof(u32);
    const size_t per_launch = (size_t)64 << (20 - 10) >> lr; // frame groups that make ~4096 blocks, as for N = 2^20
    const size_t cap = per_launch ? per_launch : 1;
    const unsigned groups = (unsigned)(nframes < cap ? nframes : cap);
    const unsigned grid = (unsigned)(((size_t)1 << (lr - 4)) * groups);
#define INTFFT_2DC(LRR)                                                                                                                   \
    if (fx) {                                                                                                                             \
        allow_max_lds(kptr(k_big2x_c<true, LRR>));                                                                                        \
        hipLaunchKernelGGL((k_big2x_c<true, LRR>), dim3(grid), dim3(512), ldsa, stream, pin, scr, tw1k, c, tw2d, nframes, groups, sl, halves);  \
    } else {                                                                                                                              \
        allow_max_lds(kptr(k_big2x_c<false, LRR>));                                                                                       \
        hipLaunchKernelGGL((k_big2x_c<false, LRR>), dim3(grid), dim3(512), ldsa, stream, pin, scr, tw1k, c, tw2d, nframes, groups, sl, halves); \
    }
    switch (lr) {
    case 10: INTFFT_2DC(10) break;
    case 11: INTFFT_2DC(11) break;
    case 12: INTFFT_2DC(12) break;
    case 13: INTFFT_2DC(13) break;
    case 14: INTFFT_2DC(14) break;
    default: return hipErrorInvalidValue;
    }
#undef INTFFT_2DC
    return hipGetLastError();
}

// N = 2^20: column pass + pass B.  tw1k / h_tw1k: the packed / host twiddle tables of the 1024-point cores
// the row cores of the 1024 x 2048 plan + the store of X[k1 + 1024 k2] (k_rows2k_tr): tw16r = the 2048-point core's packed table, h_tw its host copy
hipError_t launch_fused2d_rows2k(int twd, const u32 *prod, u32 *pout, const uint2 *tw16r, const int2 *h_tw, size_t nframes, hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    Round5Consts c;
    auto pk = [&](int idx, u32 &wa, u32 &wb) {
        const int2 w = h_tw[idx];
        wa = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        wb = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    };
    for (int i = 0; i < 16; ++i) pk(15 + i, c.wa4[i], c.wb4[i]);
    for (int i = 0; i < 8; ++i) pk(7 + i, c.wa3[i], c.wb3[i]);
    for (int i = 0; i < 4; ++i) pk(3 + i, c.wa2[i], c.wb2[i]);
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fx = twd == 16 && allow_fast;
    const size_t ldsb = (size_t)1024 * ROWY * sizeof(u32);
    const size_t pairs = nframes * 32, cap = (size_t)device_cus();
    const unsigned grid = (unsigned)(pairs < cap ? pairs : cap);
    if (fx) {
        allow_max_lds(kptr(k_rows2k_tr<true>));
        hipLaunchKernelGGL((k_rows2k_tr<true>), dim3(grid), dim3(1024), ldsb, stream, prod, pout, tw16r, c, nframes, sl);
    } else {
        allow_max_lds(kptr(k_rows2k_tr<false>));
        hipLaunchKernelGGL((k_rows2k_tr<false>), dim3(grid), dim3(1024), ldsb, stream, prod, pout, tw16r, c, nframes, sl);
    }
    return hipGetLastError();
}

// N = 2^22: the 2048-point column cores + multiplier (k_cols2k_c), then the 2048-point row cores + the store of X[k1 + 2048 k2] (k_rows2k_tr<., 11>)
hipError_t launch_fused2d_2k2k(int twd, const u32 *pin, u32 *pout, u32 *scr, const uint2 *tw16r, const int2 *h_tw, const u32 *tw2d, size_t nframes, hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    Round5Consts c;
    auto pk = [&](int idx, u32 &wa, u32 &wb) {
        const int2 w = h_tw[idx];
        wa = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        wb = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    };
    for (int i = 0; i < 16; ++i) pk(15 + i, c.wa4[i], c.wb4[i]);
    for (int i = 0; i < 8; ++i) pk(7 + i, c.wa3[i], c.wb3[i]);
    for (int i = 0; i < 4; ++i) pk(3 + i, c.wa2[i], c.wb2[i]);
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fx = twd == 16 && allow_fast;
    const size_t ldsb = (size_t)1024 * ROWY * sizeof(u32);
    const size_t ctiles = nframes << 7, cap16 = (size_t)device_cus() / 16 * 16; // 128 column chunks per frame, XCD-paired: the grid is a multiple of 16
    const unsigned gc = (unsigned)std::min(ctiles, cap16);
    const size_t pairs = nframes << 6, cap = (size_t)device_cus();
    const unsigned gr = (unsigned)(pairs < cap ? pairs : cap);
    if (fx) {
        allow_max_lds(kptr(k_cols2k_c<true>));
        allow_max_lds(kptr(k_rows2k_tr<true, 11>));
        hipLaunchKernelGGL((k_cols2k_c<true>), dim3(gc), dim3(1024), ldsb, stream, pin, scr, tw16r, c, tw2d, nframes, sl);
        hipLaunchKernelGGL((k_rows2k_tr<true, 11>), dim3(gr), dim3(1024), ldsb, stream, scr, pout, tw16r, c, nframes, sl);
    } else {
        allow_max_lds(kptr(k_cols2k_c<false>));
        allow_max_lds(kptr(k_rows2k_tr<false, 11>));
        hipLaunchKernelGGL((k_cols2k_c<false>), dim3(gc), dim3(1024), ldsb, stream, pin, scr, tw16r, c, tw2d, nframes, sl);
        hipLaunchKernelGGL((k_rows2k_tr<false, 11>), dim3(gr), dim3(1024), ldsb, stream, scr, pout, tw16r, c, nframes, sl);
    }
    return hipGetLastError();
}


hipError_t launch_fused2d(int twd, const u32 *pin, u32 *pout, u32 *scr, const uint2 *tw1k, const int2 *h_tw1k, const u32 *tw2d, size_t nframes, int halves,
                          hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    hipError_t e = launch_fused2d_cols(10, twd, pin, scr, tw1k, h_tw1k, tw2d, nframes, halves, stream);
    if (e != hipSuccess) return e;
    Round5Consts c;
    fused2d_consts(h_tw1k, c);
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fx = twd == 16 && allow_fast;
    const size_t ldsb = (size_t)512 * ROWY * sizeof(u32);
    const size_t ntiles = nframes << 5, capb = ((size_t)device_cus() * 2 + 15) / 16 * 16; // (a block of pass B takes both partner tiles)
    const unsigned gb = (unsigned)(ntiles < capb ? (ntiles + 15) / 16 * 16 : capb);
    if (fx) {
        allow_max_lds(kptr(k_big2x_b<20, true>));
        hipLaunchKernelGGL((k_big2x_b<20, true>), dim3(gb), dim3(512), ldsb, stream, scr, pout, tw1k, c, nframes, sl, 1);
    } else {
        allow_max_lds(kptr(k_big2x_b<20, false>));
        hipLaunchKernelGGL((k_big2x_b<20, false>), dim3(gb), dim3(512), ldsb, stream, scr, pout, tw1k, c, nframes, sl, 1);
    }
    return hipGetLastError();
}

// ---- the 2-D scheme at N = 2^20 = 1024 x 1024, INVERSE, in TWO launches: k_big2x_qb<20> + k_big2x_ci -----------------------------
// x[n1 N2 + n2] from X[k1 + N1 k2]: N2-point int_ifftNk over k2 for every k1 (the row cores), T = V conj(W_N^(k1 n2))
// through the re/im-swapped multiplier feed (int_dit2_fly.vhd:304-322), N1-point int_ifftNk over k1 for every n2 (the column cores).
//   k_big2x_qb<20>  pass QB as it is: its row r gathers X[brev10(pos) * 1024 + brev10(r)] -- the input of the row core k1 = brev10(r) in
//                   the bit-reversed order a DIT core takes -- and its STAGE 0..9 twiddles (index = position mod 2^s, s < 10) ARE the
//                   1024-point core's; the result is V[k1 = brev10(r)][n2] in the scratch layout [q][c][hi][k][l]
//   k_big2x_ci      pass QA's tiles (1024 rows r x 16 columns n2, XCD-paired half lines on the store side): multiply by conj W from the
//                   forward plan's table [chunk][r][16] (W_N^(brev10(r) n2): the same entries), then the column cores -- r IS the DIT
//                   position of k1 -- with the 1024-point core's own twiddles (index = r mod 2^s: STAGE 0..4 wave-uniform, STAGE 5..9
//                   per thread, frame and column invariant), natural or HALVES order out
// L2 = 11 (round 5): the column cores of the 1024 x 2048 inverse plan -- 128 chunks of 16 columns n2, rows of 2048 samples on the user side; the scratch keeps
// the [q][chunk][hi][k][l] shape (2 KiB runs per (q, chunk)), written by k_rows2k_qtr.  ROWS (L2 = 12 .. 14, round 5): the scratch holds plain rows [r][n2] as an
// N2-point inverse sub-plan leaves them (64-byte pieces, the XCD partner takes the other half of every line: plain loads)
template <bool FAST_OK, int L2 = 10, bool ROWS = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_big2x_ci(const u32 *scr, u32 *out, const uint2 *__restrict__ tw1k, const Round5Consts c,
                                                                                              const u32 *__restrict__ tw2d, size_t nframes, unsigned groups, const Slice sl,
                                                                                              int halves)
{
    constexpr int L = 10 + L2, RB = 5;
    extern __shared__ u32 lds[]; // 1024 rows x ROWX
    const int tid = threadIdx.x, l = tid & 15, hx = tid >> 4;
    const unsigned slot = blockIdx.x & 7u, part = (blockIdx.x >> 3) & 1u, G = (blockIdx.x >> 4) * 8u + slot;
    constexpr unsigned GC = 1u << (L2 - 5); // chunk pairs per frame
    const unsigned chunk = (G & (GC - 1u)) * 2u + part, grp = G >> (L2 - 5);
    const unsigned lfull = chunk * 16 + l;
    const unsigned toff = ((unsigned)hx << L2) | lfull; // user side: thread (hx = r4..r0 after the transpose, l)
    const unsigned toff2 = ROWS ? (((unsigned)hx << (5 + L2)) | lfull)
                                : ((chunk << 9) | (((unsigned)hx & 1u) << 8) | (((unsigned)hx >> 1) << 4) | (unsigned)l); // scratch side: r = hx << 5 | q
    const u32 *const twu = tw2d + ((size_t)chunk << 14); // [chunk][r = hx << 5 | q][l]
    const unsigned twoff = ((unsigned)hx << 9) + (unsigned)l;
    // round 2 (regs = r9..r5, thread = r4..r0 = hx): STAGE 5 + b on reg bit b, twiddle index (jj << 5) | hx; DIT packing; frame invariant
    u32 wa16[8], wb16[8];
    RoundTwQ t2;
    {
        auto ld = [&](unsigned idx, u32 &wa, u32 &wb) {
            const uint2 w = tw1k[idx + (unsigned)hx];
            wa = w.x, wb = w.y;
            to_dit_packing(wa, wb);
        };
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) ld(511u + ((unsigned)jj << 5), wa16[jj], wb16[jj]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ld(255u + ((unsigned)jj << 5), t2.wa8[jj], t2.wb8[jj]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) ld(127u + ((unsigned)jj << 5), t2.wa4[jj], t2.wb4[jj]);
        ld(63u, t2.wa2[0], t2.wb2[0]);
        ld(31u, t2.wa1[0], t2.wb1[0]);
    }
    u32 *const wr_base = lds + ROWX * (hx << 5) + l;  // round 1 thread (hx = r9..r5, l): row (hx << 5) + q
    const u32 *const rd_base = lds + ROWX * hx + l;   // round 2 thread (hx = r4..r0, l): row (j << 5) + hx

    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    for (size_t frame = grp; frame < nframes; frame += groups) {
        const u32 *src = scr + (frame << L);
        u32 *dst = out + (frame << L);
        unsigned toff_l = toff, toff2_l = toff2, twoff_l = twoff;
        asm volatile("" : "+v"(toff_l), "+v"(toff2_l), "+v"(twoff_l));
        u32 v[32], tw[32];
        const gptr_t<const u32> twq = at32(twu, twoff_l);
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            if constexpr (ROWS) v[q] = *at32(src + ((size_t)q << L2), toff2_l); // (plain: 190 against 176 Gsample/s at N = 2^22 with non-temporal loads)
            else v[q] = INTFFT_LD(at32(src + ((size_t)q << (L - 5)), toff2_l));
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) tw[q] = twq[16 * q]; // one SGPR base + thread offset, 64-byte steps in the immediate
        // T = V conj(W): T.re = V.re wr + V.im wi, T.im = V.im wr - V.re wi = the DIT butterfly's multiplier with Wc = (wr, wi) -- the table
        // entry itself -- and Wd = (-wi, wr); plain 16-bit results (exact extraction: DIT stages form A >> 1 and T >> 1 themselves)
        const v2s mp = {-1, 1};
#pragma unroll
        for (int q = 0; q < 32; q += 2) {
            const u32 wb0 = as_u32(as_v2s(__builtin_amdgcn_alignbit(tw[q], tw[q], 16)) * mp);
            const u32 wb1 = as_u32(as_v2s(__builtin_amdgcn_alignbit(tw[q + 1], tw[q + 1], 16)) * mp);
            u32 y0, y1;
            mul2x<16, false>(v[q], v[q], tw[q], wb0, v[q + 1], v[q + 1], tw[q + 1], wb1, sl.off_y, sl.sel, y0, y1, sl.wd);
            v[q] = y0, v[q + 1] = y1;
        }
        bool fast = false;
        {
            u32 acc = 0;
#pragma unroll
            for (int q = 0; q < 32; ++q) acc |= v[q] + sl.gbias;
            const bool bad = block_any(vote_flags, vote_phase, (acc & sl.gmask) != 0); // on the products (a rotation can use the guard bit up); also orders the previous frame's LDS reads
            fast = FAST_OK && !bad;
        }
        if (fast) dit_round5_c<FAST_OK>(v, c, sl);
        else dit_round5_c<false>(v, c, sl);
#pragma unroll
        for (int q = 0; q < 32; ++q) wr_base[ROWX * q] = v[q];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = rd_base[ROWX * (j << RB)];
        if (fast) {
            dit_round_q<FAST_OK, 0>(v, t2, sl);
            dit_round_q<FAST_OK, 16>(v, t2, sl);
            dit_top16<FAST_OK>(v, wa16, wb16, sl);
        } else {
            dit_round_q<false, 0>(v, t2, sl);
            dit_round_q<false, 16>(v, t2, sl);
            dit_top16<false>(v, wa16, wb16, sl);
        }
        if (halves) { // HALVES order out: registers j and j + 16 (n1 bit 9) are one 8-byte store
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            v2u *d2 = reinterpret_cast<v2u *>(dst);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v2u w = {v[j], v[j + 16]};
                __builtin_nontemporal_store(w, at32(d2 + ((size_t)j << (RB + L2)), toff_l));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) __builtin_nontemporal_store(v[j], at32(dst + ((size_t)j << (RB + L2)), toff_l));
        }
    }
}

// ---- the 2-D scheme at N = 2^21 = 1024 x 2048, INVERSE, in TWO launches (round 5): k_rows2k_qtr + k_big2x_ci<., 11> -----------------------------------
// The mirror of k_rows2k_tr: ONE workgroup of 1024 threads takes the 16 row cores whose k1 are consecutive (k1 = brev6(r6) << 4 | kb: 64-byte pieces of
// X[k1 + 1024 k2]) -- 128 KiB, one workgroup per CU:
//   round 1  thread = (jj = p10..p5, kb), regs q = p4..p0: position p takes X[k1 + 1024 brev11(p)]; DIT STAGE 0..4 on wave-uniform twiddles (dit_round5_c)
//   LDS      row (p10..p5) << 4 | rev4(kb), column p4..p0; read back as wave = row k = rev4(kb), lane = (p10, p4..p0), register r = (p5, p9..p6)
//   stage 5  DIT STAGE 5 on the register pairs (j, j + 16), one twiddle per lane; v_permlane32_swap then exchanges p5 (register bit 4) with p10 (lane bit 5)
//   round 2  lane = p5..p0, regs j = p10..p6: DIT STAGE 6..10 (per-lane twiddles re-read per tile, quarter-turn sharing)
//   store    V[r = k << 6 | r6][n2 = 64 j + lane] into the scratch shape k_big2x_ci reads: [q = r4..r0][chunk = n2 >> 4][r5][k = r9..r6][l]: every wave
//            instruction writes four 64-byte pieces; the 16 waves of the block complete 1 KiB runs
// The next tile's 32 loads are issued right behind the transpose's LDS writes, as in k_rows2k_tr.  (int_ifftNk.vhd:183-341 for the 2048-point core; the
// scheme itself is this library's extension)
// L1 = log2 N1 (11: the 2048 x 2048 plan); ROWS_OUT: plain rows V[r][n2] out (256 contiguous bytes per wave instruction) for k_cols2k_ci
template <bool FAST_OK, int L1 = 10, bool ROWS_OUT = false>
__global__ __launch_bounds__(1024) void k_rows2k_qtr(const u32 *in, u32 *scr, const uint2 *__restrict__ twf, const Round5Consts c, size_t nframes, const Slice sl)
{
    extern __shared__ u32 lds[]; // 1024 rows x ROWY, then round 2's per-lane twiddles (DIT packing): 16 slots x 64 lanes
    const int tid = threadIdx.x, lane = tid & 63;
    const int k = __builtin_amdgcn_readfirstlane(tid >> 6); // the row of this wave (round 2)
    // STAGE 6 + b on register bit b of round 2, table index (jx << 6) | lane, the upper half of every stage as quarter turns: 1 + 1 + 2 + 4 + 8 slots, tile
    // invariant.  Round 2 holds the next tile's 32 loads and its own 32 values: the twiddles are read from here just in time, STAGE 10's eight only
    // after STAGE 6..9 are done (all 17 pairs in registers beside the two data sets: 52 spilled VGPRs)
    uint2 *const twl = reinterpret_cast<uint2 *>(lds + 1024 * ROWY);
    {
        const int slot = tid >> 6; // 0: STAGE 6, 1: STAGE 7, 2..3: STAGE 8, 4..7: STAGE 9, 8..15: STAGE 10
        const unsigned idx = slot == 0 ? 63u : slot == 1 ? 127u : slot < 4 ? 255u + ((unsigned)(slot - 2) << 6) : slot < 8 ? 511u + ((unsigned)(slot - 4) << 6)
                                                                                                                  : 1023u + ((unsigned)(slot - 8) << 6);
        uint2 w = twf[idx + (unsigned)lane];
        to_dit_packing(w.x, w.y);
        twl[tid] = w;
    }
    u32 wa5[4], wb5[4];
    {
        uint2 w = twf[31u + (unsigned)(lane & 31)]; // STAGE 5: index p4..p0
        to_dit_packing(w.x, w.y);
        wa5[0] = wa5[1] = wa5[2] = wa5[3] = w.x;
        wb5[0] = wb5[1] = wb5[2] = wb5[3] = w.y;
    }
    // transpose, read side: element (register r = (p5, p9..p6), lane = (p10, p4..p0)) <- row ((p10, p9..p6, p5) << 4) | k
    const u32 *const rd_base = lds + ROWY * ((((lane >> 5) << 5) << 4) | k) + (lane & 31);
    const int jj = tid >> 4, kb = tid & 15;
    const int krow = ((kb & 1) << 3) | ((kb & 2) << 1) | ((kb & 4) >> 1) | ((kb & 8) >> 3);
    u32 *const wr_base = lds + ROWY * ((jj << 4) | krow);
    const unsigned rjj = __brev((unsigned)jj) >> 26;
    constexpr int RR = L1 - 5; // 2^RR tiles per frame and partner
    static_assert(L1 == 10 || ROWS_OUT, "the [q][chunk] scratch shape is k_big2x_ci<., 11>'s");
    const unsigned toff2 = (rjj << L1) | (unsigned)kb;                                       // user side (round 1 thread)
    const unsigned toff = ROWS_OUT ? (unsigned)lane : (((unsigned)lane >> 4) << 9) | ((unsigned)k << 4) | ((unsigned)lane & 15u); // scratch side (round 2 thread): chunk = 4 j + (lane >> 4)
    __shared__ __attribute__((aligned(256))) u32 vote_flags[64];
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    // XCD-paired tiles: blocks b and b + 8 (same XCD: b & 7) take the two r6 partners (r6, r6 ^ 32) = the two 64-byte halves of every line of X at
    // the same time, so that the half a block does not use is an L2 hit for its partner (the grid is a multiple of 16).  One block taking both partners
    // one after the other: 251 Gsample/s; paired, plain loads: 264; paired, non-temporal loads: 273
    const unsigned slot = blockIdx.x & 7u, part = (blockIdx.x >> 3) & 1u;
    size_t t = blockIdx.x;
    auto tile_of = [&](size_t tt) { return (tt >> 4) * 8u + slot; }; // G: frame = G >> 5, r6 = part << 5 | (G & 31)
    bool have = (tile_of(t) >> RR) < nframes;
    u32 v[32];
    auto load_tile = [&](size_t G, int q0) { // (in two halves: the second one is issued once round 2's twiddles are dead)
        const unsigned r6 = (part << RR) | ((unsigned)G & ((1u << RR) - 1u));
        const u32 *src = in + ((G >> RR) << (L1 + 11)) + ((__brev(r6) >> (36 - L1)) << 4); // wave-uniform
        unsigned toff2_l = toff2;
        asm volatile("" : "+v"(toff2_l));
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q0 + q] = INTFFT_LD(at32(src + ((size_t)rev5c(q0 + q) << (L1 + 6)), toff2_l)); // position (jj << 5 | q) = X[k1 + 1024 brev11]
    };
    if (have) load_tile(tile_of(t), 0), load_tile(tile_of(t), 16);
    while (have) {
        const size_t ct = tile_of(t);
        t += gridDim.x;
        const size_t nt = tile_of(t);
        const bool have_next = (nt >> RR) < nframes;
        bool fast = false;
        {
            u32 acc = 0;
#pragma unroll
            for (int q = 0; q < 32; ++q) acc |= v[q] + sl.gbias;
            const bool bad = block_any(vote_flags, vote_phase, (acc & sl.gmask) != 0); // also orders the previous tile's LDS reads
            fast = FAST_OK && !bad;
        }
        // One body per extraction form from here to the stores (one branch per tile): with the two forms of round 2 behind separate branches inside
        // one body the allocator spills 50-60 VGPRs (either form alone: 103 / 105)
        const unsigned r6c = (part << RR) | ((unsigned)ct & ((1u << RR) - 1u)); // the tile's rows: r = k << (L1 - 4) | r6c
        u32 *dst = ROWS_OUT ? scr + ((ct >> RR) << (L1 + 11)) + ((size_t)(((unsigned)k << (L1 - 4)) | r6c) << 11)
                            : scr + ((ct >> 5) << 21) + ((size_t)((unsigned)ct & 31u) << 16) + (part << 8); // wave-uniform (scratch shape: q = r4..r0, r5 = part)
#define INTFFT_R2K_TILE(FX)                                                                                                                    \
    {                                                                                                                                          \
        dit_round5_c<FX>(v, c, sl);                                                                                                            \
        _Pragma("unroll") for (int q = 0; q < 32; ++q) wr_base[q] = v[q];                                                                      \
        asm volatile("" ::: "memory");                                                                                                         \
        if (have_next) load_tile(nt, 0); /* flies during round 2 and the stores below */                                                       \
        __syncthreads();                                                                                                                       \
        u32 w[32];                                                                                                                             \
        _Pragma("unroll") for (int r = 0; r < 32; ++r) w[r] = rd_base[ROWY * ((((r & 15) << 1) | (r >> 4)) << 4)];                             \
        u32 wa16[8], wb16[8];                                                                                                                  \
        RoundTwQ t1;                                                                                                                           \
        {                                                                                                                                      \
            const uint2 x0 = twl[lane], x1 = twl[64 + lane];                                                                                   \
            t1.wa1[0] = x0.x, t1.wb1[0] = x0.y, t1.wa2[0] = x1.x, t1.wb2[0] = x1.y;                                                            \
            _Pragma("unroll") for (int j2 = 0; j2 < 2; ++j2)                                                                                   \
            {                                                                                                                                  \
                const uint2 x = twl[64 * (2 + j2) + lane];                                                                                     \
                t1.wa4[j2] = x.x, t1.wb4[j2] = x.y;                                                                                            \
            }                                                                                                                                  \
            _Pragma("unroll") for (int j4 = 0; j4 < 4; ++j4)                                                                                   \
            {                                                                                                                                  \
                const uint2 x = twl[64 * (4 + j4) + lane];                                                                                     \
                t1.wa8[j4] = x.x, t1.wb8[j4] = x.y;                                                                                            \
            }                                                                                                                                  \
        }                                                                                                                                      \
        /* STAGE 5: pairs (j, j + 16) */                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 16; j += 4)                                                                                      \
            group4_dit<FX, false, 0, true>(w[j], w[j + 16], w[j + 1], w[j + 17], w[j + 2], w[j + 18], w[j + 3], w[j + 19], wa5, wb5, sl);      \
        swap_guard(*reinterpret_cast<u32(*)[16]>(&w[0]));                                                                                      \
        swap_guard(*reinterpret_cast<u32(*)[16]>(&w[16]));                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) swap32(w[j], w[j + 16]); /* register bit 4: p5 -> p10 */                                \
        dit_round_q<FX, 0>(w, t1, sl);                                                                                                         \
        dit_round_q<FX, 16>(w, t1, sl);                                                                                                        \
        asm volatile("" ::: "memory"); /* STAGE 10's twiddles only now */                                                                      \
        _Pragma("unroll") for (int j8 = 0; j8 < 8; ++j8)                                                                                       \
        {                                                                                                                                      \
            const uint2 x = twl[64 * (8 + j8) + lane];                                                                                         \
            wa16[j8] = x.x, wb16[j8] = x.y;                                                                                                    \
        }                                                                                                                                      \
        dit_top16<FX>(w, wa16, wb16, sl);                                                                                                      \
        asm volatile("" ::: "memory");                                                                                                         \
        if (have_next) load_tile(nt, 16); /* the other half: the twiddles are dead */                                                          \
        unsigned toff_l = toff;                                                                                                                \
        asm volatile("" : "+v"(toff_l));                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 32; ++j) *at32(dst + ((size_t)j << (ROWS_OUT ? 6 : 11)), toff_l) = w[j];                         \
    }
        if (fast) INTFFT_R2K_TILE(FAST_OK)
        else {
            if (sl.wd != 16) wrap_inputs(v, sl.wd);
            INTFFT_R2K_TILE(false)
        }
#undef INTFFT_R2K_TILE
        have = have_next;
    }
}

// ---- the 2-D scheme at N = 2^22 = 2048 x 2048, INVERSE, in TWO launches (round 5): k_rows2k_qtr<., 11, true> + k_cols2k_ci ------------------------------
// The mirror of k_cols2k_c on its tile (2048 rows r x 16 columns n2, r = the DIT position of k1): plain rows V[r][n2] in (64-byte pieces, XCD-paired), T = V conj(W)
// from the forward plan's table, DIT STAGE 0..4 on regs r4..r0 (thread = (jj = r10..r5, kb)), the LDS transpose, then wave = r3..r0, lane = (r5 -> r10 after the
// swap, r4, kb): STAGE 5, v_permlane32_swap, STAGE 6..10 on per-thread twiddles parked in LDS (index (j << 6) | r5..r0), x[n1 N2 + n2] out as 64-byte pieces
// (non-temporal; the partner block fills the other half of every line).  Tile body per extraction form as in k_rows2k_qtr.  Measured variants (Gsample/s of the
// plan): plain loads 179, non-temporal loads 188 (shipped), the whole next tile requested after the last stage 178-185, all 32 table entries in one batch 182,
// the table entries of the next tile requested with its data 86 spilled VGPRs.
template <bool FAST_OK>
__global__ __launch_bounds__(1024) void k_cols2k_ci(const u32 *scr, u32 *out, const uint2 *__restrict__ twf, const Round5Consts c, const u32 *__restrict__ tw2d, size_t nframes,
                                                    const Slice sl)
{
    extern __shared__ u32 lds[]; // 1024 rows x ROWY, then round 2's twiddles (DIT packing): 16 slots x 64 values of r5..r0
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // r3..r0 of round 2
    const unsigned t6 = (((unsigned)lane >> 4) << 4) | (unsigned)wv;  // r5..r0 of round 2
    uint2 *const twl = reinterpret_cast<uint2 *>(lds + 1024 * ROWY);
    {
        const int slot = tid >> 6; // 0: STAGE 6, 1: STAGE 7, 2..3: STAGE 8, 4..7: STAGE 9, 8..15: STAGE 10; entry = r5..r0
        const unsigned idx = slot == 0 ? 63u : slot == 1 ? 127u : slot < 4 ? 255u + ((unsigned)(slot - 2) << 6) : slot < 8 ? 511u + ((unsigned)(slot - 4) << 6)
                                                                                                                  : 1023u + ((unsigned)(slot - 8) << 6);
        uint2 w = twf[idx + (unsigned)lane];
        to_dit_packing(w.x, w.y);
        twl[tid] = w;
    }
    u32 wa5[4], wb5[4];
    {
        uint2 w = twf[31u + (t6 & 31u)]; // STAGE 5: index r4..r0
        to_dit_packing(w.x, w.y);
        wa5[0] = wa5[1] = wa5[2] = wa5[3] = w.x;
        wb5[0] = wb5[1] = wb5[2] = wb5[3] = w.y;
    }
    // transpose, read side: element (register r = (r5, r9..r6), lane = (r10, r4, kb), wave r3..r0) <- row ((r10, r9..r6, r5) << 4) | kb, column r4..r0
    const u32 *const rd_base = lds + ROWY * ((((lane >> 5) << 5) << 4) | (lane & 15)) + ((((lane >> 4) & 1) << 4) | wv);
    const int jj = tid >> 4, kb = tid & 15;
    u32 *const wr_base = lds + ROWY * ((jj << 4) | kb);
    const unsigned soff = ((unsigned)jj << 16) | (unsigned)kb;                    // scratch side (round 1): row jj << 5 (+ q), column kb
    const unsigned twoff = ((unsigned)jj << 9) + (unsigned)kb;                    // table: [r = jj << 5 | q][kb]
    const unsigned loff = (((unsigned)lane >> 4) << 15) | ((unsigned)lane & 15u); // user side (round 2): rows r5 r4 (x 2048 samples), column kb
    __shared__ __attribute__((aligned(256))) u32 vote_flags[64];
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    const unsigned slot = blockIdx.x & 7u, part = (blockIdx.x >> 3) & 1u;
    size_t t = blockIdx.x;
    auto tile_of = [&](size_t tt) { return (tt >> 4) * 8u + slot; }; // G: frame = G >> 6, chunk = (G & 63) * 2 + part
    bool have = (tile_of(t) >> 6) < nframes;
    u32 v[32];
    auto load_tile = [&](size_t G, int q0) {
        const unsigned chunk = ((unsigned)G & 63u) * 2u + part;
        const u32 *src = scr + ((G >> 6) << 22) + chunk * 16u; // wave-uniform
        unsigned so = soff;
        asm volatile("" : "+v"(so));
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q0 + q] = INTFFT_LD(at32(src + ((size_t)(q0 + q) << 11), so));
    };
    if (have) load_tile(tile_of(t), 0), load_tile(tile_of(t), 16);
    while (have) {
        const size_t ct = tile_of(t);
        t += gridDim.x;
        const size_t nt = tile_of(t);
        const bool have_next = (nt >> 6) < nframes;
        const unsigned chunk = ((unsigned)ct & 63u) * 2u + part;
        // T = V conj(W): the DIT butterfly's multiplier with Wc = (wr, wi) -- the table entry itself -- and Wd = (-wi, wr); plain 16-bit results
        {
            const u32 *const twu = tw2d + ((size_t)chunk << 15);
            unsigned two = twoff;
            asm volatile("" : "+v"(two));
            const gptr_t<const u32> twq = at32(twu, two);
            const v2s mp = {-1, 1};
#pragma unroll
            for (int q0 = 0; q0 < 32; q0 += 8) {
                u32 tw[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) tw[i] = twq[16 * (q0 + i)];
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    const u32 wb0 = as_u32(as_v2s(__builtin_amdgcn_alignbit(tw[q], tw[q], 16)) * mp);
                    const u32 wb1 = as_u32(as_v2s(__builtin_amdgcn_alignbit(tw[q + 1], tw[q + 1], 16)) * mp);
                    u32 y0, y1;
                    mul2x<16, false>(v[q0 + q], v[q0 + q], tw[q], wb0, v[q0 + q + 1], v[q0 + q + 1], tw[q + 1], wb1, sl.off_y, sl.sel, y0, y1, sl.wd);
                    v[q0 + q] = y0, v[q0 + q + 1] = y1;
                }
            }
        }
        bool fast = false;
        {
            u32 acc = 0;
#pragma unroll
            for (int q = 0; q < 32; ++q) acc |= v[q] + sl.gbias;
            const bool bad = block_any(vote_flags, vote_phase, (acc & sl.gmask) != 0); // on the products; also orders the previous tile's LDS reads
            fast = FAST_OK && !bad;
        }
        u32 *dst = out + ((ct >> 6) << 22) + ((size_t)wv << 11) + chunk * 16u; // wave-uniform
#define INTFFT_C2KI_TILE(FX)                                                                                                                   \
    {                                                                                                                                          \
        dit_round5_c<FX>(v, c, sl);                                                                                                            \
        _Pragma("unroll") for (int q = 0; q < 32; ++q) wr_base[q] = v[q];                                                                      \
        asm volatile("" ::: "memory");                                                                                                         \
        if (have_next) load_tile(nt, 0); /* flies during round 2 and the stores below */                                                       \
        __syncthreads();                                                                                                                       \
        u32 w[32];                                                                                                                             \
        _Pragma("unroll") for (int r = 0; r < 32; ++r) w[r] = rd_base[ROWY * ((((r & 15) << 1) | (r >> 4)) << 4)];                             \
        u32 wa16[8], wb16[8];                                                                                                                  \
        RoundTwQ t1;                                                                                                                           \
        {                                                                                                                                      \
            const uint2 x0 = twl[t6], x1 = twl[64 + t6];                                                                                       \
            t1.wa1[0] = x0.x, t1.wb1[0] = x0.y, t1.wa2[0] = x1.x, t1.wb2[0] = x1.y;                                                            \
            _Pragma("unroll") for (int j2 = 0; j2 < 2; ++j2)                                                                                   \
            {                                                                                                                                  \
                const uint2 x = twl[64 * (2 + j2) + t6];                                                                                       \
                t1.wa4[j2] = x.x, t1.wb4[j2] = x.y;                                                                                            \
            }                                                                                                                                  \
            _Pragma("unroll") for (int j4 = 0; j4 < 4; ++j4)                                                                                   \
            {                                                                                                                                  \
                const uint2 x = twl[64 * (4 + j4) + t6];                                                                                       \
                t1.wa8[j4] = x.x, t1.wb8[j4] = x.y;                                                                                            \
            }                                                                                                                                  \
        }                                                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 16; j += 4)                                                                                      \
            group4_dit<FX, false, 0, true>(w[j], w[j + 16], w[j + 1], w[j + 17], w[j + 2], w[j + 18], w[j + 3], w[j + 19], wa5, wb5, sl);      \
        swap_guard(*reinterpret_cast<u32(*)[16]>(&w[0]));                                                                                      \
        swap_guard(*reinterpret_cast<u32(*)[16]>(&w[16]));                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) swap32(w[j], w[j + 16]); /* register bit 4: r5 -> r10 */                                \
        dit_round_q<FX, 0>(w, t1, sl);                                                                                                         \
        dit_round_q<FX, 16>(w, t1, sl);                                                                                                        \
        asm volatile("" ::: "memory");                                                                                                         \
        _Pragma("unroll") for (int j8 = 0; j8 < 8; ++j8)                                                                                       \
        {                                                                                                                                      \
            const uint2 x = twl[64 * (8 + j8) + t6];                                                                                           \
            wa16[j8] = x.x, wb16[j8] = x.y;                                                                                                    \
        }                                                                                                                                      \
        dit_top16<FX>(w, wa16, wb16, sl);                                                                                                      \
        asm volatile("" ::: "memory");                                                                                                         \
        if (have_next) load_tile(nt, 16);                                                                                                      \
        unsigned lo = loff;                                                                                                                    \
        asm volatile("" : "+v"(lo));                                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 32; ++j) __builtin_nontemporal_store(w[j], at32(dst + ((size_t)j << 17), lo));                   \
    }
        if (fast) INTFFT_C2KI_TILE(FAST_OK)
        else INTFFT_C2KI_TILE(false)
#undef INTFFT_C2KI_TILE
        have = have_next;
    }
}

// N = 2^22 inverse: the 2048-point row cores (k_rows2k_qtr<., 11, true>: plain rows out) + the conj multiplier and the 2048-point column cores (k_cols2k_ci)
hipError_t launch_fused2d_inv_2k2k(int twd, const u32 *pin, u32 *pout, u32 *scr, const uint2 *tw16r, const int2 *h_tw, const u32 *tw2d, size_t nframes, hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    Round5Consts c;
    auto pk = [&](int idx, u32 &wa, u32 &wb) { // DIT packing: Wc = (wr, wi), Wd = (-wi, wr)
        const int2 w = h_tw[idx];
        wa = ((u32)w.x & 0xFFFFu) | ((u32)w.y << 16);
        wb = ((u32)(-w.y) & 0xFFFFu) | ((u32)w.x << 16);
    };
    for (int i = 0; i < 16; ++i) pk(15 + i, c.wa4[i], c.wb4[i]);
    for (int i = 0; i < 8; ++i) pk(7 + i, c.wa3[i], c.wb3[i]);
    for (int i = 0; i < 4; ++i) pk(3 + i, c.wa2[i], c.wb2[i]);
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fx = twd == 16 && allow_fast;
    const size_t ldsr = (size_t)1024 * ROWY * sizeof(u32) + 1024 * sizeof(uint2);
    const size_t ntiles = nframes << 7, cap16 = (size_t)device_cus() / 16 * 16; // both kernels: 128 XCD-paired tiles per frame, the grid a multiple of 16
    const unsigned g = (unsigned)std::min(ntiles, cap16);
    if (fx) {
        allow_max_lds(kptr(k_rows2k_qtr<true, 11, true>));
        allow_max_lds(kptr(k_cols2k_ci<true>));
        hipLaunchKernelGGL((k_rows2k_qtr<true, 11, true>), dim3(g), dim3(1024), ldsr, stream, pin, scr, tw16r, c, nframes, sl);
        hipLaunchKernelGGL((k_cols2k_ci<true>), dim3(g), dim3(1024), ldsr, stream, scr, pout, tw16r, c, tw2d, nframes, sl);
    } else {
        allow_max_lds(kptr(k_rows2k_qtr<false, 11, true>));
        allow_max_lds(kptr(k_cols2k_ci<false>));
        hipLaunchKernelGGL((k_rows2k_qtr<false, 11, true>), dim3(g), dim3(1024), ldsr, stream, pin, scr, tw16r, c, nframes, sl);
        hipLaunchKernelGGL((k_cols2k_ci<false>), dim3(g), dim3(1024), ldsr, stream, scr, pout, tw16r, c, tw2d, nframes, sl);
    }
    return hipGetLastError();
}

// N = 2^20 inverse: pass QB (row cores) + the multiplier and the column cores.  tw1k / h_tw1k: the packed / host tables of the 1024-point cores
hipError_t launch_fused2d_inv(int twd, const u32 *pin, u32 *pout, u32 *scr, const uint2 *tw1k, const int2 *h_tw1k, const u32 *tw2d, size_t nframes,
                              int halves, hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    Round5Consts c;
    auto pk = [&](int idx, u32 &wa, u32 &wb) { // DIT packing: Wc = (wr, wi), Wd = (-wi, wr)
        const int2 w = h_tw1k[idx];
        wa = ((u32)w.x & 0xFFFFu) | ((u32)w.y << 16);
        wb = ((u32)(-w.y) & 0xFFFFu) | ((u32)w.x << 16);
    };
    for (int i = 0; i < 16; ++i) pk(15 + i, c.wa4[i], c.wb4[i]);
    for (int i = 0; i < 8; ++i) pk(7 + i, c.wa3[i], c.wb3[i]);
    for (int i = 0; i < 4; ++i) pk(3 + i, c.wa2[i], c.wb2[i]);
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fx = twd == 16 && allow_fast;
    const size_t ldsa = (size_t)1024 * ROWX * sizeof(u32), ldsb = (size_t)512 * ROWY * sizeof(u32);
    const size_t ntiles = nframes << 6, capb = ((size_t)device_cus() * 2 + 15) / 16 * 16;
    const unsigned gb = (unsigned)(ntiles < capb ? (ntiles + 15) / 16 * 16 : capb);
    const unsigned groups = (unsigned)(nframes < 64 ? nframes : 64);
    if (fx) {
        allow_max_lds(kptr(k_big2x_qb<20, true>));
        allow_max_lds(kptr(k_big2x_ci<true>));
        hipLaunchKernelGGL((k_big2x_qb<20, true>), dim3(gb), dim3(512), ldsb, stream, pin, scr, tw1k, c, nframes, sl);
        hipLaunchKernelGGL((k_big2x_ci<true>), dim3(64u * groups), dim3(512), ldsa, stream, scr, pout, tw1k, c, tw2d, nframes, groups, sl, halves);
    } else {
        allow_max_lds(kptr(k_big2x_qb<20, false>));
        allow_max_lds(kptr(k_big2x_ci<false>));
        hipLaunchKernelGGL((k_big2x_qb<20, false>), dim3(gb), dim3(512), ldsb, stream, pin, scr, tw1k, c, nframes, sl);
        hipLaunchKernelGGL((k_big2x_ci<false>), dim3(64u * groups), dim3(512), ldsa, stream, scr, pout, tw1k, c, tw2d, nframes, groups, sl, halves);
    }
    return hipGetLastError();
}

// N = 2^22 .. 2^24 inverse, last of three launches: the conj multiplier + the 1024-point column cores on the plain rows [r][n2] an N2-point inverse sub-plan left
hipError_t launch_fused2d_inv_cols(int l2, int twd, const u32 *rows, u32 *pout, const uint2 *tw1k, const int2 *h_tw1k, const u32 *tw2d, size_t nframes, int halves,
                                   hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    Round5Consts cc;
    auto pk = [&](int idx, u32 &wa, u32 &wb) { // DIT packing: Wc = (wr, wi), Wd = (-wi, wr)
        const int2 w = h_tw1k[idx];
        wa = ((u32)w.x & 0xFFFFu) | ((u32)w.y << 16);
        wb = ((u32)(-w.y) & 0xFFFFu) | ((u32)w.x << 16);
    };
    for (int i = 0; i < 16; ++i) pk(15 + i, cc.wa4[i], cc.wb4[i]);
    for (int i = 0; i < 8; ++i) pk(7 + i, cc.wa3[i], cc.wb3[i]);
    for (int i = 0; i < 4; ++i) pk(3 + i, cc.wa2[i], cc.wb2[i]);
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fx = twd == 16 && allow_fast;
    const size_t ldsa = (size_t)1024 * ROWX * sizeof(u32);
    const unsigned chunks = 1u << (l2 - 4);
    const unsigned groups = (unsigned)std::max<size_t>(1, std::min<size_t>(nframes, (size_t)4096 / chunks));
#define INTFFT_CI_ROWS(LL, FX)                                                                                                                  \
    {                                                                                                                                           \
        allow_max_lds(kptr(k_big2x_ci<FX, LL, true>));                                                                                          \
        hipLaunchKernelGGL((k_big2x_ci<FX, LL, true>), dim3(chunks * groups), dim3(512), ldsa, stream, rows, pout, tw1k, cc, tw2d, nframes, groups, sl, halves); \
    }
    if (l2 == 11) {
        if (fx) INTFFT_CI_ROWS(11, true) else INTFFT_CI_ROWS(11, false)
    } else if (l2 == 12) {
        if (fx) INTFFT_CI_ROWS(12, true) else INTFFT_CI_ROWS(12, false)
    } else if (l2 == 13) {
        if (fx) INTFFT_CI_ROWS(13, true) else INTFFT_CI_ROWS(13, false)
    } else if (l2 == 14) {
        if (fx) INTFFT_CI_ROWS(14, true) else INTFFT_CI_ROWS(14, false)
    } else return hipErrorInvalidValue;
#undef INTFFT_CI_ROWS
    return hipGetLastError();
}

// N = 2^21 inverse: the 2048-point row cores (k_rows2k_qtr; tw16r / h_tw2k: the packed / host tables of the 2048-point core) + the multiplier and the
// 1024-point column cores (k_big2x_ci<., 11>; tw1k / h_tw1k)
hipError_t launch_fused2d_inv21(int twd, const u32 *pin, u32 *pout, u32 *scr, const uint2 *tw1k, const int2 *h_tw1k, const uint2 *tw16r, const int2 *h_tw2k,
                                const u32 *tw2d, size_t nframes, int halves, hipStream_t stream)
{
    if (nframes == 0) return hipSuccess;
    auto consts = [](const int2 *h, Round5Consts &c) {
        auto pk = [&](int idx, u32 &wa, u32 &wb) { // DIT packing: Wc = (wr, wi), Wd = (-wi, wr)
            const int2 w = h[idx];
            wa = ((u32)w.x & 0xFFFFu) | ((u32)w.y << 16);
            wb = ((u32)(-w.y) & 0xFFFFu) | ((u32)w.x << 16);
        };
        for (int i = 0; i < 16; ++i) pk(15 + i, c.wa4[i], c.wb4[i]);
        for (int i = 0; i < 8; ++i) pk(7 + i, c.wa3[i], c.wb3[i]);
        for (int i = 0; i < 4; ++i) pk(3 + i, c.wa2[i], c.wb2[i]);
    };
    Round5Consts cr, cc;
    consts(h_tw2k, cr);
    consts(h_tw1k, cc);
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fx = twd == 16 && allow_fast;
    const size_t ldsa = (size_t)1024 * ROWX * sizeof(u32), ldsr = (size_t)1024 * ROWY * sizeof(u32) + 1024 * sizeof(uint2);
    const size_t ntiles = nframes << 6; // XCD-paired: blocks b, b + 8 take the two r6 partners; the grid is a multiple of 16
    const unsigned gr = (unsigned)std::min<size_t>(ntiles, (size_t)device_cus() / 16 * 16);
    const unsigned groups = (unsigned)(nframes < 32 ? nframes : 32);
    if (fx) {
        allow_max_lds(kptr(k_rows2k_qtr<true>));
        allow_max_lds(kptr(k_big2x_ci<true, 11>));
        hipLaunchKernelGGL((k_rows2k_qtr<true>), dim3(gr), dim3(1024), ldsr, stream, pin, scr, tw16r, cr, nframes, sl);
        hipLaunchKernelGGL((k_big2x_ci<true, 11>), dim3(128u * groups), dim3(512), ldsa, stream, scr, pout, tw1k, cc, tw2d, nframes, groups, sl, halves);
    } else {
        allow_max_lds(kptr(k_rows2k_qtr<false>));
        allow_max_lds(kptr(k_big2x_ci<false, 11>));
        hipLaunchKernelGGL((k_rows2k_qtr<false>), dim3(gr), dim3(1024), ldsr, stream, pin, scr, tw16r, cr, nframes, sl);
        hipLaunchKernelGGL((k_big2x_ci<false, 11>), dim3(128u * groups), dim3(512), ldsa, stream, scr, pout, tw1k, cc, tw2d, nframes, groups, sl, halves);
    }
    return hipGetLastError();
}

hipError_t launch_big2x_inv(int log2n, bool fx, const u32 *pin, u32 *pout, u32 *scr, const uint2 *tw16f, const int2 *h_tw, size_t nframes, const Slice &sl,
                            int halves, hipStream_t stream, bool in_bitrev)
{
    if (nframes == 0) return hipSuccess;
    Round5Consts c;
    auto pk = [&](int idx, u32 &wa, u32 &wb) { // DIT packing: Wc = (wr, wi), Wd = (-wi, wr)
        const int2 w = h_tw[idx];
        wa = ((u32)w.x & 0xFFFFu) | ((u32)w.y << 16);
        wb = ((u32)(-w.y) & 0xFFFFu) | ((u32)w.x << 16);
    };
    for (int i = 0; i < 16; ++i) pk(15 + i, c.wa4[i], c.wb4[i]);
    for (int i = 0; i < 8; ++i) pk(7 + i, c.wa3[i], c.wb3[i]);
    for (int i = 0; i < 4; ++i) pk(3 + i, c.wa2[i], c.wb2[i]);
#define INTFFT_2XQ_LAUNCH(LL, FX, RD)                                                                                              \
    {                                                                                                                              \
        constexpr int RB = LL - 15, TT = 16 << RB;                                                                                 \
        const size_t ldsa = (size_t)(32 << RB) * ROWX * sizeof(u32) + (LL == 20 ? 16 : 8) * 16 * sizeof(uint2);                    \
        const size_t ldsb = (size_t)512 * ROWY * sizeof(u32);                                                                      \
        allow_max_lds(kptr(k_big2x_qa<LL, FX, RD>));                                                                                   \
        allow_max_lds(kptr(k_big2x_qb<LL, FX, false, RD>));                                                                                   \
        allow_max_lds(kptr(k_big2x_qb<LL, FX, true, RD>));                                                                             \
        const size_t ntiles = nframes << (LL - 14), capb = ((size_t)device_cus() * 2 + 15) / 16 * 16;                              \
        const unsigned gb = (unsigned)(ntiles < capb ? (ntiles + 15) / 16 * 16 : capb);                                            \
        if (in_bitrev) hipLaunchKernelGGL((k_big2x_qb<LL, FX, true, RD>), dim3(gb), dim3(512), ldsb, stream, pin, scr, tw16f, c, nframes, sl); \
        else hipLaunchKernelGGL((k_big2x_qb<LL, FX, false, RD>), dim3(gb), dim3(512), ldsb, stream, pin, scr, tw16f, c, nframes, sl);         \
        const unsigned groups = (unsigned)(nframes < 64 ? nframes : 64);                                                           \
        hipLaunchKernelGGL((k_big2x_qa<LL, FX, RD>), dim3(64u * groups), dim3(TT), ldsa, stream, scr, pout, tw16f, nframes, groups, sl, halves); \
    }
    // RNDMODE = 1 (sl.round; 2 = on narrow data): the exact-path kernels in their ROUND instantiations
    if (log2n == 20) {
        if (sl.round == 1) INTFFT_2XQ_LAUNCH(20, false, 1) else if (sl.round == 2) INTFFT_2XQ_LAUNCH(20, false, 2)
        else if (fx) INTFFT_2XQ_LAUNCH(20, true, 0) else INTFFT_2XQ_LAUNCH(20, false, 0)
    } else {
        if (sl.round == 1) INTFFT_2XQ_LAUNCH(19, false, 1) else if (sl.round == 2) INTFFT_2XQ_LAUNCH(19, false, 2)
        else if (fx) INTFFT_2XQ_LAUNCH(19, true, 0) else INTFFT_2XQ_LAUNCH(19, false, 0)
    }
#undef INTFFT_2XQ_LAUNCH
    return hipGetLastError();
}

// the quarter-turn relation both passes rely on (stages 5 .. L-1), checked on the plan's generated tables (host copy)
bool big2x_tables_ok(int log2n, const int2 *h_tw, int twd)
{
    for (int s = 5; s < log2n; ++s) {
        const int2 *t = h_tw + ((size_t)1 << s) - 1;
        const size_t h = (size_t)1 << (s - 1);
        for (size_t k = 0; k < h; ++k) {
            const int neg = (int)(((long long)(-t[k].x) << (64 - twd)) >> (64 - twd));
            if (t[k + h].x != t[k].y || t[k + h].y != neg) return false;
            if (t[k].x == -32768 || t[k].y == -32768) return false; // the inverse negates packed 16-bit twiddle halves
        }
    }
    return true;
}

bool big2x_supported(int log2n) { return (log2n == 19 || log2n == 20) && !diag_env("INTFFT_NO_BIG2X"); }

const char *big2x_kernel_name() { return "k_big2x_a/k_big2x_b"; }

hipError_t launch_big2x(int log2n, bool fx, const u32 *pin, u32 *pout, u32 *scr, const uint2 *tw16f, const int2 *h_tw, size_t nframes, const Slice &sl,
                        int halves, hipStream_t stream, bool out_bitrev)
{
    if (nframes == 0) return hipSuccess;
    Round5Consts c;
    auto pk = [&](int idx, u32 &wa, u32 &wb) {
        const int2 w = h_tw[idx];
        wa = ((u32)w.x & 0xFFFFu) | ((u32)(-w.y) << 16);
        wb = ((u32)w.y & 0xFFFFu) | ((u32)w.x << 16);
    };
    for (int i = 0; i < 16; ++i) pk(15 + i, c.wa4[i], c.wb4[i]);
    for (int i = 0; i < 8; ++i) pk(7 + i, c.wa3[i], c.wb3[i]);
    for (int i = 0; i < 4; ++i) pk(3 + i, c.wa2[i], c.wb2[i]);
#define INTFFT_2XB_SHIFT 1 /* a block of pass B takes both partner tiles */
#define INTFFT_2XA_LAUNCH(LL, FX, RD)                                                                                                                              \
    if (LL == 20 && RD == 0 && !halves && full20) { /* one pipelined workgroup per CU on full lines (k_big2x_a1) */                                           \
        const size_t lds20 = (size_t)1024 * 33 * sizeof(u32) + 16 * 32 * sizeof(uint2);                                                                       \
        const size_t nt = nframes << 5, capa = (size_t)device_cus();                                                                                          \
        allow_max_lds(kptr(k_big2x_a1<FX>));                                                                                                                   \
        hipLaunchKernelGGL((k_big2x_a1<FX>), dim3((unsigned)(nt < capa ? nt : capa)), dim3(1024), lds20, stream, pin, scr, tw16f, nframes, sl);                \
    } else if (LL == 19 && full19) {                                                                                                                                 \
        constexpr int CB19 = LL == 19 ? 5 : 4; /* (instantiated for LL = 19 only) */                                                                            \
        const size_t lds19 = ((size_t)(32 << RB) * 33 + 1) * sizeof(u32) + 8 * 32 * sizeof(uint2);                                                             \
        allow_max_lds(kptr(k_big2x_a<LL, FX, RD, CB19>));                                                                                                      \
        hipLaunchKernelGGL((k_big2x_a<LL, FX, RD, CB19>), dim3(32u * groups), dim3(32 << RB), lds19, stream, pin, scr, tw16f, nframes, groups, sl, halves);   \
    } else hipLaunchKernelGGL((k_big2x_a<LL, FX, RD>), dim3(64u * groups), dim3(TT), ldsa, stream, pin, scr, tw16f, nframes, groups, sl, halves);
#define INTFFT_2X_LAUNCH(LL, FX, RD)                                                                                               \
    {                                                                                                                              \
        constexpr int RB = LL - 15, TT = 16 << RB;                                                                                 \
        const size_t ldsa = (size_t)(32 << RB) * ROWX * sizeof(u32) + (LL == 20 ? 16 : 8) * 16 * sizeof(uint2);                    \
        const size_t ldsb = (size_t)512 * ROWY * sizeof(u32);                                                                      \
        allow_max_lds(kptr(k_big2x_a<LL, FX, RD>));                                                                                    \
        allow_max_lds(kptr(k_big2x_b<LL, FX, false, RD>));                                                                                    \
        allow_max_lds(kptr(k_big2x_b<LL, FX, true, RD>));                                                                              \
        /* frame groups: 64 = every block takes ONE tile of a 64-frame chunk.  The partner blocks b, b + 8 then start together  \
           (per-XCD dispatch order) instead of drifting apart over a frame walk: FETCH_SIZE 387 MB against 436 MB per 2^26      \
           samples (268 MB ideal), 270 against 262 Gsample/s; the per-block twiddle parking is 16 loads of 16 threads */         \
        const size_t cap = 64;                                                                                                     \
        const unsigned groups = (unsigned)(nframes < cap ? nframes : cap);                                                         \
        const bool full19 = diag_env("INTFFT_2XA_HALF19") == nullptr; /* A/B: the half-line tiles of round 4 at N = 2^19 */        \
        const bool full20 = diag_env("INTFFT_2XA_FULL20") != nullptr; /* experiment: full-line pass A at N = 2^20 */               \
        (void)full19, (void)full20;                                                                                                \
        INTFFT_2XA_LAUNCH(LL, FX, RD)                                                                                                \
        const size_t ntiles = nframes << (LL - 14) >> INTFFT_2XB_SHIFT, capb = ((size_t)device_cus() * 2 + 15) / 16 * 16;          \
        const unsigned gb = (unsigned)(ntiles < capb ? (ntiles + 15) / 16 * 16 : capb);                                            \
        if (out_bitrev) hipLaunchKernelGGL((k_big2x_b<LL, FX, true, RD>), dim3(gb), dim3(512), ldsb, stream, scr, pout, tw16f, c, nframes, sl, 0); \
        else hipLaunchKernelGGL((k_big2x_b<LL, FX, false, RD>), dim3(gb), dim3(512), ldsb, stream, scr, pout, tw16f, c, nframes, sl, 0);      \
    }
    if (log2n == 20) {
        if (sl.round == 1) INTFFT_2X_LAUNCH(20, false, 1) else if (sl.round == 2) INTFFT_2X_LAUNCH(20, false, 2)
        else if (fx) INTFFT_2X_LAUNCH(20, true, 0) else INTFFT_2X_LAUNCH(20, false, 0)
    } else {
        if (sl.round == 1) INTFFT_2X_LAUNCH(19, false, 1) else if (sl.round == 2) INTFFT_2X_LAUNCH(19, false, 2)
        else if (fx) INTFFT_2X_LAUNCH(19, true, 0) else INTFFT_2X_LAUNCH(19, false, 0)
    }
#undef INTFFT_2X_LAUNCH
    return hipGetLastError();
}

} // namespace intfft
