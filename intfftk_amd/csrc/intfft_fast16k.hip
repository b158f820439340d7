// intfft_fast16k.hip -- ONE-pass block kernels for N = 8192 and N = 16384: int_fftNk / int_ifftNk / int_fft_ifft_pair with NFFT = 13, 14, DATA_WIDTH = 16
// (or 9 .. 15 in int16 containers), TWDL_WIDTH <= 16, scaled-truncate, natural order in and out
// (src/vhdl/fft/int_fftNk.vhd:75,184-207: NFFT is a free generic; a 32 / 64 KiB frame fits one workgroup's LDS, so these lengths need not
// take the two HBM passes of the N >= 2^15 plans).  Same packed arithmetic as intfft_fast1024.hip / intfft_fast4096.hip (intfft_pk16.hpp).
//
// One workgroup of 2^(L-5) threads owns one frame, 32 packed (re | im << 16) samples per thread, persistent loop over frames.
// L stages = three in-register rounds (5 + 5 + 4 stages at L = 14; 5 + 4 + 4 at L = 13) with two block-wide LDS transposes:
//
//   layout A  reg j = n(L-1)..n(L-5), thread = n(L-6)..n0              DIF L-1..L-5      / DIT L-5..L-1      per-thread twiddles, re-read per frame
//   layout B  reg q = n8..n4,         thread = (jx = n(L-1)..n9, l = n3..n0)   DIF 8..4 (L = 13: 7..4, n8 rides along) / DIT 4..8   twiddles per l, parked in LDS
//   layout C  reg r = n4..n0,         thread t3 = rev(n(L-1)..n5)      DIF 3..0 / DIT 0..3 on the two 16-register halves (n4 rides along), wave-uniform twiddles
//
// The forward core stores from layout C with the bit reversal (int_bitrev_order / outbuf_half_path) folded into the thread mapping:
// X[brev_L(n)] = out[rev5(r) << (L-5) | t3], every store instruction 256 contiguous bytes per wave; the inverse core loads the same way.
// Layout A <-> B goes through LDS rows (row = n(L-1)..n4, 17 dwords apart), B <-> C through rows of 32 + 1 dwords indexed by
// R = (the five low bits of t3) | (the other bits of n8..n5 in natural order) << 5: the 32 lanes of a b32 access then hit 32 banks on
// the C side and at most two lanes share one on the B side.  Both transposes share one region (4 barriers per frame).
// Twiddles: quarter-turn sharing as in intfft_big2x.hip (W[k + 2^(s-1)] = (W[k].im, -W[k].re), rom_twiddle_int.vhd:177-183, checked by the
// planner on the plan's own tables -- STAGE 11, 12, 13 are Taylor stages, row_twiddle_tay.vhd:123-268).
#include "intfft_pk16.hpp"

#include <cstdlib>

namespace intfft {

constexpr int ROWP = 17; // transpose A <-> B: LDS row stride in dwords (16 columns + 1)
constexpr int ROWQ = 33; // transpose B <-> C: 32 columns + 1

// DIF stages 3, 2, 1, 0 on v[B .. B+15] (reg = n3..n0), wave-uniform twiddles; P0: PREMASK of the inputs (0xF: they hold X >> 1)
template <bool FASTX, int B, int P0, int ROUND = 0> __device__ __forceinline__ void dif_round4_c(u32 (&v)[32], const RoundCConsts &c, const Slice &sl)
{
    const u32 wa0[4] = {c.wa3[0], c.wa3[1], c.wa3[2], c.wa3[3]}, wb0[4] = {c.wb3[0], c.wb3[1], c.wb3[2], c.wb3[3]};
    const u32 wa1[4] = {c.wa3[4], c.wa3[5], c.wa3[6], c.wa3[7]}, wb1[4] = {c.wb3[4], c.wb3[5], c.wb3[6], c.wb3[7]};
    if constexpr (ROUND != 0) { // RNDMODE = 1 (int_dif2_fly.vhd:167-219): plain values, rhu2 sums, exact extraction; STAGE 1 / 0 of all 32 registers follow in the caller
        static_assert(!FASTX, "round mode uses the exact extraction");
        group4<ROUND, 0, false, false, true, 0>(v[B + 0], v[B + 8], v[B + 1], v[B + 9], v[B + 2], v[B + 10], v[B + 3], v[B + 11], wa0, wb0, sl);
        group4<ROUND, 0, false, false, true, 0>(v[B + 4], v[B + 12], v[B + 5], v[B + 13], v[B + 6], v[B + 14], v[B + 7], v[B + 15], wa1, wb1, sl);
        group4<ROUND, 0, false, false, true, 0>(v[B + 0], v[B + 4], v[B + 1], v[B + 5], v[B + 2], v[B + 6], v[B + 3], v[B + 7], c.wa2, c.wb2, sl);
        group4<ROUND, 0, false, false, true, 0>(v[B + 8], v[B + 12], v[B + 9], v[B + 13], v[B + 10], v[B + 14], v[B + 11], v[B + 15], c.wa2, c.wb2, sl);
        return;
    }
    group4<false, FASTX, false, true, true, P0>(v[B + 0], v[B + 8], v[B + 1], v[B + 9], v[B + 2], v[B + 10], v[B + 3], v[B + 11], wa0, wb0, sl);
    group4<false, FASTX, false, true, true, P0>(v[B + 4], v[B + 12], v[B + 5], v[B + 13], v[B + 6], v[B + 14], v[B + 7], v[B + 15], wa1, wb1, sl);
    group4<false, FASTX, false, true, true, 0>(v[B + 0], v[B + 4], v[B + 1], v[B + 5], v[B + 2], v[B + 6], v[B + 3], v[B + 7], c.wa2, c.wb2, sl);
    group4<false, FASTX, false, true, true, 0xF>(v[B + 8], v[B + 12], v[B + 9], v[B + 13], v[B + 10], v[B + 14], v[B + 11], v[B + 15], c.wa2, c.wb2, sl);
#pragma unroll
    for (int g = B; g < B + 16; g += 8) { // stage 1: kind = r & 4
        bfly_triv<false, false>(v[g], v[g + 2]);
        bfly_mj<false, false>(v[g + 1], v[g + 3]);
        bfly_triv<false, true>(v[g + 4], v[g + 6]);
        bfly_mj<false, true>(v[g + 5], v[g + 7]);
    }
#pragma unroll
    for (int g = B; g < B + 16; g += 2) bfly_triv<false, false>(v[g], v[g + 1]);
}

// DIT stages 0, 1, 2, 3 on v[B .. B+15], wave-uniform twiddles in the DIT packing (DITPACK) or -- the pair, which shares one set of
// constants between its cores -- in the DIF packing through the re/im-swapped multiplier feed of int_dit2_fly.vhd:304-322
template <bool FASTX, int B, bool DITPACK = true, int ROUND = 0> __device__ __forceinline__ void dit_round4_c(u32 (&v)[32], const RoundCConsts &c, const Slice &sl)
{
    constexpr bool RD = ROUND != 0; // RNDMODE = 1 (int_dit2_fly.vhd:164-217): rhu2 sums on full-width values; ROUND == 2: + the w-bit wrap of narrow data
#pragma unroll
    for (int g = B; g < B + 16; g += 2) bfly_triv<RD, false>(v[g], v[g + 1]); // STAGE 0: T = B
    if constexpr (ROUND == 2) {
#pragma unroll
        for (int g = B + 1; g < B + 16; g += 2) v[g] = wrap_w(v[g], sl.wd);
    }
#pragma unroll
    for (int g = B; g < B + 16; g += 4) { // STAGE 1: even positions T = B, odd positions T = +j B (quirk)
        bfly_triv<RD, false>(v[g], v[g + 2]);
        bfly_pj_dit<RD>(v[g + 1], v[g + 3]);
    }
    if constexpr (ROUND == 2) {
#pragma unroll
        for (int g = B; g < B + 16; g += 4) v[g + 2] = wrap_w(v[g + 2], sl.wd), v[g + 3] = wrap_w(v[g + 3], sl.wd);
    }
    group4_dit<FASTX, true, ROUND, DITPACK>(v[B + 0], v[B + 4], v[B + 1], v[B + 5], v[B + 2], v[B + 6], v[B + 3], v[B + 7], c.wa2, c.wb2, sl);
    group4_dit<FASTX, true, ROUND, DITPACK>(v[B + 8], v[B + 12], v[B + 9], v[B + 13], v[B + 10], v[B + 14], v[B + 11], v[B + 15], c.wa2, c.wb2, sl);
    const u32 wa0[4] = {c.wa3[0], c.wa3[1], c.wa3[2], c.wa3[3]}, wb0[4] = {c.wb3[0], c.wb3[1], c.wb3[2], c.wb3[3]};
    const u32 wa1[4] = {c.wa3[4], c.wa3[5], c.wa3[6], c.wa3[7]}, wb1[4] = {c.wb3[4], c.wb3[5], c.wb3[6], c.wb3[7]};
    group4_dit<FASTX, true, ROUND, DITPACK>(v[B + 0], v[B + 8], v[B + 1], v[B + 9], v[B + 2], v[B + 10], v[B + 3], v[B + 11], wa0, wb0, sl);
    group4_dit<FASTX, true, ROUND, DITPACK>(v[B + 4], v[B + 12], v[B + 5], v[B + 13], v[B + 6], v[B + 14], v[B + 7], v[B + 15], wa1, wb1, sl);
}

__device__ __forceinline__ constexpr int rev5k(int r) { return ((r & 1) << 4) | ((r & 2) << 2) | (r & 4) | ((r & 8) >> 2) | ((r & 16) >> 4); }

enum { M16_FWD = 0, M16_INV = 1, M16_PAIR = 2 }; // PAIR: int_fft_ifft_pair (int_fft_ifft_pair.vhd:209-280): the forward core into the inverse core without leaving layout C

// ROUND: RNDMODE = 1 (the testbench's "ROUNDING" UUT) on its own instantiations (round 4): the ROUND forms of every register round -- rhu2 sums on
// full-width values, exact extraction, no pre-shifted kinds, quarter turns through the negated twiddle -- 1: 16-bit data, 2: DATA_WIDTH 9 .. 15
// The cores' own beat orders (round 4, template OB + launch argument native_orders).  Time side: a HALVES beat (x[i], x[i + N/2]) is the register pair (j, j + 16) of
// layout A, one 8-byte access per lane.  Frequency side: BITREV order is the core position n itself, so a layout-C thread owns 32
// CONSECUTIVE samples (128 bytes) -- moved straight between registers and memory every wave instruction would touch 64 lines; the chunk goes through the
// idle transpose region once more instead (thread-major rows of 33 dwords, read back position-major) and is loaded / stored 256 bytes per wave instruction.
template <int L, int MODE, bool FAST_OK, int ROUND = 0, int OB = 0> // OB: 1 = the cores' own beat orders, 2 = the same with BITREV_LANES on the frequency side
__global__ __launch_bounds__(1 << (L - 5)) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fft16k_i16(const u32 *in, u32 *out, const uint2 *__restrict__ twf,
                                                                                                          const RoundCConsts c, size_t nframes, const Slice sl, int native_orders)
{
    static_assert(!OB || MODE != M16_PAIR, "native orders: forward or inverse core alone");
    // OB instantiations serve every non-natural combination (bit 0: HALVES on the time side, bit 1: BITREV on the frequency side, wave-uniform tests);
    // the natural-order instantiations carry none of that code (a run-time `halves` test alone cost the inverse kernel two spilled registers)
    const bool halves = OB && (native_orders & 1), bitrev = OB && (native_orders & 6);
    // bit 2 (round 6): BITREV_LANES on the frequency side -- the BITREV path with core position n at memory index (n & 1) * N/2 + (n >> 1)
    constexpr bool lanes = OB == 2; // its own instantiations: a run-time stride cost the BITREV path 3 % (32 scalar adds instead of immediate offsets)
    typedef u32 v2u __attribute__((ext_vector_type(2)));
    static_assert(L == 13 || L == 14, "one workgroup per frame of 8192 / 16384 points");
    static_assert(!ROUND || !FAST_OK, "round mode: exact extraction");
    constexpr int RB = L - 9; // thread bits above l in layouts A and B
    constexpr bool FWD_PART = MODE != M16_INV, INV_PART = MODE != M16_FWD;
    constexpr int NSLOT = RB == 5 ? 16 : 8; // layout B's twiddle slots per packing
    extern __shared__ u32 lds[]; // (32 << RB) rows x ROWP (covers the 2^(L-5) rows x ROWQ of the second transpose), then layout B's twiddles (the pair: both packings)
    uint2 *const tw2 = reinterpret_cast<uint2 *>(lds + (32 << RB) * ROWP);
    const int tid = threadIdx.x, l = tid & 15, hx = tid >> 4;

    // layout B's twiddles (STAGE 4 + b on reg bit b of q, table index (rr << 4) | l): frame invariant, one set per column l, parked once.
    // slots: [STAGE 8: 8 (L = 14 only)] [STAGE 7: 4] [STAGE 6: 2] [STAGE 5: 1] [STAGE 4: 1]
    if (hx == 0) {
        int s = 0;
        auto park = [&](unsigned idx) {
            uint2 w = twf[idx + (unsigned)l];
            if (FWD_PART) tw2[16 * s + l] = w; // DIF packing {Wa, Wb}
            to_dit_packing(w.x, w.y);
            if (INV_PART) tw2[16 * (s + (FWD_PART ? NSLOT : 0)) + l] = w; // DIT packing {Wc, Wd} (the pair: behind the forward set)
            ++s;
        };
        if constexpr (RB == 5)
            for (int rr = 0; rr < 8; ++rr) park(255u + ((unsigned)rr << 4));
        for (int rr = 0; rr < 4; ++rr) park(127u + ((unsigned)rr << 4));
        for (int rr = 0; rr < 2; ++rr) park(63u + ((unsigned)rr << 4));
        park(31u);
        park(15u);
    }
    auto tw_b = [&](int s, u32(&wat)[8], u32(&wbt)[8], RoundTwQ &t) { // s: first slot of the packing wanted

        auto get = [&](u32 &wa, u32 &wb) {
            const uint2 w = tw2[16 * s++ + l];
            wa = w.x, wb = w.y;
        };
        if constexpr (RB == 5) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) get(wat[rr], wbt[rr]);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) get(t.wa8[rr], t.wb8[rr]);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) get(t.wa4[rr], t.wb4[rr]);
        get(t.wa2[0], t.wb2[0]);
        get(t.wa1[0], t.wb1[0]);
    };
    // layout A's twiddles (STAGE L-5 + b on reg bit b of j, table index (jj << (L-5)) | tid): per thread, re-read from the L2-resident table
    // in every frame (held over the loop they cost 32 VGPRs: spills, as in k_big2p_a)
    auto tw_a = [&](unsigned to, u32(&wat)[8], u32(&wbt)[8], RoundTwQ &t) { // to: the thread's BYTE offset into a stage table (8 tid)
        auto ld = [&](unsigned uniform_idx, u32 &wa, u32 &wb) {
            const uint2 w = ld2_at32b(twf + uniform_idx, to);
            wa = w.x, wb = w.y;
        };
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) ld((1u << (L - 1)) - 1u + ((unsigned)jj << (L - 5)), wat[jj], wbt[jj]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ld((1u << (L - 2)) - 1u + ((unsigned)jj << (L - 5)), t.wa8[jj], t.wb8[jj]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) ld((1u << (L - 3)) - 1u + ((unsigned)jj << (L - 5)), t.wa4[jj], t.wb4[jj]);
        ld((1u << (L - 4)) - 1u, t.wa2[0], t.wb2[0]);
        ld((1u << (L - 5)) - 1u, t.wa1[0], t.wb1[0]);
    };

    // transpose addressing.  A <-> B: element (row = n(L-1)..n4, column l)
    // (every base below is recomputed from an opaque copy of the thread index at its transpose -- `REMAT` -- instead of being held over the register
    // rounds: four resident address registers were what pushed the pair and the native-order kernels over 128 VGPRs)
#ifdef INTFFT_16K_NO_REMAT
#define INTFFT_16K_OPQ(t)
#else
#define INTFFT_16K_OPQ(t) asm volatile("" : "+v"(t))
#endif
    auto a_base_f = [&]() { unsigned t = (unsigned)tid; INTFFT_16K_OPQ(t); return lds + ROWP * (int)(t >> 4) + (int)(t & 15u); };        // layout A thread (hx = n(L-6)..n4, l): row (j << RB) | hx
    auto b_base_f = [&]() { unsigned t = (unsigned)tid; INTFFT_16K_OPQ(t); return lds + ROWP * (int)((t >> 4) << 5) + (int)(t & 15u); }; // layout B thread (jx = hx, l):         row (jx << 5) | q
    // B <-> C: row R of ROWQ dwords, column n4..n0.  Layout C thread t3 = rev(n(L-1)..n5): bit i = n(L-1-i).
    //   L = 14: R = rev5(jx) | (n8..n5 = q >> 1) << 5;           t3 = rev5(jx) | rev4(q >> 1) << 5
    //   L = 13: R = rev4(jx) | (n8 = q >> 4) << 4 | (n7..n5 = (q >> 1) & 7) << 5;   t3 = rev4(jx) | n8 << 4 | rev3(n7..n5) << 5
    auto bq_base_f = [&]() { // + ROWQ * rowq_of(q) + ((q & 1) << 4)
        unsigned t = (unsigned)tid;
        INTFFT_16K_OPQ(t);
        return lds + ROWQ * (int)(__brev(t >> 4) >> (32 - RB)) + (int)(t & 15u);
    };
    const unsigned t3 = (unsigned)tid;
    auto c_base_f = [&]() {
        unsigned t = (unsigned)tid;
        INTFFT_16K_OPQ(t);
        const unsigned rc = RB == 5 ? ((t & 31u) | ((__brev(t >> 5) >> 28) << 5)) : ((t & 31u) | ((__brev(t >> 5) >> 29) << 5));
        return lds + ROWQ * (int)rc;
    };
    // OB staging: thread t3's 32 positions in row t3; position p = 32 * brev(row) + column, read back as p = k * T + tid
    u32 *const stg_own = lds + ROWQ * (int)t3;
    u32 *const stg_lin = lds + ROWQ * (int)((__brev(t3 >> 5) >> (32 - (L - 10))) << 5) + (int)(t3 & 31u); // + ROWQ * rev5(k)
    const short s2 = (short)(1 - (hx & 1)); // L = 14, layout B: the kind of its inputs is n9 = jx bit 0
    const v2s sh2 = {s2, s2};
    const v2s none = {0, 0};
    (void)sh2;

    __shared__ __attribute__((aligned(256))) u32 vote_flags[64]; // (256 bytes: the dynamic LDS behind it keeps the alignment it had behind __syncthreads_or's own buffer)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        const u32 *src = in + (f << L); // wave-uniform
        u32 *dst = out + (f << L);
        unsigned tid_l = (unsigned)tid, twb = (unsigned)tid * 8u; // opaque copies: keep the per-access addresses out of loop-invariant VGPR pairs (see k_big2p_a)
        asm volatile("" : "+v"(tid_l), "+v"(twb));
        u32 v[32];
        if constexpr (FWD_PART) {
            if (OB && halves) { // HALVES beats: 8-byte loads of the register pairs (j, j + 16)
                const v2u *sh = reinterpret_cast<const v2u *>(src);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const v2u w = INTFFT_LD(at32(sh + ((size_t)j << (L - 5)), tid_l));
                    v[j] = w.x, v[j + 16] = w.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 32; ++j) v[j] = INTFFT_LD(at32(src + ((size_t)j << (L - 5)), tid_l)); // layout A
            }
        } else if (OB && bitrev) { // BITREV order in: memory index = core position; linear loads, handed over to layout C through the staging rows
            {   // BITREV_LANES: position k * T + tid sits at (tid & 1) * N/2 + k * T/2 + (tid >> 1) -- even threads read the first half of
                // the frame, odd ones the second (128-byte runs per wave); ONE set of 32 loads, offset and stride picked once
                unsigned off = lanes ? ((((unsigned)tid & 1u) << (L - 1)) | ((unsigned)tid >> 1)) : tid_l;
                constexpr int sh = lanes ? L - 6 : L - 5;
                asm volatile("" : "+v"(off));
#pragma unroll
                for (int k = 0; k < 32; ++k) v[k] = INTFFT_LD(at32(src + ((size_t)k << sh), off));
            }
            __syncthreads(); // the previous frame's last transpose reads
#pragma unroll
            for (int k = 0; k < 32; ++k) stg_lin[ROWQ * rev5k(k)] = v[k];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 32; ++r) v[r] = stg_own[r];
        } else {
#pragma unroll
            for (int r = 0; r < 32; ++r) v[r] = INTFFT_LD(at32(src + ((size_t)rev5k(r) << (L - 5)), tid_l)); // layout C: position (t3, r) = X[brev_L]
        }
        u32 wat[8], wbt[8];
        RoundTwQ ta;
        if constexpr (FWD_PART) tw_a(twb, wat, wbt, ta);
        // guard-bit vote of the frame (closed under all L stages); the barrier also orders the previous frame's LDS reads
        bool fast = false;
        {
            u32 acc = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) acc |= v[j] + sl.gbias;
            const bool bad = block_any(vote_flags, vote_phase, (acc & sl.gmask) != 0);
            fast = FAST_OK && !bad;
        }
        if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd); // DATA_WIDTH < 16, exact path: containers wrapped to w bits

        if constexpr (FWD_PART) {
            // ---- layout A: DIF L-1 .. L-5 ----
            if (fast) {
                dif_top16<FAST_OK, 0, false>(v, wat, wbt, sl, none);
                dif_round_q<FAST_OK, 0, 0, false>(v, ta, sl, none);
                dif_round_q<FAST_OK, 16, 0xF, false>(v, ta, sl, none);
            } else {
                dif_top16<false, 0, false, ROUND>(v, wat, wbt, sl, none);
                dif_round_q<false, 0, 0, false, 4, ROUND>(v, ta, sl, none);
                dif_round_q<false, 16, 0xF, false, 4, ROUND>(v, ta, sl, none);
            }
            {
                u32 *const a_base = a_base_f();
#pragma unroll
                for (int j = 0; j < 32; ++j) a_base[ROWP * (j << RB)] = v[j];
            }
            __syncthreads();
            u32 wa2t[8], wb2t[8];
            RoundTwQ tb;
            tw_b(0, wa2t, wb2t, tb);
            {
                u32 *const b_base = b_base_f();
#pragma unroll
                for (int q = 0; q < 32; ++q) v[q] = b_base[ROWP * q];
            }
            // ---- layout B: DIF 8 .. 4 (L = 13: 7 .. 4 on the two halves; the kind of their inputs is n8 = q bit 4) ----
            if constexpr (RB == 4) {
                if (fast) {
                    dif_round_q<FAST_OK, 0, 0, false>(v, tb, sl, none);
                    dif_round_q<FAST_OK, 16, 0xF, false>(v, tb, sl, none);
                } else {
                    dif_round_q<false, 0, 0, false, 4, ROUND>(v, tb, sl, none);
                    dif_round_q<false, 16, 0xF, false, 4, ROUND>(v, tb, sl, none);
                }
            } else {
                if (fast) {
                    dif_top16<FAST_OK, 0, true>(v, wa2t, wb2t, sl, sh2);
                    dif_round_q<FAST_OK, 0, 0, false>(v, tb, sl, none);
                    dif_round_q<FAST_OK, 16, 0xF, false>(v, tb, sl, none);
                } else {
                    dif_top16<false, 0, true, ROUND>(v, wa2t, wb2t, sl, sh2);
                    dif_round_q<false, 0, 0, false, 4, ROUND>(v, tb, sl, none);
                    dif_round_q<false, 16, 0xF, false, 4, ROUND>(v, tb, sl, none);
                }
            }
            __syncthreads(); // every thread has read its A -> B rows: the region may take the B -> C rows
            u32 *const bq_base = bq_base_f();
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int rq = RB == 5 ? ((q >> 1) << 5) : (((q >> 4) << 4) | (((q >> 1) & 7) << 5));
                bq_base[ROWQ * rq + ((q & 1) << 4)] = v[q];
            }
            __syncthreads();
            {
                u32 *const c_base = c_base_f();
#pragma unroll
                for (int r = 0; r < 32; ++r) v[r] = c_base[r];
            }
            // ---- layout C: DIF 3 .. 0 on the halves n4 = 0 / 1 (the upper half holds Y >> 1 of STAGE 4) ----
            if (fast) {
                dif_round4_c<FAST_OK, 0, 0>(v, c, sl);
                dif_round4_c<FAST_OK, 16, 0xF>(v, c, sl);
            } else {
                dif_round4_c<false, 0, 0, ROUND>(v, c, sl);
                dif_round4_c<false, 16, 0xF, ROUND>(v, c, sl);
                if constexpr (ROUND != 0) round_stages10<32, ROUND == 2>(v, sl); // STAGE 1, 0 in their round forms on both halves
            }
            if (MODE == M16_FWD && OB && bitrev) { // BITREV order out: through the staging rows, then 256 bytes per wave instruction
                unsigned st_off = lanes ? ((((unsigned)tid & 1u) << (L - 1)) | ((unsigned)tid >> 1)) : tid_l; // BITREV_LANES: two 128-byte runs
                constexpr int st_sh = lanes ? L - 6 : L - 5;                                                  // per wave, one per half of the frame
                asm volatile("" : "+v"(st_off));
                __syncthreads(); // every thread has read its layout-C row
#pragma unroll
                for (int r = 0; r < 32; ++r) stg_own[r] = v[r];
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 32; ++k) v[k] = stg_lin[ROWQ * rev5k(k)];
#pragma unroll
                for (int k = 0; k < 32; ++k) __builtin_nontemporal_store(v[k], at32(dst + ((size_t)k << st_sh), st_off));
            } else if constexpr (MODE == M16_FWD) {
#pragma unroll
                for (int r = 0; r < 32; ++r) __builtin_nontemporal_store(v[r], at32(dst + ((size_t)rev5k(r) << (L - 5)), tid_l));
            }
        }
        if constexpr (INV_PART) {
            // ---- layout C: DIT 0 .. 3 (the pair: position n holds X[brev n], what int_ifftNk takes there; the constants stay in the DIF packing) ----
            constexpr bool CP = MODE == M16_INV;
            if (fast) {
                dit_round4_c<FAST_OK, 0, CP>(v, c, sl);
                dit_round4_c<FAST_OK, 16, CP>(v, c, sl);
            } else {
                dit_round4_c<false, 0, CP, ROUND>(v, c, sl);
                dit_round4_c<false, 16, CP, ROUND>(v, c, sl);
            }
            {
                u32 *const c_base = c_base_f();
#pragma unroll
                for (int r = 0; r < 32; ++r) c_base[r] = v[r];
            }
            __syncthreads();
            u32 wa2t[8], wb2t[8];
            RoundTwQ tb;
            tw_b(FWD_PART ? NSLOT : 0, wa2t, wb2t, tb);
            u32 *const bq_base_i = bq_base_f();
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int rq = RB == 5 ? ((q >> 1) << 5) : (((q >> 4) << 4) | (((q >> 1) & 7) << 5));
                v[q] = bq_base_i[ROWQ * rq + ((q & 1) << 4)];
            }
            // ---- layout B: DIT 4 .. 8 (L = 13: 4 .. 7) ----
            if (fast) {
                dit_round_q<FAST_OK, 0>(v, tb, sl);
                dit_round_q<FAST_OK, 16>(v, tb, sl);
                if constexpr (RB == 5) dit_top16<FAST_OK>(v, wa2t, wb2t, sl);
            } else {
                dit_round_q<false, 0, ROUND>(v, tb, sl);
                dit_round_q<false, 16, ROUND>(v, tb, sl);
                if constexpr (RB == 5) dit_top16<false, ROUND>(v, wa2t, wb2t, sl);
            }
            __syncthreads();
            {
                u32 *const b_base = b_base_f();
#pragma unroll
                for (int q = 0; q < 32; ++q) b_base[ROWP * q] = v[q];
            }
            tw_a(twb, wat, wbt, ta); // in flight across the barrier (the data registers are free here)
            __syncthreads();
            {
                u32 *const a_base = a_base_f();
#pragma unroll
                for (int j = 0; j < 32; ++j) v[j] = a_base[ROWP * (j << RB)];
            }
            to_dit_packing(ta.wa1[0], ta.wb1[0]);
            to_dit_packing(ta.wa2[0], ta.wb2[0]);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) to_dit_packing(ta.wa4[jj], ta.wb4[jj]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) to_dit_packing(ta.wa8[jj], ta.wb8[jj]);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) to_dit_packing(wat[jj], wbt[jj]);
            // ---- layout A: DIT L-5 .. L-1 ----
            if (fast) {
                dit_round_q<FAST_OK, 0>(v, ta, sl);
                dit_round_q<FAST_OK, 16>(v, ta, sl);
                dit_top16<FAST_OK>(v, wat, wbt, sl);
            } else {
                dit_round_q<false, 0, ROUND>(v, ta, sl);
                dit_round_q<false, 16, ROUND>(v, ta, sl);
                dit_top16<false, ROUND>(v, wat, wbt, sl);
            }
            if (OB && halves) { // HALVES beats out: 8-byte stores of the register pairs (j, j + 16)
                v2u *dh = reinterpret_cast<v2u *>(dst);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const v2u w = {v[j], v[j + 16]};
                    __builtin_nontemporal_store(w, at32(dh + ((size_t)j << (L - 5)), tid_l));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 32; ++j) __builtin_nontemporal_store(v[j], at32(dst + ((size_t)j << (L - 5)), tid_l));
            }
        }
    }
}

bool fast16k_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int direction, int use_fly, int in_order, int out_order)
{
    return (log2n == 13 || log2n == 14) && packed_width_ok(data_width, format, rndmode) && twdl_width >= 8 && twdl_width <= 16 && format == 0 &&
           (!rndmode || !diag_env("INTFFT_NO_PACKED_ROUND")) && use_fly == 1 && !diag_env("INTFFT_NO_FAST16K") &&
           // + the cores' own beat orders: int_fftNk HALVES in / BITREV out, int_ifftNk BITREV in / HALVES out (and the mixed forms)
           (direction == 0   ? (in_order == 0 || in_order == 2) && (out_order == 0 || out_order == 1 || out_order == 3)
            : direction == 1 ? (in_order == 0 || in_order == 1 || in_order == 3) && (out_order == 0 || out_order == 2)
                             : direction == 2 && in_order == 0 && out_order == 0);
}

// the quarter-turn relation of the shared-twiddle rounds (stages 5 .. L-1), checked on the plan's generated tables (host copy); the inverse
// negates packed 16-bit twiddle halves, so no entry may be -2^15
bool fast16k_tables_ok(int log2n, const int2 *h_tw, int twd)
{
    for (int s = 5; s < log2n; ++s) {
        const int2 *t = h_tw + ((size_t)1 << s) - 1;
        const size_t h = (size_t)1 << (s - 1);
        for (size_t k = 0; k < h; ++k) {
            const int neg = (int)(((long long)(-t[k].x) << (64 - twd)) >> (64 - twd));
            if (t[k + h].x != t[k].y || t[k + h].y != neg) return false;
            if (t[k].x == -32768 || t[k].y == -32768) return false;
        }
    }
    return true;
}

const char *fast16k_kernel_name() { return "k_fft16k_i16"; }

template <int L, int MODE, bool FX, int RD = 0, int OB = 0>
static hipError_t launch16k_ob(const u32 *in, u32 *out, const uint2 *tw16f, const RoundCConsts &c, size_t nframes, const Slice &sl, hipStream_t stream, int native_orders)
{
    constexpr int RB = L - 9, T = 16 << RB;
    const size_t ldsb = (size_t)(32 << RB) * ROWP * sizeof(u32) + (size_t)(RB == 5 ? 16 : 8) * (MODE == M16_PAIR ? 2 : 1) * 16 * sizeof(uint2);
    allow_max_lds(kptr(k_fft16k_i16<L, MODE, FX, RD, OB>));
    const size_t cap = resident_blocks(kptr(k_fft16k_i16<L, MODE, FX, RD, OB>), T, RB == 5 ? 2 : 4, RB == 5 ? 2 : 4);
    const unsigned blocks = (unsigned)(nframes < cap ? nframes : cap);
    hipLaunchKernelGGL((k_fft16k_i16<L, MODE, FX, RD, OB>), dim3(blocks), dim3(T), ldsb, stream, in, out, tw16f, c, nframes, sl, native_orders);
    return hipGetLastError();
}
// native_orders: bit 0 = HALVES on the time side, bit 1 = BITREV, bit 2 = BITREV_LANES on the frequency side (single cores only): any of them -> the OB instantiation
template <int L, int MODE, bool FX, int RD = 0>
static hipError_t launch16k(const u32 *in, u32 *out, const uint2 *tw16f, const RoundCConsts &c, size_t nframes, const Slice &sl, hipStream_t stream, int native_orders)
{
    if constexpr (MODE != M16_PAIR) {
        if (native_orders & 4) return launch16k_ob<L, MODE, FX, RD, 2>(in, out, tw16f, c, nframes, sl, stream, native_orders);
        if (native_orders) return launch16k_ob<L, MODE, FX, RD, 1>(in, out, tw16f, c, nframes, sl, stream, native_orders);
    }
    return launch16k_ob<L, MODE, FX, RD, 0>(in, out, tw16f, c, nframes, sl, stream, 0);
}

hipError_t launch_fast16k(int log2n, int direction, int twd, const void *in, void *out, const uint2 *tw16f, const int2 *h_tw, size_t nframes, hipStream_t stream,
                          int data_width, int rndmode, int native_orders)
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
    if (direction == 1) to_dit_packing_host(c);
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    if (data_width != 16) sl.set_width(data_width);
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fx = twd == 16 && allow_fast;
    const u32 *pin = static_cast<const u32 *>(in);
    u32 *pout = static_cast<u32 *>(out);
    sl.round = rndmode ? (data_width != 16 ? 2 : 1) : 0;
#define INTFFT_16K_RD(LL, RD)                                                                                                       \
    {                                                                                                                              \
        if (direction == 0) return launch16k<LL, M16_FWD, false, RD>(pin, pout, tw16f, c, nframes, sl, stream, native_orders);                    \
        if (direction == 2) return launch16k<LL, M16_PAIR, false, RD>(pin, pout, tw16f, c, nframes, sl, stream, native_orders);                   \
        return launch16k<LL, M16_INV, false, RD>(pin, pout, tw16f, c, nframes, sl, stream, native_orders);                                        \
    }
    if (sl.round == 1) {
        if (log2n == 13) INTFFT_16K_RD(13, 1)
        INTFFT_16K_RD(14, 1)
    }
    if (sl.round == 2) {
        if (log2n == 13) INTFFT_16K_RD(13, 2)
        INTFFT_16K_RD(14, 2)
    }
#undef INTFFT_16K_RD
#define INTFFT_16K(LL)                                                                                                              \
    if (direction == 0) return fx ? launch16k<LL, M16_FWD, true>(pin, pout, tw16f, c, nframes, sl, stream, native_orders)                        \
                                  : launch16k<LL, M16_FWD, false>(pin, pout, tw16f, c, nframes, sl, stream, native_orders);                       \
    if (direction == 2) return fx ? launch16k<LL, M16_PAIR, true>(pin, pout, tw16f, c, nframes, sl, stream, native_orders)                       \
                                  : launch16k<LL, M16_PAIR, false>(pin, pout, tw16f, c, nframes, sl, stream, native_orders);                      \
    return fx ? launch16k<LL, M16_INV, true>(pin, pout, tw16f, c, nframes, sl, stream, native_orders)                                             \
              : launch16k<LL, M16_INV, false>(pin, pout, tw16f, c, nframes, sl, stream, native_orders);
    if (log2n == 13) {
        INTFFT_16K(13)
    }
    INTFFT_16K(14)
#undef INTFFT_16K
}

} // namespace intfft
