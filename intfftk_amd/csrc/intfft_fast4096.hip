// intfft_fast4096.hip -- packed-int16 block kernel for N = 4096 (BASELINE config 5 and its two halves):
// int_fftNk / int_ifftNk / int_fft_ifft_pair with NFFT = 12, DATA_WIDTH = 16, TWDL_WIDTH <= 16,
// scaled-truncate, natural-order input and output (src/vhdl/main/int_fft_ifft_pair.vhd:161-330).
//
// One 256-thread workgroup owns one frame: 16 packed (re | im << 16) samples per thread, persistent loop
// over frames.  Twelve radix-2 stages = three in-register rounds of four stages, with a block-wide LDS
// transpose between rounds (two alternating 20 KiB regions -> one barrier per transpose; rows of 18 dwords):
//
//   layout LA  reg = n11..8, thread = n7..0                       DIF 11,10,9,8   / DIT 8,9,10,11
//   layout LB  reg = n7..4,  thread = (n11..8, n3..0)             DIF 7,6,5,4     / DIT 4,5,6,7
//   layout LC  reg = n3..0,  thread = rev8(n11..4)                DIF 3,2,1,0     / DIT 0,1,2,3
//
// The pair never leaves LC between the forward and the inverse core: after the DIF stages position n
// holds X[bitrev(n)], which is exactly the element int_ifftNk expects at position n -- the RTL wires the
// FFT lane outputs straight into the IFFT lane inputs (int_fft_ifft_pair.vhd:242-280).  The forward
// core alone stores from LC with the bit reversal folded into the thread mapping (every store
// instruction writes 256 contiguous bytes per wave); the inverse core alone loads into LC the same way.
//
// Round-A/B twiddles are frame invariant and live in VGPRs in the DIF packing {Wa, Wb}; the DIT
// butterfly reuses them by feeding the multiplier re/im-swapped, exactly like int_dit2_fly.vhd:304-322:
//   Bs = (B.im, B.re):   T.re = dot(Bs, Wb) = B.re*wr + B.im*wi,   T.im = dot(Bs, Wa) = B.im*wr - B.re*wi.
// Arithmetic, asm blocks, pre-shifted outputs and the guard-bit fast extraction: intfft_pk16.hpp /
// intfft_fast1024.hip.  The guard-bit bound covers the pair: through a scaled DIT stage the complex
// magnitude also grows by <= 1.42, so over 24 stages M <= 23172 + 35.
#include "intfft_pk16.hpp"

#include <cstdlib>

namespace intfft {

// LDS row stride in dwords.  18 (round 4): b64 row reads; every ds_write_b32 (two groups of 32 lanes, bank = dword address mod 32) and every
// ds_read_b64 (two groups of 32 lanes, 64 banks) of the four transposes is conflict-free -- SQ_LDS_BANK_CONFLICT of the forward kernel 4.2e6 -> 0
// per 2^26 samples, of the pair 1.57e7 -> 4.2e6 -- which is worth 0-2 % (the kernels are VALU-bound: DESIGN 4.1).  -DINTFFT_4K_ROW=20 restores
// the b128 form of rounds 1-3 (2-way on every write) for A/B runs.
#ifndef INTFFT_4K_ROW
#define INTFFT_4K_ROW 18
#endif
constexpr int ROW4K = INTFFT_4K_ROW;
constexpr int ROW_CB = 20;            // the LC -> LB transpose keeps 16-byte aligned rows inside its blocks (BLK_CB)
constexpr int REGION4K = 256 * 20;    // dwords per transpose region (16 x BLK_CB = 5056 for LC -> LB)

// Diagnostics only (tools/build_variant_multi.sh <name> "-DINTFFT_4K_ABL=<bits>" intfft_fast4096.hip; never set in the product build; the
// results of an ablated kernel are WRONG, its time and counters are the measurement -- DESIGN 4.1, "where the pair's time goes"):
//   1: no block-wide barrier in the transposes    2: every transpose through conflict-free addresses (lane-linear writes, linear b128 reads)
//   4: no global loads                             8: no global stores
//   16 << k: the writes of transpose k skipped (0: LA -> LB, 1: LB -> LC, 2: LC -> LB, 3: LB -> LA) -- bank-conflict attribution by counter difference
#ifndef INTFFT_4K_ABL
#define INTFFT_4K_ABL 0
#endif
#define INTFFT_X_BARRIER() { if (!(INTFFT_4K_ABL & 1)) __syncthreads(); }
// one transposed write of register j: K = transpose number, off = the kernel's dword offset
#define INTFFT_X_WRITE(K, region, off, j, val)                                                         \
    {                                                                                                  \
        if (!((INTFFT_4K_ABL >> (4 + (K))) & 1)) (region)[(INTFFT_4K_ABL & 2) ? tid + 256 * (j) : (off)] = (val); \
    }

// ---- block-wide transposes: write 16 scattered dwords, barrier, read one padded row ------------------
// OFF(j): compile-time row offset (in rows) of register j; wr: this thread's base (dword index)
#define INTFFT_X_READ(region)                                                                          \
    {                                                                                                  \
        INTFFT_X_BARRIER()                                                                             \
        if constexpr (ROW4K % 4 == 0) {                                                                \
            const uint4 *rp = reinterpret_cast<const uint4 *>((region) + ((INTFFT_4K_ABL & 2) ? 16 * tid : ROW4K * tid)); \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                              \
            {                                                                                          \
                const uint4 x = rp[q];                                                                 \
                v[4 * q + 0] = x.x;                                                                    \
                v[4 * q + 1] = x.y;                                                                    \
                v[4 * q + 2] = x.z;                                                                    \
                v[4 * q + 3] = x.w;                                                                    \
            }                                                                                          \
        } else {                                                                                       \
            const uint2 *rp = reinterpret_cast<const uint2 *>((region) + ROW4K * tid);                 \
            _Pragma("unroll") for (int q = 0; q < 8; ++q)                                              \
            {                                                                                          \
                const uint2 x = rp[q];                                                                 \
                v[2 * q + 0] = x.x;                                                                    \
                v[2 * q + 1] = x.y;                                                                    \
            }                                                                                          \
        }                                                                                              \
    }

// The LC -> LB transpose (inverse core) writes row 16 * (n11..8) + r, column n7..4 with the lanes of a wave = (n11..8: 16 values,
// n7..6: 4 values).  With rows at ROW4K * row, 16 * ROW4K = 0 mod 64 banks puts the 16 rows of a column on ONE bank: 11x the
// bank conflicts of the forward kernel (PMC), 129 vs 112 us per 2^26 samples.  This one transpose therefore packs the blocks
// of 16 rows BLK_CB = 316 dwords apart (a block's last row needs no pad: 15 * 20 + 16) and stores column n7..4 at position
// n6 + 2 n7 + 4 n4 + 8 n5: 316 = -4 mod 64, so the wave's 64 writes land on banks (n6 + 2 n7) - 4 (n11..8) + const -- all
// distinct (N = 2048, where the wave holds n10..8 and n7..5: 2-way).  Rows stay 16-byte aligned for the b128 reads, whose
// registers come back in the permuted order; 16 * 316 dwords fit the region.
constexpr int BLK_CB = ROW4K % 4 == 0 ? 316 : 318;
// With b64 row reads (ROW4K = 18) this transpose takes blocks 318 dwords apart and column n7..4 at position n7 + 2 n6 + 4 n5 + 8 n4: a
// ds_write_b32 is served in two groups of 32 lanes on banks (a / 4) mod 32 -- the group holds n11..8 and n7, and 318 = -2 mod 32 puts its 32
// writes on 32 banks -- and a ds_read_b64 in two groups of 32 lanes on banks mod 64: rows 20 apart in two blocks 318 apart cover all 64
// (the 316-dword form was laid out for 64 banks per write: it is 2-way on every write, measured 32 conflict cycles per wave and transpose).
#define INTFFT_X_READ_CB(region)                                                                       \
    {                                                                                                  \
        INTFFT_X_BARRIER()                                                                             \
        if constexpr (ROW4K % 4 == 0) {                                                                \
            const uint4 *rp = reinterpret_cast<const uint4 *>((region) + ((INTFFT_4K_ABL & 2) ? 16 * tid : BLK_CB * (tid >> 4) + ROW_CB * (tid & 15))); \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                              \
            {                                                                                          \
                const uint4 x = rp[q];                                                                 \
                v[q] = x.x;                                                                            \
                v[q + 4] = x.y;                                                                        \
                v[q + 8] = x.z;                                                                        \
                v[q + 12] = x.w;                                                                       \
            }                                                                                          \
        } else {                                                                                       \
            const uint2 *rp = reinterpret_cast<const uint2 *>((region) + BLK_CB * (tid >> 4) + ROW_CB * (tid & 15)); \
            _Pragma("unroll") for (int q = 0; q < 8; ++q)                                              \
            {                                                                                          \
                const uint2 x = rp[q];                                                                 \
                v[rev4c(2 * q)] = x.x;                                                                 \
                v[rev4c(2 * q + 1)] = x.y;                                                             \
            }                                                                                          \
        }                                                                                              \
    }

// MODE_MID: the pair on one 4096-point block of a longer frame (middle pass of the N >= 8192 pair, intfft_big20.hip):
// like MODE_PAIR, but the block's inputs come from the DIF stage 12 of pass 1 (Y >> 1 where n12 = block index bit 0)
enum { MODE_FWD = 0, MODE_INV = 1, MODE_PAIR = 2, MODE_MID = 3 };

// L = 11 (N = 2048): the workgroup owns a chunk of two frames; index bit n11 numbers the frame and its stage is
// skipped in both cores (twiddle indices are positions mod 2^s, so nothing else changes).  lc_bit<L>(k): the bit
// of the LC thread index t'' that carries index bit n_k (k = 4..11): in-frame bits reversed in the low bits,
// frame bits on top, so that natural-order X of a frame = rev4(r) * 2^(L-4) + (low bits of t'') stays one
// contiguous run per wave.
// OB (native BITREV order on the LC side: int_fftNk output / int_ifftNk input beats, memory index = core index):
// LC thread = n11..n4 in natural bit order; two lane swaps then give every lane 4 consecutive samples (dwordx4).
template <int L, int OB = 0> __host__ __device__ constexpr int lc_bit(int k)
{
    return OB ? k - 4 : (k < L ? (L - 1) - k : (L - 4) + (k - L));
}
template <int L, int OB = 0> __host__ __device__ constexpr int lc_row_of_reg(int j) // LB register j' = n7..4 -> its LC thread bits
{
    return (((j >> 0) & 1) << lc_bit<L, OB>(4)) | (((j >> 1) & 1) << lc_bit<L, OB>(5)) | (((j >> 2) & 1) << lc_bit<L, OB>(6)) |
           (((j >> 3) & 1) << lc_bit<L, OB>(7));
}

// ROUND: RNDMODE = 1 (the testbench's "ROUNDING" UUT, fft_signle_test.vhd:93-112): rhu2 sums on full-width values, exact
// extraction, no pre-shifted outputs (so none of the per-thread shift amounts below apply)
// ROUND: 0 truncate, 1 round, 2 round on narrow data (its own instantiation: the w-bit wraps of intfft_pk16.hpp)
template <int L, int MODE, bool FAST_OK, int OB = 0, int ROUND = 0> // OB: 1 = the cores' own beat orders, 2 = the same with BITREV_LANES on the frequency side
__global__ __launch_bounds__(256) void k_fft4096_i16(const u32 *in, u32 *out, const int2 *__restrict__ twt,
                                                     const RoundCConsts c, size_t nframes_user, const Slice sl, int io_flags)
{
    const int halves = io_flags & 1;             // HALVES beats on the time side
    constexpr bool lanes = OB == 2;              // BITREV_LANES instead of BITREV on the frequency side (round 6): its own instantiations --
                                                 // as a run-time switch it cost the BITREV-in inverse 5 %
    static_assert(!OB || MODE == MODE_FWD || MODE == MODE_INV, "native orders: forward or inverse core alone");
    static_assert(L == 11 || L == 12, "block kernel: N = 2048 or 4096");
    static_assert(!ROUND || !FAST_OK, "round mode: exact extraction");
    constexpr int FP = 1 << (12 - L), NS = L - 8;        // frames per 4096-sample chunk; executed stages of round A
    const size_t nframes = (nframes_user + FP - 1) / FP; // chunks
    __shared__ __attribute__((aligned(16))) u32 lds[2 * REGION4K];
    static_assert((ROW4K == 20 || 256 * ROW4K <= REGION4K - 4) && 16 * BLK_CB <= REGION4K - 4, "vote cells: pad columns of the last row / behind the layouts of region 0");
    u32 *const reg0 = lds, *const reg1 = lds + REGION4K;
    u32 *const vote_flags = lds + (REGION4K - 4); // three pad cells of region 0's last row (columns 16..19 are never transposed; beyond the 16 x 316 dwords of the LC -> LB transpose too)
    unsigned vote_phase = 0;
    block_any_init(vote_flags);
    const int tid = threadIdx.x;
    const int lo4 = tid & 15, hi4 = tid >> 4;

    // ---- frame-invariant twiddles (DIF packing), stage s table at twt + 2^s - 1 ----
    // round A: thread = n7..0, regs n11..8 -> stage 11 index 256*jj + tid, ... stage 8 index tid
    // round B: thread low nibble q = n3..0, regs n7..4 -> stage 7 index 16*jj + q, ... stage 4 index q
    RoundTw ta, tb;
    auto ld = [&](int idx, u32 &wa, u32 &wb) {
        const int2 w = twt[idx];
        wa = pack_wa(w);
        wb = pack_wb(w);
    };
    if constexpr (L >= 12) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ld(2047 + 256 * j + tid, ta.wa8[j], ta.wb8[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) ld(1023 + 256 * j + tid, ta.wa4[j], ta.wb4[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) ld(511 + 256 * j + tid, ta.wa2[j], ta.wb2[j]);
    ld(255 + tid, ta.wa1[0], ta.wb1[0]);
#pragma unroll
    for (int j = 0; j < 8; ++j) ld(127 + 16 * j + lo4, tb.wa8[j], tb.wb8[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) ld(63 + 16 * j + lo4, tb.wa4[j], tb.wb4[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) ld(31 + 16 * j + lo4, tb.wa2[j], tb.wb2[j]);
    ld(15 + lo4, tb.wa1[0], tb.wb1[0]);
    // twiddles in the DIT packing wherever the inverse core runs: its butterflies then multiply the unswapped B; the forward core
    // of a pair produces D with its halves exchanged instead (free: op_sel of the packed subtract; group4's DPK form)
    // (round-mode pairs keep the DIF packing: measured 239 vs 221 Gsample/s with the exchanged form)
    constexpr bool DP = MODE == MODE_INV || (MODE != MODE_FWD && !ROUND);
    if constexpr (DP) {
        to_dit_packing(ta);
        to_dit_packing(tb);
    }

    // ---- transpose addressing (dword offsets of this thread; register part is compile time) ----
    // LA -> LB and LB -> LA: element (thread x, reg y) -> row 16*y + x3..0, column x7..4
    const int w_ab = ROW4K * lo4 + hi4;
    // LB -> LC: thread t' = (n11..8 = hi4, n3..0 = lo4), reg j' = n7..4 -> row = LC thread t'' (lc_bit<L>), column n3..0
    const int row_hi = ((hi4 & 1) << lc_bit<L, OB>(8)) | (((hi4 >> 1) & 1) << lc_bit<L, OB>(9)) | (((hi4 >> 2) & 1) << lc_bit<L, OB>(10)) |
                       (((hi4 >> 3) & 1) << lc_bit<L, OB>(11));
    const int w_bc = ROW4K * row_hi + lo4; // + ROW4K * lc_row_of_reg<L>(j')
    // LC -> LB: thread t'' , reg r = n3..0 -> row = LB thread 16 * (n11..8) + r, column = LB register n7..4
    auto nb = [&](int k) { return (tid >> lc_bit<L, OB>(k)) & 1; };
    const int lb_hi = nb(8) | (nb(9) << 1) | (nb(10) << 2) | (nb(11) << 3), lb_reg = nb(4) | (nb(5) << 1) | (nb(6) << 2) | (nb(7) << 3);
    const int w_cb = BLK_CB * lb_hi + (ROW4K % 4 == 0 ? (lb_reg >> 2) + 4 * (lb_reg & 3) : rev4c(lb_reg)); // see INTFFT_X_READ_CB
    // per-thread shift amounts where the value kind depends on a thread bit after a transpose
    const short shb = (short)(1 - (hi4 & 1)); // LB: kind = n8 = t'4
    const short shc = (short)(1 - nb(4));     // LC: kind = n4
    const v2s sh_b = {shb, shb}, sh_c = {shc, shc};
    // LC <-> natural-order X: index = rev4(r) * 2^(L-4) + lc_off (in-frame bits reversed, frame bits in place)
    int lc_off = 0, lc_frame = 0;
#pragma unroll
    for (int k = 4; k < 12; ++k) {
        lc_off += nb(k) * (k >= L ? (1 << k) : (1 << (L - 1 - k)));
        if (k >= L) lc_frame += nb(k) << (k - L);
    }

    // OB: x4 unit index of the lane's vector q = (n9 n8): (n11 n10 | q | n7..n4 | n3 n2); its frame = n11..nL
    const int ob_unit = ((tid >> 6) << 8) | ((tid & 15) << 2) | ((tid >> 4) & 3);
    auto load_frame = [&](u32(&v)[16], size_t f) { // global loads only (the prefetch below keeps them in flight)
        const u32 *src = in + f * 4096;
        const bool partial = L < 12 && (f + 1) * FP > nframes_user; // last chunk: the absent frame reads as 0
        const bool lc_ok = !partial || f * FP + (size_t)lc_frame < nframes_user;
        if (MODE == MODE_INV && OB) { // memory index = n: x4 loads (regs n9 n8 n1 n0); two lane swaps follow
            typedef u32 v4u __attribute__((ext_vector_type(4)));
            const v4u *s4 = reinterpret_cast<const v4u *>(src) + ob_unit;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4u x = {0u, 0u, 0u, 0u};
                if (!partial || f * FP + (size_t)((4 * (ob_unit + 64 * q)) >> L) < nframes_user) {
                    if (lanes) { // core position n sits at memory index (n & 1) * N/2 + (n >> 1): two 8-byte pieces, one per half of the frame
                        typedef u32 v2u __attribute__((ext_vector_type(2)));
                        const int n0 = 4 * (ob_unit + 64 * q), nl = n0 & ((1 << L) - 1);
                        const u32 *const d = src + (n0 - nl) + (nl >> 1);
                        const v2u ev = INTFFT_LD(reinterpret_cast<const v2u *>(d)), od = INTFFT_LD(reinterpret_cast<const v2u *>(d + (1 << (L - 1))));
                        x = v4u{ev.x, od.x, ev.y, od.y};
                    } else
                        x = INTFFT_LD(s4 + 64 * q);
                }
                v[4 * q] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
            }
        } else if (MODE == MODE_INV) { // LC: v[r] = X[brev_L(n)] of the thread's frame (one test around the 16 loads:
            // tested one by one they are issued one by one)
            if (lc_ok) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = INTFFT_LD(src + (rev4c(r) << (L - 4)) + lc_off);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = 0u;
            }
        } else if (MODE == MODE_FWD && halves) {
            // HALVES: beat q = 256 jj + tid of the chunk holds (x[i], x[i + N/2]) of frame q >> (L-1): thread tid of the
            // LA registers j0 = (frame << (L-8)) | (i >> 8) and j0 | 2^(L-9)
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            const v2u *src2 = reinterpret_cast<const v2u *>(src) + tid;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                constexpr int HB = 1 << (L - 9);
                const int j0 = ((jj >> (L - 9)) << (L - 8)) | (jj & (HB - 1));
                v2u w = {0u, 0u};
                if (!partial || f * FP + (size_t)(jj >> (L - 9)) < nframes_user) w = INTFFT_LD(src2 + 256 * jj);
                v[j0] = w.x;
                v[j0 | HB] = w.y;
            }
        } else if (!partial) { // LA: v[j] = x[256 j + tid]
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = (INTFFT_4K_ABL & 4) ? (u32)(tid * 0x00030005u + j * 0x00110007u + (u32)f) & 0x1fff1fffu : INTFFT_LD(src + 256 * j + tid);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[j] = f * FP + (size_t)((256 * j + tid) >> L) < nframes_user ? INTFFT_LD(src + 256 * j + tid) : 0u;
        }
    };
    // Round mode, one core, N = 4096: the workgroup's next frame is loaded into 16 more registers while this one is computed (the
    // barriers of the transposes wait for LDS only, so the loads stay in flight): +5..7 %.  The truncate-mode kernels gain nothing
    // from it (and the N = 2048 ones pass 128 VGPRs with it), the pair has no registers to spare.
    constexpr bool PIPE = ROUND && L == 12 && (MODE == MODE_FWD || MODE == MODE_INV);
    u32 pre[16];
    if (PIPE && blockIdx.x < nframes) load_frame(pre, blockIdx.x);

    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        u32 v[16];
        u32 *dst = out + f * 4096;
        const bool partial = L < 12 && (f + 1) * FP > nframes_user;
        const bool lc_ok = !partial || f * FP + (size_t)lc_frame < nframes_user;
        if (PIPE) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = pre[j];
            if (f + gridDim.x < nframes) load_frame(pre, f + gridDim.x);
        } else {
            load_frame(v, f);
        }
        if (MODE == MODE_INV && OB) { // -> regs n3..0
#pragma unroll
            for (int g = 0; g < 16; g += 8)
#pragma unroll
                for (int r = 0; r < 4; ++r) swap16(v[g + r], v[g + r + 4]);
#pragma unroll
            for (int r = 0; r < 8; ++r) swap32(v[r], v[r + 8]);
        }

        // guard-bit test of the whole frame (block-uniform)
        bool fast = false;
        const short sm = (short)(1 - (int)(f & 1)); // MODE_MID: shift amount of the first stage's inputs
        const v2s sh_m = {sm, sm};
        if (FAST_OK) { // one barrier (block_any; round 3: two + the one at the end of the frame): it also orders the previous frame's LDS reads
            bool bad = guard_acc(v, sl.gbias, sl.gmask) != 0;
            if (MODE == MODE_MID && (f & 1)) { // Y >> 1 inputs: |v| < 2^13
                u32 acc = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc |= v[j] + sl.gbias1;
                bad = (acc & sl.gmask1) != 0;
            }
            fast = !block_any(vote_flags, vote_phase, bad);
        }
        if (MODE != MODE_MID && !fast && sl.wd != 16) wrap_inputs(v, sl.wd); // DATA_WIDTH < 16: containers wrapped to w bits

#define INTFFT_BODY(FX, RD)                                                                              \
    {                                                                                                   \
        if (MODE != MODE_INV) {                                                                         \
            if (MODE == MODE_MID) dif_round<FX, true, NS, RD, DP>(v, ta, sl, sh_m); /* round mode: plain inputs */ \
            else dif_round<FX, false, NS, RD, DP>(v, ta, sl, sh_b);                                      \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) INTFFT_X_WRITE(0, reg0, w_ab + ROW4K * 16 * j, j, v[j]) \
            INTFFT_X_READ(reg0)                                                                         \
            dif_round<FX, true, 4, RD, DP>(v, tb, sl, sh_b);                                             \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) INTFFT_X_WRITE(1, reg1, (w_bc + ROW4K * lc_row_of_reg<L, OB>(j)), j, v[j]) \
            INTFFT_X_READ(reg1)                                                                         \
            dif_round_c<FX, RD, DP>(v, c, sl, sh_c);                                                     \
        }                                                                                               \
        if (MODE == MODE_FWD && OB) { /* memory index = n: two lane swaps, dwordx4 stores (1 KiB per wave) */ \
            swap_guard(v);                                                                              \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) swap32(v[r], v[r + 8]);                       \
            _Pragma("unroll") for (int g = 0; g < 16; g += 8)                                           \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) swap16(v[g + r], v[g + r + 4]);           \
            typedef u32 v4u __attribute__((ext_vector_type(4)));                                        \
            v4u *d4 = reinterpret_cast<v4u *>(dst) + ob_unit;                                           \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                               \
            {                                                                                           \
                const v4u x = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};                     \
                if (!partial || f * FP + (size_t)((4 * (ob_unit + 64 * q)) >> L) < nframes_user) {      \
                    if (lanes) { /* BITREV_LANES: even / odd core positions to the two halves of the frame */ \
                        typedef u32 v2u __attribute__((ext_vector_type(2)));                            \
                        const int n0 = 4 * (ob_unit + 64 * q), nl = n0 & ((1 << L) - 1);                \
                        u32 *const dl = dst + (n0 - nl) + (nl >> 1);                                    \
                        const v2u ev = {x.x, x.z}, od = {x.y, x.w};                                     \
                        __builtin_nontemporal_store(ev, reinterpret_cast<v2u *>(dl));                   \
                        __builtin_nontemporal_store(od, reinterpret_cast<v2u *>(dl + (1 << (L - 1))));  \
                    } else                                                                              \
                        __builtin_nontemporal_store(x, d4 + 64 * q);                                    \
                }                                                                                       \
            }                                                                                           \
        } else if (MODE == MODE_FWD) {                                                                  \
            if (lc_ok) {                                                                                \
                _Pragma("unroll") for (int r = 0; r < 16; ++r)                                          \
                    __builtin_nontemporal_store(v[r], dst + (rev4c(r) << (L - 4)) + lc_off);            \
            }                                                                                           \
        } else {                                                                                        \
            dit_round_c<FX, RD, DP>(v, c, sl);                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) INTFFT_X_WRITE(2, reg0, w_cb + ROW_CB * r, r, v[r]) \
            INTFFT_X_READ_CB(reg0)                                                                      \
            dit_round<FX, 4, RD, DP>(v, tb, sl);                                                     \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) INTFFT_X_WRITE(3, reg1, w_ab + ROW4K * 16 * j, j, v[j]) \
            INTFFT_X_READ(reg1)                                                                         \
            dit_round<FX, NS, RD, DP>(v, ta, sl);                                                    \
            if (MODE == MODE_INV && halves) { /* HALVES beats, mirror of the forward load */            \
                typedef u32 v2u __attribute__((ext_vector_type(2)));                                    \
                v2u *d2 = reinterpret_cast<v2u *>(dst) + tid;                                           \
                _Pragma("unroll") for (int jj = 0; jj < 8; ++jj)                                        \
                {                                                                                       \
                    constexpr int HB = 1 << (L - 9);                                                    \
                    const int j0 = ((jj >> (L - 9)) << (L - 8)) | (jj & (HB - 1));                      \
                    const v2u w = {v[j0], v[j0 | HB]};                                                  \
                    if (!partial || f * FP + (size_t)(jj >> (L - 9)) < nframes_user)                    \
                        __builtin_nontemporal_store(w, d2 + 256 * jj);                                  \
                }                                                                                       \
            } else {                                                                                    \
            if (INTFFT_4K_ABL & 8) {                                                                    \
                u32 acc = 0;                                                                            \
                _Pragma("unroll") for (int j = 0; j < 16; ++j) acc ^= v[j];                             \
                if (acc == 0x12345u) dst[tid] = acc;                                                    \
            } else                                                                                      \
            _Pragma("unroll") for (int j = 0; j < 16; ++j)                                              \
                if (!partial || f * FP + (size_t)((256 * j + tid) >> L) < nframes_user)                 \
                    __builtin_nontemporal_store(v[j], dst + 256 * j + tid);                             \
            }                                                                                           \
        }                                                                                               \
    }
        if (FAST_OK && fast) INTFFT_BODY(FAST_OK, ROUND)
        else INTFFT_BODY(false, ROUND)
#undef INTFFT_BODY
        if (!FAST_OK) __syncthreads(); // region reuse by the next frame (the vote's barrier does it where there is a vote)
    }
}

bool fast4096_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int direction, int use_fly,
                        int in_order, int out_order)
{
    if (!((log2n == 12 || log2n == 11) && packed_width_ok(data_width, format, rndmode) && twdl_width >= 8 && twdl_width <= 16 && format == 0 && use_fly == 1))
        return false;
    if (rndmode && diag_env("INTFFT_NO_PACKED_ROUND")) return false; // ROUNDING: all three directions, the cores' native orders too
    if (direction == 0) return (in_order == 0 || in_order == 2) && (out_order == 0 || out_order == 1 || out_order == 3); // + HALVES in, BITREV / BITREV_LANES out
    if (direction == 1) return (in_order == 0 || in_order == 1 || in_order == 3) && (out_order == 0 || out_order == 2); // + BITREV / BITREV_LANES in, HALVES out
    return in_order == 0 && out_order == 0;
}

const char *fast4096_kernel_name() { return "k_fft4096_i16"; }

template <int L, int MODE, bool FAST_OK, int OB = 0, int ROUND = 0>
static hipError_t launch4k(const u32 *in, u32 *out, const int2 *tw, const RoundCConsts &c, size_t nframes,
                           const Slice &sl, hipStream_t stream, int halves = 0)
{
    const size_t cap = resident_blocks(kptr(k_fft4096_i16<L, MODE, FAST_OK, OB, ROUND>), 256, 2);
    const size_t chunks = (nframes + ((size_t)1 << (12 - L)) - 1) >> (12 - L);
    const unsigned blocks = (unsigned)(chunks < cap ? chunks : cap);
    hipLaunchKernelGGL((k_fft4096_i16<L, MODE, FAST_OK, OB, ROUND>), dim3(blocks), dim3(256), 0, stream, in, out, tw, c, nframes, sl,
                       halves);
    return hipGetLastError();
}

hipError_t launch_fast4096_mid(int twd, void *scratch, size_t nblocks4k, const int2 *tw_all, const int2 *h_tw, hipStream_t stream,
                               int data_width, int rndmode)
{
    if (nblocks4k == 0) return hipSuccess;
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
    if (!rndmode) to_dit_packing_host(c); // MODE_MID runs both cores: DIT packing (see the kernel); round mode: DIF packing
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    if (data_width != 16) sl.set_width(data_width);
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    u32 *p = static_cast<u32 *>(scratch);
    if (rndmode) return launch4k<12, MODE_MID, false, false, 1>(p, p, tw_all, c, nblocks4k, sl, stream);
    return (twd == 16 && allow_fast) ? launch4k<12, MODE_MID, true>(p, p, tw_all, c, nblocks4k, sl, stream)
                                     : launch4k<12, MODE_MID, false>(p, p, tw_all, c, nblocks4k, sl, stream);
}

template <int L>
static hipError_t launch4k_l(int direction, bool fast_ok, const u32 *pin, u32 *pout, const int2 *tw_all, const RoundCConsts &c,
                             size_t nframes, const Slice &sl, hipStream_t stream, int lc_bitrev, int halves, int round)
{
    // lc_bitrev: 0 natural, 1 BITREV, 2 BITREV_LANES on the frequency side (single cores only) -> the OB = 0 / 1 / 2 instantiations
#define INTFFT_4K_OB(MODE, FX, RD)                                                                                                      \
    (lc_bitrev == 2   ? launch4k<L, MODE, FX, 2, RD>(pin, pout, tw_all, c, nframes, sl, stream, halves)                                 \
     : lc_bitrev == 1 ? launch4k<L, MODE, FX, 1, RD>(pin, pout, tw_all, c, nframes, sl, stream, halves)                                 \
                      : launch4k<L, MODE, FX, 0, RD>(pin, pout, tw_all, c, nframes, sl, stream, halves))
    if (round) {
        switch (direction) {
        case 0: return sl.wd != 16 ? INTFFT_4K_OB(MODE_FWD, false, 2) : INTFFT_4K_OB(MODE_FWD, false, 1);
        case 1: return sl.wd != 16 ? INTFFT_4K_OB(MODE_INV, false, 2) : INTFFT_4K_OB(MODE_INV, false, 1);
        default:
            return sl.wd != 16 ? launch4k<L, MODE_PAIR, false, 0, 2>(pin, pout, tw_all, c, nframes, sl, stream)
                               : launch4k<L, MODE_PAIR, false, 0, 1>(pin, pout, tw_all, c, nframes, sl, stream);
        }
    }
    switch (direction) {
    case 0: return fast_ok ? INTFFT_4K_OB(MODE_FWD, true, 0) : INTFFT_4K_OB(MODE_FWD, false, 0);
    case 1: return fast_ok ? INTFFT_4K_OB(MODE_INV, true, 0) : INTFFT_4K_OB(MODE_INV, false, 0);
    default:
        return fast_ok ? launch4k<L, MODE_PAIR, true>(pin, pout, tw_all, c, nframes, sl, stream)
                       : launch4k<L, MODE_PAIR, false>(pin, pout, tw_all, c, nframes, sl, stream);
    }
#undef INTFFT_4K_OB
}

hipError_t launch_fast4096(int log2n, int direction, int twd, int lc_bitrev, int halves, const void *in, void *out, const int2 *tw_all, const int2 *h_tw,
                           size_t nframes, hipStream_t stream, int round, int data_width)
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
    if (direction == 1 || (direction == 2 && !round)) to_dit_packing_host(c); // kernels with DP (see k_fft4096_i16) hold the DIT packing
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    if (data_width != 16) sl.set_width(data_width);
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fast_ok = twd == 16 && allow_fast;
    const u32 *pin = static_cast<const u32 *>(in);
    u32 *pout = static_cast<u32 *>(out);
    if (log2n == 11) return launch4k_l<11>(direction, fast_ok, pin, pout, tw_all, c, nframes, sl, stream, lc_bitrev, halves, round);
    return launch4k_l<12>(direction, fast_ok, pin, pout, tw_all, c, nframes, sl, stream, lc_bitrev, halves, round);
}

} // namespace intfft
