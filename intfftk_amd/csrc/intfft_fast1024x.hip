// intfft_fast1024x.hip -- packed-int16 wave kernel for N = 1024, inverse core and FFT->IFFT pair:
// int_ifftNk / int_fft_ifft_pair with NFFT = 10, DATA_WIDTH = 16, TWDL_WIDTH <= 16, scaled-truncate,
// natural-order input and output (src/vhdl/fft/int_ifftNk.vhd:71-343,
// src/vhdl/main/int_fft_ifft_pair.vhd:209-280).  (The forward core alone is intfft_fast1024.hip.)
//
// Same wave-per-frame mapping as the forward kernel, run in both directions:
//
//   L1  reg = n9..6, lane = n5..0                              DIF 9,8,7,6         / DIT 6,7,8,9
//       v_permlane32_swap (lane5 <-> reg bit 3): stage 5;  v_permlane16_swap (lane4 <-> reg bit 2): stage 4
//   LC  reg = n3..0, lane = rev6(n9..4)  (via the wave-private LDS transpose)   DIF 3..0 / DIT 0..3
//
// The pair stays in LC between the two cores (position n holds X[bitrev(n)], exactly what int_ifftNk
// expects there: int_fft_ifft_pair.vhd:242-280).  The inverse alone loads into LC with the bit reversal
// folded into the addressing (256 contiguous bytes per load instruction).  Twiddles are held once, in
// the DIF packing, and shared by both directions (re/im-swapped multiplier feed, int_dit2_fly.vhd:304-322).
#include "intfft_pk16.hpp"

#include <cstdlib>

namespace intfft {

constexpr int ROWX = 20;
// Row r of the wave's 64 x 16 transpose tile starts at dword rowx(r): 20-dword rows (conflict-free b128 row reads) plus 16
// dwords per block of 16 rows -- 16 * ROWX = 0 mod 64 banks, and the inverse core's column writes have the block number
// (n9, n8) in lane bits: without the pad those four lanes share a bank (4-way conflict on every write).
constexpr int XBLK = 16 * ROWX + 16, XWAVE = 4 * XBLK;
__device__ __forceinline__ constexpr int rowx(int r) { return ROWX * r + 16 * (r >> 4); }

enum { X_INV = 1, X_PAIR = 2 };

// L < 10 (64 <= N < 1024, natural order only): 2^(10-L) frames share the wave as in intfft_fast1024.hip -- the
// stages of the frame-number bits are skipped in both cores.  The pair needs no other change (it loads and stores
// in the L1 layout); the inverse alone loads X[brev_L(n)] with the mirror image of the forward kernel's
// short-frame store (dwordx4 loads + two lane swaps, LC lane bits per lane_bit<L>()).
// ROUND: RNDMODE = 1 (rhu2 sums on full-width values, exact extraction, no pre-shifted outputs)
template <int L, int MODE, bool FAST_OK, int ROUND = 0>
#ifdef INTFFT_WPE_X /* A/B: tools/build_variant.sh wpe5 intfft_fast1024x.hip -DINTFFT_WPE_X=5 */
#define INTFFT_WPE_X_ATTR __attribute__((amdgpu_waves_per_eu(INTFFT_WPE_X)))
#else
#define INTFFT_WPE_X_ATTR
#endif
__global__ __launch_bounds__(256) INTFFT_WPE_X_ATTR void k_fft1024x_i16(const u32 *in, u32 *out, const int2 *__restrict__ twt,
                                                      const RoundCConsts c, size_t nframes_user, const Slice sl,
                                                      int in_bitrev, int out_halves)
{
    static_assert(!ROUND || !FAST_OK, "round mode uses the exact extraction");
    constexpr int FP = 1 << (10 - L), NS = L - 6;        // frames per chunk; executed stages among 9..6
    const size_t nframes = (nframes_user + FP - 1) / FP; // chunks
    __shared__ __attribute__((aligned(16))) u32 lds_all[4 * XWAVE];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32 *lds = lds_all + wv * XWAVE;

    // frame-invariant twiddles: stages 9..6 (reg offsets 8,4,2,1; index 64*jj + lane), 5 and 4
    RoundTw ta;
    u32 wa5, wb5, wa4, wb4;
    auto ld = [&](int idx, u32 &wa, u32 &wb) {
        const int2 w = twt[idx];
        wa = pack_wa(w);
        wb = pack_wb(w);
    };
    if constexpr (L >= 10) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ld(511 + 64 * j + lane, ta.wa8[j], ta.wb8[j]);
    }
    if constexpr (L >= 9) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ld(255 + 64 * j + lane, ta.wa4[j], ta.wb4[j]);
    }
    if constexpr (L >= 8) {
#pragma unroll
        for (int j = 0; j < 2; ++j) ld(127 + 64 * j + lane, ta.wa2[j], ta.wb2[j]);
    }
    if constexpr (L >= 7) ld(63 + lane, ta.wa1[0], ta.wb1[0]);
    ld(31 + (lane & 31), wa5, wb5);
    ld(15 + (lane & 15), wa4, wb4);
    // twiddles in the DIT packing: the inverse core multiplies the unswapped B; in a (truncate-mode) pair the forward core
    // produces D with its halves exchanged instead (free: op_sel of the packed subtract; group4's DPK form)
    constexpr bool DP = MODE == X_INV || !ROUND;
    if constexpr (DP) {
        to_dit_packing(ta);
        to_dit_packing(wa5, wb5);
        to_dit_packing(wa4, wb4);
    }
    const u32 w5a[4] = {wa5, wa5, wa5, wa5}, w5b[4] = {wb5, wb5, wb5, wb5};
    const u32 w4a[4] = {wa4, wa4, wa4, wa4}, w4b[4] = {wb4, wb4, wb4, wb4};

    // LDS transpose addressing.  "mid" layout (after / before the lane swaps): lane5 = n9, lane4 = n8,
    // lane3..0 = n3..0; reg j3 = n5, j2 = n4, j1 = n7, j0 = n6.   LC: reg = n3..0, lane bit i = n(9-i).
    // forward (mid -> LC): element (lane t, reg j) -> row = LC lane, column = n3..0
    const int t5 = lane >> 5, t4 = (lane >> 4) & 1;
    u32 *wr_f = lds + ROWX * (t5 + 2 * t4) + (lane & 15); // + rowx(4 j1 + 8 j0 + 16 j3 + 32 j2)
    // inverse (LC -> mid): element (lane l, reg r) -> row = mid lane (32 n9 + 16 n8 + r), column = mid reg
    const int l0 = lane & 1, l1 = (lane >> 1) & 1, l2 = (lane >> 2) & 1, l3 = (lane >> 3) & 1, l4 = (lane >> 4) & 1,
              l5 = lane >> 5; // l_i = n(9-i)
    // NATURAL input: LC lane bit i = n(9-i).  BITREV input (memory index = n): LC lane = n9..n4 (bit i = n(4+i))
    u32 *wr_i = in_bitrev ? lds + XBLK * (2 * l5 + l4) + ((l1 << 3) | (l0 << 2) | (l3 << 1) | l2)
                          : lds + XBLK * (2 * l0 + l1) + ((l4 << 3) | (l5 << 2) | (l2 << 1) | l3); // + ROWX * r
    int lane_off = 0, lane_frame = 0; // short-frame inverse: see intfft_fast1024.hip
    if (L < 10 && MODE == X_INV && !in_bitrev) {
        auto a = [&](int k) { return (lane >> lane_bit<L>(k)) & 1; }; // LC lane bit lane_bit<L>(k) = a_k
        wr_i = lds + XBLK * (2 * a(9) + a(8)) + ((a(5) << 3) | (a(4) << 2) | (a(7) << 1) | a(6));
        // while loading (before the swaps) lane bit 5 = a3 and lane bit 4 = a2
        lane_off = ((lane >> 5) & 1) * out_weight<L>(3) + ((lane >> 4) & 1) * out_weight<L>(2);
#pragma unroll
        for (int k = 4; k < 10; ++k) {
            if (k == L - 1 || k == L - 2) continue;
            lane_off += a(k) * out_weight<L>(k);
            if (k >= L) lane_frame += a(k) << (k - L);
        }
    }
    const uint4 *rd_base = reinterpret_cast<const uint4 *>(lds + rowx(lane));
    const short s3 = (short)(1 - (lane >> 5)); // LC: kind = n4 = lane bit 5
    const v2s sh3 = {s3, s3};

    const size_t wave0 = (size_t)blockIdx.x * 4 + wv, nwaves = (size_t)gridDim.x * 4;
    for (size_t f = wave0; f < nframes; f += nwaves) {
        u32 v[16];
        const u32 *src = in + f * 1024;
        const bool partial = L < 10 && (f + 1) * FP > nframes_user; // last chunk: absent frames read as 0, not stored
        if (L < 10 && MODE == X_INV && !in_bitrev) {
            typedef u32 v4u __attribute__((ext_vector_type(4)));
            const bool ok = !partial || f * FP + (size_t)lane_frame < nframes_user;
            if constexpr (L == 6) {
                // N = 64: a lane's four vectors lie in four quarters of its 256-byte frame (a load instruction would read sixteen 64-byte
                // runs): the chunk comes in as 1 KiB per instruction and is handed over through the wave's LDS tile in memory order
                v4u y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 64 * i + lane; // 16-byte piece of the chunk: frame e >> 4 within it
                    y[i] = v4u{0u, 0u, 0u, 0u};
                    if (!partial || f * FP + (size_t)(e >> (L - 2)) < nframes_user) y[i] = INTFFT_LD(reinterpret_cast<const v4u *>(src) + e);
                }
                wave_lds_fence(); // the previous frame's transposition reads
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<v4u *>(lds + 4 * (64 * i + lane)) = y[i];
                wave_lds_fence();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4u x = *reinterpret_cast<const v4u *>(lds + lane_off + (q & 1) * out_weight<L>(0) + (q >> 1) * out_weight<L>(1));
                    v[q] = x.x;
                    v[q + 8] = x.y;
                    v[q + 4] = x.z;
                    v[q + 12] = x.w;
                }
                wave_lds_fence();
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v4u x = {0u, 0u, 0u, 0u};
                    if (ok)
                        x = INTFFT_LD(
                            reinterpret_cast<const v4u *>(src + lane_off + (q & 1) * out_weight<L>(0) + (q >> 1) * out_weight<L>(1)));
                    v[q] = x.x;
                    v[q + 8] = x.y;
                    v[q + 4] = x.z;
                    v[q + 12] = x.w;
                }
            }
#pragma unroll
            for (int g = 0; g < 16; g += 8)
#pragma unroll
                for (int r = 0; r < 4; ++r) swap16(v[g + r], v[g + r + 4]);
#pragma unroll
            for (int r = 0; r < 8; ++r) swap32(v[r], v[r + 8]);
        } else if (MODE == X_INV && in_bitrev) {
            // memory index = n: x4 loads give (regs n9 n8 n1 n0, lane n3 n2 n7..4); two lane swaps -> (regs n3..0, lane n9..4)
            typedef u32 v4u __attribute__((ext_vector_type(4)));
            typedef u32 v2u __attribute__((ext_vector_type(2)));
            const v4u *s4 = reinterpret_cast<const v4u *>(src) + (((lane & 15) << 2) | (lane >> 4));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4u x = {0u, 0u, 0u, 0u}; // short frames: the vector's frame = bits a9..aL of (q, lane & 15)
                if (!partial || f * FP + (size_t)(((q << 8) | ((lane & 15) << 4)) >> L) < nframes_user) {
                    if (in_bitrev == 2) {
                        // BITREV_LANES in (the serial stream of outbuf_half_path.vhd:160-172): core position n sits at memory index
                        // (n & 1) * N/2 + (n >> 1), so the vector's four consecutive n are two 8-byte pieces, one per half of its frame
                        const int n0 = 4 * (64 * q + (((lane & 15) << 2) | (lane >> 4))), nl = n0 & ((1 << L) - 1);
                        const u32 *const d = src + (n0 - nl) + (nl >> 1);
                        const v2u ev = INTFFT_LD(reinterpret_cast<const v2u *>(d)), od = INTFFT_LD(reinterpret_cast<const v2u *>(d + (1 << (L - 1))));
                        x = v4u{ev.x, od.x, ev.y, od.y};
                    } else
                        x = INTFFT_LD(s4 + 64 * q);
                }
                v[4 * q] = x.x;
                v[4 * q + 1] = x.y;
                v[4 * q + 2] = x.z;
                v[4 * q + 3] = x.w;
            }
#pragma unroll
            for (int g = 0; g < 16; g += 8)
#pragma unroll
                for (int r = 0; r < 4; ++r) swap16(v[g + r], v[g + r + 4]);
#pragma unroll
            for (int r = 0; r < 8; ++r) swap32(v[r], v[r + 8]);
        } else if (partial) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[j] = f * FP + (size_t)((64 * j + lane) >> L) < nframes_user ? src[64 * j + lane] : 0u;
        } else if (MODE == X_INV) { // LC: v[r] = X[rev10(n)] = X[64 * rev4(r) + lane]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
                v[r] = INTFFT_LD(src + 64 * rr + lane);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = INTFFT_LD(src + 64 * j + lane);
        }
        const bool fast = FAST_OK && frame_has_guard_bit(v, sl.gbias, sl.gmask);
        if (!fast && sl.wd != 16) wrap_inputs(v, sl.wd); // DATA_WIDTH < 16: containers wrapped to w bits (exact path)

#define INTFFT_XBODY(FX, RD)                                                                             \
    {                                                                                                   \
        if (MODE == X_PAIR) { /* forward core: L1 -> LC */                                              \
            dif_round<FX, false, NS, RD, DP>(v, ta, sl, sh3);                                                \
            swap_guard(v);                                                                              \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) swap32(v[j], v[j + 8]);                       \
            group4<RD, FX, false, !RD, false, (NS >= 1 && !RD ? 0xA : 0), false, DP>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], w5a, w5b, sl); \
            group4<RD, FX, false, !RD, false, (NS >= 1 && !RD ? 0xA : 0), false, DP>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], w5a, w5b, sl); \
            swap_guard(v);                                                                              \
            _Pragma("unroll") for (int g = 0; g < 16; g += 8)                                           \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) swap16(v[g + j], v[g + j + 4]);           \
            group4<RD, FX, false, !RD, false, 0x0, false, DP>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], w4a, w4b, sl); \
            group4<RD, FX, false, !RD, false, (RD ? 0x0 : 0xF), false, DP>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], w4a, w4b, sl); \
            wave_lds_fence();                                                              \
            _Pragma("unroll") for (int j = 0; j < 16; ++j)                                              \
            {                                                                                           \
                const int j0 = j & 1, j1 = (j >> 1) & 1, j2 = (j >> 2) & 1, j3 = (j >> 3) & 1;          \
                wr_f[rowx(4 * j1 + 8 * j0 + 16 * j3 + 32 * j2)] = v[j];                                    \
            }                                                                                           \
            wave_lds_fence();                                                              \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                               \
            {                                                                                           \
                const uint4 x = rd_base[q];                                                             \
                v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;             \
            }                                                                                           \
            wave_lds_fence();                                                              \
            dif_round_c<FX, RD, DP>(v, c, sl, sh3);                                                      \
        }                                                                                               \
        /* inverse core: LC -> L1 */                                                                    \
        dit_round_c<FX, RD, DP>(v, c, sl);                                                           \
        wave_lds_fence();                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) wr_i[ROWX * r] = v[r];                           \
        wave_lds_fence();                                                                  \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                   \
        {                                                                                               \
            const uint4 x = rd_base[q];                                                                 \
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;                 \
        }                                                                                               \
        wave_lds_fence();                                                                  \
        group4_dit<FX, false, RD, DP>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7], w4a, w4b, sl);     \
        group4_dit<FX, false, RD, DP>(v[8], v[12], v[9], v[13], v[10], v[14], v[11], v[15], w4a, w4b, sl); \
        swap_guard(v);                                                                                  \
        _Pragma("unroll") for (int g = 0; g < 16; g += 8)                                               \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) swap16(v[g + j], v[g + j + 4]);               \
        group4_dit<FX, false, RD, DP>(v[0], v[8], v[1], v[9], v[2], v[10], v[3], v[11], w5a, w5b, sl);   \
        group4_dit<FX, false, RD, DP>(v[4], v[12], v[5], v[13], v[6], v[14], v[7], v[15], w5a, w5b, sl); \
        swap_guard(v);                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) swap32(v[j], v[j + 8]);                           \
        dit_round<FX, NS, RD, DP>(v, ta, sl);                                                        \
    }
        if (FAST_OK && fast) INTFFT_XBODY(FAST_OK, ROUND)
        else INTFFT_XBODY(false, ROUND)
#undef INTFFT_XBODY
        if (out_halves) { // HALVES: beat q = 64 jj + lane holds (x[i], x[i + N/2]) = (v[j0], v[j0 | 2^(L-7)]), see the forward kernel
            if constexpr (L >= 7) {
                typedef u32 v2u __attribute__((ext_vector_type(2)));
                v2u *d2 = reinterpret_cast<v2u *>(out + f * 1024) + lane;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    constexpr int HB = 1 << (L - 7);
                    const int j0 = ((jj >> (L - 7)) << (L - 6)) | (jj & (HB - 1));
                    const v2u w = {v[j0], v[j0 | HB]};
                    if (!partial || f * FP + (size_t)(jj >> (L - 7)) < nframes_user) __builtin_nontemporal_store(w, d2 + 64 * jj);
                }
            }
        } else if (partial) {
            u32 *dst = out + f * 1024 + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (f * FP + (size_t)((64 * j + lane) >> L) < nframes_user) dst[64 * j] = v[j];
        } else {
            u32 *dst = out + f * 1024 + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j) __builtin_nontemporal_store(v[j], dst + 64 * j);
        }
    }
}

bool fast1024x_supported(int log2n, int data_width, int twdl_width, int format, int rndmode, int direction,
                         int use_fly, int in_order, int out_order)
{
    if (!(packed_width_ok(data_width, format, rndmode) && twdl_width >= 8 && twdl_width <= 16 && format == 0 && use_fly == 1))
        return false;
    if (rndmode && diag_env("INTFFT_NO_PACKED_ROUND")) return false; // ROUNDING: the same orders as truncate mode
    if (log2n == 6) return (direction == 1 || direction == 2) && in_order == 0 && out_order == 0;
    // N >= 128: the inverse also takes BITREV (native int_ifftNk beats) or BITREV_LANES (their serial form) in and HALVES (native) out
    return log2n >= 7 && log2n <= 10 &&
           ((direction == 2 && in_order == 0 && out_order == 0) ||
            (direction == 1 && (in_order == 0 || in_order == 1 || in_order == 3) && (out_order == 0 || out_order == 2)));
}

const char *fast1024x_kernel_name() { return "k_fft1024x_i16"; }

template <int L, int MODE, bool FAST_OK, int ROUND = 0>
static hipError_t launchx(const u32 *in, u32 *out, const int2 *tw, const RoundCConsts &c, size_t nframes,
                          const Slice &sl, int in_bitrev, int out_halves, hipStream_t stream)
{
    const size_t cap = resident_blocks(kptr(k_fft1024x_i16<L, MODE, FAST_OK, ROUND>), 256, 4);
    const size_t chunks = (nframes + ((size_t)1 << (10 - L)) - 1) >> (10 - L);
    const size_t need = (chunks + 3) / 4;
    const unsigned blocks = (unsigned)(need < cap ? need : cap);
    hipLaunchKernelGGL((k_fft1024x_i16<L, MODE, FAST_OK, ROUND>), dim3(blocks), dim3(256), 0, stream, in, out, tw, c, nframes, sl,
                       in_bitrev, out_halves);
    return hipGetLastError();
}

template <int L>
static hipError_t launchx_l(int direction, bool fast_ok, const u32 *pin, u32 *pout, const int2 *tw_all, const RoundCConsts &c,
                            size_t nframes, const Slice &sl, int in_bitrev, int out_halves, hipStream_t stream, int round)
{
    if (round)
        return sl.wd != 16 ? (direction == 1 ? launchx<L, X_INV, false, 2>(pin, pout, tw_all, c, nframes, sl, in_bitrev, out_halves, stream)
                                             : launchx<L, X_PAIR, false, 2>(pin, pout, tw_all, c, nframes, sl, 0, 0, stream))
               : direction == 1 ? launchx<L, X_INV, false, 1>(pin, pout, tw_all, c, nframes, sl, in_bitrev, out_halves, stream)
                                : launchx<L, X_PAIR, false, 1>(pin, pout, tw_all, c, nframes, sl, 0, 0, stream);
    if (direction == 1)
        return fast_ok ? launchx<L, X_INV, true>(pin, pout, tw_all, c, nframes, sl, in_bitrev, out_halves, stream)
                       : launchx<L, X_INV, false>(pin, pout, tw_all, c, nframes, sl, in_bitrev, out_halves, stream);
    return fast_ok ? launchx<L, X_PAIR, true>(pin, pout, tw_all, c, nframes, sl, 0, 0, stream)
                   : launchx<L, X_PAIR, false>(pin, pout, tw_all, c, nframes, sl, 0, 0, stream);
}

hipError_t launch_fast1024x(int log2n, int direction, int twd, int in_bitrev, int out_halves, const void *in, void *out,
                            const int2 *tw_all, const int2 *h_tw, size_t nframes, hipStream_t stream, int round, int data_width)
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
    if (direction == 1 || !round) to_dit_packing_host(c); // kernels with DP (see k_fft1024x_i16) hold their twiddles in the DIT packing
    Slice sl{twd - 1, twd, 0x05040100u, 0x07060302u};
    if (data_width != 16) sl.set_width(data_width);
    static const int allow_fast = diag_env("INTFFT_FAST_EXTRACT") ? atoi(diag_env("INTFFT_FAST_EXTRACT")) : 1;
    const bool fast_ok = twd == 16 && allow_fast;
    const u32 *pin = static_cast<const u32 *>(in);
    u32 *pout = static_cast<u32 *>(out);
    switch (log2n) {
    case 6: return launchx_l<6>(direction, fast_ok, pin, pout, tw_all, c, nframes, sl, 0, 0, stream, round);
    case 7: return launchx_l<7>(direction, fast_ok, pin, pout, tw_all, c, nframes, sl, in_bitrev, out_halves, stream, round);
    case 8: return launchx_l<8>(direction, fast_ok, pin, pout, tw_all, c, nframes, sl, in_bitrev, out_halves, stream, round);
    case 9: return launchx_l<9>(direction, fast_ok, pin, pout, tw_all, c, nframes, sl, in_bitrev, out_halves, stream, round);
    default: return launchx_l<10>(direction, fast_ok, pin, pout, tw_all, c, nframes, sl, in_bitrev, out_halves, stream, round);
    }
}

} // namespace intfft
